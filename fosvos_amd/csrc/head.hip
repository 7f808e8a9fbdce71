// Fused side-output head: 4 x (per-channel transposed conv k=2f, stride f) + centre crop + concat
// + 1x1 fuse, and the optional score_dsn -> 1-channel transposed conv -> crop branch
// (reference: src/networks/osvos_vgg.py:69-82, src/layers/osvos_layers.py:47-54).
//
// The reference materialises 4 x [N,16,~H,~W] deconv outputs, 4 crops and a [N,64,H,W] concat
// (about 310 MB fp32 per 480x854 frame).  Here one kernel reads the four small side maps (8.7 MB,
// L2-resident) and writes the logit maps only; HBM-bound on the fp32 output.
//
// Geometry per scale s (f = 2^(s+1), k = 2f): deconv output size (h+1)*f; output pixel Yo receives
// input rows i1 = Yo/f (tap ky = Yo%f) and i0 = i1-1 (tap ky+f); the crop removes
// top = floor(((h+1)f - H)/2) leading rows (same for columns).  Upscale weights enter as their
// diagonal, channel fastest: filt[s] = [k][k][16] (one k x k filter per channel), filt1[s] = [k][k].
#include "common.hpp"

using namespace fosvos;

namespace {
struct HeadGeom {
    int hs[4], ws[4];    // side map sizes
    int top[4], left[4]; // crop offsets in deconv-output coordinates
};

struct HeadPtrs {
    const float *side[4];
    const float *filt[4];
    const float *filt1[4];
};

__host__ inline bool make_geom(const int hs[4], const int ws[4], int H, int W, HeadGeom &g) {
    for (int s = 0; s < 4; ++s) {
        const int f = 2 << s;
        const int uh = (hs[s] + 1) * f, uw = (ws[s] + 1) * f;
        if (hs[s] <= 0 || ws[s] <= 0 || uh < H || uw < W) return false;
        g.hs[s] = hs[s];
        g.ws[s] = ws[s];
        g.top[s] = (uh - H) / 2;   // floor(d/2) leading pixels removed (src/layers/osvos_layers.py:47-54)
        g.left[s] = (uw - W) / 2;
    }
    return true;
}

// ---------------------------------------------------------------------------------------- forward
// Workgroup = 16 x 64 output pixels.  Thread (y, xr) owns the 4 pixels x = xr + 16 m of row y: they share their
// residues modulo every scale's stride (f divides 16), hence the SAME filter taps at every scale, so the
// 4 taps x 16 channels of filter (pre-multiplied by the fuse weights) are fetched once per scale and thread and
// reused for the 4 pixels.  The side-map windows of all four scales (at most 10x34, 6x18, 4x10, 3x6 low-res pixels
// x 16 channels, zero outside the map) are staged in LDS once per workgroup.  The per-pixel formulation read
// 128 x 16 bytes through the texture path for every output pixel (840 MB per frame: 58 us).
constexpr int HF_TY = 16, HF_TX = 64;
__host__ __device__ constexpr int hf_rows(int s) { return HF_TY / (2 << s) + 2; }
__host__ __device__ constexpr int hf_cols(int s) { return HF_TX / (2 << s) + 2; }
__host__ __device__ constexpr int hf_off(int s) {  // float offset of scale s's window in LDS
    int o = 0;
    for (int i = 0; i < s; ++i) o += hf_rows(i) * hf_cols(i) * 16;
    return o;
}

template <bool WITH_SIDE_OUT>
__global__ __launch_bounds__(256) void k_head_fwd(HeadPtrs p, HeadGeom g, const float *__restrict__ dsn_w,
                                                   const float *__restrict__ dsn_b, const float *__restrict__ fuse_w,
                                                   const float *__restrict__ fuse_b, float *__restrict__ fused,
                                                   float *so0, float *so1, float *so2, float *so3, int H, int W,
                                                   int umask) {
    // umask bit s: the 16 per-channel filters of scale s are IDENTICAL (the caller's promise: what interp_surgery writes and
    // lr 0 keeps, src/layers/osvos_layers.py:70-81, src/util/network_provider.py:154-155).  Then
    //   sum_c filt[ky][kx][c] fw[c] side[i][j][c] = filt[ky][kx][0] * z[i][j],   z = sum_c fw[c] side[i][j][c]:
    // the channel contraction is done once per LOW-RES pixel while the window is staged (s_z: one float per pixel) and an
    // output pixel costs 4 multiply-adds and 4 four-byte LDS reads per scale instead of 64 and 16 sixteen-byte reads (the
    // general form is bound by LDS bandwidth: 36-44 us for five 480x854 frames on the path between the passes).
    __shared__ __attribute__((aligned(16))) float s_win[hf_off(4)];
    __shared__ float s_z[hf_off(4) / 16], s_zs[WITH_SIDE_OUT ? hf_off(4) / 16 : 1];
    __shared__ float s_fw[64], s_dw[64];
    const int tid = threadIdx.x;
    const int n = blockIdx.z, Y0 = blockIdx.y * HF_TY, X0 = blockIdx.x * HF_TX;
    if (tid < 64) {
        s_fw[tid] = fuse_w[tid];
        s_dw[tid] = WITH_SIDE_OUT ? dsn_w[tid] : 0.f;
    }
    if (umask) __syncthreads();  // (the contraction below reads s_fw)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int f = 2 << s, NR = hf_rows(s), NC = hf_cols(s);
        const int i_lo = (Y0 + g.top[s]) / f - 1, j_lo = (X0 + g.left[s]) / f - 1;
        const float4 *sd = reinterpret_cast<const float4 *>(p.side[s] + (int64_t)n * g.hs[s] * g.ws[s] * 16);
        if ((umask >> s) & 1) {
            // four consecutive lanes hold the four channel quads of one window pixel (NR * NC * 4 is a multiple of 4 and so
            // is 256: a group is complete or absent)
            const float db = WITH_SIDE_OUT ? dsn_b[s] : 0.f;
            for (int e = tid; e < NR * NC * 4; e += 256) {
                const int q = e & 3, px = e >> 2;
                const int r = px / NC, c = px - r * NC;
                const int i = i_lo + r, j = j_lo + c;
                const bool in = i >= 0 && i < g.hs[s] && j >= 0 && j < g.ws[s];
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (in) v = sd[((int64_t)i * g.ws[s] + j) * 4 + q];
                const float *fwq = s_fw + 16 * s + 4 * q, *dwq = s_dw + 16 * s + 4 * q;
                float d = fwq[0] * v.x + fwq[1] * v.y + fwq[2] * v.z + fwq[3] * v.w;
                d += __shfl_xor(d, 1, 64);
                d += __shfl_xor(d, 2, 64);
                float sc = 0.f;
                if (WITH_SIDE_OUT) {
                    sc = dwq[0] * v.x + dwq[1] * v.y + dwq[2] * v.z + dwq[3] * v.w;
                    sc += __shfl_xor(sc, 1, 64);
                    sc += __shfl_xor(sc, 2, 64);
                }
                if (q == 0) {
                    s_z[hf_off(s) / 16 + px] = d;
                    if (WITH_SIDE_OUT) s_zs[hf_off(s) / 16 + px] = in ? sc + db : 0.f;
                }
            }
            continue;
        }
        // the window is kept as four channel-quad planes [q][pixel]: neighbouring pixels sit 16 B apart, so the taps' reads
        // have no bank conflicts (pixel-major, at a 64-B pitch, 47 % of this kernel's LDS cycles were conflict cycles)
        f32x4 *dst = reinterpret_cast<f32x4 *>(s_win + hf_off(s));
        for (int e = tid; e < NR * NC * 4; e += 256) {
            const int q = e & 3, px = e >> 2;
            const int r = px / NC, c = px - r * NC;
            const int i = i_lo + r, j = j_lo + c;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i >= 0 && i < g.hs[s] && j >= 0 && j < g.ws[s]) v = sd[((int64_t)i * g.ws[s] + j) * 4 + q];
            dst[q * (NR * NC) + px] = f32x4{v.x, v.y, v.z, v.w};
        }
    }
    __syncthreads();
    const int y = tid >> 4, xr = tid & 15;
    const int Y = Y0 + y;
    const bool row_ok = Y < H;
    int64_t idx[4];
    bool px_ok[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int X = X0 + xr + 16 * m;
        px_ok[m] = row_ok && X < W;
        idx[m] = ((int64_t)n * H + Y) * W + X;
    }
    float out[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) out[m] = fuse_b[0];
    // the scale and tap loops stay rolled: unrolled, hipcc hoists every filter row of every scale (512 VGPRs + scratch)
#pragma unroll 1
    for (int s = 0; s < 4; ++s) {
        const int f = 2 << s, k = 4 << s, NC = HF_TX / f + 2;
        const int i_lo = (Y0 + g.top[s]) / f - 1, j_lo = (X0 + g.left[s]) / f - 1;
        const int Yo = Y + g.top[s], Xo = X0 + xr + g.left[s];
        const int i1 = Yo / f, ky1 = Yo - i1 * f;
        const int kx1 = Xo % f;  // the same for x + 16 m
        int woff = 0;
        for (int t = 0; t < s; ++t) woff += (HF_TY / (2 << t) + 2) * (HF_TX / (2 << t) + 2) * 16;
        const f32x4 *win = reinterpret_cast<const f32x4 *>(s_win + woff);
        const int npx = (HF_TY / f + 2) * NC;
        const float *filt = p.filt[s];
        const float *filt1 = WITH_SIDE_OUT ? p.filt1[s] : nullptr;
        const float db = WITH_SIDE_OUT ? dsn_b[s] : 0.f;
        if ((umask >> s) & 1) {
            const float *zw = s_z + woff / 16, *zsw = s_zs + (WITH_SIDE_OUT ? woff / 16 : 0);
            float so_u[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ab = 0; ab < 4; ++ab) {
                const int a = ab >> 1, b = ab & 1;
                const int i = i1 - a, ky = ky1 + a * f, kx = kx1 + b * f;
                const float h = filt[(ky * k + kx) * 16];
                const float f1 = WITH_SIDE_OUT ? filt1[ky * k + kx] : 0.f;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int j = (Xo + 16 * m) / f - b;
                    const int at = (i - i_lo) * NC + (j - j_lo);
                    out[m] += h * zw[at];
                    if (WITH_SIDE_OUT) so_u[m] += f1 * zsw[at];
                }
            }
            if (WITH_SIDE_OUT) {
                float *so = s == 0 ? so0 : s == 1 ? so1 : s == 2 ? so2 : so3;
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    if (px_ok[m]) so[idx[m]] = so_u[m];
            }
            continue;
        }
        float fw[16], dw[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            fw[c] = s_fw[16 * s + c];
            dw[c] = s_dw[16 * s + c];
        }
        float so_acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int ab = 0; ab < 4; ++ab) {
            const int a = ab >> 1, b = ab & 1;
            const int i = i1 - a, ky = ky1 + a * f, kx = kx1 + b * f;
            float gq[16];  // this tap's filter row, pre-multiplied by the fuse weights of scale s
            const float4 *fp = reinterpret_cast<const float4 *>(filt + (ky * k + kx) * 16);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 t = fp[q];
                gq[4 * q] = t.x * fw[4 * q];
                gq[4 * q + 1] = t.y * fw[4 * q + 1];
                gq[4 * q + 2] = t.z * fw[4 * q + 2];
                gq[4 * q + 3] = t.w * fw[4 * q + 3];
            }
            const float f1 = WITH_SIDE_OUT ? filt1[ky * k + kx] : 0.f;
            const bool i_ok = i >= 0 && i < g.hs[s];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int j = (Xo + 16 * m) / f - b;
                const f32x4 *vp = win + (i - i_lo) * NC + (j - j_lo);
                float dot = 0.f, score = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = vp[q * npx];
                    dot += gq[4 * q] * v[0] + gq[4 * q + 1] * v[1] + gq[4 * q + 2] * v[2] + gq[4 * q + 3] * v[3];
                    if (WITH_SIDE_OUT)
                        score += dw[4 * q] * v[0] + dw[4 * q + 1] * v[1] + dw[4 * q + 2] * v[2] + dw[4 * q + 3] * v[3];
                }
                out[m] += dot;
                if (WITH_SIDE_OUT && i_ok && j >= 0 && j < g.ws[s]) so_acc[m] += (score + db) * f1;
            }
        }
        if (WITH_SIDE_OUT) {
            float *so = s == 0 ? so0 : s == 1 ? so1 : s == 2 ? so2 : so3;
#pragma unroll
            for (int m = 0; m < 4; ++m)
                if (px_ok[m]) so[idx[m]] = so_acc[m];
        }
    }
#pragma unroll
    for (int m = 0; m < 4; ++m)
        if (px_ok[m]) fused[idx[m]] = out[m];
}

// ---------------------------------------------------------------------------------------- backward
// Per scale S (f = 2^(S+1), k = 2f): a workgroup owns a TI x TJ tile of low-res pixels.  It stages the
// (TI+1)f x (TJ+1)f window of upstream gradients (zero outside the image) and the whole [k][k][16]
// filter in LDS, then thread (pixel p, channel c) evaluates
//   T_c = sum_window filt[ky][kx][c] * d_fused[Y,X]      G = sum_window filt1[ky][kx] * d_side_out[Y,X]
//   d_side[c]  = fuse_w[16s+c] * T_c + dsn_w[s][c] * G
//   d_fuse_w[16s+c] += side[c] * T_c;  d_dsn_w[s][c] += side[c] * G;  d_dsn_b[s] += G
// from LDS only.  The tap sums are taken by thread (tap slice ts, pixel pl, channel quad cq): one 16-byte filter read serves
// four channels and (scales 1-3) one 16-byte window read four taps - 16 multiply-adds per five LDS instructions, where one
// thread per (pixel, channel) walking all k x k taps paid two LDS reads per multiply-add and was bound by LDS issue at every
// scale (52 / 25 / 18 / 27 us for five 480x854 frames, on the critical path between the passes).  The large filters are cut
// into TS tap slices (row ranges) whose partial sums meet in LDS in a fixed order.
// Block partials go to the workspace as slabs [block][3][16]; k_head_finish sums them in order.
template <int S> struct HeadTile;
template <> struct HeadTile<0> { static constexpr int TI = 8, TJ = 32, TS = 1; };
template <> struct HeadTile<1> { static constexpr int TI = 4, TJ = 16, TS = 1; };
template <> struct HeadTile<2> { static constexpr int TI = 4, TJ = 4, TS = 4; };
template <> struct HeadTile<3> { static constexpr int TI = 2, TJ = 8, TS = 4; };

// UNIFORM: the scale's 16 per-channel filters are identical (see k_head_fwd): T_c is ONE number per pixel, the filter is
// staged as k x k floats instead of k x k x 16 (64 KB at scale 3: one workgroup per CU) and the tap sum is split over all
// 256 threads as (tap slice u = 4 ts + cq, pixel).
template <int S, bool WITH_SIDE_OUT, bool UNIFORM>
__global__ __launch_bounds__(256) void k_head_bwd_scale(const float *__restrict__ side, const float *__restrict__ filt,
                                                         const float *__restrict__ filt1,
                                                         const float *__restrict__ fuse_w16,
                                                         const float *__restrict__ dsn_w16,
                                                         const float *__restrict__ d_fused,
                                                         const float *__restrict__ d_so, uint16_t *__restrict__ d_side,
                                                         float *__restrict__ slabs, int hs, int ws, int top, int left,
                                                         int H, int W) {
    constexpr int f = 2 << S, k = 4 << S;
    constexpr int TI = HeadTile<S>::TI, TJ = HeadTile<S>::TJ;
    constexpr int WR = (TI + 1) * f, WC = (TJ + 1) * f;
    extern __shared__ __attribute__((aligned(16))) float smem_h[];
    float *sF = smem_h;                 // [k*k][16]   (UNIFORM: [k*k])
    float *sW = sF + k * k * (UNIFORM ? 1 : 16);  // [WR][WC] window of d_fused
    float *sW1 = sW + WR * WC;          // [WR][WC] window of d_side_out   (WITH_SIDE_OUT)
    float *sF1 = sW1 + (WITH_SIDE_OUT ? WR * WC : 0);  // [k*k]            (WITH_SIDE_OUT)
    const int tid = threadIdx.x;
    const int n = blockIdx.z, i0 = blockIdx.y * TI, j0 = blockIdx.x * TJ;
    const int Y0 = i0 * f - top, X0 = j0 * f - left;
    // Staging: every global load of a thread is issued before the first LDS store (unconditional loads from clamped
    // addresses).  Rolled, these loops waited out one memory round trip per element: 16 + 27 in a row at scale 3,
    // which was the whole 24 us of the kernel.
    if constexpr (UNIFORM) {
        constexpr int FN = k * k, FIT = (FN + 255) / 256;
        float tf[FIT];
#pragma unroll
        for (int it = 0; it < FIT; ++it) tf[it] = filt[(int64_t)min(it * 256 + tid, FN - 1) * 16];  // channel 0 stands for all
#pragma unroll
        for (int it = 0; it < FIT; ++it)
            if (it * 256 + tid < FN) sF[it * 256 + tid] = tf[it];
    } else {
        constexpr int FN = k * k * 4, FIT = (FN + 255) / 256;
        float4 tf[FIT];
#pragma unroll
        for (int it = 0; it < FIT; ++it) tf[it] = reinterpret_cast<const float4 *>(filt)[min(it * 256 + tid, FN - 1)];
#pragma unroll
        for (int it = 0; it < FIT; ++it)
            if (it * 256 + tid < FN) reinterpret_cast<float4 *>(sF)[it * 256 + tid] = tf[it];
    }
    {
        constexpr int WN = WR * WC, WIT = (WN + 255) / 256;
        float tw[WIT], tw1[WITH_SIDE_OUT ? WIT : 1];
        bool okv[WIT];
#pragma unroll
        for (int it = 0; it < WIT; ++it) {
            const int e = min(it * 256 + tid, WN - 1);
            const int r = e / WC, cidx = e - r * WC;
            const int Y = Y0 + r, X = X0 + cidx;
            okv[it] = Y >= 0 && Y < H && X >= 0 && X < W;
            const int64_t off = ((int64_t)n * H + min(max(Y, 0), H - 1)) * W + min(max(X, 0), W - 1);
            tw[it] = d_fused ? d_fused[off] : 0.f;
            if (WITH_SIDE_OUT) tw1[it] = d_so[off];
        }
#pragma unroll
        for (int it = 0; it < WIT; ++it) {
            const int e = it * 256 + tid;
            if (e < WN) {
                sW[e] = okv[it] ? tw[it] : 0.f;
                if (WITH_SIDE_OUT) sW1[e] = okv[it] ? tw1[it] : 0.f;
            }
        }
    }
    if (WITH_SIDE_OUT)
        for (int e = tid; e < k * k; e += 256) sF1[e] = filt1[e];
    __syncthreads();
    // ---- the tile's pixels in chunks of PXC = 64 / TS.  Phase 1: partial tap sums, thread (ts, pl, cq): pixel pl of the
    // chunk, channels 4 cq .. 4 cq + 3, filter rows ts k / TS .. (ts + 1) k / TS - 1
    constexpr int TS = HeadTile<S>::TS, NPX = TI * TJ, PXC = 64 / TS;
    static_assert(NPX % PXC == 0 && PXC % 16 == 0, "whole chunks; phase 2 walks a chunk 16 pixels at a time");
    __shared__ __attribute__((aligned(16))) float sP[UNIFORM ? 1 : TS][UNIFORM ? 1 : PXC][16];
    __shared__ float sG[UNIFORM ? 1 : TS][UNIFORM ? 1 : PXC];
    constexpr int NU = 4 * TS, RU = k / NU;  // UNIFORM: tap slices (filter rows u RU .. u RU + RU - 1) and their partial sums
    static_assert(k % NU == 0, "whole filter rows per slice");
    __shared__ float sTu[UNIFORM ? NU : 1][UNIFORM ? PXC : 1], sGu[UNIFORM ? NU : 1][UNIFORM ? PXC : 1];
    const int c = tid & 15, pl = tid >> 4;
    const float fw = fuse_w16[c];
    const float dw = WITH_SIDE_OUT ? dsn_w16[c] : 0.f;
    float a = 0.f, b = 0.f, g = 0.f;
    // DIRECT (scale 0, one filter for all channels: 16 taps, the most pixels): thread (pixel, channel half) takes the whole
    // tap sum itself - no partial sums, no barriers - reads its 8 side values as two 16-byte loads and writes its 8
    // gradients (and 8 padding zeros) as 16-byte stores.  (One thread per (pixel, channel) stored 2 bytes per lane: 50-56 us
    // for five frames at the head of the weight-gradient stream, against ~75 MB of traffic.)
    constexpr bool DIRECT = UNIFORM && S == 0;
    __shared__ float red[4][3][16];
    if constexpr (DIRECT) {
        const int h8 = (tid & 1) * 8, pl2 = tid >> 1;
        float fw8[8], dw8[8], a8[8], b8[8], g1 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            fw8[e] = fuse_w16[h8 + e];
            dw8[e] = WITH_SIDE_OUT ? dsn_w16[h8 + e] : 0.f;
            a8[e] = 0.f;
            b8[e] = 0.f;
        }
        for (int p = pl2; p < NPX; p += 128) {
            const int il = p / TJ, jl = p - il * TJ;
            const int i = i0 + il, j = j0 + jl;
            if (i >= hs || j >= ws) continue;
            const float *wp = sW + il * f * WC + jl * f;
            const float *wp1 = sW1 + il * f * WC + jl * f;
            float T = 0.f, G = 0.f;
#pragma unroll
            for (int ky = 0; ky < k; ++ky)
#pragma unroll
                for (int kx = 0; kx < k; ++kx) {
                    T += sF[ky * k + kx] * wp[ky * WC + kx];
                    if (WITH_SIDE_OUT) G += sF1[ky * k + kx] * wp1[ky * WC + kx];
                }
            const int64_t pix = ((int64_t)n * hs + i) * ws + j;
            const float4 s0 = *reinterpret_cast<const float4 *>(side + pix * 16 + h8);
            const float4 s1 = *reinterpret_cast<const float4 *>(side + pix * 16 + h8 + 4);
            const float sv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
            float ds[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                ds[e] = fw8[e] * T + dw8[e] * G;
                a8[e] += sv[e] * T;
                if (WITH_SIDE_OUT) b8[e] += sv[e] * G;
            }
            if (WITH_SIDE_OUT && h8 == 0) g1 += G;
            *reinterpret_cast<uint4 *>(d_side + pix * 32 + h8) = pack8(ds);           // channels 8 h .. 8 h + 7
            *reinterpret_cast<uint4 *>(d_side + pix * 32 + 16 + h8) = make_uint4(0, 0, 0, 0);  // ... and their padding twins
        }
        // sums over the 32 pixel lanes of a wave (lanes of equal parity), then across the waves through `red`
#pragma unroll
        for (int o = 2; o < 64; o <<= 1) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                a8[e] += __shfl_xor(a8[e], o, 64);
                if (WITH_SIDE_OUT) b8[e] += __shfl_xor(b8[e], o, 64);
            }
            if (WITH_SIDE_OUT) g1 += __shfl_xor(g1, o, 64);
        }
        const int wave_d = tid >> 6, lane_d = tid & 63;
        if (lane_d < 2) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                red[wave_d][0][8 * lane_d + e] = a8[e];
                red[wave_d][1][8 * lane_d + e] = b8[e];
                red[wave_d][2][8 * lane_d + e] = (lane_d == 0 && e == 0) ? g1 : 0.f;
            }
        }
    }
#pragma unroll 1
    for (int p0 = 0; p0 < (DIRECT ? 0 : NPX); p0 += PXC) {
    if (p0) __syncthreads();  // the previous chunk's partial sums have been read
    if constexpr (UNIFORM) {
        const int cq = tid & 3, pl1 = (tid >> 2) % PXC, ts = tid / (4 * PXC), u = 4 * ts + cq;
        const int il = (p0 + pl1) / TJ, jl = (p0 + pl1) - il * TJ;
        const float *wp = sW + il * f * WC + jl * f;
        const float *wp1 = sW1 + il * f * WC + jl * f;
        float T1 = 0.f, G1 = 0.f;
#pragma unroll
        for (int ky = u * RU; ky < (u + 1) * RU; ++ky) {
            if constexpr (S >= 1) {
#pragma unroll
                for (int kx = 0; kx < k; kx += 4) {
                    const float4 w4 = *reinterpret_cast<const float4 *>(wp + ky * WC + kx);
                    const float4 h4 = *reinterpret_cast<const float4 *>(sF + ky * k + kx);
                    T1 += h4.x * w4.x + h4.y * w4.y + h4.z * w4.z + h4.w * w4.w;
                    if (WITH_SIDE_OUT) {
                        const float4 v4 = *reinterpret_cast<const float4 *>(wp1 + ky * WC + kx);
                        G1 += sF1[ky * k + kx] * v4.x + sF1[ky * k + kx + 1] * v4.y + sF1[ky * k + kx + 2] * v4.z +
                              sF1[ky * k + kx + 3] * v4.w;
                    }
                }
            } else {
#pragma unroll
                for (int kx = 0; kx < k; ++kx) {
                    T1 += sF[ky * k + kx] * wp[ky * WC + kx];
                    if (WITH_SIDE_OUT) G1 += sF1[ky * k + kx] * wp1[ky * WC + kx];
                }
            }
        }
        sTu[u][pl1] = T1;
        if (WITH_SIDE_OUT) sGu[u][pl1] = G1;
    } else {
        const int cq = tid & 3, pl1 = (tid >> 2) % PXC, ts = tid / (4 * PXC);
        const int il = (p0 + pl1) / TJ, jl = (p0 + pl1) - il * TJ;
        const float *wp = sW + il * f * WC + jl * f;
        const float *wp1 = sW1 + il * f * WC + jl * f;
        const float4 *sF4 = reinterpret_cast<const float4 *>(sF);
        float4 T4 = make_float4(0.f, 0.f, 0.f, 0.f);
        float G1 = 0.f;
        constexpr int KY = k / TS;
#pragma unroll 2
        for (int ky = ts * KY; ky < (ts + 1) * KY; ++ky) {
            if constexpr (S >= 1) {  // the window row of a pixel starts on a 16-byte boundary: four taps per read
#pragma unroll
                for (int kx = 0; kx < k; kx += 4) {
                    const float4 w4 = *reinterpret_cast<const float4 *>(wp + ky * WC + kx);
                    const float wv[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float4 f4 = sF4[(ky * k + kx + e) * 4 + cq];
                        T4.x += f4.x * wv[e]; T4.y += f4.y * wv[e]; T4.z += f4.z * wv[e]; T4.w += f4.w * wv[e];
                    }
                    if (WITH_SIDE_OUT && cq == 0) {
                        const float4 v4 = *reinterpret_cast<const float4 *>(wp1 + ky * WC + kx);
                        G1 += sF1[ky * k + kx] * v4.x + sF1[ky * k + kx + 1] * v4.y + sF1[ky * k + kx + 2] * v4.z +
                              sF1[ky * k + kx + 3] * v4.w;
                    }
                }
            } else {
#pragma unroll
                for (int kx = 0; kx < k; ++kx) {
                    const float wv = wp[ky * WC + kx];
                    const float4 f4 = sF4[(ky * k + kx) * 4 + cq];
                    T4.x += f4.x * wv; T4.y += f4.y * wv; T4.z += f4.z * wv; T4.w += f4.w * wv;
                    if (WITH_SIDE_OUT && cq == 0) G1 += sF1[ky * k + kx] * wp1[ky * WC + kx];
                }
            }
        }
        *reinterpret_cast<float4 *>(&sP[ts][pl1][4 * cq]) = T4;
        if (WITH_SIDE_OUT && cq == 0) sG[ts][pl1] = G1;
    }
    __syncthreads();
    // ---- phase 2: thread (pixel pl + 16 m of the chunk, channel c) adds the slices in order and finishes its element
    for (int pc = pl; pc < PXC; pc += 16) {
        const int p = p0 + pc;
        const int il = p / TJ, jl = p - il * TJ;
        const int i = i0 + il, j = j0 + jl;
        if (i >= hs || j >= ws) continue;
        float T, G;
        if constexpr (UNIFORM) {
            T = sTu[0][pc];
            G = WITH_SIDE_OUT ? sGu[0][pc] : 0.f;
#pragma unroll
            for (int u = 1; u < NU; ++u) {
                T += sTu[u][pc];
                if (WITH_SIDE_OUT) G += sGu[u][pc];
            }
        } else {
            T = sP[0][pc][c];
            G = WITH_SIDE_OUT ? sG[0][pc] : 0.f;
#pragma unroll
            for (int ts = 1; ts < TS; ++ts) {
                T += sP[ts][pc][c];
                if (WITH_SIDE_OUT) G += sG[ts][pc];
            }
        }
        const int64_t pix = ((int64_t)n * hs + i) * ws + j;
        const float sv = side[pix * 16 + c];
        const float ds = fw * T + dw * G;
        d_side[pix * 32 + c] = f2bf(ds);   // padded 32-channel bf16 NHWC image: channel c ...
        d_side[pix * 32 + 16 + c] = 0;     // ... and a zero in channel c+16
        a += sv * T;
        if (WITH_SIDE_OUT) {
            b += sv * G;
            if (c == 0) g += G;
        }
    }
    }  // chunks
    // block reduction over lanes with equal c: xor 16, 32 inside the wave, then LDS across waves
    if constexpr (!DIRECT) {
        a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
        b += __shfl_xor(b, 16, 64); b += __shfl_xor(b, 32, 64);
        g += __shfl_xor(g, 16, 64); g += __shfl_xor(g, 32, 64);
        const int wave = tid >> 6, lane = tid & 63;
        if (lane < 16) {
            red[wave][0][lane] = a;
            red[wave][1][lane] = b;
            red[wave][2][lane] = g;
        }
    }
    __syncthreads();
    if (tid < 48) {
        const int q = tid / 16, cc = tid % 16;
        const int blk = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        slabs[(int64_t)blk * 48 + tid] = (red[0][q][cc] + red[1][q][cc]) + (red[2][q][cc] + red[3][q][cc]);
    }
}

template <int S>
struct HeadBwdLaunch {
    static int blocks(int N, int hs, int ws) {
        return (int)(cdiv(ws, HeadTile<S>::TJ) * cdiv(hs, HeadTile<S>::TI) * N);
    }
    template <bool SO, bool UNI>
    static size_t lds_bytes() {
        constexpr int f = 2 << S, k = 4 << S;
        constexpr int WR = (HeadTile<S>::TI + 1) * f, WC = (HeadTile<S>::TJ + 1) * f;
        return sizeof(float) * (size_t)(k * k * (UNI ? 1 : 16) + WR * WC * (SO ? 2 : 1) + (SO ? k * k : 0));
    }
    template <bool SO, bool UNI>
    static int run(const float *side, const float *filt, const float *filt1, const float *fw16, const float *dw16,
                   const float *d_fused, const float *d_so, uint16_t *d_side, float *slabs, int N, int hs, int ws,
                   int top, int left, int H, int W, hipStream_t st) {
        const size_t lds = lds_bytes<SO, UNI>();
        auto kern = k_head_bwd_scale<S, SO, UNI>;
        if (lds > 64 * 1024)
            FOSVOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        dim3 grid((unsigned)cdiv(ws, HeadTile<S>::TJ), (unsigned)cdiv(hs, HeadTile<S>::TI), (unsigned)N);
        FOSVOS_PROF("k_head_bwd_scale", st, 0.0);
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, side, filt, filt1, fw16, dw16, d_fused, d_so, d_side, slabs, hs,
                           ws, top, left, H, W);
        FOSVOS_LAUNCH_CHECK();
        return FOSVOS_OK;
    }
};

// per-block partial sums of d_fused (for d_fuse_b)
__global__ __launch_bounds__(256) void k_sum_partials(const float *__restrict__ x, int64_t n, float *__restrict__ out) {
    float a = 0.f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) a += x[i];
    a = wave_sum(a);
    __shared__ float s[4];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = (s[0] + s[1]) + (s[2] + s[3]);
}

struct FinishArgs {
    const float *slabs[4];
    int n_slabs[4];
    const float *bias_partials;
    int n_bias;
    int accumulate;  // add into d_* instead of overwriting
};

// One block per scale (+ one for the bias): 16 row groups x 48 columns; group g sums slabs g, g+16, ...
// (fixed order, fp64), then the 16 group sums are added in order: deterministic.
__global__ __launch_bounds__(768) void k_head_finish(FinishArgs fa, float *__restrict__ d_fuse_w,
                                                      float *__restrict__ d_fuse_b, float *__restrict__ d_dsn_w,
                                                      float *__restrict__ d_dsn_b) {
    const int s = blockIdx.x;
    __shared__ double red[16][48];
    if (s == 4) {
        double acc = 0.0;
        for (int i = threadIdx.x; i < fa.n_bias; i += 768) acc += (double)fa.bias_partials[i];
        acc = wave_sum(acc);
        if ((threadIdx.x & 63) == 0) red[0][threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
            for (int w = 0; w < 12; ++w) t += red[0][w];
            d_fuse_b[0] = fa.accumulate ? d_fuse_b[0] + (float)t : (float)t;
        }
        return;
    }
    const int col = threadIdx.x % 48, g = threadIdx.x / 48;
    double acc = 0.0;
    // eight slabs in flight per thread, added in the same order as one at a time (a rolled loop paid one memory round trip
    // per slab: 131 in a row at scale 0 with five frames - the whole 50 us of this kernel, at the head of the
    // weight-gradient stream)
    const int n_slabs = fa.n_slabs[s];
    for (int b0 = g; b0 < n_slabs; b0 += 16 * 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fa.slabs[s][(int64_t)min(b0 + 16 * j, n_slabs - 1) * 48 + col];
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (b0 + 16 * j < n_slabs) acc += (double)v[j];
    }
    red[g][col] = acc;
    __syncthreads();
    if (g != 0) return;
    double t = 0.0;
    for (int k = 0; k < 16; ++k) t += red[k][col];
    const int q = col / 16, c = col % 16;
    const float tf = (float)t;
    if (q == 0) d_fuse_w[16 * s + c] = fa.accumulate ? d_fuse_w[16 * s + c] + tf : tf;
    if (q == 1 && d_dsn_w) d_dsn_w[16 * s + c] = fa.accumulate ? d_dsn_w[16 * s + c] + tf : tf;
    if (q == 2 && d_dsn_b && c == 0) d_dsn_b[s] = fa.accumulate ? d_dsn_b[s] + tf : tf;
}

constexpr int kBiasBlocks = 512;
}  // namespace

extern "C" int fosvos_head_fwd(const float *const side[4], const int hs[4], const int ws[4],
                               const float *const filt[4], const float *const filt1[4], const float *dsn_w,
                               const float *dsn_b, const float *fuse_w, const float *fuse_b, float *fused,
                               float *const side_out[4], int N, int H, int W, int filt_uniform, int device, void *stream) {
    FOSVOS_REQUIRE(side && hs && ws && filt && fuse_w && fuse_b && fused, FOSVOS_E_ARG, "head_fwd: null pointer");
    FOSVOS_REQUIRE(N > 0 && H > 0 && W > 0, FOSVOS_E_SHAPE, "head_fwd: bad shape N=%d H=%d W=%d", N, H, W);
    const bool with_so = side_out && side_out[0];
    HeadPtrs p;
    for (int s = 0; s < 4; ++s) {
        FOSVOS_REQUIRE(side[s] && filt[s], FOSVOS_E_ARG, "head_fwd: null side/filter pointer at scale %d", s);
        p.side[s] = side[s];
        p.filt[s] = filt[s];
        p.filt1[s] = (with_so && filt1) ? filt1[s] : nullptr;
        if (with_so)
            FOSVOS_REQUIRE(side_out[s] && filt1 && filt1[s] && dsn_w && dsn_b, FOSVOS_E_ARG,
                           "head_fwd: side outputs need all four buffers, filt1, dsn_w and dsn_b");
    }
    HeadGeom g;
    FOSVOS_REQUIRE(make_geom(hs, ws, H, W, g), FOSVOS_E_SHAPE,
                   "head_fwd: side map sizes (%d,%d),(%d,%d),(%d,%d),(%d,%d) do not cover %dx%d", hs[0], ws[0], hs[1],
                   ws[1], hs[2], ws[2], hs[3], ws[3], H, W);
    FOSVOS_ENTER(device);
    FOSVOS_REQUIRE(N <= 65535, FOSVOS_E_SHAPE, "head_fwd: batch %d", N);
    const dim3 grid((unsigned)cdiv(W, HF_TX), (unsigned)cdiv(H, HF_TY), (unsigned)N);
    FOSVOS_PROF("k_head_fwd", stream, 0.0);
    if (with_so)
        hipLaunchKernelGGL(k_head_fwd<true>, grid, dim3(256), 0, (hipStream_t)stream, p, g, dsn_w, dsn_b, fuse_w, fuse_b,
                           fused, side_out[0], side_out[1], side_out[2], side_out[3], H, W, filt_uniform & 15);
    else
        hipLaunchKernelGGL(k_head_fwd<false>, grid, dim3(256), 0, (hipStream_t)stream, p, g, dsn_w, dsn_b, fuse_w, fuse_b,
                           fused, (float *)nullptr, (float *)nullptr, (float *)nullptr, (float *)nullptr, H, W,
                           filt_uniform & 15);
    FOSVOS_LAUNCH_CHECK();
    return FOSVOS_OK;
}

static inline size_t head_slab_floats(int N, int H, int W, size_t off[5]) {
    // side sizes are ceil-halved four times; take the worst case the model can produce
    size_t total = 0;
    int h = H, w = W;
    for (int s = 0; s < 4; ++s) {
        h = (h + 1) / 2;
        w = (w + 1) / 2;
        off[s] = total;
        const int nb = s == 0 ? HeadBwdLaunch<0>::blocks(N, h, w) : s == 1 ? HeadBwdLaunch<1>::blocks(N, h, w)
                     : s == 2 ? HeadBwdLaunch<2>::blocks(N, h, w) : HeadBwdLaunch<3>::blocks(N, h, w);
        total += (size_t)nb * 48;
    }
    off[4] = total;
    total += kBiasBlocks;
    return total;
}

extern "C" size_t fosvos_head_bwd_workspace_bytes(int N, int H, int W) {
    size_t off[5];
    return head_slab_floats(N, H, W, off) * sizeof(float);
}

extern "C" int fosvos_head_bwd(const float *const side[4], const int hs[4], const int ws[4],
                               const float *const filt[4], const float *const filt1[4], const float *dsn_w,
                               const float *fuse_w, const float *d_fused, const float *const d_side_out[4],
                               uint16_t *const d_side[4], float *d_fuse_w, float *d_fuse_b, float *d_dsn_w,
                               float *d_dsn_b, int N, int H, int W, int filt_uniform, void *workspace,
                               size_t workspace_bytes, int device, void *stream) {
    const HeadBwdArgs a{side, hs, ws, filt, filt1, dsn_w, fuse_w, d_fused, d_side_out, d_side, d_fuse_w, d_fuse_b,
                        d_dsn_w, d_dsn_b, N, H, W, 0, workspace, workspace_bytes, device, filt_uniform & 15};
    if (int rc = head_bwd_check(a)) return rc;
    for (int s = 0; s < 4; ++s)
        if (int rc = head_bwd_scale(a, s, stream)) return rc;
    return head_bwd_finish(a, stream);
}

namespace {
inline bool head_with_so(const HeadBwdArgs &a) { return a.d_side_out && a.d_side_out[0]; }
}

int fosvos::head_bwd_check(const HeadBwdArgs &a) {
    FOSVOS_REQUIRE(a.side && a.hs && a.ws && a.filt && a.fuse_w && a.d_side && a.d_fuse_w && a.d_fuse_b && a.workspace,
                   FOSVOS_E_ARG, "head_bwd: null pointer");
    FOSVOS_REQUIRE(a.N > 0 && a.H > 0 && a.W > 0, FOSVOS_E_SHAPE, "head_bwd: bad shape N=%d H=%d W=%d", a.N, a.H, a.W);
    const bool with_so = head_with_so(a);
    FOSVOS_REQUIRE(a.d_fused || with_so, FOSVOS_E_ARG, "head_bwd: no upstream gradient given");
    HeadGeom g;
    FOSVOS_REQUIRE(make_geom(a.hs, a.ws, a.H, a.W, g), FOSVOS_E_SHAPE, "head_bwd: side map sizes do not cover %dx%d", a.H,
                   a.W);
    size_t off[5];
    const size_t need = head_slab_floats(a.N, a.H, a.W, off) * sizeof(float);
    FOSVOS_REQUIRE(a.workspace_bytes >= need, FOSVOS_E_WORKSPACE, "head_bwd: workspace %zu < %zu", a.workspace_bytes, need);
    for (int s = 0; s < 4; ++s) {
        FOSVOS_REQUIRE(a.side[s] && a.filt[s] && a.d_side[s], FOSVOS_E_ARG, "head_bwd: null pointer at scale %d", s);
        FOSVOS_REQUIRE(a.hs[s] <= (((a.H - 1) >> (s + 1)) + 1) && a.ws[s] <= (((a.W - 1) >> (s + 1)) + 1), FOSVOS_E_SHAPE,
                       "head_bwd: side map %d larger than the ceil-pooled size", s);
        if (with_so)
            FOSVOS_REQUIRE(a.d_side_out[s] && a.filt1 && a.filt1[s] && a.dsn_w && a.d_dsn_w && a.d_dsn_b, FOSVOS_E_ARG,
                           "head_bwd: side-output gradients need all four, filt1, dsn_w, d_dsn_w and d_dsn_b");
    }
    return FOSVOS_OK;
}

// d_side[s] and the scale's partial sums for the fuse / score_dsn weight gradients (slabs in the workspace)
int fosvos::head_bwd_scale(const HeadBwdArgs &a, int s, void *stream) {
    FOSVOS_REQUIRE(s >= 0 && s < 4, FOSVOS_E_ARG, "head_bwd_scale: scale %d", s);
    const bool with_so = head_with_so(a);
    HeadGeom g;
    make_geom(a.hs, a.ws, a.H, a.W, g);
    size_t off[5];
    head_slab_floats(a.N, a.H, a.W, off);
    FOSVOS_ENTER(a.device);
    hipStream_t st = (hipStream_t)stream;
    float *wsf = reinterpret_cast<float *>(a.workspace);
    const float *f1 = with_so ? a.filt1[s] : nullptr;
    const float *dw16 = with_so ? a.dsn_w + 16 * s : nullptr;
    const float *dso = with_so ? a.d_side_out[s] : nullptr;
    const bool uni = (a.filt_uniform >> s) & 1;
#define FOSVOS_HEAD_RUN(SS, SO_, UNI_)                                                                                \
    HeadBwdLaunch<SS>::run<SO_, UNI_>(a.side[s], a.filt[s], f1, a.fuse_w + 16 * s, dw16, a.d_fused, dso, a.d_side[s],   \
                                      wsf + off[s], a.N, a.hs[s], a.ws[s], g.top[s], g.left[s], a.H, a.W, st)
#define FOSVOS_HEAD_CASE(SS)                                                                                          \
    case SS:                                                                                                          \
        return with_so ? (uni ? FOSVOS_HEAD_RUN(SS, true, true) : FOSVOS_HEAD_RUN(SS, true, false))                   \
                       : (uni ? FOSVOS_HEAD_RUN(SS, false, true) : FOSVOS_HEAD_RUN(SS, false, false));
    switch (s) {
        FOSVOS_HEAD_CASE(0)
        FOSVOS_HEAD_CASE(1)
        FOSVOS_HEAD_CASE(2)
        FOSVOS_HEAD_CASE(3)
    }
#undef FOSVOS_HEAD_CASE
#undef FOSVOS_HEAD_RUN
    return FOSVOS_OK;
}

// d_fuse_w, d_fuse_b, d_dsn_w, d_dsn_b from the four scales' slabs (fixed-order sums)
int fosvos::head_bwd_finish(const HeadBwdArgs &a, void *stream) {
    const bool with_so = head_with_so(a);
    size_t off[5];
    head_slab_floats(a.N, a.H, a.W, off);
    FOSVOS_ENTER(a.device);
    hipStream_t st = (hipStream_t)stream;
    float *wsf = reinterpret_cast<float *>(a.workspace);
    FinishArgs fa;
    for (int s = 0; s < 4; ++s) {
        fa.slabs[s] = wsf + off[s];
        fa.n_slabs[s] = s == 0 ? HeadBwdLaunch<0>::blocks(a.N, a.hs[s], a.ws[s])
                      : s == 1 ? HeadBwdLaunch<1>::blocks(a.N, a.hs[s], a.ws[s])
                      : s == 2 ? HeadBwdLaunch<2>::blocks(a.N, a.hs[s], a.ws[s])
                               : HeadBwdLaunch<3>::blocks(a.N, a.hs[s], a.ws[s]);
    }
    fa.bias_partials = wsf + off[4];
    fa.n_bias = 0;
    fa.accumulate = a.accumulate;
    if (a.d_fused) {
        fa.n_bias = kBiasBlocks;
        FOSVOS_PROF("k_sum_partials", st, 0.0);
        hipLaunchKernelGGL(k_sum_partials, dim3(kBiasBlocks), dim3(256), 0, st, a.d_fused, (int64_t)a.N * a.H * a.W,
                           wsf + off[4]);
        FOSVOS_LAUNCH_CHECK();
    }
    FOSVOS_PROF("k_head_finish", st, 0.0);
    hipLaunchKernelGGL(k_head_finish, dim3(5), dim3(768), 0, st, fa, a.d_fuse_w, a.d_fuse_b,
                       with_so ? a.d_dsn_w : nullptr, with_so ? a.d_dsn_b : nullptr);
    FOSVOS_LAUNCH_CHECK();
    return FOSVOS_OK;
}
