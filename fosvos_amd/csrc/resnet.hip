// Inference kernels of the thin-channel ResNet family (OSVOS_RESNET: src/networks/osvos_resnet.py:15-150, and the
// filter-pruned nets src/prune.py:297-481 derives from it).  Eval-mode BatchNorm is folded into the conv
// that feeds it when the weights are packed, the residual add and the ReLU ride in the conv epilogue.
//
// These layers are thin (8..128 channels once scale_down_exponent or pruning has been applied) and their channel counts
// are arbitrary, so they do not go through the MFMA implicit GEMM (Ci % 32, Co % 64).  The contraction runs on the vector
// ALU instead, two bf16 products per lane and clock (v_dot2c_f32_bf16, fp32 accumulate):
//   * a thread owns ONE output pixel and COB output channels (8..64 accumulators);
//   * activations are bf16 NHWC with the channel count padded to a multiple of 8, so a tap of 8 input channels is one
//     16-byte load per thread, straight from global memory (neighbouring threads share taps through L1; the maps of a
//     thin net fit L2);
//   * weights are the same for every lane of a wave: they are indexed by block and loop counters only, so hipcc fetches
//     them with scalar loads and feeds them to the dot instruction as its SGPR operand - no LDS, no broadcast reads.
// Padded channels hold exact zeros everywhere (zero weights and bias), so no kernel ever masks a channel.
#include "common.hpp"


using namespace fosvos;

namespace {

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float dot2(uint32_t a, uint32_t w, float acc) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, w), acc, false);
}

// ------------------------------------------------------------------------------------------ weight packing
// OIHW fp32 -> [ci chunk of 8][tap][ci pair][Cop] dwords (low half = even input channel), scaled by the BatchNorm that
// follows the conv; bias_out[co] = bn_bias - mean * s (+ conv_bias * s), s = bn_weight / sqrt(var + eps).
__global__ void k_pack_conv2d_bn(const float *__restrict__ w, int Co, int Ci, int k, const float *__restrict__ conv_bias,
                                 const float *__restrict__ bn_w, const float *__restrict__ bn_b,
                                 const float *__restrict__ bn_m, const float *__restrict__ bn_v, float eps,
                                 uint32_t *__restrict__ wp, float *__restrict__ bias_out, int Cop, int cic, int64_t total,
                                 int bias_len) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < bias_len) {
        float b = 0.f;
        if (i < Co) {
            const float s = bn_w ? bn_w[i] / sqrtf(bn_v[i] + eps) : 1.f;
            b = (bn_w ? bn_b[i] - bn_m[i] * s : 0.f) + (conv_bias ? conv_bias[i] * s : 0.f);
        }
        bias_out[i] = b;
    }
    if (i >= total) return;
    const int T = k * k;
    const int co = (int)(i % Cop);
    int64_t r = i / Cop;
    const int p = (int)(r & 3);
    r >>= 2;
    const int t = (int)(r % T);
    const int c = (int)(r / T);
    uint32_t v = 0;
    if (c < cic && co < Co) {
        const float s = bn_w ? bn_w[co] / sqrtf(bn_v[co] + eps) : 1.f;
        const int ci = c * 8 + p * 2;
        const float lo = ci < Ci ? w[((int64_t)co * Ci + ci) * T + t] * s : 0.f;
        const float hi = ci + 1 < Ci ? w[((int64_t)co * Ci + ci + 1) * T + t] * s : 0.f;
        v = pack2bf(lo, hi);
    }
    wp[i] = v;
}

// The fold alone, fp32 OIHW -> fp32 OIHW (layers that continue through the MFMA path's own packing).
__global__ void k_fold_conv_bn(const float *__restrict__ w, int Co, int per_co, const float *__restrict__ conv_bias,
                               const float *__restrict__ bn_w, const float *__restrict__ bn_b,
                               const float *__restrict__ bn_m, const float *__restrict__ bn_v, float eps,
                               float *__restrict__ wo, float *__restrict__ bias_out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < Co) {
        const float s = bn_w ? bn_w[i] / sqrtf(bn_v[i] + eps) : 1.f;
        bias_out[i] = (bn_w ? bn_b[i] - bn_m[i] * s : 0.f) + (conv_bias ? conv_bias[i] * s : 0.f);
    }
    if (i >= (int64_t)Co * per_co) return;
    const int co = (int)(i / per_co);
    wo[i] = w[i] * (bn_w ? bn_w[co] / sqrtf(bn_v[co] + eps) : 1.f);
}

// First layer: [Co][3][7][7] fp32 -> two images.  (1) fp32 [tap * 3 + ci][Cop] for the vector-ALU kernel; (2) bf16 MFMA
// fragments [k-step 6][k-group 4][channel NC][8] over K = 8 rows x 24 (a filter row is kx-major, channel-minor: 21 values +
// 3 zeros; the 8th row is zeros), at float offset 147 * Cop + 64 of the same buffer, NC = Cop rounded up to 16.
__host__ __device__ constexpr int f7_nc(int Cop) { return (Cop + 15) / 16 * 16; }
__global__ void k_pack_conv7x7_bn(const float *__restrict__ w, int Co, const float *__restrict__ bn_w,
                                  const float *__restrict__ bn_b, const float *__restrict__ bn_m,
                                  const float *__restrict__ bn_v, float eps, float *__restrict__ wp,
                                  float *__restrict__ bias_out, int Cop, int total, int bias_len) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < bias_len) {
        float b = 0.f;
        if (i < Co && bn_w) b = bn_b[i] - bn_m[i] * bn_w[i] / sqrtf(bn_v[i] + eps);
        bias_out[i] = b;
    }
    const int NC = f7_nc(Cop);
    if (i < 6 * 4 * NC * 8) {
        const int e = i & 7, n = (i >> 3) % NC, kgs = (i >> 3) / NC;  // kgs = k-step * 4 + k-group
        const int k = kgs * 8 + e, ky = k / 24, j = k % 24;
        float v = 0.f;
        if (ky < 7 && j < 21 && n < Co) {
            const float s = bn_w ? bn_w[n] / sqrtf(bn_v[n] + eps) : 1.f;
            v = w[((int64_t)n * 3 + j % 3) * 49 + ky * 7 + j / 3] * s;
        }
        reinterpret_cast<uint16_t *>(wp + 147 * Cop + 64)[i] = f2bf(v);
    }
    if (i >= total) return;
    const int co = i % Cop, r = i / Cop;  // r = tap * 3 + ci
    float v = 0.f;
    if (r < 147 && co < Co) {
        const int ci = r % 3, t = r / 3;
        const float s = bn_w ? bn_w[co] / sqrtf(bn_v[co] + eps) : 1.f;
        v = w[((int64_t)co * 3 + ci) * 49 + t] * s;
    }
    wp[i] = v;
}

// ------------------------------------------------------------------------------------------ k x k conv, stride S
// blockDim = (threads, KS): threadIdx.x picks the pixel, threadIdx.y a slice of the input-channel chunks (c = y, y + KS,
// ...).  KS > 1 is for the deep layers, whose few thousand pixels cannot fill 1024 SIMDs on their own: the slices'
// partial sums meet in LDS and slice 0 runs the epilogue.
template <int KS, int S, int COB, bool OUT_F32, bool SLICED>
__global__ __launch_bounds__(SLICED ? 512 : 256) void k_conv2d(const uint4 *__restrict__ x, const uint32_t *__restrict__ wp,
                                                const float *__restrict__ bias, const uint4 *__restrict__ addend,
                                                void *__restrict__ y, int N, int H, int W, int Ho, int Wo, int cic,
                                                int Cop, int relu) {
    extern __shared__ float red[];  // [slice - 1][COB][blockDim.x] when blockDim.y > 1
    constexpr int P = KS / 2, T = KS * KS;
    const int64_t npix = (int64_t)N * Ho * Wo;
    const int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = pix < npix;
    const int64_t pc = live ? pix : npix - 1;
    const int ox = (int)(pc % Wo), oy = (int)((pc / Wo) % Ho), n = (int)(pc / ((int64_t)Wo * Ho));
    const int cb = blockIdx.y;
    // (a wave never spans two slices - blockDim.x is a multiple of 64 - so the slice index is wave-uniform; saying so
    // keeps the weight addresses scalar)
    const int slice = SLICED ? __builtin_amdgcn_readfirstlane(threadIdx.y) : 0, slices = SLICED ? (int)blockDim.y : 1;

    int off[T];
    bool ok[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int iy = oy * S + t / KS - P, ix = ox * S + t % KS - P;
        ok[t] = iy >= 0 && iy < H && ix >= 0 && ix < W;
        off[t] = ((n * H + min(max(iy, 0), H - 1)) * W + min(max(ix, 0), W - 1)) * cic;
    }
    float acc[COB];
#pragma unroll
    for (int j = 0; j < COB; ++j) acc[j] = 0.f;

    const uint32_t *wb = wp + cb * COB;
    // SLICED launches run few waves per SIMD, so the next chunk's taps are fetched while this one is multiplied; the
    // others hide the latency behind their neighbours and keep the registers (4 waves per SIMD instead of 2).
    uint4 a[T];
    if (SLICED) {
#pragma unroll
        for (int t = 0; t < T; ++t) a[t] = x[off[t] + slice];
    }
    for (int c = slice; c < cic; c += slices) {
        uint4 an[T];
        if (SLICED) {
            const int cn = min(c + slices, cic - 1);
#pragma unroll
            for (int t = 0; t < T; ++t) an[t] = x[off[t] + cn];
        } else {
#pragma unroll
            for (int t = 0; t < T; ++t) {
                a[t] = x[off[t] + c];
                if (!ok[t]) a[t] = make_uint4(0, 0, 0, 0);
            }
        }
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const uint32_t *wt = wb + (int64_t)((c * T + t) * 4) * Cop;  // uniform across the wave: scalar loads
            const uint4 at = (SLICED && !ok[t]) ? make_uint4(0, 0, 0, 0) : a[t];
            const uint32_t av[4] = {at.x, at.y, at.z, at.w};
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int j = 0; j < COB; ++j) acc[j] = dot2(av[p], wt[p * Cop + j], acc[j]);
        }
        if (SLICED) {
#pragma unroll
            for (int t = 0; t < T; ++t) a[t] = an[t];
        }
    }
    if (SLICED) {
        const int bx = blockDim.x;
        if (slice > 0) {
#pragma unroll
            for (int j = 0; j < COB; ++j) red[((slice - 1) * COB + j) * bx + threadIdx.x] = acc[j];
        }
        __syncthreads();
        if (slice > 0) return;
        for (int q = 0; q < slices - 1; ++q)
#pragma unroll
            for (int j = 0; j < COB; ++j) acc[j] += red[(q * COB + j) * bx + threadIdx.x];
    }
    if (!live) return;
    const int cop8 = Cop >> 3;
#pragma unroll
    for (int g = 0; g < COB / 8; ++g) {
        const int co8 = cb * (COB / 8) + g;
        if (co8 >= cop8) break;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = acc[g * 8 + e] + bias[cb * COB + g * 8 + e];
        if (addend) {
            float r[8];
            unpack8(addend[pix * cop8 + co8], r);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += r[e];
        }
        if (relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (OUT_F32) {
            float4 *o = reinterpret_cast<float4 *>(y) + (pix * cop8 + co8) * 2;
            o[0] = make_float4(v[0], v[1], v[2], v[3]);
            o[1] = make_float4(v[4], v[5], v[6], v[7]);
        } else {
            reinterpret_cast<uint4 *>(y)[pix * cop8 + co8] = pack8(v);
        }
    }
}

// ------------------------------------------------------------------------------------------ 7x7 stride-2 first layer
// frame fp32 NCHW [N,3,H,W] -> bf16 NHWC [N,Ho,Wo,Cop]; fp32 multiply-add with the weights as scalar operands.  A
// workgroup owns 8 x 32 output pixels and stages their 21 x 69 x 3 input patch in LDS (coalesced row reads, zeros
// outside the frame); the 147 taps of a thread then come from LDS (odd row pitch: the stride-2 reads of a wave's two
// half-rows fall on distinct banks) instead of 147 half-used global loads.
constexpr int F7_TH = 8, F7_TW = 32, F7_PH = F7_TH * 2 + 5, F7_PW = F7_TW * 2 + 5;

template <int COB>
__global__ __launch_bounds__(256) void k_conv7x7s2_first(const float *__restrict__ frame, const float *__restrict__ wp,
                                                         const float *__restrict__ bias, uint4 *__restrict__ y, int H,
                                                         int W, int Ho, int Wo, int tiles_x, int Cop, int relu) {
    __shared__ float sIn[3 * F7_PH * F7_PW];
    const int tile = blockIdx.x, cb = blockIdx.y, n = blockIdx.z;
    const int oy0 = (tile / tiles_x) * F7_TH, ox0 = (tile % tiles_x) * F7_TW;
    const int64_t plane = (int64_t)H * W;
    const float *f0 = frame + (int64_t)n * 3 * plane;
    for (int idx = threadIdx.x; idx < 3 * F7_PH * F7_PW; idx += 256) {
        const int ci = idx / (F7_PH * F7_PW), r = (idx / F7_PW) % F7_PH, c = idx % F7_PW;
        const int iy = oy0 * 2 - 3 + r, ix = ox0 * 2 - 3 + c;
        sIn[idx] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? f0[ci * plane + (int64_t)iy * W + ix] : 0.f;
    }
    __syncthreads();
    const int lx = threadIdx.x % F7_TW, ly = threadIdx.x / F7_TW;
    float acc[COB];
#pragma unroll
    for (int j = 0; j < COB; ++j) acc[j] = 0.f;
#pragma unroll 1
    for (int ky = 0; ky < 7; ++ky) {
        const float *row = sIn + (ly * 2 + ky) * F7_PW + lx * 2;
        float v[21];
#pragma unroll
        for (int kx = 0; kx < 7; ++kx)
#pragma unroll
            for (int ci = 0; ci < 3; ++ci) v[kx * 3 + ci] = row[ci * (F7_PH * F7_PW) + kx];
        const float *wr = wp + (int64_t)(ky * 21) * Cop + cb * COB;  // uniform: scalar loads
#pragma unroll
        for (int q = 0; q < 21; ++q)
#pragma unroll
            for (int j = 0; j < COB; ++j) acc[j] = fmaf(v[q], wr[q * Cop + j], acc[j]);
    }
    const int oy = oy0 + ly, ox = ox0 + lx;
    if (oy >= Ho || ox >= Wo) return;
    const int64_t pix = ((int64_t)n * Ho + oy) * Wo + ox;
    const int cop8 = Cop >> 3;
#pragma unroll
    for (int g = 0; g < COB / 8; ++g) {
        const int co8 = cb * (COB / 8) + g;
        if (co8 >= cop8) break;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            v[e] = acc[g * 8 + e] + bias[cb * COB + g * 8 + e];
            if (relu) v[e] = fmaxf(v[e], 0.f);
        }
        y[pix * cop8 + co8] = pack8(v);
    }
}

// The same layer on the matrix cores: frame and weights as bf16, K = 7 x 7 x 3 laid out as 8 rows x 24 (192 = 6 steps of
// v_mfma_f32_16x16x32_bf16).  The patch is staged in LDS as bf16, pixel-major with the 3 channels interleaved, so the 21
// values one output pixel needs from a patch row are contiguous and a lane's 8 consecutive k never leave a row; weights are
// the row operand (a lane ends up with 4 consecutive channels of one pixel: an 8-byte store), all 6 x NFB weight fragments
// stay in registers.  A workgroup owns 8 x 32 output pixels x all channels (Cop <= 64), a wave 2 rows of them.
// Walks the [3 channels][rows][69 columns] input patch in steps of 256 elements without a division per element (the
// decomposition of a flat index by constant divisions cost ~50 vector instructions per element: 1.2e7 per launch, more than
// anything else in these kernels).  256 = 3 * 69 + 49.
struct PatchWalk {
    int ci, r, c;
    __device__ __forceinline__ void start(int idx, int rows) {
        ci = idx / (rows * F7_PW);
        const int rem = idx - ci * rows * F7_PW;
        r = rem / F7_PW;
        c = rem - r * F7_PW;
    }
    __device__ __forceinline__ void step256(int rows) {
        c += 49;
        r += 3;
        if (c >= F7_PW) {
            c -= F7_PW;
            r += 1;
        }
        if (r >= rows) {
            r -= rows;
            ci += 1;
        }
    }
};

constexpr int F7_PITCH = 208;  // bf16 per patch row: 69 pixels x 3 channels = 207, +1

// Stage a [3][ROWS][69] patch of the fp32 NCHW frame (top-left input pixel (y_base, x_base), zeros outside the frame) as
// bf16 [row][69 x 3 interleaved] in LDS.  Frames whose width is a multiple of 4 take 16-byte loads (an aligned group of 4
// columns is inside the frame or outside it as a whole): 19 groups per patch row instead of 69 scalar loads, and the
// address arithmetic - most of these kernels' vector instructions - once per 4 elements.
template <int ROWS>
__device__ __forceinline__ bool stage_patch_vec4(const float *__restrict__ f0, int64_t plane, int y_base, int x_base, int H,
                                                 int W, uint16_t *sIn, int tid) {
    if ((W & 3) || (reinterpret_cast<uintptr_t>(f0) & 15)) return false;
    const int xa = (x_base >= 0 ? x_base : x_base - 3) / 4 * 4;  // floor to a multiple of 4
    constexpr int G = 19, ITEMS = 3 * ROWS * G;
    for (int item = tid; item < ITEMS; item += 256) {
        const int rp = item / G, g = item - rp * G;
        const int ci = rp / ROWS, r = rp - ci * ROWS;
        const int iy = y_base + r, x0 = xa + 4 * g;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (iy >= 0 && iy < H && x0 >= 0 && x0 < W) v = *reinterpret_cast<const float4 *>(f0 + ci * plane + (int64_t)iy * W + x0);
        const float e[4] = {v.x, v.y, v.z, v.w};
        const int c0 = x0 - x_base;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = c0 + q;
            if (c >= 0 && c < F7_PW) sIn[r * F7_PITCH + c * 3 + ci] = f2bf(e[q]);
        }
    }
    return true;
}


template <int NFB>
__global__ __launch_bounds__(256) void k_conv7x7s2_first_mfma(const float *__restrict__ frame, const uint4 *__restrict__ wimg,
                                                              const float *__restrict__ bias, uint16_t *__restrict__ y,
                                                              int H, int W, int Ho, int Wo, int tiles_x, int Cop, int relu) {
    constexpr int ROWS = F7_PH + 1, NC = NFB * 16;
    __shared__ __attribute__((aligned(16))) uint16_t sIn[ROWS * F7_PITCH + 8];
    const int tile = blockIdx.x, n = blockIdx.z, tid = threadIdx.x;
    const int oy0 = (tile / tiles_x) * F7_TH, ox0 = (tile % tiles_x) * F7_TW;
    const int64_t plane = (int64_t)H * W;
    const float *f0 = frame + (int64_t)n * 3 * plane;
    // patch loads in batches of 6 from clamped addresses (zero padding applied as a select): rolled, with the load under
    // a bounds branch, every element waited out its own trip to memory - 17 in a row per thread
    constexpr int F7_N = 3 * F7_PH * F7_PW, F7_B = 6;
    PatchWalk pw;
    pw.start(tid, F7_PH);
    const bool staged = stage_patch_vec4<F7_PH>(f0, plane, oy0 * 2 - 3, ox0 * 2 - 3, H, W, sIn, tid);
    for (int base = tid; !staged && base < F7_N; base += 256 * F7_B) {
        float v[F7_B];
        int dst[F7_B];
#pragma unroll
        for (int u = 0; u < F7_B; ++u) {
            const int ci = min(pw.ci, 2), r = pw.r, c = pw.c;
            const int iy = oy0 * 2 - 3 + r, ix = ox0 * 2 - 3 + c;
            const float t = f0[ci * plane + (int64_t)min(max(iy, 0), H - 1) * W + min(max(ix, 0), W - 1)];
            v[u] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? t : 0.f;
            dst[u] = pw.ci < 3 ? r * F7_PITCH + c * 3 + ci : -1;
            pw.step256(F7_PH);
        }
#pragma unroll
        for (int u = 0; u < F7_B; ++u)
            if (dst[u] >= 0) sIn[dst[u]] = f2bf(v[u]);
    }
    // what only zero weights ever multiply must still be finite: the pad column of every row, the 8th filter row's patch
    // row and the few elements a last lane reads past it
    for (int idx = tid; idx < F7_PITCH + 8 + ROWS; idx += 256) {
        if (idx < F7_PITCH + 8) sIn[F7_PH * F7_PITCH + idx] = 0;
        else sIn[(idx - F7_PITCH - 8) * F7_PITCH + F7_PITCH - 1] = 0;
    }
    const int wave = tid >> 6, lane = tid & 63, l16 = lane & 15, kg = lane >> 4;
    bf16x8 wf[6][NFB];
#pragma unroll
    for (int ks = 0; ks < 6; ++ks)
#pragma unroll
        for (int nf = 0; nf < NFB; ++nf) wf[ks][nf] = __builtin_bit_cast(bf16x8, wimg[(ks * 4 + kg) * NC + nf * 16 + l16]);
    f32x4 acc[4][NFB];
#pragma unroll
    for (int mf = 0; mf < 4; ++mf)
#pragma unroll
        for (int nf = 0; nf < NFB; ++nf) acc[mf][nf] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) {
        const int k0 = ks * 32 + kg * 8, ky = k0 / 24, j0 = k0 % 24;
#pragma unroll
        for (int mf = 0; mf < 4; ++mf) {
            const int ly = wave * 2 + (mf >> 1), lx = (mf & 1) * 16 + l16;
            const uint32_t *p = reinterpret_cast<const uint32_t *>(sIn + (2 * ly + ky) * F7_PITCH + 6 * lx + j0);
            const uint4 av = make_uint4(p[0], p[1], p[2], p[3]);
            const bf16x8 a = __builtin_bit_cast(bf16x8, av);
#pragma unroll
            for (int nf = 0; nf < NFB; ++nf)
                acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][nf], a, acc[mf][nf], 0, 0, 0);
        }
    }
#pragma unroll
    for (int mf = 0; mf < 4; ++mf) {
        const int oy = oy0 + wave * 2 + (mf >> 1), ox = ox0 + (mf & 1) * 16 + l16;
        if (oy >= Ho || ox >= Wo) continue;
        uint16_t *yp = y + (((int64_t)n * Ho + oy) * Wo + ox) * Cop;
#pragma unroll
        for (int nf = 0; nf < NFB; ++nf) {
            const int ch0 = nf * 16 + kg * 4;
            if (ch0 >= Cop) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = acc[mf][nf][r] + bias[ch0 + r];
                if (relu) v[r] = fmaxf(v[r], 0.f);
            }
            *reinterpret_cast<uint2 *>(yp + ch0) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
        }
    }
}

// First layer AND its 3x3 stride-2 max pool in one launch: the 540 x 960 conv map of a 1080p frame is never written (33-66
// MB out and back in) - a workgroup owns 8 x 15 POOLED pixels, computes the 17 x 31 conv pixels under them (18 x 32 are
// computed: 36 fragments of 16, 9 per wave) exactly as k_conv7x7s2_first_mfma does (same K order, so the same bits), keeps
// them in LDS as bf16 after bias + ReLU, and pools from there.  Conv positions outside the map count as 0, which is
// neutral under max because every window holds at least one real, ReLU'd (>= 0) value.
constexpr int FP_PH = 8, FP_PW = 15, FP_CR = 2 * FP_PH + 2, FP_CC = 32, FP_ROWS = 2 * (FP_CR - 1) + 7;  // 18 x 32 conv, 41 rows

template <int NFB>
__global__ __launch_bounds__(256) void k_conv7x7s2_pool_first_mfma(const float *__restrict__ frame,
                                                                   const uint4 *__restrict__ wimg,
                                                                   const float *__restrict__ bias, uint16_t *__restrict__ yp,
                                                                   int H, int W, int Ho, int Wo, int Hp, int Wp, int tiles_x,
                                                                   int Cop) {
    constexpr int NC = NFB * 16;
    __shared__ __attribute__((aligned(16))) uint16_t sIn[(FP_ROWS + 1) * F7_PITCH + 8];
    __shared__ __attribute__((aligned(16))) uint16_t sC[(FP_CR - 1) * FP_CC * NC];  // (row 17 is computed, never pooled)
    const int tile = blockIdx.x, n = blockIdx.z, tid = threadIdx.x;
    const int py0 = (tile / tiles_x) * FP_PH, px0 = (tile % tiles_x) * FP_PW;
    const int cr0 = 2 * py0 - 1, cc0 = 2 * px0 - 1;  // conv position of local (0, 0)
    const int64_t plane = (int64_t)H * W;
    const float *f0 = frame + (int64_t)n * 3 * plane;
    constexpr int FP_N = 3 * FP_ROWS * F7_PW, FP_B = 6;  // batched, unconditional loads: see k_conv7x7s2_first_mfma
    PatchWalk pw;
    pw.start(tid, FP_ROWS);
    const bool staged = stage_patch_vec4<FP_ROWS>(f0, plane, cr0 * 2 - 3, cc0 * 2 - 3, H, W, sIn, tid);
    for (int base = tid; !staged && base < FP_N; base += 256 * FP_B) {
        float v[FP_B];
        int dst[FP_B];
#pragma unroll
        for (int u = 0; u < FP_B; ++u) {
            const int ci = min(pw.ci, 2), r = pw.r, c = pw.c;
            const int iy = cr0 * 2 - 3 + r, ix = cc0 * 2 - 3 + c;
            const float t = f0[ci * plane + (int64_t)min(max(iy, 0), H - 1) * W + min(max(ix, 0), W - 1)];
            v[u] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? t : 0.f;
            dst[u] = pw.ci < 3 ? r * F7_PITCH + c * 3 + ci : -1;
            pw.step256(FP_ROWS);
        }
#pragma unroll
        for (int u = 0; u < FP_B; ++u)
            if (dst[u] >= 0) sIn[dst[u]] = f2bf(v[u]);
    }
    for (int idx = tid; idx < F7_PITCH + 8 + FP_ROWS + 1; idx += 256) {  // see k_conv7x7s2_first_mfma
        if (idx < F7_PITCH + 8) sIn[FP_ROWS * F7_PITCH + idx] = 0;
        else sIn[(idx - F7_PITCH - 8) * F7_PITCH + F7_PITCH - 1] = 0;
    }
    const int wave = tid >> 6, lane = tid & 63, l16 = lane & 15, kg = lane >> 4;
    bf16x8 wf[6][NFB];
#pragma unroll
    for (int ks = 0; ks < 6; ++ks)
#pragma unroll
        for (int nf = 0; nf < NFB; ++nf) wf[ks][nf] = __builtin_bit_cast(bf16x8, wimg[(ks * 4 + kg) * NC + nf * 16 + l16]);
    float bv[NFB][4];
#pragma unroll
    for (int nf = 0; nf < NFB; ++nf)
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[nf][r] = bias[nf * 16 + kg * 4 + r];
    __syncthreads();
    // three fragments at a time: three independent MFMA chains hide each other's LDS reads (one at a time, a wave sat out
    // every read's latency)
#pragma unroll 1
    for (int m0 = 0; m0 < 9; m0 += 3) {
        f32x4 acc[3][NFB];
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int nf = 0; nf < NFB; ++nf) acc[u][nf] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) {
            const int k0 = ks * 32 + kg * 8, ky = k0 / 24, j0 = k0 % 24;
            bf16x8 a[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int frag = wave * 9 + m0 + u, ly = frag >> 1, lx = (frag & 1) * 16 + l16;
                const uint32_t *p = reinterpret_cast<const uint32_t *>(sIn + (2 * ly + ky) * F7_PITCH + 6 * lx + j0);
                a[u] = __builtin_bit_cast(bf16x8, make_uint4(p[0], p[1], p[2], p[3]));
            }
#pragma unroll
            for (int u = 0; u < 3; ++u)
#pragma unroll
                for (int nf = 0; nf < NFB; ++nf)
                    acc[u][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][nf], a[u], acc[u][nf], 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int frag = wave * 9 + m0 + u, ly = frag >> 1, lx = (frag & 1) * 16 + l16;
            const int cr = cr0 + ly, cc = cc0 + lx;
            const bool real = cr >= 0 && cr < Ho && cc >= 0 && cc < Wo;
            if (ly >= FP_CR - 1) continue;
#pragma unroll
            for (int nf = 0; nf < NFB; ++nf) {
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = real ? fmaxf(acc[u][nf][r] + bv[nf][r], 0.f) : 0.f;
                *reinterpret_cast<uint2 *>(sC + (ly * FP_CC + lx) * NC + nf * 16 + kg * 4) =
                    make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
            }
        }
    }
    __syncthreads();
    const int cg = Cop >> 3;  // 8-channel groups that exist in the output
    for (int idx = tid; idx < FP_PH * FP_PW * cg; idx += 256) {
        const int g = idx % cg, q = idx / cg, pxl = q % FP_PW, pyl = q / FP_PW;
        const int py = py0 + pyl, px = px0 + pxl;
        if (py >= Hp || px >= Wp) continue;
        float mx[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) mx[e] = 0.f;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                float v[8];
                unpack8(*reinterpret_cast<const uint4 *>(sC + ((2 * pyl + dy) * FP_CC + 2 * pxl + dx) * NC + g * 8), v);
#pragma unroll
                for (int e = 0; e < 8; ++e) mx[e] = fmaxf(mx[e], v[e]);
            }
        *reinterpret_cast<uint4 *>(yp + (((int64_t)n * Hp + py) * Wp + px) * Cop + g * 8) = pack8(mx);
    }
}

// ------------------------------------------------------------------------------------------ 3x3 stride-2 max pool, pad 1
__global__ __launch_bounds__(256) void k_maxpool3x3s2(const uint4 *__restrict__ x, uint4 *__restrict__ y, int N, int H,
                                                      int W, int Ho, int Wo, int c8) {
    const int64_t total = (int64_t)N * Ho * Wo * c8;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int g = (int)(i % c8);
    int64_t r = i / c8;
    const int ox = (int)(r % Wo);
    r /= Wo;
    const int oy = (int)(r % Ho), n = (int)(r / Ho);
    float m[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
        const int iy = oy * 2 + dy;
        if (iy < 0 || iy >= H) continue;
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int ix = ox * 2 + dx;
            if (ix < 0 || ix >= W) continue;
            float v[8];
            unpack8(x[(((int64_t)n * H + iy) * W + ix) * c8 + g], v);
#pragma unroll
            for (int e = 0; e < 8; ++e) m[e] = fmaxf(m[e], v[e]);
        }
    }
    y[i] = pack8(m);
}

// ------------------------------------------------------------------------------------------ side-output head
// Four scales of transposed conv (kernel 2f, stride f), centre crop, concat and 1x1 fuse, with the 16 -> 16 upscale
// filter and the fuse weights already contracted into one [k][k][16] filter per scale on the host (both are linear).
struct DeconvHeadArgs {
    const float *side[4];
    const float *filt[4];
    const float *filt1[4];
    float *side_out[4];
    float *fused;
    const float *dsn_w, *dsn_b, *fuse_b;
    int hs[4], ws[4], f[4], top[4], left[4];
    int N, H, W, with_side_out;
};

// A workgroup owns 4 rows x 640 columns (three per row of a 1920-wide frame: 810 workgroups, which all fit the chip at
// once - with 512 columns the 1080 workgroups needed a second round for the last 56: 51 vs 44 us; 960 columns: 47 us);
// a thread owns DH_PX pixels of one row, 64 columns apart: every stride
// divides 64, so they share the filter phase (ky, kx) at every scale and the 4 taps x 16 channels of filter live in
// registers for all of them.  Per scale the side-map window under the tile (3 rows x (640 / f + 2) pixels, zeros
// outside the map) is staged in LDS together with its score_dsn value d = dsn_b + dsn_w . side, computed once per
// source pixel instead of once per output pixel and tap; the taps then read LDS (groups of f lanes share an address).
constexpr int DH_PX = 10, DH_ROWS = 4, DH_COLS = 64 * DH_PX;
constexpr int DH_NR = 3, DH_NC = DH_COLS / 4 + 2;  // window bound at the smallest stride (4)
constexpr int DH_WIN = DH_NR * DH_NC;
#define V4(v) f32x4{(v).x, (v).y, (v).z, (v).w}

__global__ __launch_bounds__(256) void k_deconv_head(const DeconvHeadArgs g) {
    __shared__ f32x4 sS[DH_NR * DH_NC * 4];  // a true vector type: float4 (a struct) is scalarised and re-sliced into 4-byte reads
    __shared__ float sD[DH_NR * DH_NC];
    const int Xb = blockIdx.x * DH_COLS, Yb = blockIdx.y * DH_ROWS, n = blockIdx.z;
    const int X0 = Xb + threadIdx.x, Y = Yb + threadIdx.y;
    const int tid = threadIdx.y * 64 + threadIdx.x;
    // two partial sums per pixel (even / odd channels): the operands of v_pk_fma_f32 are then register pairs as loaded,
    // where one running sum per pixel made hipcc pair neighbouring PIXELS and shuffle their operands together (half of
    // the kernel's 3900 vector instructions per wave were moves)
    f32x2 fused[DH_PX];
    const float fb = g.fuse_b[0];
#pragma unroll
    for (int p = 0; p < DH_PX; ++p) fused[p] = f32x2{fb, 0.f};
#pragma unroll 1
    for (int s = 0; s < 4; ++s) {
        const int f = g.f[s], k = 2 * f, hs = g.hs[s], ws = g.ws[s], jstep = 64 / f;
        const int r_lo = (Yb + g.top[s]) / f - 1, c_lo = (Xb + g.left[s]) / f - 1, nc = DH_COLS / f + 2;
        const int yy = Y + g.top[s], xx = X0 + g.left[s];
        const int i0 = yy / f, ky0 = yy - i0 * f, j0 = xx / f, kx0 = xx - j0 * f;
        // the filter of tap (a, b) sits at [ky0 + a f][kx0 + b f]; tap 0's is requested before the window is staged and
        // tap t + 1's while tap t is multiplied, so no tap waits out a global round trip
        const f32x4 *fbase = reinterpret_cast<const f32x4 *>(g.filt[s]);
        const float *f1base = g.filt1[s];
        int fidx = ky0 * k + kx0;
        f32x4 n0 = fbase[fidx * 4 + 0], n1 = fbase[fidx * 4 + 1], n2 = fbase[fidx * 4 + 2], n3 = fbase[fidx * 4 + 3];
        float nw = g.with_side_out ? f1base[fidx] : 0.f;
        __syncthreads();  // the previous scale's window has been consumed
        for (int idx = tid; idx < DH_NR * nc; idx += 256) {
            const int r = idx / nc, c = idx - r * nc, i = r_lo + r, j = c_lo + c;
            float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0, v2 = v0, v3 = v0;
            float d = 0.f;
            if (i >= 0 && i < hs && j >= 0 && j < ws) {
                const float4 *sp = reinterpret_cast<const float4 *>(g.side[s]) + (((int64_t)n * hs + i) * ws + j) * 4;
                v0 = sp[0], v1 = sp[1], v2 = sp[2], v3 = sp[3];
                if (g.with_side_out) {
                    const float4 *dw = reinterpret_cast<const float4 *>(g.dsn_w) + s * 4;
                    const float4 w0 = dw[0], w1 = dw[1], w2 = dw[2], w3 = dw[3];
                    d = g.dsn_b[s];
                    d = fmaf(v0.x, w0.x, d); d = fmaf(v0.y, w0.y, d); d = fmaf(v0.z, w0.z, d); d = fmaf(v0.w, w0.w, d);
                    d = fmaf(v1.x, w1.x, d); d = fmaf(v1.y, w1.y, d); d = fmaf(v1.z, w1.z, d); d = fmaf(v1.w, w1.w, d);
                    d = fmaf(v2.x, w2.x, d); d = fmaf(v2.y, w2.y, d); d = fmaf(v2.z, w2.z, d); d = fmaf(v2.w, w2.w, d);
                    d = fmaf(v3.x, w3.x, d); d = fmaf(v3.y, w3.y, d); d = fmaf(v3.z, w3.z, d); d = fmaf(v3.w, w3.w, d);
                }
            }
            sS[0 * DH_WIN + idx] = V4(v0);  // channel-quad planes: neighbouring source pixels sit 16 B apart, so the taps' reads
            sS[1 * DH_WIN + idx] = V4(v1);  // are conflict-free at any width hipcc splits them into (pixel-major, its 4-byte
            sS[2 * DH_WIN + idx] = V4(v2);  // reads at a 64-B pitch spent half of all LDS cycles on bank conflicts)
            sS[3 * DH_WIN + idx] = V4(v3);
            sD[idx] = d;
        }
        __syncthreads();
        float so[DH_PX];
#pragma unroll
        for (int p = 0; p < DH_PX; ++p) so[p] = 0.f;
        // (the tap loop stays rolled: unrolled, hipcc keeps all four taps' filters and window reads live - 512 VGPRs and
        // scratch)
#pragma unroll 1
        for (int tap = 0; tap < 4; ++tap) {
            const int a = tap >> 1, b = tap & 1;
            const int r = i0 - a - r_lo;  // 0..2
            const f32x4 f0 = n0, f1 = n1, f2 = n2, f3 = n3;
            const float w1 = nw;
            if (tap < 3) {
                const int ta = (tap + 1) >> 1, tb = (tap + 1) & 1;
                fidx = (ky0 + ta * f) * k + kx0 + tb * f;
                n0 = fbase[fidx * 4 + 0], n1 = fbase[fidx * 4 + 1], n2 = fbase[fidx * 4 + 2], n3 = fbase[fidx * 4 + 3];
                nw = g.with_side_out ? f1base[fidx] : 0.f;
            }
            const int base = r * nc + (j0 - b - c_lo);
#pragma unroll
            for (int p = 0; p < DH_PX; ++p) {
                const int idx = base + p * jstep;
                const f32x4 s0 = sS[idx], s1 = sS[DH_WIN + idx], s2 = sS[2 * DH_WIN + idx], s3 = sS[3 * DH_WIN + idx];
                f32x2 t = fused[p];
                t += s0.lo * f0.lo; t += s0.hi * f0.hi;
                t += s1.lo * f1.lo; t += s1.hi * f1.hi;
                t += s2.lo * f2.lo; t += s2.hi * f2.hi;
                t += s3.lo * f3.lo; t += s3.hi * f3.hi;
                fused[p] = t;
                so[p] = fmaf(w1, sD[idx], so[p]);
            }
        }
        if (g.with_side_out && Y < g.H) {
#pragma unroll
            for (int p = 0; p < DH_PX; ++p)
                if (X0 + 64 * p < g.W) g.side_out[s][((int64_t)n * g.H + Y) * g.W + X0 + 64 * p] = so[p];
        }
    }
    if (Y >= g.H) return;
#pragma unroll
    for (int p = 0; p < DH_PX; ++p)
        if (X0 + 64 * p < g.W) g.fused[((int64_t)n * g.H + Y) * g.W + X0 + 64 * p] = fused[p][0] + fused[p][1];
}

// ------------------------------------------------------------------------------------------ launch planning
// Output-channel block of a launch: the widest of 64/32/16/8 that wastes at most an eighth of its lanes on channel
// padding and still gives the chip 2 waves per SIMD; thin or deep layers fall back to narrower blocks (more waves), and
// when even 8-channel blocks leave SIMDs empty the input channels are split over up to 8 waves of a workgroup.
struct ConvLaunch {
    int cob, threads, slices;
};
ConvLaunch plan_conv2d(int64_t npix, int Cop, int cic) {
    const int cand[4] = {64, 32, 16, 8};
    int pick = 8;
    for (int q = 0; q < 4; ++q) {
        const int cob = cand[q], nb = (Cop + cob - 1) / cob;
        if (cob != 8 && (nb * cob - Cop) * 8 > Cop) continue;
        pick = cob;
        if (cdiv(npix, 64) * nb >= 2048) break;
    }
#ifdef FOSVOS_CONV2D_LAB
    if (const char *e = lab_env("FOSVOS_COB")) pick = atoi(e);
#endif
    const int nb = (Cop + pick - 1) / pick;
    const int64_t waves = cdiv(npix, 64) * nb;
#ifdef FOSVOS_CONV2D_LAB
    if (const char *e = lab_env("FOSVOS_THREADS")) return {pick, atoi(e), 1};
#endif
    // (measured: at 2028 waves - 32 -> 32 channels on 135 x 240 - four slices cost 20 us against 15 unsliced; at 512 waves -
    // 128 -> 128 on 34 x 60 - eight slices take 32 us against 83)
    if (waves >= 1024 || cic < 2) return {pick, cdiv(npix, 256) * nb >= 1024 ? 256 : 64, 1};
    int slices = 2;
    while (slices < 8 && slices * 2 <= cic && waves * slices < 4096) slices *= 2;
    return {pick, 64, slices};
}

template <int KS, int S, bool OUT_F32>
void launch_conv2d(const ConvLaunch &L, dim3 grid, hipStream_t st, const uint4 *x, const uint32_t *wp, const float *bias,
                   const uint4 *addend, void *y, int N, int H, int W, int Ho, int Wo, int cic, int Cop, int relu) {
    const size_t lds = L.slices > 1 ? (size_t)(L.slices - 1) * L.cob * L.threads * sizeof(float) : 0;  // <= 7*64*64*4
#define FOSVOS_GO(COB, SLICED)                                                                                          \
    hipLaunchKernelGGL((k_conv2d<KS, S, COB, OUT_F32, SLICED>), grid, dim3(L.threads, L.slices), lds, st, x, wp, bias,      \
                       addend, y, N, H, W, Ho, Wo, cic, Cop, relu)
    if (L.slices > 1) {  // (only ever planned with 8-channel blocks)
        FOSVOS_GO(8, true);
        return;
    }
    switch (L.cob) {
        case 64: FOSVOS_GO(64, false); break;
        case 32: FOSVOS_GO(32, false); break;
        case 16: FOSVOS_GO(16, false); break;
        default: FOSVOS_GO(8, false); break;
    }
#undef FOSVOS_GO
}

}  // namespace

// ------------------------------------------------------------------------------------------ C ABI
extern "C" size_t fosvos_conv2d_packed_dwords(int out_ch, int in_ch, int k) {
    if (out_ch <= 0 || in_ch <= 0 || (k != 1 && k != 3)) return 0;
    return (size_t)(roundup(in_ch, 8) / 8) * k * k * 4 * roundup(out_ch, 8) + 64;  // + slack a partial block may read
}
extern "C" size_t fosvos_conv2d_bias_elems(int out_ch) { return out_ch > 0 ? (size_t)roundup(out_ch, 64) + 64 : 0; }

extern "C" int fosvos_pack_conv2d_bn(const float *w_oihw, int Co, int Ci, int k, const float *conv_bias,
                                     const float *bn_weight, const float *bn_bias, const float *bn_mean,
                                     const float *bn_var, float eps, uint32_t *w_packed, float *bias_out, int device,
                                     void *stream) {
    FOSVOS_REQUIRE(w_oihw && w_packed && bias_out, FOSVOS_E_ARG, "pack_conv2d_bn: null pointer");
    FOSVOS_REQUIRE(Co > 0 && Ci > 0 && (k == 1 || k == 3), FOSVOS_E_ARG, "pack_conv2d_bn: Co=%d Ci=%d k=%d", Co, Ci, k);
    FOSVOS_REQUIRE(!bn_weight || (bn_bias && bn_mean && bn_var), FOSVOS_E_ARG, "pack_conv2d_bn: partial BatchNorm");
    FOSVOS_ENTER(device);
    const int Cop = roundup(Co, 8), cic = roundup(Ci, 8) / 8;
    const int64_t total = (int64_t)fosvos_conv2d_packed_dwords(Co, Ci, k);
    const int bias_len = (int)fosvos_conv2d_bias_elems(Co);
    const int64_t n = std::max<int64_t>(total, bias_len);
    hipLaunchKernelGGL(k_pack_conv2d_bn, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, w_oihw, Co, Ci, k,
                       conv_bias, bn_weight, bn_bias, bn_mean, bn_var, eps, w_packed, bias_out, Cop, cic, total, bias_len);
    FOSVOS_LAUNCH_CHECK();
    return 0;
}

extern "C" int fosvos_fold_conv_bn(const float *w_oihw, int Co, int Ci, int k, const float *conv_bias,
                                   const float *bn_weight, const float *bn_bias, const float *bn_mean,
                                   const float *bn_var, float eps, float *w_folded, float *bias_out, int device,
                                   void *stream) {
    FOSVOS_REQUIRE(w_oihw && w_folded && bias_out, FOSVOS_E_ARG, "fold_conv_bn: null pointer");
    FOSVOS_REQUIRE(Co > 0 && Ci > 0 && k > 0, FOSVOS_E_ARG, "fold_conv_bn: Co=%d Ci=%d k=%d", Co, Ci, k);
    FOSVOS_REQUIRE(!bn_weight || (bn_bias && bn_mean && bn_var), FOSVOS_E_ARG, "fold_conv_bn: partial BatchNorm");
    FOSVOS_ENTER(device);
    const int per_co = Ci * k * k;
    hipLaunchKernelGGL(k_fold_conv_bn, dim3((unsigned)cdiv((int64_t)Co * per_co, 256)), dim3(256), 0, (hipStream_t)stream,
                       w_oihw, Co, per_co, conv_bias, bn_weight, bn_bias, bn_mean, bn_var, eps, w_folded, bias_out);
    FOSVOS_LAUNCH_CHECK();
    return 0;
}

extern "C" int fosvos_conv2d_fwd(const uint16_t *x, const uint32_t *w_packed, const float *bias, const uint16_t *addend,
                                 void *y, int N, int H, int W, int Ci, int Co, int k, int stride, unsigned flags,
                                 int device, void *stream) {
    FOSVOS_REQUIRE(x && w_packed && bias && y, FOSVOS_E_ARG, "conv2d_fwd: null pointer");
    FOSVOS_REQUIRE(N > 0 && H > 0 && W > 0 && Ci > 0 && Co > 0, FOSVOS_E_ARG, "conv2d_fwd: N=%d H=%d W=%d Ci=%d Co=%d", N,
                   H, W, Ci, Co);
    FOSVOS_REQUIRE((k == 1 || k == 3) && (stride == 1 || stride == 2), FOSVOS_E_SHAPE,
                   "conv2d_fwd: k=%d stride=%d (1 or 3, 1 or 2)", k, stride);
    FOSVOS_REQUIRE((flags & ~(unsigned)(FOSVOS_CONV_RELU | FOSVOS_CONV_OUT_F32)) == 0, FOSVOS_E_ARG,
                   "conv2d_fwd: unknown flags 0x%x", flags);
    const bool f32 = flags & FOSVOS_CONV_OUT_F32;
    FOSVOS_REQUIRE(!(f32 && addend), FOSVOS_E_SHAPE, "conv2d_fwd: fp32 output with a residual operand");
    const int P = k / 2, Ho = (H + 2 * P - k) / stride + 1, Wo = (W + 2 * P - k) / stride + 1;
    const int Cop = roundup(Co, 8), cic = roundup(Ci, 8) / 8;
    FOSVOS_REQUIRE((int64_t)N * H * W * cic < (int64_t)1 << 31 && (int64_t)N * Ho * Wo * Cop < (int64_t)1 << 34, FOSVOS_E_ARG,
                   "conv2d_fwd: tensor too large for 32-bit tap offsets");
    FOSVOS_ENTER(device);
    const int64_t npix = (int64_t)N * Ho * Wo;
    const ConvLaunch L = plan_conv2d(npix, Cop, cic);
    const dim3 grid((unsigned)cdiv(npix, L.threads), (unsigned)cdiv(Cop, L.cob));
    const int relu = (flags & FOSVOS_CONV_RELU) ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
    const uint4 *xv = reinterpret_cast<const uint4 *>(x), *av = reinterpret_cast<const uint4 *>(addend);
#define FOSVOS_ARGS L, grid, st, xv, w_packed, bias, av, y, N, H, W, Ho, Wo, cic, Cop, relu
    if (k == 3 && stride == 1) {
        if (f32) launch_conv2d<3, 1, true>(FOSVOS_ARGS);
        else launch_conv2d<3, 1, false>(FOSVOS_ARGS);
    } else if (k == 3) {
        FOSVOS_REQUIRE(!f32, FOSVOS_E_SHAPE, "conv2d_fwd: fp32 output only for 3x3 stride 1");
        launch_conv2d<3, 2, false>(FOSVOS_ARGS);
    } else if (stride == 1) {
        FOSVOS_REQUIRE(!f32, FOSVOS_E_SHAPE, "conv2d_fwd: fp32 output only for 3x3 stride 1");
        launch_conv2d<1, 1, false>(FOSVOS_ARGS);
    } else {
        FOSVOS_REQUIRE(!f32, FOSVOS_E_SHAPE, "conv2d_fwd: fp32 output only for 3x3 stride 1");
        launch_conv2d<1, 2, false>(FOSVOS_ARGS);
    }
#undef FOSVOS_ARGS
    FOSVOS_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t fosvos_conv7x7_packed_elems(int out_ch) {  // fp32 image + slack, then the bf16 MFMA image (2 per float)
    if (out_ch <= 0) return 0;
    const int Cop = roundup(out_ch, 8);
    return (size_t)147 * Cop + 64 + (size_t)6 * 4 * f7_nc(Cop) * 8 / 2;
}

extern "C" int fosvos_pack_conv7x7_bn(const float *w_oihw, int Co, const float *bn_weight, const float *bn_bias,
                                      const float *bn_mean, const float *bn_var, float eps, float *w_packed,
                                      float *bias_out, int device, void *stream) {
    FOSVOS_REQUIRE(w_oihw && w_packed && bias_out && Co > 0, FOSVOS_E_ARG, "pack_conv7x7_bn: bad argument");
    FOSVOS_REQUIRE(!bn_weight || (bn_bias && bn_mean && bn_var), FOSVOS_E_ARG, "pack_conv7x7_bn: partial BatchNorm");
    FOSVOS_ENTER(device);
    const int Cop = roundup(Co, 8), total = 147 * Cop + 64;
    const int bias_len = (int)fosvos_conv2d_bias_elems(Co);
    const int n = std::max(std::max(total, bias_len), 6 * 4 * f7_nc(Cop) * 8);
    hipLaunchKernelGGL(k_pack_conv7x7_bn, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, w_oihw, Co,
                       bn_weight, bn_bias, bn_mean, bn_var, eps, w_packed, bias_out, Cop, total, bias_len);
    FOSVOS_LAUNCH_CHECK();
    return 0;
}

extern "C" int fosvos_conv7x7s2_first_fwd(const float *frame, const float *w_packed, const float *bias, uint16_t *y, int N,
                                          int H, int W, int Co, unsigned flags, int device, void *stream) {
    FOSVOS_REQUIRE(frame && w_packed && bias && y, FOSVOS_E_ARG, "conv7x7s2_first_fwd: null pointer");
    FOSVOS_REQUIRE(N > 0 && H > 0 && W > 0 && Co > 0, FOSVOS_E_ARG, "conv7x7s2_first_fwd: N=%d H=%d W=%d Co=%d", N, H, W, Co);
    FOSVOS_REQUIRE((flags & ~(unsigned)(FOSVOS_CONV_RELU | FOSVOS_CONV_FP32_MATH)) == 0, FOSVOS_E_ARG,
                   "conv7x7s2_first_fwd: flags 0x%x", flags);
    FOSVOS_ENTER(device);
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1, Cop = roundup(Co, 8);
    FOSVOS_REQUIRE(N < 65536, FOSVOS_E_ARG, "conv7x7s2_first_fwd: N=%d", N);
    const int tiles_x = (int)cdiv(Wo, F7_TW), tiles_y = (int)cdiv(Ho, F7_TH);
    const int relu = (flags & FOSVOS_CONV_RELU) ? 1 : 0;
    if (!(flags & FOSVOS_CONV_FP32_MATH) && Cop <= 64) {  // bf16 MFMA form
        const uint4 *wimg = reinterpret_cast<const uint4 *>(w_packed + 147 * Cop + 64);
        const dim3 g((unsigned)(tiles_x * tiles_y), 1, (unsigned)N);
#define FOSVOS_GO(NFB)                                                                                                   \
    hipLaunchKernelGGL(k_conv7x7s2_first_mfma<NFB>, g, dim3(256), 0, (hipStream_t)stream, frame, wimg, bias, y, H, W, Ho, Wo, \
                       tiles_x, Cop, relu)
        switch (f7_nc(Cop) / 16) {
            case 1: FOSVOS_GO(1); break;
            case 2: FOSVOS_GO(2); break;
            case 3: FOSVOS_GO(3); break;
            default: FOSVOS_GO(4); break;
        }
#undef FOSVOS_GO
        FOSVOS_LAUNCH_CHECK();
        return 0;
    }
    const ConvLaunch L = plan_conv2d((int64_t)N * Ho * Wo, Cop, 1);
    const dim3 grid((unsigned)(tiles_x * tiles_y), (unsigned)cdiv(Cop, L.cob), (unsigned)N);
    uint4 *yv = reinterpret_cast<uint4 *>(y);
#define FOSVOS_GO(COB)                                                                                                \
    hipLaunchKernelGGL(k_conv7x7s2_first<COB>, grid, dim3(256), 0, (hipStream_t)stream, frame, w_packed, bias, yv, H, W, Ho, \
                       Wo, tiles_x, Cop, relu)
    switch (L.cob) {
        case 64: FOSVOS_GO(64); break;
        case 32: FOSVOS_GO(32); break;
        case 16: FOSVOS_GO(16); break;
        default: FOSVOS_GO(8); break;
    }
#undef FOSVOS_GO
    FOSVOS_LAUNCH_CHECK();
    return 0;
}

extern "C" int fosvos_conv7x7s2_pool_first_fwd(const float *frame, const float *w_packed, const float *bias,
                                               uint16_t *y_pooled, int N, int H, int W, int Co, int device, void *stream) {
    FOSVOS_REQUIRE(frame && w_packed && bias && y_pooled, FOSVOS_E_ARG, "conv7x7s2_pool_first_fwd: null pointer");
    FOSVOS_REQUIRE(N > 0 && N < 65536 && H > 0 && W > 0 && Co > 0, FOSVOS_E_ARG, "conv7x7s2_pool_first_fwd: N=%d H=%d W=%d Co=%d",
                   N, H, W, Co);
    const int Cop = roundup(Co, 8);
    FOSVOS_REQUIRE(Cop <= 64, FOSVOS_E_SHAPE, "conv7x7s2_pool_first_fwd: at most 64 channels (got %d)", Co);
    FOSVOS_ENTER(device);
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1, Hp = (Ho - 1) / 2 + 1, Wp = (Wo - 1) / 2 + 1;
    const int tiles_x = (int)cdiv(Wp, FP_PW), tiles_y = (int)cdiv(Hp, FP_PH);
    const uint4 *wimg = reinterpret_cast<const uint4 *>(w_packed + 147 * Cop + 64);
    const dim3 g((unsigned)(tiles_x * tiles_y), 1, (unsigned)N);
#define FOSVOS_GO(NFB)                                                                                                    \
    hipLaunchKernelGGL(k_conv7x7s2_pool_first_mfma<NFB>, g, dim3(256), 0, (hipStream_t)stream, frame, wimg, bias, y_pooled, H, \
                       W, Ho, Wo, Hp, Wp, tiles_x, Cop)
    switch (f7_nc(Cop) / 16) {
        case 1: FOSVOS_GO(1); break;
        case 2: FOSVOS_GO(2); break;
        case 3: FOSVOS_GO(3); break;
        default: FOSVOS_GO(4); break;
    }
#undef FOSVOS_GO
    FOSVOS_LAUNCH_CHECK();
    return 0;
}

extern "C" int fosvos_maxpool3x3s2_fwd(const uint16_t *x, uint16_t *y, int N, int H, int W, int C, int device,
                                       void *stream) {
    FOSVOS_REQUIRE(x && y, FOSVOS_E_ARG, "maxpool3x3s2_fwd: null pointer");
    FOSVOS_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, FOSVOS_E_ARG, "maxpool3x3s2_fwd: N=%d H=%d W=%d C=%d", N, H,
                   W, C);
    FOSVOS_ENTER(device);
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const int64_t total = (int64_t)N * Ho * Wo * (C / 8);
    hipLaunchKernelGGL(k_maxpool3x3s2, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const uint4 *>(x), reinterpret_cast<uint4 *>(y), N, H, W, Ho, Wo, C / 8);
    FOSVOS_LAUNCH_CHECK();
    return 0;
}

extern "C" int fosvos_deconv_head_fwd(const float *const side[4], const int hs[4], const int ws[4], const int stride[4],
                                      const float *const filt[4], const float *const filt1[4], const float *dsn_w,
                                      const float *dsn_b, const float *fuse_b, float *fused, float *const side_out[4],
                                      int N, int H, int W, int device, void *stream) {
    FOSVOS_REQUIRE(side && hs && ws && stride && filt && fuse_b && fused, FOSVOS_E_ARG, "deconv_head_fwd: null pointer");
    FOSVOS_REQUIRE(N > 0 && H > 0 && W > 0 && N < 65536, FOSVOS_E_ARG, "deconv_head_fwd: N=%d H=%d W=%d", N, H, W);
    DeconvHeadArgs g;
    const bool so = side_out && side_out[0];
    for (int s = 0; s < 4; ++s) {
        FOSVOS_REQUIRE(side[s] && filt[s], FOSVOS_E_ARG, "deconv_head_fwd: null side/filter pointer at scale %d", s);
        FOSVOS_REQUIRE(hs[s] > 0 && ws[s] > 0 && stride[s] > 0, FOSVOS_E_ARG, "deconv_head_fwd: scale %d is %dx%d stride %d",
                       s, hs[s], ws[s], stride[s]);
        FOSVOS_REQUIRE(stride[s] >= 4 && 64 % stride[s] == 0, FOSVOS_E_SHAPE,
                       "deconv_head_fwd: stride %d of scale %d (4, 8, 16, 32 or 64)", stride[s], s);
        const int dh = (hs[s] + 1) * stride[s], dw = (ws[s] + 1) * stride[s];  // (h-1) f + 2f
        FOSVOS_REQUIRE(dh >= H && dw >= W, FOSVOS_E_ARG,
                       "deconv_head_fwd: scale %d upsamples to %dx%d, smaller than the %dx%d frame", s, dh, dw, H, W);
        if (so)
            FOSVOS_REQUIRE(side_out[s] && filt1 && filt1[s] && dsn_w && dsn_b, FOSVOS_E_ARG,
                           "deconv_head_fwd: side outputs need side_out[%d], filt1, dsn_w, dsn_b", s);
        g.side[s] = side[s];
        g.filt[s] = filt[s];
        g.filt1[s] = so ? filt1[s] : nullptr;
        g.side_out[s] = so ? side_out[s] : nullptr;
        g.hs[s] = hs[s];
        g.ws[s] = ws[s];
        g.f[s] = stride[s];
        g.top[s] = (dh - H) / 2;  // the reference's centre crop drops floor(d/2) leading rows / columns
        g.left[s] = (dw - W) / 2;
    }
    g.fused = fused;
    g.dsn_w = dsn_w;
    g.dsn_b = dsn_b;
    g.fuse_b = fuse_b;
    g.N = N;
    g.H = H;
    g.W = W;
    g.with_side_out = so ? 1 : 0;
    FOSVOS_ENTER(device);
    hipLaunchKernelGGL(k_deconv_head, dim3((unsigned)cdiv(W, DH_COLS), (unsigned)cdiv(H, DH_ROWS), (unsigned)N), dim3(64, DH_ROWS), 0,
                       (hipStream_t)stream, g);
    FOSVOS_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------ whole-network forward
namespace {

struct ActShape {
    int h, w, c;  // c = real channels
    size_t bytes(int N) const { return (size_t)N * h * w * roundup(c, 8) * sizeof(uint16_t); }
};
inline int conv_out(int h, int k, int stride) { return (h + 2 * (k / 2) - k) / stride + 1; }
inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// Arena: [first conv output][4 rotating activation slots of the largest block tensor][4 fp32 side maps]
struct ResnetLayout {
    size_t first_bytes, slot_bytes, side_off[4], side_bytes[4], ws_off, ws_bytes, aux_ws_off, aux_ws_bytes, total;
    int hs[4], ws[4];
};

int resnet_layout(const fosvos_resnet_net *net, int N, int H, int W, ResnetLayout *L) {
    FOSVOS_REQUIRE(net && net->blocks && net->first_w && net->first_b && net->first_co > 0, FOSVOS_E_ARG,
                   "resnet: null net / blocks / first layer");
    FOSVOS_REQUIRE(N > 0 && H > 0 && W > 0, FOSVOS_E_ARG, "resnet: N=%d H=%d W=%d", N, H, W);
    ActShape a{(H - 1) / 2 + 1, (W - 1) / 2 + 1, net->first_co};
    L->first_bytes = align256(a.bytes(N));
    a.h = (a.h - 1) / 2 + 1;
    a.w = (a.w - 1) / 2 + 1;
    size_t slot = a.bytes(N), ws = 0, aux_ws = 0;
    int b = 0;
    for (int s = 0; s < 4; ++s) {
        FOSVOS_REQUIRE(net->blocks_per_stage[s] > 0, FOSVOS_E_ARG, "resnet: stage %d has no blocks", s);
        for (int j = 0; j < net->blocks_per_stage[s]; ++j, ++b) {
            const fosvos_resnet_block &blk = net->blocks[b];
            FOSVOS_REQUIRE(blk.n_convs == 2 || blk.n_convs == 3, FOSVOS_E_ARG, "resnet: block %d has %d convs", b, blk.n_convs);
            ActShape y = a;
            for (int q = 0; q < blk.n_convs; ++q) {
                const fosvos_conv2d_desc &c = blk.conv[q];
                FOSVOS_REQUIRE(c.w_packed && c.bias && c.Ci == y.c, FOSVOS_E_SHAPE,
                               "resnet: block %d conv %d takes %d channels, its input has %d", b, q + 1, c.Ci, y.c);
                if (c.kind == 1) {
                    FOSVOS_REQUIRE(c.k == 3 && (c.stride == 1 || (c.stride == 2 && q < blk.n_convs - 1)) && c.Ci % 32 == 0 &&
                                       (c.Co % 64 == 0 || c.Co == 32),
                                   FOSVOS_E_SHAPE,
                                   "resnet: block %d conv %d (%d -> %d, k %d, stride %d) does not fit the MFMA path", b, q + 1,
                                   c.Ci, c.Co, c.k, c.stride);
                } else {
                    FOSVOS_REQUIRE(c.kind == 0, FOSVOS_E_ARG, "resnet: block %d conv %d has kind %d", b, q + 1, c.kind);
                }
                y = ActShape{conv_out(y.h, c.k, c.stride), conv_out(y.w, c.k, c.stride), c.Co};
                slot = std::max(slot, y.bytes(N));
            }
            ActShape r = a;
            if (blk.has_down) {
                FOSVOS_REQUIRE(blk.down.w_packed && blk.down.bias && blk.down.Ci == a.c && blk.down.kind == 0, FOSVOS_E_SHAPE,
                               "resnet: block %d downsample takes %d channels, its input has %d", b, blk.down.Ci, a.c);
                r = ActShape{conv_out(a.h, blk.down.k, blk.down.stride), conv_out(a.w, blk.down.k, blk.down.stride), blk.down.Co};
                slot = std::max(slot, r.bytes(N));
            }
            FOSVOS_REQUIRE(r.h == y.h && r.w == y.w && r.c == y.c, FOSVOS_E_SHAPE,
                           "resnet: block %d residual is %dx%dx%d, the conv branch %dx%dx%d", b, r.h, r.w, r.c, y.h, y.w, y.c);
            a = y;
        }
        const fosvos_conv2d_desc &sp = net->side[s];
        FOSVOS_REQUIRE(sp.w_packed && sp.bias && sp.Co == 16 && sp.k == 3 && sp.stride == 1 && (sp.kind == 0 || sp.kind == 1),
                       FOSVOS_E_SHAPE, "resnet: side_prep %d must be a 3x3 stride-1 conv to 16 channels", s);
        if (sp.kind == 1) {
            FOSVOS_REQUIRE(sp.Ci % 32 == 0, FOSVOS_E_SHAPE, "resnet: side_prep %d (%d channels in) does not fit the MFMA path", s,
                           sp.Ci);
            aux_ws = std::max(aux_ws, fosvos_conv3x3_workspace_bytes(N, a.h, a.w, sp.Ci, 16));
        }
        FOSVOS_REQUIRE(sp.Ci == a.c, FOSVOS_E_SHAPE, "resnet: side_prep %d expects %d input channels, the stage produces %d", s,
                       sp.Ci, a.c);
        L->hs[s] = a.h;
        L->ws[s] = a.w;
        L->side_bytes[s] = align256((size_t)N * a.h * a.w * 16 * sizeof(float));
    }
    L->slot_bytes = align256(slot);
    size_t off = L->first_bytes + 4 * L->slot_bytes;
    for (int s = 0; s < 4; ++s) {
        L->side_off[s] = off;
        off += L->side_bytes[s];
    }
    L->ws_off = off;
    L->ws_bytes = align256(ws);
    L->aux_ws_off = off + L->ws_bytes;  // the side_prep convs may run beside the trunk (aux_stream): their own split-K slabs
    L->aux_ws_bytes = align256(aux_ws);
    L->total = L->aux_ws_off + L->aux_ws_bytes + 256;
    return 0;
}

}  // namespace

extern "C" size_t fosvos_resnet_arena_bytes(const fosvos_resnet_net *net, int N, int H, int W) {
    ResnetLayout L;
    return resnet_layout(net, N, H, W, &L) ? 0 : L.total;
}

// events of the caller's context (fosvos_ctx::resnet_ev) when an aux_stream is given:
// 0: fork point, 1: downsample done, 2..5: side map s done, 6..9: side conv s may start
static_assert(kFosvosResnetEvents == 12, "event slots of resnet_forward");

extern "C" int fosvos_resnet_forward(const fosvos_resnet_net *net, const float *frame, int N, int H, int W, void *arena,
                                     size_t arena_bytes, float *fused, float *const side_out[4], int device,
                                     void *stream, fosvos_ctx *ctx, void *aux_stream) {
    ResnetLayout L;
    if (int rc = resnet_layout(net, N, H, W, &L)) return rc;
    FOSVOS_REQUIRE(frame && arena && fused, FOSVOS_E_ARG, "resnet_forward: null pointer");
    FOSVOS_REQUIRE(arena_bytes >= L.total, FOSVOS_E_WORKSPACE, "resnet_forward: arena of %zu bytes, %zu needed", arena_bytes,
                   L.total);
    FOSVOS_ENTER(device);
    hipStream_t sm = (hipStream_t)stream, sa = (hipStream_t)aux_stream;
    const bool par = aux_stream != nullptr && aux_stream != stream;
    hipEvent_t *ev = nullptr;
    if (par) {
        if (int rc = ctx_check(ctx, "resnet_forward (aux_stream given)")) return rc;
        FOSVOS_REQUIRE(ctx->device == device, FOSVOS_E_ARG, "resnet_forward: context of device %d used on device %d",
                       ctx->device, device);
        ev = ctx->resnet_ev;
    }
    char *base = reinterpret_cast<char *>(((uintptr_t)arena + 255) & ~(uintptr_t)255);
    uint16_t *first = reinterpret_cast<uint16_t *>(base);
    uint16_t *slot[4];
    for (int q = 0; q < 4; ++q) slot[q] = reinterpret_cast<uint16_t *>(base + L.first_bytes + q * L.slot_bytes);
    auto other = [](int a, int b, int c) {  // a slot that is none of a, b, c
        for (int q = 0; q < 4; ++q)
            if (q != a && q != b && q != c) return q;
        return 0;
    };
    // a slot the auxiliary stream is still reading (the last stage output, under its side_prep conv): the trunk waits
    // for that conv before the first kernel that writes the slot again
    int busy_slot = -1, busy_ev = -1;
    auto before_write = [&](int q) -> int {
        if (par && q == busy_slot) {
            FOSVOS_HIP_CHECK(hipStreamWaitEvent(sm, ev[busy_ev], 0));
            busy_slot = -1;
        }
        return 0;
    };

    int h = (H - 1) / 2 + 1, w = (W - 1) / 2 + 1;
    // (fused with the pool up to 32 channels: above, the conv tile in LDS leaves one workgroup per CU and the fused launch
    // loses to the two kernels - 0.578 vs 0.55 ms per 1080p frame at 64 channels)
    if (!net->first_fp32_math && !net->first_unfused && roundup(net->first_co, 8) <= 32) {
        if (int rc = fosvos_conv7x7s2_pool_first_fwd(frame, net->first_w, net->first_b, slot[0], N, H, W, net->first_co, device,
                                                     stream))
            return rc;
    } else {
        if (int rc = fosvos_conv7x7s2_first_fwd(frame, net->first_w, net->first_b, first, N, H, W, net->first_co,
                                                FOSVOS_CONV_RELU | (net->first_fp32_math ? FOSVOS_CONV_FP32_MATH : 0u), device,
                                                stream))
            return rc;
        if (int rc = fosvos_maxpool3x3s2_fwd(first, slot[0], N, h, w, roundup(net->first_co, 8), device, stream)) return rc;
    }
    h = (h - 1) / 2 + 1;
    w = (w - 1) / 2 + 1;
    int cur = 0, b = 0;
    const float *side[4];
    for (int s = 0; s < 4; ++s) {
        for (int j = 0; j < net->blocks_per_stage[s]; ++j, ++b) {
            const fosvos_resnet_block &blk = net->blocks[b];
            int res = cur;
            if (blk.has_down) {
                // the 1x1 downsample conv of the residual branch runs beside conv1 (both read the block input)
                res = other(cur, -1, -1);
                if (int rc = before_write(res)) return rc;
                if (par) {
                    FOSVOS_HIP_CHECK(hipEventRecord(ev[0], sm));
                    FOSVOS_HIP_CHECK(hipStreamWaitEvent(sa, ev[0], 0));
                }
                if (int rc = fosvos_conv2d_fwd(slot[cur], reinterpret_cast<const uint32_t *>(blk.down.w_packed), blk.down.bias,
                                               nullptr, slot[res], N, h, w, blk.down.Ci, blk.down.Co, blk.down.k, blk.down.stride, 0,
                                               device, par ? aux_stream : stream))
                    return rc;
                if (par) FOSVOS_HIP_CHECK(hipEventRecord(ev[1], sa));
            }
            int in = cur, ih = h, iw = w;
            for (int q = 0; q < blk.n_convs; ++q) {
                const fosvos_conv2d_desc &c = blk.conv[q];
                const bool last = q == blk.n_convs - 1;
                const int out = other(in, res, last ? -1 : cur);  // (the block input stays live until the residual is taken)
                if (int rc = before_write(out)) return rc;
                if (last && blk.has_down && par) FOSVOS_HIP_CHECK(hipStreamWaitEvent(sm, ev[1], 0));
                const uint16_t *add = last ? slot[res] : nullptr;
                const int rc = c.kind == 1 && c.stride == 2
                                   ? fosvos_conv3x3_s2_fwd(slot[in], reinterpret_cast<const uint16_t *>(c.w_packed), c.bias, slot[out],
                                                           N, ih, iw, c.Ci, c.Co, FOSVOS_CONV_RELU, base + L.ws_off, L.ws_bytes,
                                                           device, stream)
                               : c.kind == 1
                                   ? fosvos_conv3x3_fwd_add(slot[in], reinterpret_cast<const uint16_t *>(c.w_packed), c.bias, add,
                                                            slot[out], N, ih, iw, c.Ci, c.Co, FOSVOS_CONV_RELU, base + L.ws_off,
                                                            L.ws_bytes, device, stream)
                                   : fosvos_conv2d_fwd(slot[in], reinterpret_cast<const uint32_t *>(c.w_packed), c.bias, add,
                                                       slot[out], N, ih, iw, c.Ci, c.Co, c.k, c.stride, FOSVOS_CONV_RELU, device,
                                                       stream);
                if (rc) return rc;
                ih = conv_out(ih, c.k, c.stride);
                iw = conv_out(iw, c.k, c.stride);
                in = out;
            }
            cur = in;
            h = ih;
            w = iw;
        }
        // side_prep of this stage: beside the next stage's blocks
        float *smap = reinterpret_cast<float *>(base + L.side_off[s]);
        const fosvos_conv2d_desc &sp = net->side[s];
        void *st = stream;
        if (par) {
            if (busy_slot >= 0) {  // one stage output at a time is tracked: retire the previous one first
                FOSVOS_HIP_CHECK(hipStreamWaitEvent(sm, ev[busy_ev], 0));
                busy_slot = -1;
            }
            FOSVOS_HIP_CHECK(hipEventRecord(ev[6 + s], sm));
            FOSVOS_HIP_CHECK(hipStreamWaitEvent(sa, ev[6 + s], 0));
            st = aux_stream;
        }
        const int rc = sp.kind == 1 ? fosvos_conv3x3_fwd_add(slot[cur], reinterpret_cast<const uint16_t *>(sp.w_packed), sp.bias,
                                                             nullptr, smap, N, h, w, sp.Ci, 16, FOSVOS_CONV_OUT_F32,
                                                             base + L.aux_ws_off, L.aux_ws_bytes, device, st)
                                    : fosvos_conv2d_fwd(slot[cur], reinterpret_cast<const uint32_t *>(sp.w_packed), sp.bias, nullptr,
                                                        smap, N, h, w, sp.Ci, sp.Co, 3, 1, FOSVOS_CONV_OUT_F32, device, st);
        if (rc) return rc;
        if (par) {
            FOSVOS_HIP_CHECK(hipEventRecord(ev[2 + s], sa));
            busy_slot = cur;
            busy_ev = 2 + s;
        }
        side[s] = smap;
    }
    if (par) {
        for (int s = 0; s < 4; ++s) FOSVOS_HIP_CHECK(hipStreamWaitEvent(sm, ev[2 + s], 0));
    }
    return fosvos_deconv_head_fwd(side, L.hs, L.ws, net->stride, net->filt, net->filt1, net->dsn_w, net->dsn_b, net->fuse_b,
                                  fused, side_out, N, H, W, device, stream);
}
