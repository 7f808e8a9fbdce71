// Weight / bias gradient of conv1_1 (3 -> 64 channels on the fp32 NCHW frame; autograd's backward-weight of Conv2d in
// stages[0][0], src/networks/osvos_vgg.py:92).
//
//   dw[co][ci][ky][kx] = sum_{n,y,x} dy[n,y,x,co] * frame[n,ci,y+ky-1,x+kx-1]        db[co] = sum dy[.,co]
//
// GEMM view: M = 64 co, N = 27 = (ci, ky, kx) padded to 32, K = 410k pixels: ONE v_mfma_f32_32x32x16_bf16 covers all 27
// filter taps of 32 output channels for 16 pixels, so the arithmetic is nothing (51k MFMAs per frame) and the kernel is
// bound by streaming dy once: HBM roofline, 52.5 MB of bf16 dy + 4.9 MB of frame per 480x854 frame.  (Round 1 ran the
// layer on the 64-channel kernel with a 16-channel zero-padded copy of the frame: 9 taps x 16 padded channels per MFMA
// step, a layout pass, 2 fold launches - 115 us per step.)
//
// Tile = 8 rows x 16 pixels.  dy is staged like in conv_wgrad.hip ([32-channel half][pixel][64 B], transposed reads).
// The frame halo (3 x 10 x 18 fp32) is rounded to bf16 (the same rounding the forward kernel's MFMA consumers see
// nowhere: conv1_1 forward is fp32 - the weight gradient of this one layer therefore carries one bf16 rounding of the
// frame, as in round 1) and written THREE times, shifted by kx, as B[kx][ci][halo row][16 pixels]: the B fragment of
// filter tap column n = ci*9 + ky*3 + kx for tile row r is then one aligned 16-byte read per lane.  Wave w owns tile
// rows 2w, 2w+1 (both co halves); the four waves' accumulators meet in LDS in a fixed order and the workgroup writes
// one slab [64][27] laid out like dw itself; the reduction over workgroups is conv_wgrad.hip's (queued with the other
// layers).
#include "common.hpp"

using namespace fosvos;

namespace {
constexpr int TH = 8, TPIX = 128, HALO_W = 18, NHALO = 3 * (TH + 2) * HALO_W;  // 540 frame values per tile
constexpr int CO = 64, NTAP = 27;
constexpr int Y_HALF = TPIX * 64 + 64, Y_BYTES = 2 * Y_HALF;   // dy image (see conv_wgrad.hip)
constexpr int B_BYTES = 3 * 3 * (TH + 2) * 16 * 2;             // B[kx][ci][hy][16] bf16 = 2880
constexpr int BUF_BYTES = Y_BYTES + B_BYTES;                   // 19392
constexpr int LDS_BYTES = 2 * BUF_BYTES > 4 * CO * 32 * 4 ? 2 * BUF_BYTES : 4 * CO * 32 * 4;  // tile images / final reduce

struct FirstArgs {
    const float *frame;  // [N,3,H,W]
    const uint16_t *dy;  // [N,H,W,64]
    float *slabs;        // [S][64][27]
    float *bias_part;    // [S][64] or null
    int N, H, W;
    int tiles_x, tiles_y, n_tiles, tiles_per_split;
    int newest_first;  // tile walk: 0 = a contiguous ascending range per workgroup; G > 0 = chunks of G tiles, newest first (see the kernel)
};

typedef __attribute__((address_space(3))) s16x4 *lds_s16x4_ptr;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ bf16x8 tr_pair(const char *p) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p + 256));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

__device__ uint4 g_zero16_f;

__global__ __launch_bounds__(256) void k_wgrad_first(const FirstArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_f[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int split = blockIdx.x;
    const int H = a.H, W = a.W;

    f32x16 acc[2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[h][r] = 0.f;
    float bsum[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bsum[e] = 0.f;
    const bool do_bias = a.bias_part != nullptr;

    // dy staging plan: piece i = it * 256 + tid -> pixel i / 8, 16-byte chunk i % 8 (4 pieces per thread)
    const int yc = tid & 7, ypix0 = tid >> 3;  // piece `it`: pixel ypix0 + 32 it = tile row (ypix0 >> 4) + 2 it
    const int y_ty0 = ypix0 >> 4, y_tx = ypix0 & 15;
    const int y_lds0 = (yc >> 2) * Y_HALF + ypix0 * 64 + (yc & 3) * 16;  // + it * 32 * 64
    // frame staging plan: value e = k * 256 + tid -> (ci, hy, hx)
    int f_ci[3], f_hy[3], f_hx[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int e = min(k * 256 + tid, NHALO - 1);
        f_ci[k] = e / ((TH + 2) * HALO_W);
        const int rem = e - f_ci[k] * (TH + 2) * HALO_W;
        f_hy[k] = rem / HALO_W;
        f_hx[k] = rem - f_hy[k] * HALO_W;
    }
    // fragment reads: A as in conv_wgrad.hip; B: lane (n = l & 31, h = l >> 5) reads 8 pixels of column n
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int rd_y = ((g >> 1) * 8 + q) * 64 + (g & 1) * 32 + p * 8;  // + half * Y_HALF + row * 16 * 64
    const int n_col = min(lane & 31, NTAP - 1);                       // columns 27..31 are padding: any in-bounds read
    const int n_ci = n_col / 9, n_ky = (n_col % 9) / 3, n_kx = n_col % 3;
    const int rd_b = Y_BYTES + ((n_kx * 3 + n_ci) * (TH + 2) + n_ky) * 32 + (lane >> 5) * 16;  // + row * 32

    // newest_first = G > 0: the tiles are taken in chunks of G consecutive ones (neighbours in a tile row: they share the
    // 128-byte lines of the fp32 frame rows and their halo columns), workgroup s takes chunks n_chunks - 1 - s, - S, ... -
    // every workgroup starts in the part of dy its producer (conv1_2's data gradient, which ends right in front of this
    // launch) wrote LAST, the part that may still sit in the memory-side cache, and all of them work their way back to the
    // oldest part together.  (With a contiguous range per workgroup the launch reads all of dy at once, most of it long
    // evicted.)  The loop below counts tile - t_begin.
    const int S_wg = gridDim.x, G_ch = a.newest_first;
    const int n_chunks = G_ch ? (a.n_tiles + G_ch - 1) / G_ch : 0;
    const int my_chunks = (G_ch && split < n_chunks) ? (n_chunks - split + S_wg - 1) / S_wg : 0;
    const int t_begin = G_ch ? 0 : split * a.tiles_per_split;
    const int t_end = G_ch ? my_chunks * G_ch - ((my_chunks && split == 0) ? n_chunks * G_ch - a.n_tiles : 0)
                           : min(t_begin + a.tiles_per_split, a.n_tiles);
    int lt_chunk = n_chunks - 1 - split, lt_j = 0;
    int lt_lin = G_ch ? max(lt_chunk, 0) * G_ch : t_begin;
    int lt_x = lt_lin % a.tiles_x, lt_y = (lt_lin / a.tiles_x) % a.tiles_y, lt_n = lt_lin / (a.tiles_x * a.tiles_y);

    uint4 py0, py1, py2, py3;
    float pf0, pf1, pf2;
    py0 = py1 = py2 = py3 = make_uint4(0, 0, 0, 0);
    pf0 = pf1 = pf2 = 0.f;
    const void *zero = &g_zero16_f;
#define FOSVOS_F_LDY(i_)                                                                                  \
    {                                                                                                     \
        const bool ok_ = y_ty0 + 2 * (i_) < vrows_ && y_tx < vcols_ && live_;                             \
        py##i_ = *reinterpret_cast<const uint4 *>(                                                        \
            ok_ ? (const void *)(ybase_ + ((int64_t)(y_ty0 + 2 * (i_)) * W + y_tx) * CO + yc * 8) : zero); \
    }
#define FOSVOS_F_LDF(k_)                                                                                  \
    {                                                                                                     \
        const int gy_ = y0_ + f_hy[k_] - 1, gx_ = x0_ + f_hx[k_] - 1;                                     \
        const bool ok_ = gy_ >= 0 && gy_ < H && gx_ >= 0 && gx_ < W && live_ && (k_) * 256 + tid < NHALO; \
        pf##k_ = *(ok_ ? fbase_ + ((int64_t)f_ci[k_] * H + gy_) * W + gx_ : reinterpret_cast<const float *>(zero)); \
    }
#define FOSVOS_F_LOAD_TILE()                                                                              \
    {                                                                                                     \
        const int y0_ = lt_y * TH, x0_ = lt_x * 16;                                                       \
        const int vrows_ = H - y0_, vcols_ = W - x0_;                                                     \
        const uint16_t *ybase_ = a.dy + (((int64_t)lt_n * H + y0_) * W + x0_) * CO;                       \
        const float *fbase_ = a.frame + (int64_t)lt_n * 3 * H * W;                                        \
        FOSVOS_F_LDY(0) FOSVOS_F_LDY(1) FOSVOS_F_LDY(2) FOSVOS_F_LDY(3)                                   \
        FOSVOS_F_LDF(0) FOSVOS_F_LDF(1) FOSVOS_F_LDF(2)                                                   \
        if (G_ch) {                                                                                       \
            ++lt_lin;                                                                                     \
            if (++lt_j == G_ch || lt_lin >= a.n_tiles) {                                                  \
                lt_chunk -= S_wg;                                                                         \
                lt_lin = max(lt_chunk, 0) * G_ch;                                                         \
                lt_j = 0;                                                                                 \
            }                                                                                             \
            lt_x = lt_lin % a.tiles_x;                                                                    \
            lt_y = (lt_lin / a.tiles_x) % a.tiles_y;                                                      \
            lt_n = lt_lin / (a.tiles_x * a.tiles_y);                                                      \
        } else if (++lt_x == a.tiles_x) {                                                                 \
            lt_x = 0;                                                                                     \
            if (++lt_y == a.tiles_y) { lt_y = 0; ++lt_n; }                                                \
        }                                                                                                 \
    }
#define FOSVOS_F_STF(k_, img_)                                                                            \
    if ((k_) * 256 + tid < NHALO) {                                                                       \
        const uint16_t v_ = f2bf(pf##k_);                                                                 \
        _Pragma("unroll") for (int kx = 0; kx < 3; ++kx) {                                                \
            const int c_ = f_hx[k_] - kx;                                                                 \
            if (c_ >= 0 && c_ < 16)                                                                       \
                *reinterpret_cast<uint16_t *>((img_) + Y_BYTES + (((kx * 3 + f_ci[k_]) * (TH + 2) + f_hy[k_]) * 16 + c_) * 2) = v_; \
        }                                                                                                 \
    }
#define FOSVOS_F_STORE_TILE(img_)                                                                         \
    {                                                                                                     \
        *reinterpret_cast<uint4 *>((img_) + y_lds0 + 0 * 32 * 64) = py0;                                  \
        *reinterpret_cast<uint4 *>((img_) + y_lds0 + 1 * 32 * 64) = py1;                                  \
        *reinterpret_cast<uint4 *>((img_) + y_lds0 + 2 * 32 * 64) = py2;                                  \
        *reinterpret_cast<uint4 *>((img_) + y_lds0 + 3 * 32 * 64) = py3;                                  \
        FOSVOS_F_STF(0, img_) FOSVOS_F_STF(1, img_) FOSVOS_F_STF(2, img_)                                 \
    }

    if (t_begin < t_end) {
        const bool live_ = true;
        FOSVOS_F_LOAD_TILE()
        FOSVOS_F_STORE_TILE(smem_f)
    }
    __syncthreads();
    for (int tile = t_begin; tile < t_end; ++tile) {
        char *cur = smem_f + ((tile - t_begin) & 1) * BUF_BYTES;
        char *nxt = smem_f + (((tile - t_begin) & 1) ^ 1) * BUF_BYTES;
        const bool live_ = tile + 1 < t_end;
        if (live_) FOSVOS_F_LOAD_TILE()
        if (do_bias) {  // column sums of the dy tile: thread t keeps chunk t % 8 (8 channels), pixels t / 8 + 32 k
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float f[8];
                unpack8(*reinterpret_cast<const uint4 *>(cur + y_lds0 + k * 32 * 64), f);
#pragma unroll
                for (int e = 0; e < 8; ++e) bsum[e] += f[e];
            }
        }
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {  // this wave's two tile rows
            const int row = 2 * wave + rr;
            const bf16x8 bfr = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(cur + rd_b + row * 32));
            const bf16x8 a_lo = tr_pair(cur + rd_y + row * 16 * 64);
            const bf16x8 a_hi = tr_pair(cur + rd_y + Y_HALF + row * 16 * 64);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo, bfr, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, bfr, acc[1], 0, 0, 0);
        }
        if (live_) FOSVOS_F_STORE_TILE(nxt)
        __syncthreads();
    }
    // ---- the four waves' partial sums meet in LDS: red[wave][co][32 columns], summed in wave order
    float *red = reinterpret_cast<float *>(smem_f);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = h * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            red[(wave * CO + co) * 32 + (lane & 31)] = acc[h][r];
        }
    __syncthreads();
    float *slab = a.slabs + (int64_t)split * CO * NTAP;
    for (int e = tid; e < CO * NTAP; e += 256) {
        const int co = e / NTAP, n = e - co * NTAP;
        float v = red[(0 * CO + co) * 32 + n];
        v += red[(1 * CO + co) * 32 + n];
        v += red[(2 * CO + co) * 32 + n];
        v += red[(3 * CO + co) * 32 + n];
        slab[e] = v;
    }
    if (do_bias) {
        __syncthreads();
        float *sb = reinterpret_cast<float *>(smem_f);  // [256][9] floats
#pragma unroll
        for (int e = 0; e < 8; ++e) sb[tid * 9 + e] = bsum[e];
        __syncthreads();
        if (tid < CO) {
            const int c = tid >> 3, e = tid & 7;
            float acc_b = 0.f;
            for (int t2 = c; t2 < 256; t2 += 8) acc_b += sb[t2 * 9 + e];
            a.bias_part[(int64_t)split * CO + tid] = acc_b;
        }
    }
}

struct FirstPlan {
    int tiles_x, tiles_y, n_tiles, tps, S;
    size_t slab_bytes, bias_bytes;
};
FirstPlan make_first_plan(int N, int H, int W) {
    FirstPlan p;
    p.tiles_x = (int)cdiv(W, 16);
    p.tiles_y = (int)cdiv(H, TH);
    p.n_tiles = p.tiles_x * p.tiles_y * N;
    int S = 512;  // two workgroups per CU (19 KB of LDS x 2 images each): the kernel only streams
    if (S > p.n_tiles) S = p.n_tiles;
    p.tps = (int)cdiv(p.n_tiles, S);
    p.S = (int)cdiv(p.n_tiles, p.tps);
    p.slab_bytes = ((size_t)p.S * CO * NTAP * sizeof(float) + 255) / 256 * 256;
    p.bias_bytes = (size_t)p.S * CO * sizeof(float);
    return p;
}
}  // namespace

extern "C" size_t fosvos_conv3x3_first_wgrad_workspace_bytes(int N, int H, int W, int Co) {
    if (N <= 0 || H <= 0 || W <= 0 || Co != CO) return 0;
    const FirstPlan p = make_first_plan(N, H, W);
    return p.slab_bytes + p.bias_bytes;
}

extern "C" int fosvos_conv3x3_first_wgrad(const float *frame, const uint16_t *dy, float *dw, float *db, int N, int H,
                                          int W, int Co, void *workspace, size_t workspace_bytes, int device,
                                          void *stream) {
    return fosvos::first_wgrad_impl(frame, dy, dw, db, N, H, W, Co, 0, workspace, workspace_bytes, device, stream, nullptr);
}

int fosvos::first_wgrad_impl(const float *frame, const uint16_t *dy, float *dw, float *db, int N, int H, int W, int Co,
                             int accumulate, void *workspace, size_t workspace_bytes, int device, void *stream,
                             WgradReduceTable *reduce) {
    FOSVOS_REQUIRE(frame && dy && dw && workspace, FOSVOS_E_ARG, "conv3x3_first_wgrad: null pointer");
    FOSVOS_REQUIRE(Co == CO, FOSVOS_E_SHAPE, "conv3x3_first_wgrad: Co=%d, only %d is built", Co, CO);
    FOSVOS_REQUIRE(N > 0 && H > 0 && W > 0, FOSVOS_E_SHAPE, "conv3x3_first_wgrad: bad shape N=%d H=%d W=%d", N, H, W);
    FOSVOS_REQUIRE((int64_t)H * W * CO < 0x7fffffffLL, FOSVOS_E_SHAPE, "conv3x3_first_wgrad: one image exceeds 2^31 elements");
    const FirstPlan p = make_first_plan(N, H, W);
    const size_t need = p.slab_bytes + p.bias_bytes;
    FOSVOS_REQUIRE(workspace_bytes >= need, FOSVOS_E_WORKSPACE, "conv3x3_first_wgrad: workspace %zu < %zu",
                   workspace_bytes, need);
    FOSVOS_ENTER(device);
    hipStream_t st = (hipStream_t)stream;
    char *ws = reinterpret_cast<char *>(workspace);
    FirstArgs a;
    a.frame = frame; a.dy = dy; a.slabs = reinterpret_cast<float *>(ws);
    a.bias_part = db ? reinterpret_cast<float *>(ws + p.slab_bytes) : nullptr;
    a.N = N; a.H = H; a.W = W;
    a.tiles_x = p.tiles_x; a.tiles_y = p.tiles_y; a.n_tiles = p.n_tiles; a.tiles_per_split = p.tps;
    a.newest_first = lab_env_int("FOSVOS_WGRAD_FIRST_ORDER", 4);  // lab switch: 0 = contiguous ranges, G = chunk size
    FOSVOS_PROF("k_wgrad_first", st, 2.0 * N * H * W * 27.0 * CO);
    hipLaunchKernelGGL(k_wgrad_first, dim3((unsigned)p.S), dim3(256), LDS_BYTES, st, a);
    FOSVOS_LAUNCH_CHECK();
    return fosvos::wgrad_queue_reduce(a.slabs, a.bias_part, dw, db, p.S, p.S, (int64_t)CO * NTAP, (int64_t)CO * NTAP, CO, CO,
                                      accumulate, reduce, device, stream);
}
