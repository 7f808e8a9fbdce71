// Weight / bias gradient of the 3x3 conv on bf16 NHWC activations (autograd's backward-weight of
// Conv2d in stages[*] / side_prep[*], src/networks/osvos_vgg.py:42,92).
//
//   dw[co][ci][tap] = sum_{n,y,x} dy[n,y,x,co] * x[n,y+ky-1,x+kx-1,ci]        db[co] = sum dy[.,co]
//
// GEMM view: M = co, N = ci (x 9 taps), K = pixels.  Both operands are stored channel-contiguous
// (NHWC) while the contraction runs over pixels, so both MFMA operands need a transpose: the tiles
// are staged as [16-channel block][pixel][16 ch] (32-byte rows) and read with
// ds_read_b64_tr_b16, which hands lane (i) the 4 pixels x channel i column of a 4x16 block.
// A k-step is 2 image rows x 16 pixels; lane group g takes pixels 4g..4g+3 of the first row
// (k elements 0-3) and of the second row (k elements 4-7).  A and B use the same assignment, and a
// 32-lane half reads 8 consecutive 32-byte rows = one 256-byte bank row: conflict-free for every tap.
//
// Workgroup = 4 waves: co block (64 or 16) shared by all waves, wave w owns ci 16w..16w+15 of a
// 64-wide ci block, all 9 taps: 4 x 9 accumulators (144 VGPRs).  The pixel range is split over
// blockIdx.x; each split writes one fp32 slab [tap][co][ci]; k_wgrad_fold / k_wgrad_final sum the slabs in
// a fixed order, so results are bitwise reproducible (no float atomics).
#include "common.hpp"

using namespace fosvos;

namespace {
constexpr int TH = 8;           // tile rows (4 k-steps)
constexpr int TPIX = TH * 16;   // 128 tile pixels
constexpr int HALO_W = 18;
constexpr int NPH = (TH + 2) * HALO_W;  // 180 halo pixels
constexpr int BCI = 64;

struct WgArgs {
    const uint16_t *x;   // [N,H,W,Ci]
    const uint16_t *dy;  // [N,H,W,Cy]  (Cy = roundup(Co,32))
    float *slabs;        // [S][9][Cor][Ci]
    float *bias_part;    // [S][Cor] per-split column sums of dy, or null
    int N, H, W, Ci, Cy, Cor;
    int tiles_x, tiles_y, n_tiles, tiles_per_split;
};

typedef __attribute__((address_space(3))) s16x4 *lds_s16x4_ptr;

__device__ __forceinline__ bf16x8 tr_pair(const char *p0, const char *p1) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p1));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

// 16 zero bytes every out-of-image lane of an LDS-DMA load reads instead (DMA cannot synthesise padding)
__device__ uint4 g_zero16;

typedef __attribute__((address_space(3))) void *lds_void_ptr;
typedef const __attribute__((address_space(1))) void *glb_void_ptr;

// One wave instruction: lane l moves the 16 bytes at `src` to LDS `dst_wave_base + 16 l` without touching a VGPR.
__device__ __forceinline__ void dma16(const void *src, char *dst_wave_base) {
    __builtin_amdgcn_global_load_lds((glb_void_ptr)src, (lds_void_ptr)dst_wave_base, 16, 0, 0);
}

// Tile pipeline: LDS-DMA (global_load_lds_dwordx4) into a DOUBLE-buffered pair of tile images, one barrier
// per tile: while the MFMAs of tile t read image t&1, the DMA of tile t+1 lands in the other image; the
// __syncthreads() at the end of the iteration drains it (hipcc emits vmcnt(0) in front of the barrier while a DMA
// is in flight) and retires image t&1.  No prefetch registers, no zero-selects, no ds_write phase.
// A DMA instruction writes 64 consecutive 16-byte slots = 32 pixel rows of one 16-channel block, which is
// exactly how the images are laid out ([block][pixel][32 B]); lane l <-> (pixel l>>1, half l&1).  Wave w moves
// dy pixel group w (tile rows 2w, 2w+1) for every co block and the WHOLE x halo block w - the one it consumes.
constexpr int NPHP = (NPH + 31) / 32 * 32;  // 192: x image pixel count padded to whole DMA instructions

// KSPLIT (the 16-channel-padded conv1_1 case, Ci = 16 = ONE ci fragment): the four waves share x block 0 and
// split the tile's k-steps instead (wave w = k-step w); each wave writes its own slab (slab index split*4 + w).
template <int BCO, bool KSPLIT = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void k_wgrad(const WgArgs a) {
    constexpr int CF = BCO / 16;
    constexpr int XB = KSPLIT ? 1 : 4;  // x channel blocks per image
    constexpr int Y_BYTES = CF * TPIX * 32, X_BYTES = XB * NPHP * 32, BUF_BYTES = Y_BYTES + X_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem_w[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
    const int split = blockIdx.x;
    const int ci0 = KSPLIT ? 0 : blockIdx.y * BCI;
    const int co0 = blockIdx.z * BCO;
    const int H = a.H, W = a.W;
    const int xblk = KSPLIT ? 0 : wave;  // x channel block this wave consumes

    f32x4 acc[CF][9];
#pragma unroll
    for (int i = 0; i < CF; ++i)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // per-lane byte offsets inside a tile image for the transposed reads
    const int rd_y = (4 * g + q) * 32 + p * 8;                          // + (cb*TPIX + row*16) * 32
    const int rd_x = Y_BYTES + (xblk * NPHP + 4 * g + q) * 32 + p * 8;  // + ((row+ky)*18 + kx) * 32

    const bool do_bias = a.bias_part != nullptr && blockIdx.y == 0;
    float bsum[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bsum[e] = 0.f;

    // ---- tile-independent DMA descriptors of this lane
    const int half = lane & 1, pl = lane >> 1;  // pl: pixel inside a 32-pixel DMA group
    // dy: pixel group = wave -> tile row 2*wave + (pl>>4), column pl&15
    const int y_ty = 2 * wave + (pl >> 4), y_tx = pl & 15;
    const int y_off = (y_ty * W + y_tx) * a.Cy + co0 + half * 8;  // + cb*16 per instruction
    // x: 6 groups of 32 halo pixels of block `wave`
    int x_off[6], x_rc[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int pix = k * 32 + pl;
        const int hy = pix / HALO_W, hx = pix - hy * HALO_W;
        x_rc[k] = pix < NPH ? ((hy << 16) | hx) : (0x7fff << 16);  // padding slots: never "in image"
        x_off[k] = ((hy - 1) * W + (hx - 1)) * a.Ci + ci0 + xblk * 16 + half * 8;
    }
    const void *zero = &g_zero16;

    const int t_begin = split * a.tiles_per_split;
    const int t_end = min(t_begin + a.tiles_per_split, a.n_tiles);
    int lt_x = t_begin % a.tiles_x, lt_y = (t_begin / a.tiles_x) % a.tiles_y, lt_n = t_begin / (a.tiles_x * a.tiles_y);

    auto issue_tile = [&](char *img) {
        const int y0 = lt_y * TH, x0 = lt_x * 16;
        const int vrows = H - y0, vcols = W - x0;
        const int64_t org = ((int64_t)lt_n * H + y0) * W + x0;
        const uint16_t *ybase = a.dy + org * a.Cy;
        const uint16_t *xbase = a.x + org * a.Ci;
        const bool y_ok = y_ty < vrows && y_tx < vcols;
#pragma unroll
        for (int cb = 0; cb < CF; ++cb)
            dma16(y_ok ? (const void *)(ybase + y_off + cb * 16) : zero, img + (cb * TPIX + wave * 32) * 32);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            if (KSPLIT && (k & 3) != wave) continue;  // one shared block: its 6 groups are dealt over the 4 waves
            const int hy = x_rc[k] >> 16, hx = x_rc[k] & 0xffff;
            const bool ok = hy >= 1 - y0 && hy <= vrows && hx >= 1 - x0 && hx <= vcols;
            dma16(ok ? (const void *)(xbase + x_off[k]) : zero, img + Y_BYTES + (xblk * NPHP + k * 32) * 32);
        }
        if (++lt_x == a.tiles_x) {
            lt_x = 0;
            if (++lt_y == a.tiles_y) { lt_y = 0; ++lt_n; }
        }
    };

    // Register-staged tile pipeline (the LDS-DMA form above is kept for reference builds with more than 4 co blocks): an
    // LDS-DMA piece costs the issuing wave ~100 clocks (1400 per tile, measured with a timing-only build), a
    // global_load_dwordx4 + ds_write_b128 pair a third of that.  Tile t+1 is loaded into 4 + 6 named registers at the
    // top of tile t (latency hidden by the MFMA loop) and written to the other image behind it.
    constexpr bool REGSTAGE = CF <= 4;
    uint4 py0, py1, py2, py3, px0, px1, px2, px3, px4, px5;
    py0 = py1 = py2 = py3 = px0 = px1 = px2 = px3 = px4 = px5 = make_uint4(0, 0, 0, 0);
#define FOSVOS_WG_LDY(cb_) \
    if constexpr ((cb_) < CF) py##cb_ = *reinterpret_cast<const uint4 *>(y_ok_ ? (const void *)(ybase_ + y_off + (cb_) * 16) : zero);
#define FOSVOS_WG_LDX(k_)                                                                               \
    if (!KSPLIT || ((k_) & 3) == wave) { /* KSPLIT: the shared block's 6 groups are dealt over the 4 waves */ \
        const int hy_ = x_rc[k_] >> 16, hx_ = x_rc[k_] & 0xffff;                                        \
        const bool ok_ = hy_ >= 1 - y0_ && hy_ <= vrows_ && hx_ >= 1 - x0_ && hx_ <= vcols_;            \
        px##k_ = *reinterpret_cast<const uint4 *>(ok_ ? (const void *)(xbase_ + x_off[k_]) : zero);     \
    }
#define FOSVOS_WG_LOAD_TILE()                                                                           \
    {                                                                                                   \
        const int y0_ = lt_y * TH, x0_ = lt_x * 16;                                                     \
        const int vrows_ = H - y0_, vcols_ = W - x0_;                                                   \
        const int64_t org_ = ((int64_t)lt_n * H + y0_) * W + x0_;                                       \
        const uint16_t *ybase_ = a.dy + org_ * a.Cy;                                                    \
        const uint16_t *xbase_ = a.x + org_ * a.Ci;                                                     \
        const bool y_ok_ = y_ty < vrows_ && y_tx < vcols_;                                              \
        FOSVOS_WG_LDY(0) FOSVOS_WG_LDY(1) FOSVOS_WG_LDY(2) FOSVOS_WG_LDY(3)                             \
        FOSVOS_WG_LDX(0) FOSVOS_WG_LDX(1) FOSVOS_WG_LDX(2) FOSVOS_WG_LDX(3) FOSVOS_WG_LDX(4) FOSVOS_WG_LDX(5) \
        if (++lt_x == a.tiles_x) {                                                                      \
            lt_x = 0;                                                                                   \
            if (++lt_y == a.tiles_y) { lt_y = 0; ++lt_n; }                                              \
        }                                                                                               \
    }
#define FOSVOS_WG_STY(cb_, img_) \
    if constexpr ((cb_) < CF) *reinterpret_cast<uint4 *>((img_) + ((cb_) * TPIX + wave * 32) * 32 + lane * 16) = py##cb_;
#define FOSVOS_WG_STX(k_, img_)       \
    if (!KSPLIT || ((k_) & 3) == wave) \
        *reinterpret_cast<uint4 *>((img_) + Y_BYTES + (xblk * NPHP + (k_) * 32) * 32 + lane * 16) = px##k_;
#define FOSVOS_WG_STORE_TILE(img_)                                                                      \
    {                                                                                                   \
        FOSVOS_WG_STY(0, img_) FOSVOS_WG_STY(1, img_) FOSVOS_WG_STY(2, img_) FOSVOS_WG_STY(3, img_)     \
        FOSVOS_WG_STX(0, img_) FOSVOS_WG_STX(1, img_) FOSVOS_WG_STX(2, img_) FOSVOS_WG_STX(3, img_)     \
        FOSVOS_WG_STX(4, img_) FOSVOS_WG_STX(5, img_)                                                   \
    }

    if constexpr (REGSTAGE) {
        if (t_begin < t_end) {
            FOSVOS_WG_LOAD_TILE()
            FOSVOS_WG_STORE_TILE(smem_w)
        }
    } else {
        if (t_begin < t_end) issue_tile(smem_w);
    }
    __syncthreads();
    for (int tile = t_begin; tile < t_end; ++tile) {
        char *cur = smem_w + ((tile - t_begin) & 1) * BUF_BYTES;
        char *nxt = smem_w + (((tile - t_begin) & 1) ^ 1) * BUF_BYTES;
        if constexpr (REGSTAGE) {
            if (tile + 1 < t_end) FOSVOS_WG_LOAD_TILE()
        } else {
            if (tile + 1 < t_end) issue_tile(nxt);
        }
        if (do_bias) {  // column sums of the dy tile: thread (cb = tid>>6, slot tid&63) keeps channels (cb, half) fixed
            if ((tid >> 6) < CF) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float f[8];
                    unpack8(*reinterpret_cast<const uint4 *>(cur + ((tid >> 6) * 256 + (tid & 63) + 64 * i) * 16), f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) bsum[e] += f[e];
                }
            }
        }
#pragma unroll
        for (int ks = 0; ks < TH / 2; ++ks) {
            if (KSPLIT && ks != wave) continue;
            bf16x8 af[CF];
#pragma unroll
            for (int i = 0; i < CF; ++i) {
                const char *base = cur + rd_y + (i * TPIX + ks * 32) * 32;
                af[i] = tr_pair(base, base + 16 * 32);
            }
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap % 3;
                const char *base = cur + rd_x + ((2 * ks + ky) * HALO_W + kx) * 32;
                const bf16x8 bfr = tr_pair(base, base + HALO_W * 32);
#pragma unroll
                for (int i = 0; i < CF; ++i)
                    acc[i][tap] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr, acc[i][tap], 0, 0, 0);
            }
        }
        if constexpr (REGSTAGE) {
            // image `nxt` was last read during tile-1 (every wave has passed that tile's barrier)
            if (tile + 1 < t_end) FOSVOS_WG_STORE_TILE(nxt)
        }
        __syncthreads();  // tile+1 is in place (DMA path: drains it, vmcnt(0)) and image `cur` is retired
    }
    if (do_bias) {
        // 256 x 8 partials -> BCO channel sums in a fixed order: thread t owns channels (cb = t>>6, half = t&1)
        float *sb = reinterpret_cast<float *>(smem_w);  // [256][9] floats
#pragma unroll
        for (int e = 0; e < 8; ++e) sb[tid * 9 + e] = bsum[e];
        __syncthreads();
        if (tid < BCO) {
            const int cb = tid >> 4, hf = (tid >> 3) & 1, e = tid & 7;
            float acc_b = 0.f;
            for (int t2 = cb * 64 + hf; t2 < cb * 64 + 64; t2 += 2) acc_b += sb[t2 * 9 + e];
            a.bias_part[(int64_t)split * a.Cor + co0 + tid] = acc_b;
        }
    }
    // ---- slab write in FRAGMENT-NATIVE order: a lane's 4 accumulator registers (4 consecutive co of one ci)
    // go out as one 16-byte store, 1 KB contiguous per wave instruction (4-byte [tap][co][ci] stores were
    // store-issue-bound: ~20 us per launch).  Element (tap, co, ci) of a split's slab lives at
    //   (((cb * (Ci/16) + ci/16) * 9 + tap) * BCO*16) + (((co%BCO)/16 * 4 + (co%16)/4) * 16 + ci%16) * 4 + co%4
    // with cb = co / BCO; k_wgrad_final undoes the permutation.
    float *slab = a.slabs + (int64_t)(KSPLIT ? split * 4 + wave : split) * 9 * a.Cor * a.Ci;
    const int wci = KSPLIT ? 0 : blockIdx.y * 4 + wave;  // global 16-wide ci fragment
    float *blk = slab + ((int64_t)blockIdx.z * (a.Ci / 16) + wci) * 9 * (BCO * 16);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int i = 0; i < CF; ++i)
            *reinterpret_cast<f32x4 *>(blk + tap * (BCO * 16) + (i * 64 + lane) * 4) = acc[i][tap];
}

// ---- slab reduction, two fully coalesced stages
// stage 1 (only when S > kFoldTo): fold the S slabs into kFoldTo partial slabs, slab s -> partial s % kFoldTo,
// each summed in increasing s (fixed order).  float4 per thread, grid = (E/1024, kFoldTo).
constexpr int kFoldTo = 8;

__device__ __forceinline__ void fold_body(const float *__restrict__ slabs, int S, int64_t E,
                                          float *__restrict__ folded, int64_t bx, int y) {
    const int64_t i4 = bx * 256LL + threadIdx.x;  // float4 index
    if (i4 * 4 >= E) return;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = y; s < S; s += kFoldTo) {
        const float4 v = *reinterpret_cast<const float4 *>(slabs + (int64_t)s * E + i4 * 4);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    *reinterpret_cast<float4 *>(folded + (int64_t)y * E + i4 * 4) = acc;
}

__global__ __launch_bounds__(256) void k_wgrad_fold(const float *__restrict__ slabs, int S, int64_t E,
                                                     float *__restrict__ folded) {
    fold_body(slabs, S, E, folded, blockIdx.x, blockIdx.y);
}

// stage 2: block = (4 consecutive co = one accumulator float4, 64 consecutive ci): sum the <= kFoldTo slabs
// (16-byte coalesced reads of the fragment-native layout), transpose through LDS and write 4 runs of 576
// contiguous floats dw[(co*Ci + ci0)*9 ...] (OIHW).
__device__ __forceinline__ void final_body(const float *__restrict__ slabs, int S, int Co, int Cor, int Ci, int Ci_real,
                                           int bco, int accumulate, float *__restrict__ dw,
                                           const float *__restrict__ bias_part, int S_bias, float *__restrict__ db,
                                           int bx, int by) {
    // Ci = channel count of the slabs (a multiple of 64, or 16 for the padded conv1_1 image); Ci_real = channels
    // of dw (OIHW rows of Ci_real*9 floats)
    __shared__ float tile[4][64 * 9];
    const int cw = Ci < 64 ? Ci : 64;  // ci columns handled by this block
    const int ci0 = bx * 64, co4 = by * 4;
    if (db && bx == 0) {  // the 4 channels' bias: S_bias per-split partials each, fixed-order tree
        __shared__ float redb[256];
        const int r = threadIdx.x >> 6, t = threadIdx.x & 63;
        float acc_b = 0.f;
        if (co4 + r < Co)
            for (int s = t; s < S_bias; s += 64) acc_b += bias_part[(int64_t)s * Cor + co4 + r];
        redb[threadIdx.x] = acc_b;
        __syncthreads();
#pragma unroll
        for (int w = 32; w > 0; w >>= 1) {
            if (t < w) redb[threadIdx.x] += redb[threadIdx.x + w];
            __syncthreads();
        }
        if (t == 0 && co4 + r < Co) db[co4 + r] = accumulate ? db[co4 + r] + redb[threadIdx.x] : redb[threadIdx.x];
    }
    const int64_t E = 9LL * Cor * Ci;
    const int cb = co4 / bco, cf = (co4 % bco) / 16, g = (co4 % 16) / 4;
    for (int e = threadIdx.x; e < 9 * cw; e += 256) {
        const int tap = e / cw, cil = e - tap * cw;
        const int ci = ci0 + cil;
        const float *p = slabs + (((int64_t)cb * (Ci / 16) + ci / 16) * 9 + tap) * (bco * 16) +
                         ((cf * 4 + g) * 16 + (ci & 15)) * 4;
        float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int s = 0; s < S; ++s) {
            const float4 v = *reinterpret_cast<const float4 *>(p + (int64_t)s * E);
            a4.x += v.x; a4.y += v.y; a4.z += v.z; a4.w += v.w;
        }
        tile[0][cil * 9 + tap] = a4.x;
        tile[1][cil * 9 + tap] = a4.y;
        tile[2][cil * 9 + tap] = a4.z;
        tile[3][cil * 9 + tap] = a4.w;
    }
    __syncthreads();
    const int wr = (Ci_real - ci0 < cw ? Ci_real - ci0 : cw) * 9;  // floats of this block that exist in dw
    for (int e = threadIdx.x; e < 4 * 9 * cw; e += 256) {
        const int r = e / (9 * cw), k = e - r * (9 * cw);
        if (co4 + r >= Co || k >= wr) continue;
        float *d = dw + ((int64_t)(co4 + r) * Ci_real + ci0) * 9 + k;
        *d = accumulate ? *d + tile[r][k] : tile[r][k];
    }
}

__global__ __launch_bounds__(256) void k_wgrad_final(const float *__restrict__ slabs, int S, int Co, int Cor, int Ci,
                                                      int Ci_real, int bco, int accumulate, float *__restrict__ dw,
                                                      const float *__restrict__ bias_part, int S_bias,
                                                      float *__restrict__ db) {
    final_body(slabs, S, Co, Cor, Ci, Ci_real, bco, accumulate, dw, bias_part, S_bias, db, blockIdx.x, blockIdx.y);
}

// ---- the same two stages for MANY layers in one launch each (vgg_net.hip queues every layer's slabs and reduces them
// behind the last MFMA kernel: 2 launches instead of 26 on the weight-gradient stream, and the small layers'
// reductions run beside the large ones)
__global__ __launch_bounds__(256) void k_wgrad_fold_all(const WgradReduceTable t) {
    int e = 0;
    while (e + 1 < t.n && (int)blockIdx.x >= t.e[e + 1].fold_begin) ++e;
    const WgradReduceEntry &q = t.e[e];
    const int local = blockIdx.x - q.fold_begin;
    if (local >= q.fold_blocks) return;  // entries without a fold stage own no blocks
    const int64_t E = 9LL * q.Cor * q.Ci;
    const int bx_count = q.fold_blocks / kFoldTo;
    fold_body(q.slabs, q.S, E, q.slabs + (int64_t)q.S * E, local % bx_count, local / bx_count);
}

__global__ __launch_bounds__(256) void k_wgrad_final_all(const WgradReduceTable t) {
    int e = 0;
    while (e + 1 < t.n && (int)blockIdx.x >= t.e[e + 1].final_begin) ++e;
    const WgradReduceEntry &q = t.e[e];
    const int local = blockIdx.x - q.final_begin;
    const int64_t E = 9LL * q.Cor * q.Ci;
    const bool folded = q.fold_blocks > 0;
    const int nx = q.Ci < 64 ? 1 : q.Ci / 64;
    final_body(folded ? q.slabs + (int64_t)q.S * E : q.slabs, folded ? kFoldTo : q.S, q.Co, q.Cor, q.Ci, q.Ci_real, q.bco,
               q.accumulate, q.dw, q.bias_part, q.S_bias, q.db, local % nx, local / nx);
}

// Workgroups per launch.  Slab traffic = workgroups x block bytes, and the wgrad kernels run on the auxiliary
// stream BESIDE the dgrad chain (vgg_net.hip), so they need not fill the chip alone: 128 fat workgroups halve the
// slab bytes of 256 at the same step time; 64 make the auxiliary stream the critical path (measured).
constexpr int kTargetBlocks = 128;
constexpr int kMainBCO = 64;  // co block of the main instantiation: 2 co fragments x 9 taps = 72 accumulator VGPRs

struct Plan {
    int Cor, Cy, bco, tiles_x, tiles_y, n_tiles, tps, S;
    size_t slab_bytes, bias_bytes;
};

Plan make_plan(int N, int H, int W, int Ci, int Co) {
    Plan p;
    p.Cor = roundup(Co, 16);
    p.Cy = roundup(Co, 32);
    p.bco = (p.Cor % 32 == 0) ? kMainBCO : 16;
    p.tiles_x = (int)cdiv(W, 16);
    p.tiles_y = (int)cdiv(H, TH);
    p.n_tiles = p.tiles_x * p.tiles_y * N;
    const int out_blocks = (p.Cor / p.bco) * (Ci / BCI);
    int S = (int)cdiv(kTargetBlocks, out_blocks);
    if (S > p.n_tiles) S = p.n_tiles;
    if (S < 1) S = 1;
    p.tps = (int)cdiv(p.n_tiles, S);
    p.S = (int)cdiv(p.n_tiles, p.tps);
    // S slabs + kFoldTo folded slabs when a fold stage is needed
    p.slab_bytes = (size_t)(p.S + (p.S > kFoldTo ? kFoldTo : 0)) * 9 * p.Cor * Ci * sizeof(float);
    p.bias_bytes = (size_t)p.S * p.Cor * sizeof(float);
    return p;
}
}  // namespace

extern "C" size_t fosvos_conv3x3_wgrad_workspace_bytes(int N, int H, int W, int Ci, int Co) {
    if (N <= 0 || H <= 0 || W <= 0 || Ci <= 0 || Co <= 0 || Ci % BCI != 0) return 0;
    const Plan p = make_plan(N, H, W, Ci, Co);
    return p.slab_bytes + p.bias_bytes;
}

extern "C" int fosvos_conv3x3_wgrad(const uint16_t *x, const uint16_t *dy, float *dw, float *db, int N, int H, int W,
                                    int Ci, int Co, int accumulate, void *workspace, size_t workspace_bytes, int device,
                                    void *stream) {
    return fosvos::wgrad_impl(x, dy, dw, db, N, H, W, Ci, Co, accumulate, workspace, workspace_bytes, device, stream,
                              nullptr);
}

namespace {
// Run (reduce == nullptr) or queue the fold / final passes of one layer whose slabs the MFMA kernel just wrote.
int finish_or_queue(const WgradReduceEntry &q0, WgradReduceTable *reduce, hipStream_t st) {
    WgradReduceEntry q = q0;
    const int64_t E = 9LL * q.Cor * q.Ci;
    const int nx = q.Ci < 64 ? 1 : q.Ci / 64;
    q.fold_blocks = q.S > kFoldTo ? (int)cdiv(E / 4, 256) * kFoldTo : 0;
    q.final_blocks = nx * (q.Cor / 4);
    if (reduce) {
        FOSVOS_REQUIRE(reduce->n < 20, FOSVOS_E_ARG, "wgrad: reduction queue full");
        const WgradReduceEntry *prev = reduce->n ? &reduce->e[reduce->n - 1] : nullptr;
        q.fold_begin = prev ? prev->fold_begin + prev->fold_blocks : 0;
        q.final_begin = prev ? prev->final_begin + prev->final_blocks : 0;
        reduce->e[reduce->n++] = q;
        return FOSVOS_OK;
    }
    const float *src = q.slabs;
    int n_src = q.S;
    if (q.fold_blocks) {
        float *folded = q.slabs + (int64_t)q.S * E;
        hipLaunchKernelGGL(k_wgrad_fold, dim3((unsigned)cdiv(E / 4, 256), kFoldTo), dim3(256), 0, st, q.slabs, q.S, E,
                           folded);
        FOSVOS_LAUNCH_CHECK();
        src = folded;
        n_src = kFoldTo;
    }
    hipLaunchKernelGGL(k_wgrad_final, dim3((unsigned)nx, (unsigned)(q.Cor / 4)), dim3(256), 0, st, src, n_src, q.Co, q.Cor,
                       q.Ci, q.Ci_real, q.bco, q.accumulate, q.dw, q.bias_part, q.S_bias, q.db);
    FOSVOS_LAUNCH_CHECK();
    return FOSVOS_OK;
}
}  // namespace

namespace {
WgradReduceEntry make_entry(const Plan &p, void *workspace, float *dw, float *db, int Ci, int Co, int accumulate) {
    WgradReduceEntry q{};
    q.slabs = reinterpret_cast<float *>(workspace);
    q.dw = dw;
    q.db = db;
    q.bias_part = db ? reinterpret_cast<float *>(reinterpret_cast<char *>(workspace) + p.slab_bytes) : nullptr;
    q.S = p.S; q.S_bias = p.S; q.Co = Co; q.Cor = p.Cor; q.Ci = Ci; q.Ci_real = Ci; q.bco = p.bco; q.accumulate = accumulate;
    return q;
}
}  // namespace

extern "C" int fosvos_conv3x3_wgrad_reduce(float *dw, float *db, int N, int H, int W, int Ci, int Co, int accumulate,
                                           void *workspace, size_t workspace_bytes, int device, void *stream) {
    FOSVOS_REQUIRE(dw && workspace, FOSVOS_E_ARG, "conv3x3_wgrad_reduce: null pointer");
    FOSVOS_REQUIRE(N > 0 && H > 0 && W > 0 && Ci > 0 && Co > 0 && Ci % BCI == 0 && (Co % 64 == 0 || Co == 16), FOSVOS_E_SHAPE,
                   "conv3x3_wgrad_reduce: bad shape");
    const Plan p = make_plan(N, H, W, Ci, Co);
    FOSVOS_REQUIRE(workspace_bytes >= p.slab_bytes + p.bias_bytes, FOSVOS_E_WORKSPACE,
                   "conv3x3_wgrad_reduce: workspace %zu < %zu", workspace_bytes, p.slab_bytes + p.bias_bytes);
    FOSVOS_ENTER(device);
    return finish_or_queue(make_entry(p, workspace, dw, db, Ci, Co, accumulate), nullptr, (hipStream_t)stream);
}

extern "C" int fosvos_conv3x3_wgrad_slabs(const uint16_t *x, const uint16_t *dy, int with_bias, int N, int H, int W, int Ci,
                                          int Co, void *workspace, size_t workspace_bytes, int device, void *stream) {
    // the queue form of wgrad_impl with a throw-away queue: the MFMA kernel runs, the reduction is left to the caller
    WgradReduceTable sink;
    sink.n = 0;
    float dummy = 0.f;  // never dereferenced on the host; only "non-null" matters for the bias partials
    return fosvos::wgrad_impl(x, dy, &dummy, with_bias ? &dummy : nullptr, N, H, W, Ci, Co, 0, workspace, workspace_bytes,
                              device, stream, &sink);
}

int fosvos::wgrad_reduce_all(WgradReduceTable *reduce, int device, void *stream) {
    FOSVOS_REQUIRE(reduce, FOSVOS_E_ARG, "wgrad_reduce_all: null table");
    if (reduce->n == 0) return FOSVOS_OK;
    FOSVOS_ENTER(device);
    hipStream_t st = (hipStream_t)stream;
    const WgradReduceEntry &last = reduce->e[reduce->n - 1];
    const int fold_total = last.fold_begin + last.fold_blocks, final_total = last.final_begin + last.final_blocks;
    if (fold_total > 0) {
        hipLaunchKernelGGL(k_wgrad_fold_all, dim3((unsigned)fold_total), dim3(256), 0, st, *reduce);
        FOSVOS_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_wgrad_final_all, dim3((unsigned)final_total), dim3(256), 0, st, *reduce);
    FOSVOS_LAUNCH_CHECK();
    reduce->n = 0;
    return FOSVOS_OK;
}

int fosvos::wgrad_impl(const uint16_t *x, const uint16_t *dy, float *dw, float *db, int N, int H, int W, int Ci, int Co,
                       int accumulate, void *workspace, size_t workspace_bytes, int device, void *stream,
                       WgradReduceTable *reduce) {
    FOSVOS_REQUIRE(x && dy && dw && workspace, FOSVOS_E_ARG, "conv3x3_wgrad: null pointer");
    FOSVOS_REQUIRE(N > 0 && H > 0 && W > 0 && Ci > 0 && Co > 0, FOSVOS_E_SHAPE, "conv3x3_wgrad: bad shape");
    FOSVOS_REQUIRE(Ci % BCI == 0, FOSVOS_E_SHAPE, "conv3x3_wgrad: Ci=%d must be a multiple of %d", Ci, BCI);
    FOSVOS_REQUIRE(Co % 64 == 0 || Co == 16, FOSVOS_E_SHAPE, "conv3x3_wgrad: Co=%d must be 16 or a multiple of 64", Co);
    const Plan p = make_plan(N, H, W, Ci, Co);
    FOSVOS_REQUIRE(workspace_bytes >= p.slab_bytes + p.bias_bytes, FOSVOS_E_WORKSPACE,
                   "conv3x3_wgrad: workspace %zu < %zu", workspace_bytes, p.slab_bytes + p.bias_bytes);
    FOSVOS_ENTER(device);
    hipStream_t st = (hipStream_t)stream;
    WgArgs a;
    a.x = x; a.dy = dy; a.slabs = reinterpret_cast<float *>(workspace);
    a.bias_part = db ? reinterpret_cast<float *>(reinterpret_cast<char *>(workspace) + p.slab_bytes) : nullptr;
    a.N = N; a.H = H; a.W = W; a.Ci = Ci; a.Cy = p.Cy; a.Cor = p.Cor;
    a.tiles_x = p.tiles_x; a.tiles_y = p.tiles_y; a.n_tiles = p.n_tiles; a.tiles_per_split = p.tps;
    const dim3 grid((unsigned)p.S, (unsigned)(Ci / BCI), (unsigned)(p.Cor / p.bco));
    // two tile images (dy + x) per workgroup; never less than the bias scratch [256][9] floats
    auto lds_bytes = [](int cf) { return (size_t)2 * (cf * TPIX + 4 * NPHP) * 32; };
    if (p.bco == 64) {
        static bool once[64];  // per device: opt in to 80 KB of dynamic LDS
        if (device >= 0 && device < 64 && !once[device]) {
            FOSVOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_wgrad<64>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes(4)));
            once[device] = true;
        }
        hipLaunchKernelGGL(k_wgrad<64>, grid, dim3(256), lds_bytes(4), st, a);
    } else if (p.bco == 32) {
        hipLaunchKernelGGL(k_wgrad<32>, grid, dim3(256), lds_bytes(2), st, a);
    } else {
        hipLaunchKernelGGL(k_wgrad<16>, grid, dim3(256), lds_bytes(1), st, a);
    }
    FOSVOS_LAUNCH_CHECK();
    WgradReduceEntry q{};
    q.slabs = a.slabs; q.dw = dw; q.db = db; q.bias_part = a.bias_part;
    q.S = p.S; q.S_bias = p.S; q.Co = Co; q.Cor = p.Cor; q.Ci = Ci; q.Ci_real = Ci; q.bco = p.bco; q.accumulate = accumulate;
    return finish_or_queue(q, reduce, st);
}

// ---------------------------------------------------------------------------------------------- conv1_1
// dw[Co,3,3,3], db[Co] of the first layer on the same MFMA kernel: the fp32 NCHW frame is first written as a
// 16-channel zero-padded bf16 NHWC image (13 MB per 480x854 frame; the other wgrads read bf16 activations too),
// then k_wgrad<64, KSPLIT> runs with Ci = 16 and the final pass keeps channels 0..2.
namespace {
struct FirstPlan {
    int tiles_x, tiles_y, n_tiles, tps, S;
    size_t frame_bytes, slab_bytes, bias_bytes;
};
FirstPlan make_first_plan(int N, int H, int W, int Co) {
    FirstPlan p;
    p.tiles_x = (int)cdiv(W, 16);
    p.tiles_y = (int)cdiv(H, TH);
    p.n_tiles = p.tiles_x * p.tiles_y * N;
    int S = kTargetBlocks / (Co / 64);
    if (S > p.n_tiles) S = p.n_tiles;
    if (S < 1) S = 1;
    p.tps = (int)cdiv(p.n_tiles, S);
    p.S = (int)cdiv(p.n_tiles, p.tps);
    p.frame_bytes = ((size_t)N * H * W * 16 * sizeof(uint16_t) + 255) / 256 * 256;
    p.slab_bytes = (size_t)(4 * p.S + kFoldTo) * 9 * Co * 16 * sizeof(float);
    p.bias_bytes = (size_t)p.S * Co * sizeof(float);
    return p;
}
}  // namespace

extern "C" size_t fosvos_conv3x3_first_wgrad_workspace_bytes(int N, int H, int W, int Co) {
    if (N <= 0 || H <= 0 || W <= 0 || Co <= 0 || Co % 64 != 0) return 0;
    const FirstPlan p = make_first_plan(N, H, W, Co);
    return p.frame_bytes + p.slab_bytes + p.bias_bytes;
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
    FOSVOS_REQUIRE(Co % 64 == 0, FOSVOS_E_SHAPE, "conv3x3_first_wgrad: Co=%d must be a multiple of 64", Co);
    FOSVOS_REQUIRE(N > 0 && H > 0 && W > 0, FOSVOS_E_SHAPE, "conv3x3_first_wgrad: bad shape N=%d H=%d W=%d", N, H, W);
    const FirstPlan p = make_first_plan(N, H, W, Co);
    const size_t need = p.frame_bytes + p.slab_bytes + p.bias_bytes;
    FOSVOS_REQUIRE(workspace_bytes >= need, FOSVOS_E_WORKSPACE, "conv3x3_first_wgrad: workspace %zu < %zu",
                   workspace_bytes, need);
    char *ws = reinterpret_cast<char *>(workspace);
    uint16_t *frame16 = reinterpret_cast<uint16_t *>(ws);
    if (int rc = fosvos_nchw_f32_to_nhwc_bf16(frame, frame16, N, 3, H, W, 16, device, stream)) return rc;
    FOSVOS_ENTER(device);
    hipStream_t st = (hipStream_t)stream;
    WgArgs a;
    a.x = frame16; a.dy = dy; a.slabs = reinterpret_cast<float *>(ws + p.frame_bytes);
    a.bias_part = db ? reinterpret_cast<float *>(ws + p.frame_bytes + p.slab_bytes) : nullptr;
    a.N = N; a.H = H; a.W = W; a.Ci = 16; a.Cy = Co; a.Cor = Co;
    a.tiles_x = p.tiles_x; a.tiles_y = p.tiles_y; a.n_tiles = p.n_tiles; a.tiles_per_split = p.tps;
    const size_t lds = (size_t)2 * (4 * TPIX + 1 * NPHP) * 32;
    hipLaunchKernelGGL((k_wgrad<64, true>), dim3((unsigned)p.S, 1, (unsigned)(Co / 64)), dim3(256), lds, st, a);
    FOSVOS_LAUNCH_CHECK();
    WgradReduceEntry q{};
    q.slabs = a.slabs; q.dw = dw; q.db = db; q.bias_part = a.bias_part;
    q.S = 4 * p.S; q.S_bias = p.S; q.Co = Co; q.Cor = Co; q.Ci = 16; q.Ci_real = 3; q.bco = 64; q.accumulate = accumulate;
    return finish_or_queue(q, reduce, st);
}
