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

template <int BCO>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 3))) void k_wgrad(const WgArgs a) {
    constexpr int CF = BCO / 16;
    constexpr int CF_LOG = CF == 4 ? 2 : (CF == 2 ? 1 : 0);
    extern __shared__ __attribute__((aligned(16))) char smem_w[];
    char *sY = smem_w;                     // [CF][TPIX][32 B]
    char *sX = smem_w + CF * TPIX * 32;    // [4][NPH][32 B]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
    const int split = blockIdx.x;
    const int ci0 = blockIdx.y * BCI;
    const int co0 = blockIdx.z * BCO;
    const int H = a.H, W = a.W;

    f32x4 acc[CF][9];
#pragma unroll
    for (int i = 0; i < CF; ++i)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // per-lane byte offsets inside a tile image for the transposed reads
    const int rd_y = (4 * g + q) * 32 + p * 8;               // + (cb*TPIX + row*16) * 32
    const int rd_x = (wave * NPH + 4 * g + q) * 32 + p * 8;  // + ((row+ky)*18 + kx) * 32

    // bias gradient: the staging loop below gives every thread the SAME 8 dy channels (cb, half) for every
    // pixel it loads, so their column sums accumulate in registers; only the ci-block-0 workgroups keep them
    const bool do_bias = a.bias_part != nullptr && blockIdx.y == 0;
    float bsum[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bsum[e] = 0.f;

    const int t_begin = split * a.tiles_per_split;
    const int t_end = min(t_begin + a.tiles_per_split, a.n_tiles);

    // ---- software pipeline: register prefetch of tile t+1 under the MFMAs of tile t, one LDS image.
    // This kernel runs ONE wave per SIMD (144 accumulators), so every non-MFMA instruction in the tile loop
    // is exposed.  Everything tile-independent is therefore hoisted: per staged 16-byte unit the thread keeps
    // its global element offset relative to the tile origin, its LDS byte offset and its (row, column) inside
    // the tile; per tile only a base pointer, two bounds and a select per unit remain.  Prefetch registers and
    // descriptors are NAMED scalars (macro-unrolled): hipcc demotes the array form to scratch memory.
    constexpr int Y_IT = (TPIX * CF * 2 + 255) / 256;
    constexpr int X_IT = (NPH * 4 * 2 + 255) / 256;
    static_assert(Y_IT <= 4 && X_IT <= 6, "prefetch register file");
    uint4 py0, py1, py2, py3, px0, px1, px2, px3, px4, px5;
    py0 = py1 = py2 = py3 = px0 = px1 = px2 = px3 = px4 = px5 = make_uint4(0, 0, 0, 0);
    unsigned ok_y = 0, ok_x = 0;  // bit i: unit i of the prefetched tile is a real pixel (else zero padding)
#define FOSVOS_WG_DESC_Y(i)                                                                                       \
    int yoff##i = 0, ylds##i = -1, yrc##i = 0;                                                                    \
    if constexpr (i < Y_IT) {                                                                                     \
        const int idx_ = i * 256 + tid;                                                                           \
        const int half_ = idx_ & 1, pl_ = (idx_ >> 1) & 3, cb_ = (idx_ >> 3) & (CF - 1), ph_ = idx_ >> (3 + CF_LOG); \
        const int pix_ = ph_ * 4 + pl_;                                                                           \
        if (idx_ < TPIX * CF * 2) {                                                                               \
            yrc##i = ((pix_ >> 4) << 8) | (pix_ & 15);                                                            \
            yoff##i = ((pix_ >> 4) * W + (pix_ & 15)) * a.Cy + co0 + cb_ * 16 + half_ * 8;                        \
            ylds##i = (cb_ * TPIX + pix_) * 32 + half_ * 16;                                                      \
        }                                                                                                         \
    }
#define FOSVOS_WG_DESC_X(i)                                                                                       \
    int xoff##i = 0, xlds##i = -1, xrc##i = 0;                                                                    \
    if constexpr (i < X_IT) {                                                                                     \
        const int idx_ = i * 256 + tid;                                                                           \
        const int half_ = idx_ & 1, pl_ = (idx_ >> 1) & 3, cb_ = (idx_ >> 3) & 3, ph_ = idx_ >> 5;                 \
        const int pix_ = ph_ * 4 + pl_;                                                                           \
        if (idx_ < NPH * 8) {                                                                                     \
            const int hy_ = pix_ / HALO_W, hx_ = pix_ - hy_ * HALO_W;                                             \
            xrc##i = (hy_ << 8) | hx_;                                                                            \
            xoff##i = ((hy_ - 1) * W + (hx_ - 1)) * a.Ci + ci0 + cb_ * 16 + half_ * 8;                            \
            xlds##i = (cb_ * NPH + pix_) * 32 + half_ * 16;                                                       \
        }                                                                                                         \
    }
    FOSVOS_WG_DESC_Y(0) FOSVOS_WG_DESC_Y(1) FOSVOS_WG_DESC_Y(2) FOSVOS_WG_DESC_Y(3)
    FOSVOS_WG_DESC_X(0) FOSVOS_WG_DESC_X(1) FOSVOS_WG_DESC_X(2) FOSVOS_WG_DESC_X(3) FOSVOS_WG_DESC_X(4) FOSVOS_WG_DESC_X(5)

    // loads are unconditional (a load under a branch makes hipcc drain vmcnt(0) inside the prefetch block):
    // out-of-image units read the tile's first pixel instead and are zeroed when written to LDS
#define FOSVOS_WG_LDY(i)                                                                                          \
    if constexpr (i < Y_IT) {                                                                                     \
        const bool ok_ = ((yrc##i >> 8) < vrows_) && ((yrc##i & 255) < vcols_);                                   \
        ok_y = ok_ ? (ok_y | (1u << i)) : (ok_y & ~(1u << i));                                                    \
        py##i = *reinterpret_cast<const uint4 *>(ybase_ + (ok_ ? yoff##i : co0));                                 \
    }
#define FOSVOS_WG_LDX(i)                                                                                          \
    if constexpr (i < X_IT) {                                                                                     \
        const int hy_ = xrc##i >> 8, hx_ = xrc##i & 255;                                                          \
        const bool ok_ = (hy_ >= ylo_) && (hy_ <= vrows_) && (hx_ >= xlo_) && (hx_ <= vcols_);                    \
        ok_x = ok_ ? (ok_x | (1u << i)) : (ok_x & ~(1u << i));                                                    \
        px##i = *reinterpret_cast<const uint4 *>(xbase_ + (ok_ ? xoff##i : ci0));                                 \
    }
    // (n, y0, x0) of the tile to load; vrows/vcols = image rows/columns left from the tile origin
#define FOSVOS_WG_LOAD_TILE()                                                                                     \
    {                                                                                                             \
        const int y0_ = lt_y * TH, x0_ = lt_x * 16;                                                               \
        const int vrows_ = H - y0_, vcols_ = W - x0_;                                                             \
        const int ylo_ = 1 - y0_, xlo_ = 1 - x0_;                                                                 \
        const int64_t org_ = ((int64_t)lt_n * H + y0_) * W + x0_;                                                 \
        const uint16_t *ybase_ = a.dy + org_ * a.Cy;                                                              \
        const uint16_t *xbase_ = a.x + org_ * a.Ci;                                                               \
        FOSVOS_WG_LDY(0) FOSVOS_WG_LDY(1) FOSVOS_WG_LDY(2) FOSVOS_WG_LDY(3)                                       \
        FOSVOS_WG_LDX(0) FOSVOS_WG_LDX(1) FOSVOS_WG_LDX(2) FOSVOS_WG_LDX(3) FOSVOS_WG_LDX(4) FOSVOS_WG_LDX(5)     \
        if (++lt_x == a.tiles_x) {                                                                                \
            lt_x = 0;                                                                                             \
            if (++lt_y == a.tiles_y) { lt_y = 0; ++lt_n; }                                                        \
        }                                                                                                         \
    }
#define FOSVOS_WG_STY(i)                                                                                          \
    if constexpr (i < Y_IT) {                                                                                     \
        if (ylds##i >= 0) {                                                                                       \
            const bool ok_ = (ok_y >> i) & 1u;                                                                    \
            uint4 v_ = py##i;                                                                                     \
            v_.x = ok_ ? v_.x : 0u; v_.y = ok_ ? v_.y : 0u; v_.z = ok_ ? v_.z : 0u; v_.w = ok_ ? v_.w : 0u;       \
            *reinterpret_cast<uint4 *>(sY + ylds##i) = v_;                                                        \
            if (do_bias) {                                                                                        \
                float f_[8];                                                                                      \
                unpack8(v_, f_);                                                                                  \
                _Pragma("unroll") for (int e = 0; e < 8; ++e) bsum[e] += f_[e];                                   \
            }                                                                                                     \
        }                                                                                                         \
    }
#define FOSVOS_WG_STX(i)                                                                                          \
    if constexpr (i < X_IT) {                                                                                     \
        if (xlds##i >= 0) {                                                                                       \
            const bool ok_ = (ok_x >> i) & 1u;                                                                    \
            uint4 v_ = px##i;                                                                                     \
            v_.x = ok_ ? v_.x : 0u; v_.y = ok_ ? v_.y : 0u; v_.z = ok_ ? v_.z : 0u; v_.w = ok_ ? v_.w : 0u;       \
            *reinterpret_cast<uint4 *>(sX + xlds##i) = v_;                                                        \
        }                                                                                                         \
    }
#define FOSVOS_WG_STORE_TILE()                                                                                    \
    {                                                                                                             \
        FOSVOS_WG_STY(0) FOSVOS_WG_STY(1) FOSVOS_WG_STY(2) FOSVOS_WG_STY(3)                                       \
        FOSVOS_WG_STX(0) FOSVOS_WG_STX(1) FOSVOS_WG_STX(2) FOSVOS_WG_STX(3) FOSVOS_WG_STX(4) FOSVOS_WG_STX(5)     \
    }
#define FOSVOS_WG_KSTEP(ks)                                                                                       \
    {                                                                                                             \
        bf16x8 af[CF];                                                                                            \
        _Pragma("unroll") for (int i = 0; i < CF; ++i) {                                                          \
            const char *base = sY + rd_y + (i * TPIX + (ks) * 32) * 32;                                           \
            af[i] = tr_pair(base, base + 16 * 32);                                                                \
        }                                                                                                         \
        _Pragma("unroll") for (int tap = 0; tap < 9; ++tap) {                                                     \
            const int ky = tap / 3, kx = tap % 3;                                                                 \
            const char *base = sX + rd_x + ((2 * (ks) + ky) * HALO_W + kx) * 32;                                  \
            const bf16x8 bfr = tr_pair(base, base + HALO_W * 32);                                                 \
            _Pragma("unroll") for (int i = 0; i < CF; ++i)                                                        \
                acc[i][tap] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr, acc[i][tap], 0, 0, 0);          \
        }                                                                                                         \
    }

    // tile walker: decoded once (the only divisions), then advanced incrementally by FOSVOS_WG_LOAD_TILE
    int lt_x = t_begin % a.tiles_x, lt_y = (t_begin / a.tiles_x) % a.tiles_y, lt_n = t_begin / (a.tiles_x * a.tiles_y);
    if (t_begin < t_end) FOSVOS_WG_LOAD_TILE()
    for (int tile = t_begin; tile < t_end; ++tile) {
        if (tile > t_begin) __syncthreads();  // every wave finished reading the previous tile
        FOSVOS_WG_STORE_TILE()
        __syncthreads();
        FOSVOS_WG_KSTEP(0)
        // the next tile's loads go out after the first k-step so that their address arithmetic issues in the
        // shadow of MFMAs already in flight
        if (tile + 1 < t_end) FOSVOS_WG_LOAD_TILE()
        FOSVOS_WG_KSTEP(1)
        FOSVOS_WG_KSTEP(2)
        FOSVOS_WG_KSTEP(3)
    }
    static_assert(TH == 8, "four k-steps per tile");
    if (do_bias) {
        // threads with equal (tid & 31 with the pixel bits masked) share channels: reduce the 256 x 8 partials in
        // LDS in a fixed order: thread (unit u = half + 2*cb) <- all threads t with the same u
        __syncthreads();
        float *sb = reinterpret_cast<float *>(smem_w);  // [256][9] floats
#pragma unroll
        for (int e = 0; e < 8; ++e) sb[tid * 9 + e] = bsum[e];
        __syncthreads();
        if (tid < BCO) {
            const int cb = tid >> 4, half = (tid >> 3) & 1, e = tid & 7;
            float acc_b = 0.f;
            // staging index bits: [half][pixel low 2][cb (CF_LOG bits)][pixel high]: enumerate the owners in order
            for (int t2 = 0; t2 < 256; ++t2)
                if ((t2 & 1) == half && ((t2 >> 3) & (CF - 1)) == cb) acc_b += sb[t2 * 9 + e];
            a.bias_part[(int64_t)split * a.Cor + co0 + tid] = acc_b;
        }
    }
    // ---- slab write in FRAGMENT-NATIVE order: a lane's 4 accumulator registers (4 consecutive co of one ci)
    // go out as one 16-byte store, 1 KB contiguous per wave instruction (4-byte [tap][co][ci] stores were
    // store-issue-bound: ~20 us per launch).  Element (tap, co, ci) of a split's slab lives at
    //   (((cb * (Ci/16) + ci/16) * 9 + tap) * BCO*16) + (((co%BCO)/16 * 4 + (co%16)/4) * 16 + ci%16) * 4 + co%4
    // with cb = co / BCO; k_wgrad_final undoes the permutation.
    float *slab = a.slabs + (int64_t)split * 9 * a.Cor * a.Ci;
    const int wci = blockIdx.y * 4 + wave;  // global 16-wide ci fragment
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

__global__ __launch_bounds__(256) void k_wgrad_fold(const float *__restrict__ slabs, int S, int64_t E,
                                                     float *__restrict__ folded) {
    const int64_t i4 = blockIdx.x * 256LL + threadIdx.x;  // float4 index
    if (i4 * 4 >= E) return;
    const int y = blockIdx.y;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = y; s < S; s += kFoldTo) {
        const float4 v = *reinterpret_cast<const float4 *>(slabs + (int64_t)s * E + i4 * 4);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    *reinterpret_cast<float4 *>(folded + (int64_t)y * E + i4 * 4) = acc;
}

// stage 2: block = (4 consecutive co = one accumulator float4, 64 consecutive ci): sum the <= kFoldTo slabs
// (16-byte coalesced reads of the fragment-native layout), transpose through LDS and write 4 runs of 576
// contiguous floats dw[(co*Ci + ci0)*9 ...] (OIHW).
__global__ __launch_bounds__(256) void k_wgrad_final(const float *__restrict__ slabs, int S, int Co, int Cor, int Ci,
                                                      int bco, int accumulate, float *__restrict__ dw,
                                                      const float *__restrict__ bias_part, int S_bias,
                                                      float *__restrict__ db) {
    __shared__ float tile[4][64 * 9];
    const int ci0 = blockIdx.x * 64, co4 = blockIdx.y * 4;
    if (db && blockIdx.x == 0) {  // the 4 channels' bias: S_bias per-split partials each, fixed-order tree
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
    for (int e = threadIdx.x; e < 576; e += 256) {
        const int tap = e >> 6, cil = e & 63;
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
    for (int e = threadIdx.x; e < 4 * 576; e += 256) {
        const int r = e / 576, k = e - r * 576;
        if (co4 + r >= Co) continue;
        float *d = dw + ((int64_t)(co4 + r) * Ci + ci0) * 9 + k;
        *d = accumulate ? *d + tile[r][k] : tile[r][k];
    }
}

// k_wgrad<64> holds 144 accumulators + the prefetched tile: one workgroup per CU.  One full wave of
// workgroups (256 CUs) keeps the slab traffic (splits x |dw|) minimal.
constexpr int kTargetBlocks = 256;
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
    if (p.bco == 64) {
        const size_t lds = (size_t)(4 * TPIX + 4 * NPH) * 32;
        hipLaunchKernelGGL(k_wgrad<64>, grid, dim3(256), lds, st, a);
    } else if (p.bco == 32) {
        const size_t lds = (size_t)(2 * TPIX + 4 * NPH) * 32;
        hipLaunchKernelGGL(k_wgrad<32>, grid, dim3(256), lds, st, a);
    } else {
        const size_t lds = (size_t)(1 * TPIX + 4 * NPH) * 32;
        hipLaunchKernelGGL(k_wgrad<16>, grid, dim3(256), lds, st, a);
    }
    FOSVOS_LAUNCH_CHECK();
    {
        const int64_t E = 9LL * p.Cor * Ci;
        const float *src = a.slabs;
        int n_src = p.S;
        if (p.S > kFoldTo) {
            float *folded = a.slabs + (int64_t)p.S * E;
            hipLaunchKernelGGL(k_wgrad_fold, dim3((unsigned)cdiv(E / 4, 256), kFoldTo), dim3(256), 0, st, a.slabs, p.S, E,
                               folded);
            FOSVOS_LAUNCH_CHECK();
            src = folded;
            n_src = kFoldTo;
        }
        hipLaunchKernelGGL(k_wgrad_final, dim3((unsigned)(Ci / 64), (unsigned)(p.Cor / 4)), dim3(256), 0, st, src, n_src, Co,
                           p.Cor, Ci, p.bco, accumulate, dw, (const float *)a.bias_part, p.S, db);
        FOSVOS_LAUNCH_CHECK();
    }
    return FOSVOS_OK;
}
