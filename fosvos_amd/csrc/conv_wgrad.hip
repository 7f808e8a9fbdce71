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
__global__ __launch_bounds__(256) void k_wgrad(const WgArgs a) {
    constexpr int CF = BCO / 16;
    constexpr int CF_LOG = CF == 4 ? 2 : 0;
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

    const int t_begin = split * a.tiles_per_split;
    const int t_end = min(t_begin + a.tiles_per_split, a.n_tiles);
    for (int tile = t_begin; tile < t_end; ++tile) {
        int t = tile;
        const int tx_i = t % a.tiles_x;
        t /= a.tiles_x;
        const int ty_i = t % a.tiles_y;
        const int n = t / a.tiles_y;
        const int y0 = ty_i * TH, x0 = tx_i * 16;
        const uint16_t *dyn = a.dy + (int64_t)n * H * W * a.Cy;
        const uint16_t *xn = a.x + (int64_t)n * H * W * a.Ci;
        __syncthreads();
        // ---- stage dy tile: 16-byte units, lane order [half][pixel low 2 bits][channel block][pixel high]
        for (int idx = tid; idx < TPIX * CF * 2; idx += 256) {
            const int half = idx & 1, pl = (idx >> 1) & 3, cb = (idx >> 3) & (CF - 1), ph = idx >> (3 + CF_LOG);
            const int pix = ph * 4 + pl;
            const int gy = y0 + (pix >> 4), gx = x0 + (pix & 15);
            uint4 v = make_uint4(0, 0, 0, 0);
            if (gy < H && gx < W)
                v = *reinterpret_cast<const uint4 *>(dyn + ((int64_t)gy * W + gx) * a.Cy + co0 + cb * 16 + half * 8);
            *reinterpret_cast<uint4 *>(sY + (cb * TPIX + pix) * 32 + half * 16) = v;
        }
        // ---- stage x halo tile (zero padding outside the image)
        for (int idx = tid; idx < NPH * 4 * 2; idx += 256) {
            const int half = idx & 1, pl = (idx >> 1) & 3, cb = (idx >> 3) & 3, ph = idx >> 5;
            const int pix = ph * 4 + pl;
            const int hy = pix / HALO_W, hx = pix - hy * HALO_W;
            const int gy = y0 + hy - 1, gx = x0 + hx - 1;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (gy >= 0 && gy < H && gx >= 0 && gx < W)
                v = *reinterpret_cast<const uint4 *>(xn + ((int64_t)gy * W + gx) * a.Ci + ci0 + cb * 16 + half * 8);
            *reinterpret_cast<uint4 *>(sX + (cb * NPH + pix) * 32 + half * 16) = v;
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < TH / 2; ++ks) {
            bf16x8 af[CF];
#pragma unroll
            for (int i = 0; i < CF; ++i) {
                const char *base = sY + rd_y + (i * TPIX + ks * 32) * 32;
                af[i] = tr_pair(base, base + 16 * 32);
            }
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap % 3;
                const char *base = sX + rd_x + ((2 * ks + ky) * HALO_W + kx) * 32;
                const bf16x8 bfr = tr_pair(base, base + HALO_W * 32);
#pragma unroll
                for (int i = 0; i < CF; ++i)
                    acc[i][tap] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr, acc[i][tap], 0, 0, 0);
            }
        }
    }
    // ---- slab write: D row = co (4*(lane>>4)+r within the fragment), D col = ci (lane&15)
    float *slab = a.slabs + (int64_t)split * 9 * a.Cor * a.Ci;
    const int ci = ci0 + wave * 16 + li;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int i = 0; i < CF; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + i * 16 + g * 4 + r;
                slab[((int64_t)tap * a.Cor + co) * a.Ci + ci] = acc[i][tap][r];
            }
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

// stage 2: block = (co, 64 consecutive ci): sum the <= kFoldTo slabs [tap][co][ci] (64-float coalesced runs),
// transpose through LDS and write the 576 contiguous floats dw[(co*Ci + ci0)*9 ...] (OIHW).
__global__ __launch_bounds__(256) void k_wgrad_final(const float *__restrict__ slabs, int S, int Co, int Cor, int Ci,
                                                      int accumulate, float *__restrict__ dw) {
    __shared__ float tile[64 * 9];
    const int ci0 = blockIdx.x * 64, co = blockIdx.y;
    const int64_t E = 9LL * Cor * Ci;
    for (int e = threadIdx.x; e < 576; e += 256) {
        const int tap = e >> 6, cil = e & 63;
        const float *p = slabs + ((int64_t)tap * Cor + co) * Ci + ci0 + cil;
        float a = 0.f;
        for (int s = 0; s < S; ++s) a += p[(int64_t)s * E];
        tile[cil * 9 + tap] = a;
    }
    __syncthreads();
    float *d = dw + ((int64_t)co * Ci + ci0) * 9;
    for (int e = threadIdx.x; e < 576; e += 256) d[e] = accumulate ? d[e] + tile[e] : tile[e];
}

// ---- bias gradient: column sums of the [pixels][Cy] bf16 matrix.  Thread = 8 channels; a block
// sweeps rows blockIdx.x, +gridDim.x, ...; partials [block][Cy] then a fixed-order final sum.
__global__ __launch_bounds__(256) void k_colsum_partial(const uint16_t *__restrict__ dy, int64_t rows, int Cy,
                                                         float *__restrict__ partial) {
    const int groups = Cy >> 3;            // threads per row
    const int rows_per_it = 256 / groups;  // groups is 4, 8, 16, 32 or 64
    const int gsel = threadIdx.x % groups, rsel = threadIdx.x / groups;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    if (rsel < rows_per_it) {
        for (int64_t r = (int64_t)blockIdx.x * rows_per_it + rsel; r < rows; r += (int64_t)gridDim.x * rows_per_it) {
            float f[8];
            unpack8(*reinterpret_cast<const uint4 *>(dy + r * Cy + gsel * 8), f);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += f[j];
        }
    }
    __shared__ float s[256][9];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[threadIdx.x][j] = acc[j];
    __syncthreads();
    for (int c = threadIdx.x; c < Cy; c += 256) {
        const int gq = c >> 3, j = c & 7;
        float v = 0.f;
        for (int r = 0; r < rows_per_it; ++r) v += s[r * groups + gq][j];
        partial[(int64_t)blockIdx.x * Cy + c] = v;
    }
}

// block = 64 channels x 4 row groups: group g sums partial blocks g, g+4, ... (fixed order), then LDS
__global__ __launch_bounds__(256) void k_colsum_final(const float *__restrict__ partial, int n_blocks, int Cy, int Co,
                                                       int accumulate, float *__restrict__ db) {
    __shared__ double s[4][64];
    const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    double acc = 0.0;
    if (c < Cy)
        for (int b = g; b < n_blocks; b += 4) acc += (double)partial[(int64_t)b * Cy + c];
    s[g][cl] = acc;
    __syncthreads();
    if (g == 0 && c < Co) {
        const float v = (float)((s[0][cl] + s[1][cl]) + (s[2][cl] + s[3][cl]));
        db[c] = accumulate ? db[c] + v : v;
    }
}

constexpr int kColsumBlocks = 256;

struct Plan {
    int Cor, Cy, bco, tiles_x, tiles_y, n_tiles, tps, S;
    size_t slab_bytes, bias_bytes;
};

Plan make_plan(int N, int H, int W, int Ci, int Co) {
    Plan p;
    p.Cor = roundup(Co, 16);
    p.Cy = roundup(Co, 32);
    p.bco = (p.Cor % 64 == 0) ? 64 : 16;
    p.tiles_x = (int)cdiv(W, 16);
    p.tiles_y = (int)cdiv(H, TH);
    p.n_tiles = p.tiles_x * p.tiles_y * N;
    const int out_blocks = (p.Cor / p.bco) * (Ci / BCI);
    int S = (int)cdiv(512, out_blocks);
    if (S > p.n_tiles) S = p.n_tiles;
    if (S < 1) S = 1;
    p.tps = (int)cdiv(p.n_tiles, S);
    p.S = (int)cdiv(p.n_tiles, p.tps);
    // S slabs + kFoldTo folded slabs when a fold stage is needed
    p.slab_bytes = (size_t)(p.S + (p.S > kFoldTo ? kFoldTo : 0)) * 9 * p.Cor * Ci * sizeof(float);
    p.bias_bytes = (size_t)kColsumBlocks * p.Cy * sizeof(float);
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
    a.N = N; a.H = H; a.W = W; a.Ci = Ci; a.Cy = p.Cy; a.Cor = p.Cor;
    a.tiles_x = p.tiles_x; a.tiles_y = p.tiles_y; a.n_tiles = p.n_tiles; a.tiles_per_split = p.tps;
    const dim3 grid((unsigned)p.S, (unsigned)(Ci / BCI), (unsigned)(p.Cor / p.bco));
    if (p.bco == 64) {
        const size_t lds = (size_t)(4 * TPIX + 4 * NPH) * 32;
        hipLaunchKernelGGL(k_wgrad<64>, grid, dim3(256), lds, st, a);
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
        hipLaunchKernelGGL(k_wgrad_final, dim3((unsigned)(Ci / 64), (unsigned)Co), dim3(256), 0, st, src, n_src, Co, p.Cor,
                           Ci, accumulate, dw);
        FOSVOS_LAUNCH_CHECK();
    }
    if (db) {
        float *partial = reinterpret_cast<float *>(reinterpret_cast<char *>(workspace) + p.slab_bytes);
        const int64_t rows = (int64_t)N * H * W;
        hipLaunchKernelGGL(k_colsum_partial, dim3(kColsumBlocks), dim3(256), 0, st, dy, rows, p.Cy, partial);
        FOSVOS_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_colsum_final, dim3((unsigned)cdiv(p.Cy, 64)), dim3(256), 0, st, partial, kColsumBlocks, p.Cy,
                           Co, accumulate, db);
        FOSVOS_LAUNCH_CHECK();
    }
    return FOSVOS_OK;
}
