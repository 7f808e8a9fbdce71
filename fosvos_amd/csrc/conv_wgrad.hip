// Weight / bias gradient of the 3x3 conv on bf16 NHWC activations (autograd's backward-weight of
// Conv2d in stages[*] / side_prep[*], src/networks/osvos_vgg.py:42,92), Ci a multiple of 64.
//
//   dw[co][ci][ky][kx] = sum_{n,y,x} dy[n,y,x,co] * x[n,y+ky-1,x+kx-1,ci]        db[co] = sum dy[.,co]
//
// GEMM view: M = co, N = ci (x 9 taps), K = pixels: MFMA roofline.  Both operands are stored channel-contiguous (NHWC) while
// the contraction runs over pixels, so both are transposed on the fly: a tile is staged in LDS pixel-major and read with
// ds_read_b64_tr_b16, which hands a lane the 4 pixels x 1 channel column of a 4 x 16 block.
//
// Operand reuse is what the kernel is built around.  A tile is 8 rows x 16 pixels of dy and its 10 x 18 halo of x.  The MFMA
// loop walks the HALO rows: the three x fragments of a halo row (kx = 0, 1, 2) meet the dy fragments of the tile rows one, two
// and three above it in the walk (ky = 0, 1, 2), so every x fragment is read from LDS once and used three times, and a dy
// fragment stays in registers for three rows.
//
// The pixel range is split over workgroups; each split writes one fp32 slab laid out like dw itself ([co][ci][9], OIHW), so
// the reduction over splits is a plain elementwise sum in split order (k_wgrad_fold / k_wgrad_reduce: bitwise reproducible,
// no float atomics, no transposes), queued per layer and run for many layers in one launch.
#include <stdlib.h>

#include "common.hpp"

using namespace fosvos;

namespace {
constexpr int TH = 8;                    // tile rows
constexpr int TPIX = TH * 16;            // 128 tile pixels
constexpr int HALO_W = 18;
constexpr int BCI = 64;                  // ci granularity of the grid (and of the API contract)

struct WgArgs {
    const uint16_t *x;   // [N,H,W,Ci]
    const uint16_t *dy;  // [N,H,W,Cy]  (Cy = roundup(Co,32))
    float *slabs;        // [S][Cor][Ci][9]
    float *bias_part;    // [S * n_ci][Cor] column sums of dy per split and ci block (each sums its share of the tiles), or null
    int N, H, W, Ci, Cy, Cor;
    int tiles_x, tiles_y, n_tiles, tiles_per_split;
    int lab;  // timing-only switch of lab builds (-DFOSVOS_LAB_BUILD, FOSVOS_WGRAD_LAB; wrong results): 1 no slab store; always 0 in the shipped library
    int S, n_ci, n_co, xcd_order;  // 1-D grid of S * n_ci * n_co workgroups, decoded in the kernel (see there)
};

typedef __attribute__((address_space(3))) s16x4 *lds_s16x4_ptr;

// =====================================================================================================================
// The MFMA kernel (round 3; round 2's ran one 280-register wave per SIMD with register-staged tiles: 33-35 % of the MFMA
// cycles when alone on the chip, 25 % of its wave cycles in s_waitcnt / s_barrier, 2x its algorithmic bytes from beyond L2):
//   * staging is LDS-DMA (buffer_load ... lds): the next tile goes from L2 straight into the other LDS image while the
//     MFMA loop runs - no staging registers (-40), no ds_write, nothing to wait for behind the loop but the DMA itself;
//     pieces are 8 pixels x 128 B, i.e. whole 128-byte lines of the NHWC tensors;
//   * 8 waves per workgroup, two per SIMD, each 32 co x 16 ci x 9 taps on v_mfma_f32_16x16x32_bf16 (18 accumulators of 4
//     registers): ~150 registers per wave, so one of these workgroups and one 192-register igemm workgroup still share a CU;
//   * the bias gradient (column sums of dy) is four v_dot2c_f32_bf16 per dy fragment, taken in turn by the waves.
// LDS image of a tile: [pixel][128 B = 64 channels], 16-byte chunk c of pixel px stored at chunk slot c ^ 2((px >> 1) & 3).
// An LDS-DMA instruction writes lane i's 16 bytes at base + 16 i, so the swizzle is applied to the SOURCE address; the
// transposed reads apply the same XOR.  Row strides (16 px for dy, 24 px for the x halo: 18 used) are multiples of 8 px, so
// the XOR term of a lane's address does not change from row to row.  With it a 32-lane half of a ds_read_b64_tr_b16 covers
// 8 consecutive pixels x 32 B on 64 distinct banks for every tap shift (simulated with the bank rule of the guide).
// k index of an MFMA (32 pixels): lane group g, element j  <->  tile row r + 4 (g >> 1), pixel 4 (g & 1) + (j & 3) + 8 (j >> 2):
// any bijection does as long as both operands use the same one - this one makes the halves of a read contiguous.
// The loop walks R = 0..5: the x fragments of halo rows (R, R + 4) meet the dy fragments of tile rows (R - ky, R + 4 - ky).
// SIDE = true is the side_prep form (16 output channels in a 32-wide dy image, 64-byte pixel rows): four waves, each
// 16 co x 16 ci x 9 taps, 64 ci per workgroup; everything else - tile, images, walk, slabs - is the same code.
namespace v2 {
constexpr int XROW = 24;                          // pixels per halo row in LDS (18 used)
constexpr int X_BYTES = (TH + 2) * XROW * 128;    // 30 KB
constexpr int X_PIECES = (TH + 2) * XROW / 8;     // 30 pieces of 8 pixels x 128 B

template <bool SIDE>
struct Cfg2 {
    static constexpr int NW = SIDE ? 4 : 8;                 // waves per workgroup
    static constexpr int NT = 64 * NW;
    static constexpr int NU = SIDE ? 1 : 2;                 // 16-channel dy sub-blocks per wave
    static constexpr int BCO = SIDE ? 16 : 64;              // output channels per workgroup
    static constexpr int YPX = SIDE ? 64 : 128;             // bytes of a dy pixel row in LDS
    static constexpr int Y_BYTES = TPIX * YPX;              // 8 / 16 KB
    static constexpr int BUF_BYTES = Y_BYTES + X_BYTES;
    static constexpr int LDS_BYTES = 2 * BUF_BYTES;         // 76 / 92 KB: two tile images
    static constexpr int Y_PIECES = Y_BYTES / 1024;         // 8 (16 px x 64 B) / 16 (8 px x 128 B)
    static constexpr int NXIT = (X_PIECES + NW - 1) / NW;   // x pieces per wave: 8 / 4
    static constexpr int N_IT = 2 + NXIT;                   // pieces per wave and tile: two of dy, then x
    static_assert(Y_PIECES == 2 * NW, "two dy pieces per wave, four tile rows apart");
};

typedef __attribute__((address_space(3))) void lds_void;

// 8 pixels of one channel: p .. +3 and the 4 pixels 8 further along the row (`second` bytes on)
__device__ __forceinline__ bf16x8 tr_pair16(const char *p, int second) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p + second));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

// One LDS-DMA instruction, hidden from the compiler: with the builtin form hipcc cannot tell the image being filled from
// the image being read and puts s_waitcnt vmcnt(0) in front of the first transposed read of every tile, which serialises the
// DMA with the MFMA loop.  M0 carries the LDS destination (wave-uniform); lane i's 16 bytes land at M0 + 16 i.  The wave
// waits for its own DMA (s_waitcnt vmcnt(0), also inline) in front of the barrier that publishes the image.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void dma16(const __amdgpu_buffer_rsrc_t rsrc, unsigned lds_dst, unsigned voff, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds_dst), "v"(voff), "s"(rsrc), "s"(soff) : "memory", "m0");
}
#pragma clang diagnostic pop

template <bool SIDE>
__global__ __launch_bounds__(Cfg2<SIDE>::NT) __attribute__((amdgpu_waves_per_eu(SIDE ? 2 : 3, SIDE ? 4 : 3))) void k_wgrad3x3_v2(const WgArgs a) {
    using C = Cfg2<SIDE>;
    constexpr int NW = C::NW, NU = C::NU, YPX = C::YPX, Y_BYTES = C::Y_BYTES, BUF_BYTES = C::BUF_BYTES, N_IT = C::N_IT;
    extern __shared__ __attribute__((aligned(16))) char smem_w[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = SIDE ? 0 : wave >> 2, wi = wave & 3;  // 32-co block (backbone), 16-ci block of this wave
    // Workgroup -> (pixel split, ci block, co block).  The n_ci * n_co workgroups of a split walk the same tiles: each x slice
    // is read by n_co of them, each dy slice by n_ci.  The dispatcher deals consecutive workgroup ids round-robin over the 8
    // XCDs (each with its own L2), so with a plain 3-D grid the workgroups that share operands never meet in an L2 and every
    // slice is fetched from beyond L2 once per reader (measured: 347 MB per launch instead of 203).  Remapped - a speed choice,
    // never correctness: the ids that share an XCD (id % 8) take one contiguous eighth of the (split, ci, co) space, co
    // fastest, and start together.
    int bid = blockIdx.x;
    if (a.xcd_order) {
        const int n_wg = gridDim.x, q8 = n_wg >> 3, r8 = n_wg & 7, xcd = bid & 7;
        bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);  // bijective for any n_wg
    }
    const int co_blk = bid % a.n_co, ci_blk = (bid / a.n_co) % a.n_ci, split = bid / (a.n_co * a.n_ci);
    const int ci0 = ci_blk * 64, co0 = co_blk * C::BCO;
    const int H = a.H, W = a.W;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void *)smem_w);

    f32x4 acc[9][NU];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int u = 0; u < NU; ++u) acc[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum[NU];  // bias gradient: this lane's share (its 8 pixels of a k-step) of sum over pixels of dy[., co]
#pragma unroll
    for (int u = 0; u < NU; ++u) bsum[u] = 0.f;

    // ---- transposed-read addresses of this lane (bytes inside a tile image).  The chunk swizzle of a 128-byte pixel row
    // (8 chunks) is c ^ 2((px >> 1) & 3); of a 64-byte row (4 chunks: the side form's dy) c ^ 2((px >> 2) & 1): either way a
    // 32-lane half of a read - 8 consecutive pixels x 32 B - lands on 64 distinct banks
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int pxl = 4 * (g & 1) + q;                                   // pixel inside the row, first read
    auto swz128 = [](int px, int c) { return (c ^ (2 * ((px >> 1) & 3))) * 16; };
    auto swz64 = [](int px, int c) { return (c ^ (2 * ((px >> 2) & 1))) * 16; };
    int rd_y[NU], rd_x[3];
#pragma unroll
    for (int u = 0; u < NU; ++u)
        rd_y[u] = ((4 * (g >> 1)) * 16 + pxl) * YPX + (SIDE ? swz64(pxl, p >> 1) : swz128(pxl, 2 * (2 * wc + u) + (p >> 1))) + (p & 1) * 8;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
        rd_x[kx] = Y_BYTES + ((4 * (g >> 1)) * XROW + pxl + kx) * 128 + swz128(pxl + kx, 2 * wi + (p >> 1)) + (p & 1) * 8;

    // ---- staging plan.  Piece pc of a tile image is its pc-th kilobyte: lane i fetches the 16 bytes that belong at LDS
    // offset 1024 pc + 16 i, i.e. the source chunk (slot ^ swizzle) of its pixel.  dy: pieces it * NW + wave, it = 0, 1 - the
    // two are four tile rows apart (same columns), so one offset register serves both.  x: pieces (it - 2) * NW + wave.
    const int ypx = SIDE ? lane >> 2 : lane >> 3, yslot = SIDE ? lane & 3 : lane & 7;  // pixel / chunk slot inside a dy piece
    const int ppx = lane >> 3, slot = lane & 7;                                         // ... inside an x piece
    constexpr int YPP = SIDE ? 16 : 8;                                                  // pixels per dy piece
    unsigned goff_y, goff_x[C::NXIT];
    {
        const int pix = wave * YPP + ypx, ty = pix >> 4, tx = pix & 15;
        const int c = SIDE ? yslot ^ (2 * ((tx >> 2) & 1)) : yslot ^ (2 * ((tx >> 1) & 3));
        goff_y = (unsigned)(((ty * W + tx) * a.Cy + co0 + c * 8) * 2);
    }
#pragma unroll
    for (int it = 2; it < N_IT; ++it) {
        const int pix = ((it - 2) * NW + wave) * 8 + ppx, hy = pix / XROW, hx = pix - hy * XROW;
        const bool used = (it - 2) * NW + wave < X_PIECES && hx < HALO_W;
        goff_x[it - 2] = used ? (unsigned)(((hy * W + hx) * a.Ci + ci0 + (slot ^ (2 * ((hx >> 1) & 3))) * 8) * 2) : ~0u;
    }
    const unsigned y_step = (unsigned)(4 * W * a.Cy * 2);  // dy piece it = 1: four tile rows further
    const int64_t halo_shift = (int64_t)(W + 1) * a.Ci;
    const unsigned y_total = (unsigned)((int64_t)a.N * H * W * a.Cy * 2), x_total = (unsigned)(((int64_t)a.N * H * W * a.Ci + halo_shift) * 2);
    const auto y_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(a.dy), 0, y_total, 0x00020000);
    const auto x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(a.x) - halo_shift, 0, x_total, 0x00020000);

    const int t_begin = split * a.tiles_per_split;
    const int t_end = min(t_begin + a.tiles_per_split, a.n_tiles);
    int lt_x = t_begin % a.tiles_x, lt_y = (t_begin / a.tiles_x) % a.tiles_y, lt_n = t_begin / (a.tiles_x * a.tiles_y);

    // DMA of tile (lt_x, lt_y, lt_n) into the image at LDS offset `img_`; out-of-image pieces read offset ~0 = zeros (the
    // descriptor's range check).  Edge tiles re-derive a piece's (row, column) from the lane: no registers held for it.
#define FOSVOS_W2_PIECE_Y(it_, img_)                                                                            \
    {                                                                                                           \
        unsigned v_ = goff_y + (it_) * y_step;                                                                  \
        if (!interior_) {                                                                                       \
            const int pix_ = ((it_) * NW + wave) * YPP + ypx;                                                   \
            v_ = ((pix_ >> 4) < vrows_ && (pix_ & 15) < vcols_) ? v_ : ~0u;                                     \
        }                                                                                                       \
        dma16(y_rsrc, (img_) + ((it_) * NW + wave) * 1024, v_, ysoff_);                                         \
    }
#define FOSVOS_W2_PIECE_X(it_, img_)                                                                            \
    if (((it_) - 1) * NW <= X_PIECES || ((it_) - 2) * NW + wave < X_PIECES) {                                   \
        unsigned v_ = goff_x[(it_) - 2];                                                                        \
        if (!interior_) {                                                                                       \
            const int pix_ = (((it_) - 2) * NW + wave) * 8 + ppx, hy_ = pix_ / XROW, hx_ = pix_ - hy_ * XROW;   \
            v_ = (hy_ >= 1 - y0_ && hy_ <= vrows_ && hx_ >= 1 - x0_ && hx_ <= vcols_) ? v_ : ~0u;               \
        }                                                                                                       \
        dma16(x_rsrc, (img_) + Y_BYTES + (((it_) - 2) * NW + wave) * 1024, v_, xsoff_);                         \
    }
#define FOSVOS_W2_TILE_SCALARS()                                                                                \
    const int y0_ = lt_y * TH, x0_ = lt_x * 16;                                                                 \
    const int vrows_ = H - y0_, vcols_ = W - x0_;                                                               \
    const int64_t org_ = ((int64_t)lt_n * H + y0_) * W + x0_;                                                   \
    const unsigned ysoff_ = (unsigned)(org_ * a.Cy * 2), xsoff_ = (unsigned)(org_ * a.Ci * 2);                  \
    const bool interior_ = y0_ >= 1 && y0_ + TH < H && x0_ >= 1 && x0_ + 16 < W;
#define FOSVOS_W2_ADVANCE()                                                                                     \
    if (++lt_x == a.tiles_x) {                                                                                  \
        lt_x = 0;                                                                                               \
        if (++lt_y == a.tiles_y) { lt_y = 0; ++lt_n; }                                                          \
    }
    // piece `it_` (of the tile the scalars in scope describe), if this form has that many
#define FOSVOS_W2_PIECE_ANY(it_, img_)                                                                          \
    if constexpr ((it_) < 2) { FOSVOS_W2_PIECE_Y(it_, img_) }                                                   \
    else if constexpr ((it_) < N_IT) { FOSVOS_W2_PIECE_X(it_, img_) }
#define FOSVOS_W2_STAGE(img_)                                                                                   \
    {                                                                                                           \
        FOSVOS_W2_TILE_SCALARS()                                                                                \
        FOSVOS_W2_PIECE_ANY(0, img_) FOSVOS_W2_PIECE_ANY(1, img_) FOSVOS_W2_PIECE_ANY(2, img_)                  \
        FOSVOS_W2_PIECE_ANY(3, img_) FOSVOS_W2_PIECE_ANY(4, img_) FOSVOS_W2_PIECE_ANY(5, img_)                  \
        FOSVOS_W2_PIECE_ANY(6, img_) FOSVOS_W2_PIECE_ANY(7, img_) FOSVOS_W2_PIECE_ANY(8, img_)                  \
        FOSVOS_W2_PIECE_ANY(9, img_)                                                                            \
        FOSVOS_W2_ADVANCE()                                                                                     \
    }
    static_assert(N_IT <= 10, "piece macros cover ten pieces per wave");
    // two pieces of the NEXT tile, issued from inside the MFMA loop in front of row R (see there)
#define FOSVOS_W2_PIECES_OF_ROW(R, img_)                                                                        \
    if (has_next) { FOSVOS_W2_PIECE_ANY(2 * (R), img_) FOSVOS_W2_PIECE_ANY(2 * (R) + 1, img_) }

    if (t_begin < t_end) FOSVOS_W2_STAGE(lds0)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA has landed ...
    __syncthreads();                                   // ... and everyone's

    const bool do_bias = a.bias_part != nullptr;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    // sum of the 8 bf16 of a fragment into `acc_` (v_dot2c_f32_bf16 against (1, 1): two elements per instruction)
    auto add8 = [](const bf16x8 &f, float acc_) {
        const bf16x2_t one2 = __builtin_bit_cast(bf16x2_t, 0x3f803f80u);
        acc_ = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(f, f, 0, 1), one2, acc_, false);
        acc_ = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(f, f, 2, 3), one2, acc_, false);
        acc_ = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(f, f, 4, 5), one2, acc_, false);
        acc_ = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(f, f, 6, 7), one2, acc_, false);
        return acc_;
    };

    for (int tile = t_begin; tile < t_end; ++tile) {
        const int par = (tile - t_begin) & 1;
        const char *cur = smem_w + par * BUF_BYTES;
        // The other image was last read during tile - 1 and every wave has passed that tile's barrier, so the next tile's DMA
        // may start now.  Its pieces are NOT issued in one burst: an LDS-DMA instruction holds its wave for ~60-180 clocks
        // while the memory pipe takes it in, and all waves bursting together left the matrix pipe idle for the first ~1000
        // clocks of every tile.  Two pieces ride in front of each of the first rows (three rows for the backbone form's six);
        // the last ones then have the remaining rows of matrix work to land under before the wave waits for them (one piece
        // per row, the last issued in the last row, was 8 % slower: measured).
        const bool has_next = tile + 1 < t_end;
        const unsigned nxt_ = lds0 + (par ^ 1) * BUF_BYTES;
        FOSVOS_W2_TILE_SCALARS()
        // the bias sums of a tile are taken by one ci block of the grid and, inside it, by one of the four ci waves in turn
        const bool my_bias = do_bias && (tile % a.n_ci) == ci_blk && ((tile / a.n_ci) & 3) == wi;

        const char *yb[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) yb[u] = cur + rd_y[u];
        const char *xb0 = cur + rd_x[0], *xb1 = cur + rd_x[1], *xb2 = cur + rd_x[2];
        bf16x8 a0[NU], a1[NU], a2[NU], b[3], bn[3];
#pragma unroll
        for (int u = 0; u < NU; ++u) a0[u] = a1[u] = a2[u] = bf16x8{};
        bn[0] = bn[1] = bn[2] = bf16x8{};
        b[0] = tr_pair16(xb0, 1024);
        b[1] = tr_pair16(xb1, 1024);
        b[2] = tr_pair16(xb2, 1024);
        // Row R: the dy fragments of row R are requested in front of the row's MFMAs and used by its LAST ones (ky = 0); the x
        // fragments of row R + 1 are requested behind the first ones (ky = 2), whose dy registers they may then take, and have
        // the others to land under: every request has matrix work of this very wave to cover it (the waves of a SIMD run in
        // lockstep - same program, one barrier per tile - and do not cover each other's LDS latency).  A compiler fence per
        // row keeps the reads in their row.
#define FOSVOS_W2_ROW(R)                                                                                        \
        {                                                                                                           \
            _Pragma("unroll") for (int u = 0; u < NU; ++u) { a2[u] = a1[u]; a1[u] = a0[u]; }                        \
            FOSVOS_W2_PIECES_OF_ROW(R, nxt_)                                                                        \
            if constexpr ((R) < 4) {                                                                                \
                _Pragma("unroll") for (int u = 0; u < NU; ++u) a0[u] = tr_pair16(yb[u] + (R) * 16 * YPX, 8 * YPX);  \
            }                                                                                                       \
            if constexpr ((R) >= 2) {                                                                               \
                _Pragma("unroll") for (int kx = 0; kx < 3; ++kx)                                                    \
                _Pragma("unroll") for (int u = 0; u < NU; ++u)                                                      \
                    acc[6 + kx][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[u], b[kx], acc[6 + kx][u], 0, 0, 0); \
                __builtin_amdgcn_sched_barrier(0);  /* the registers of a2 are free from here: the next reads may take them */ \
            }                                                                                                       \
            if constexpr ((R) + 1 < 6) {                                                                            \
                bn[0] = tr_pair16(xb0 + ((R) + 1) * XROW * 128, 1024);                                              \
                bn[1] = tr_pair16(xb1 + ((R) + 1) * XROW * 128, 1024);                                              \
                bn[2] = tr_pair16(xb2 + ((R) + 1) * XROW * 128, 1024);                                              \
            }                                                                                                       \
            if constexpr ((R) >= 1 && (R) <= 4) {                                                                   \
                _Pragma("unroll") for (int kx = 0; kx < 3; ++kx)                                                    \
                _Pragma("unroll") for (int u = 0; u < NU; ++u)                                                      \
                    acc[3 + kx][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[u], b[kx], acc[3 + kx][u], 0, 0, 0); \
            }                                                                                                       \
            if constexpr ((R) < 4) {                                                                                \
                _Pragma("unroll") for (int kx = 0; kx < 3; ++kx)                                                    \
                _Pragma("unroll") for (int u = 0; u < NU; ++u)                                                      \
                    acc[kx][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[u], b[kx], acc[kx][u], 0, 0, 0);        \
                if (my_bias) {                                                                                      \
                    _Pragma("unroll") for (int u = 0; u < NU; ++u) bsum[u] = add8(a0[u], bsum[u]);                  \
                }                                                                                                   \
            }                                                                                                       \
            asm volatile("" ::: "memory");                                                                          \
            b[0] = bn[0]; b[1] = bn[1]; b[2] = bn[2];                                                               \
        }
        FOSVOS_W2_ROW(0) FOSVOS_W2_ROW(1) FOSVOS_W2_ROW(2) FOSVOS_W2_ROW(3) FOSVOS_W2_ROW(4) FOSVOS_W2_ROW(5)
        if (has_next) FOSVOS_W2_ADVANCE()
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // tile + 1: this wave's DMA has landed ...
        __syncthreads();                                   // ... everyone's has, and image `cur` is retired
    }

    // ---- bias partials: lane (g, i) of wave (wc, wi) holds, for channel 32 wc + 16 u + i, the sum over its quarter of the
    // k index (group g) of the tiles that wave took: 16 partials per channel, added in a fixed order
    if (do_bias) {
        float *sb = reinterpret_cast<float *>(smem_w);  // [4 wi][4 g][BCO]
#pragma unroll
        for (int u = 0; u < NU; ++u) sb[(wi * 4 + g) * C::BCO + wc * 32 + u * 16 + (lane & 15)] = bsum[u];
        __syncthreads();
        if (tid < C::BCO && co0 + tid < a.Cor) {
            float acc_b = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) acc_b += sb[k * C::BCO + tid];
            a.bias_part[((int64_t)split * a.n_ci + ci_blk) * a.Cor + co0 + tid] = acc_b;
        }
    }
    // ---- slab write, laid out like dw (OIHW): register r of accumulator (t, u) of lane l is
    //   co = co0 + 32 wc + 16 u + 4 (l >> 4) + r,  ci = ci0 + 16 wi + (l & 15),  tap t: the 9 taps of an element are contiguous
    float *slab = a.slabs + (int64_t)split * a.Cor * a.Ci * 9;
    const int ci = ci0 + wi * 16 + (lane & 15);
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = co0 + wc * 32 + u * 16 + 4 * g + r;
            if (co >= a.Cor || (a.lab & 1)) continue;
            float *d = slab + ((int64_t)co * a.Ci + ci) * 9;
#pragma unroll
            for (int t = 0; t < 9; ++t) d[t] = acc[t][u][r];
        }
#undef FOSVOS_W2_ROW
#undef FOSVOS_W2_PIECES_OF_ROW
#undef FOSVOS_W2_PIECE_ANY
#undef FOSVOS_W2_TILE_SCALARS
#undef FOSVOS_W2_ADVANCE
#undef FOSVOS_W2_STAGE
#undef FOSVOS_W2_PIECE_X
#undef FOSVOS_W2_PIECE_Y
}
}  // namespace v2

// ---- reduction over splits for MANY layers per launch, two coalesced stages, every sum in a fixed order
// stage A (layers with more than kFold splits): slab 8j += slabs 8j+1 .. 8j+7, in place - one thread per float4 and
// group, so the early layers (few elements, hundreds of splits) spread over thousands of threads instead of a few dozen
// blocks walking 250 slabs each;  stage B: dw (+)= the remaining <= 32 partial slabs, db (+)= the bias partials.
constexpr int kFold = 8;

__global__ __launch_bounds__(256) void k_wgrad_fold(const WgradReduceTable t) {
    int e = 0;
    while (e + 1 < t.n && (int)blockIdx.x >= t.e[e + 1].fold_begin) ++e;
    const WgradReduceEntry &q = t.e[e];
    const int local = blockIdx.x - q.fold_begin;
    if (local >= q.fold_blocks) return;  // layers without a fold stage own no blocks
    const int per_group = q.n_blocks;     // blocks that cover the layer's elements once
    const int grp = local / per_group;
    const int64_t i4 = (int64_t)(local - grp * per_group) * 256 + threadIdx.x;
    if (i4 * 4 >= q.E_real) return;
    float *base = const_cast<float *>(q.slabs) + (int64_t)grp * kFold * q.E_pad + i4 * 4;
    const int n = min(kFold, q.S - grp * kFold);
    float4 v[kFold];
#pragma unroll
    for (int j = 0; j < kFold; ++j) v[j] = *reinterpret_cast<const float4 *>(base + (int64_t)min(j, n - 1) * q.E_pad);
    float4 acc = v[0];
#pragma unroll
    for (int j = 1; j < kFold; ++j)
        if (j < n) { acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w; }
    *reinterpret_cast<float4 *>(base) = acc;
}

__global__ __launch_bounds__(256) void k_wgrad_reduce(const WgradReduceTable t) {
    int e = 0;
    while (e + 1 < t.n && (int)blockIdx.x >= t.e[e + 1].block_begin) ++e;
    const WgradReduceEntry &q = t.e[e];
    const int local = blockIdx.x - q.block_begin;
    const int64_t i4 = (int64_t)local * 256 + threadIdx.x;  // float4 index inside the layer
    const bool folded = q.fold_blocks > 0;
    const int n_src = folded ? (q.S + kFold - 1) / kFold : q.S;
    const int64_t stride = (folded ? kFold : 1) * q.E_pad;
    if (i4 * 4 < q.E_real) {
        const float *src = q.slabs + i4 * 4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int s0 = 0; s0 < n_src; s0 += 8) {  // 8 slabs in flight, added in increasing s
            float4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const float4 *>(src + (int64_t)min(s0 + j, n_src - 1) * stride);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (s0 + j < n_src) { acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w; }
        }
        float4 *d = reinterpret_cast<float4 *>(q.dw + i4 * 4);
        if (q.accumulate) {
            const float4 o = *d;
            acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
        }
        *d = acc;
    }
    if (local == 0 && q.db) {  // the layer's bias gradient: S_bias partials per channel, fixed order
        // The 256 threads are P parts x Cg channels (Cg = Co rounded up to a power of two, at most 256): part p adds its
        // contiguous range of the partials with 16 loads in flight (a rolled loop paid one L2 round trip per partial: 65 us at
        // 250 splits; one thread per channel still left 16 of 256 threads walking 750 partials for a side_prep layer), then
        // the parts are added in order.
        __shared__ float s_part[256];
        int Cg = 1;
        while (Cg < q.Co && Cg < 256) Cg <<= 1;
        const int P = 256 / Cg, part = threadIdx.x / Cg, cl = threadIdx.x - part * Cg;
        const int per = (q.S_bias + P - 1) / P, s_lo = part * per, s_hi = min(s_lo + per, q.S_bias);
        for (int co0 = 0; co0 < q.Co; co0 += Cg) {
            const int co = co0 + cl;
            float acc_b = 0.f;
            if (co < q.Co) {
                for (int s0 = s_lo; s0 < s_hi; s0 += 16) {
                    float v[16];
#pragma unroll
                    for (int j = 0; j < 16; ++j) v[j] = q.bias_part[(int64_t)min(s0 + j, s_hi - 1) * q.Cor + co];
#pragma unroll
                    for (int j = 0; j < 16; ++j)
                        if (s0 + j < s_hi) acc_b += v[j];
                }
            }
            if (co0) __syncthreads();  // (the previous round's parts have been read)
            s_part[threadIdx.x] = acc_b;
            __syncthreads();
            if (part == 0 && co < q.Co) {
                float t = s_part[cl];
                for (int pp = 1; pp < P; ++pp) t += s_part[pp * Cg + cl];
                q.db[co] = q.accumulate ? q.db[co] + t : t;
            }
        }
    }
}

int target_blocks() {  // FOSVOS_WGRAD_BLOCKS: lab switch, read once
    static const int v = [] {
        const char *e = lab_env("FOSVOS_WGRAD_BLOCKS");
        const int n = e ? atoi(e) : 0;
        // The weight-gradient kernels run BESIDE the data-gradient chain (vgg_net.hip), and a CU that hosts one of these
        // workgroups has room for one igemm workgroup instead of two; slab bytes grow with the workgroup count.  Measured on
        // the fine-tune step with three frames per pass: 128 -> 927, 192 -> 952, 256 -> 945 frames/s
        return n > 0 ? n : 192;
    }();
    return v;
}

struct Plan {
    int Cor, Cy, side, tiles_x, tiles_y, n_tiles, tps, S, n_ci;
    size_t slab_bytes, bias_bytes;
};

// alone: the launch will find the chip idle (the cycle's last backward pass has run out of data-gradient kernels by the
// time stages 1-2 get their weight gradients): 256 workgroups, one per CU, instead of the shared-chip default
Plan make_plan(int N, int H, int W, int Ci, int Co, bool alone = false) {
    Plan p;
    p.side = Co % 64 != 0;            // side_prep: 16 outputs in a 32-wide dy image, the four-wave form of the kernel
    p.Cor = p.side ? Co : roundup(Co, 64);  // slab / bias-partial rows: side_prep keeps its 16 real channels only
    p.Cy = roundup(Co, 32);
    p.tiles_x = (int)cdiv(W, 16);
    p.tiles_y = (int)cdiv(H, TH);
    p.n_tiles = p.tiles_x * p.tiles_y * N;
    // workgroups per pixel split: 64 co x 64 ci each (side_prep: its 16 co x 64 ci)
    const int out_blocks = p.side ? Ci / 64 : (p.Cor / 64) * (Ci / 64);
    int S = (int)cdiv(alone ? std::max(256, target_blocks()) : target_blocks(), out_blocks);
    if (S > p.n_tiles) S = p.n_tiles;
    if (S < 1) S = 1;
    p.tps = (int)cdiv(p.n_tiles, S);
    p.S = (int)cdiv(p.n_tiles, p.tps);
    p.slab_bytes = (size_t)p.S * 9 * p.Cor * Ci * sizeof(float);
    p.n_ci = Ci / 64;  // the ci blocks of a split share out its bias sums
    p.bias_bytes = (size_t)p.S * p.n_ci * p.Cor * sizeof(float);
    return p;
}

int check_shape(int N, int H, int W, int Ci, int Co, const char *who) {
    FOSVOS_REQUIRE(N > 0 && H > 0 && W > 0 && Ci > 0 && Co > 0, FOSVOS_E_SHAPE, "%s: bad shape", who);
    FOSVOS_REQUIRE(Ci % BCI == 0, FOSVOS_E_SHAPE, "%s: Ci=%d must be a multiple of %d", who, Ci, BCI);
    FOSVOS_REQUIRE(Co % 64 == 0 || Co == 16, FOSVOS_E_SHAPE, "%s: Co=%d must be a multiple of 64, or 16 (side_prep)", who, Co);
    FOSVOS_REQUIRE(((int64_t)N * H * W + W + 1) * std::max(Ci, roundup(Co, 32)) * 2 < 0xffffffffLL, FOSVOS_E_SHAPE,
                   "%s: a tensor of %d x %d x %d x %d bf16 exceeds the 4 GB a buffer descriptor addresses", who, N, H, W,
                   std::max(Ci, roundup(Co, 32)));
    return FOSVOS_OK;
}

// Queue the reduction of the layer whose slabs the MFMA kernel just wrote, or (reduce == nullptr) run it now.
int finish_or_queue(const Plan &p, float *slabs, float *bias_part, float *dw, float *db, int Ci, int Co, int accumulate,
                    WgradReduceTable *reduce, int device, hipStream_t st) {
    return fosvos::wgrad_queue_reduce(slabs, bias_part, dw, db, p.S, p.S * p.n_ci, 9LL * Co * Ci, 9LL * p.Cor * Ci, Co, p.Cor,
                                      accumulate, reduce, device, st);
}
}  // namespace

extern "C" size_t fosvos_conv3x3_wgrad_workspace_bytes(int N, int H, int W, int Ci, int Co) {
    if (N <= 0 || H <= 0 || W <= 0 || Ci <= 0 || Co <= 0 || Ci % BCI != 0) return 0;
    const Plan p = make_plan(N, H, W, Ci, Co), q = make_plan(N, H, W, Ci, Co, true);  // room for either split count
    return std::max(p.slab_bytes + p.bias_bytes, q.slab_bytes + q.bias_bytes);
}

extern "C" int fosvos_conv3x3_wgrad(const uint16_t *x, const uint16_t *dy, float *dw, float *db, int N, int H, int W,
                                    int Ci, int Co, int accumulate, void *workspace, size_t workspace_bytes, int device,
                                    void *stream) {
    return fosvos::wgrad_impl(x, dy, dw, db, N, H, W, Ci, Co, accumulate, workspace, workspace_bytes, device, stream,
                              nullptr);
}

extern "C" int fosvos_conv3x3_wgrad_reduce(float *dw, float *db, int N, int H, int W, int Ci, int Co, int accumulate,
                                           void *workspace, size_t workspace_bytes, int device, void *stream) {
    FOSVOS_REQUIRE(dw && workspace, FOSVOS_E_ARG, "conv3x3_wgrad_reduce: null pointer");
    if (int rc = check_shape(N, H, W, Ci, Co, "conv3x3_wgrad_reduce")) return rc;
    const Plan p = make_plan(N, H, W, Ci, Co);
    FOSVOS_REQUIRE(workspace_bytes >= p.slab_bytes + p.bias_bytes, FOSVOS_E_WORKSPACE,
                   "conv3x3_wgrad_reduce: workspace %zu < %zu", workspace_bytes, p.slab_bytes + p.bias_bytes);
    FOSVOS_ENTER(device);
    float *slabs = reinterpret_cast<float *>(workspace);
    float *bias_part = reinterpret_cast<float *>(reinterpret_cast<char *>(workspace) + p.slab_bytes);
    return finish_or_queue(p, slabs, bias_part, dw, db, Ci, Co, accumulate, nullptr, device, (hipStream_t)stream);
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

int fosvos::wgrad_queue_reduce(const float *slabs, const float *bias_part, float *dw, float *db, int S, int S_bias,
                               int64_t E_real,
                               int64_t E_pad, int Co, int Cor, int accumulate, WgradReduceTable *reduce, int device,
                               void *stream) {
    WgradReduceTable local;
    local.n = 0;
    WgradReduceTable *t = reduce ? reduce : &local;
    FOSVOS_REQUIRE(t->n < kWgradReduceMax, FOSVOS_E_ARG, "wgrad: reduction queue full");
    FOSVOS_REQUIRE(E_real % 4 == 0 && E_pad % 4 == 0, FOSVOS_E_SHAPE, "wgrad: slab size must be a multiple of 4 floats");
    WgradReduceEntry &q = t->e[t->n];
    q.slabs = slabs; q.dw = dw; q.db = db; q.bias_part = db ? bias_part : nullptr;
    q.S = S; q.S_bias = S_bias; q.Co = Co; q.Cor = Cor; q.accumulate = accumulate;
    q.E_real = E_real;
    q.E_pad = E_pad;
    q.n_blocks = (int)cdiv(E_real / 4, 256);
    q.block_begin = t->n ? t->e[t->n - 1].block_begin + t->e[t->n - 1].n_blocks : 0;
    q.fold_blocks = S > kFold ? (int)cdiv(S, kFold) * q.n_blocks : 0;
    q.fold_begin = t->n ? t->e[t->n - 1].fold_begin + t->e[t->n - 1].fold_blocks : 0;
    ++t->n;
    return reduce ? FOSVOS_OK : fosvos::wgrad_reduce_all(t, device, stream);
}

int fosvos::wgrad_reduce_all(WgradReduceTable *reduce, int device, void *stream) {
    FOSVOS_REQUIRE(reduce, FOSVOS_E_ARG, "wgrad_reduce_all: null table");
    if (reduce->n == 0) return FOSVOS_OK;
    FOSVOS_ENTER(device);
    const WgradReduceEntry &last = reduce->e[reduce->n - 1];
    if (last.fold_begin + last.fold_blocks > 0) {
        FOSVOS_PROF("k_wgrad_fold", stream, 0.0);
        hipLaunchKernelGGL(k_wgrad_fold, dim3((unsigned)(last.fold_begin + last.fold_blocks)), dim3(256), 0,
                           (hipStream_t)stream, *reduce);
        FOSVOS_LAUNCH_CHECK();
    }
    FOSVOS_PROF("k_wgrad_reduce", stream, 0.0);
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)(last.block_begin + last.n_blocks)), dim3(256), 0, (hipStream_t)stream,
                       *reduce);
    FOSVOS_LAUNCH_CHECK();
    reduce->n = 0;
    return FOSVOS_OK;
}

int fosvos::wgrad_impl(const uint16_t *x, const uint16_t *dy, float *dw, float *db, int N, int H, int W, int Ci, int Co,
                       int accumulate, void *workspace, size_t workspace_bytes, int device, void *stream,
                       WgradReduceTable *reduce, bool alone) {
    FOSVOS_REQUIRE(x && dy && dw && workspace, FOSVOS_E_ARG, "conv3x3_wgrad: null pointer");
    if (int rc = check_shape(N, H, W, Ci, Co, "conv3x3_wgrad")) return rc;
    const Plan p = make_plan(N, H, W, Ci, Co, alone);
    FOSVOS_REQUIRE(workspace_bytes >= p.slab_bytes + p.bias_bytes, FOSVOS_E_WORKSPACE,
                   "conv3x3_wgrad: workspace %zu < %zu", workspace_bytes, p.slab_bytes + p.bias_bytes);
    FOSVOS_ENTER(device);
    hipStream_t st = (hipStream_t)stream;
    WgArgs a;
    a.x = x; a.dy = dy; a.slabs = reinterpret_cast<float *>(workspace);
    a.bias_part = db ? reinterpret_cast<float *>(reinterpret_cast<char *>(workspace) + p.slab_bytes) : nullptr;
    a.N = N; a.H = H; a.W = W; a.Ci = Ci; a.Cy = p.Cy; a.Cor = p.Cor;
    a.tiles_x = p.tiles_x; a.tiles_y = p.tiles_y; a.n_tiles = p.n_tiles; a.tiles_per_split = p.tps;
    a.S = p.S; a.n_ci = Ci / 64; a.n_co = p.side ? 1 : p.Cor / 64;
    {
#ifdef FOSVOS_LAB_BUILD  // timing experiments only (never in the shipped library: the switch drops the slab stores)
        static const int lab = lab_env_int("FOSVOS_WGRAD_LAB", 0);
        a.lab = lab;
#else
        a.lab = 0;
#endif
        const char *e = lab_env("FOSVOS_WGRAD_XCD");  // lab switch (read per call: one process can A/B): 0 = plain workgroup order
        a.xcd_order = !(e && atoi(e) == 0);
    }
    static bool once[64][2];  // per device: opt in to the dynamic LDS size
    const double flops = 2.0 * N * H * W * 9.0 * Ci * Co;
    const dim3 grid((unsigned)(p.S * a.n_ci * a.n_co));
    if (p.side) {
        using C2 = v2::Cfg2<true>;
        if (device >= 0 && device < 64 && !once[device][1]) {
            FOSVOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(v2::k_wgrad3x3_v2<true>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, C2::LDS_BYTES));
            once[device][1] = true;
        }
        FOSVOS_PROF("k_wgrad3x3_v2<true>", st, flops);
        hipLaunchKernelGGL(v2::k_wgrad3x3_v2<true>, grid, dim3(C2::NT), C2::LDS_BYTES, st, a);
    } else {
        using C2 = v2::Cfg2<false>;
        if (device >= 0 && device < 64 && !once[device][0]) {
            FOSVOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(v2::k_wgrad3x3_v2<false>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, C2::LDS_BYTES));
            once[device][0] = true;
        }
        FOSVOS_PROF("k_wgrad3x3_v2<false>", st, flops);
        hipLaunchKernelGGL(v2::k_wgrad3x3_v2<false>, grid, dim3(C2::NT), C2::LDS_BYTES, st, a);
    }
    FOSVOS_LAUNCH_CHECK();
    return finish_or_queue(p, a.slabs, a.bias_part, dw, db, Ci, Co, accumulate, reduce, device, st);
}
