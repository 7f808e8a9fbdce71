// Weight / bias gradient of the 3x3 conv on bf16 NHWC activations (autograd's backward-weight of
// Conv2d in stages[*] / side_prep[*], src/networks/osvos_vgg.py:42,92), Ci a multiple of 64.
//
//   dw[co][ci][ky][kx] = sum_{n,y,x} dy[n,y,x,co] * x[n,y+ky-1,x+kx-1,ci]        db[co] = sum dy[.,co]
//
// GEMM view: M = co, N = ci (x 9 taps), K = pixels: MFMA roofline.  v_mfma_f32_32x32x16_bf16, one k-step = 16
// consecutive pixels of one image row.  Both operands are stored channel-contiguous (NHWC) while the contraction runs
// over pixels, so both are transposed on the fly: a tile is staged in LDS as [32-channel half][pixel][64 B] and read with
// ds_read_b64_tr_b16, which hands a lane the 4 pixels x 1 channel column of a 4 x 16 block.  A 32-lane half of a read
// covers 4 pixels x 64 B = 256 contiguous bytes: conflict-free for every tap shift.
//
// Operand reuse is what this kernel is built around.  A tile is 8 rows x 16 pixels of dy and its 10 x 18 halo of x.  The
// MFMA loop walks the HALO rows: the three x fragments of halo row R (kx = 0,1,2) meet the dy fragments of tile rows
// R, R-1, R-2 (ky = 0,1,2), so every x fragment is read from LDS once and used three times, and a dy fragment stays in
// registers for three halo rows: 76 transposed reads for 72 MFMAs per wave and tile (the 16x16x32 form of round 1 needed
// 104 for 144, at twice the issue cost per FLOP).
//
// Workgroup = 4 waves in a WCO x WCI grid, wave (wc, wi) owns co 32wc..+31 x ci 32wi..+31 x 9 taps = nine 32x32
// accumulators (144 VGPRs).  <2,2>: 64 co x 64 ci (the backbone layers); <1,4>: 32 x 128 (side_prep: 16 real output
// channels in a 32-wide dy image).  The pixel range is split over blockIdx.x; each split writes one fp32 slab laid out
// like dw itself ([co][ci][9], OIHW), so the reduction over splits is a plain elementwise sum in split order
// (k_wgrad_reduce: bitwise reproducible, no float atomics, no transposes), queued per layer and run for many layers
// in one launch.
#include <stdlib.h>

#include "common.hpp"

using namespace fosvos;

namespace {
constexpr int TH = 8;                    // tile rows
constexpr int TPIX = TH * 16;            // 128 tile pixels
constexpr int HALO_W = 18;
constexpr int NPH = (TH + 2) * HALO_W;   // 180 halo pixels
constexpr int BCI = 64;                  // ci granularity of the grid (and of the API contract)

struct WgArgs {
    const uint16_t *x;   // [N,H,W,Ci]
    const uint16_t *dy;  // [N,H,W,Cy]  (Cy = roundup(Co,32))
    float *slabs;        // [S][Cor][Ci][9]
    float *bias_part;    // [S * gridDim.y][Cor] column sums of dy per split and ci block (each sums its share of the tiles), or null
    int N, H, W, Ci, Cy, Cor;
    int tiles_x, tiles_y, n_tiles, tiles_per_split;
#ifdef FOSVOS_WG_STAMP
    unsigned long long *stamps;  // diagnostic build only (tools/wgrad_stamp_lab.hip): 8 sums per workgroup
#endif
    int lab;  // timing-only switches (FOSVOS_WGRAD_LAB; wrong results): 1 no slab store, 2 loads from the zero page
    int S, n_ci, n_co, xcd_order;  // v2: 1-D grid of S * n_ci * n_co workgroups, decoded in the kernel (see there)
};

typedef __attribute__((address_space(3))) s16x4 *lds_s16x4_ptr;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// 8 bf16 of one channel: pixels p..p+3 (first read) and p+4..p+7 (second read, 4 x 64 B further)
__device__ __forceinline__ bf16x8 tr_pair(const char *p) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p + 256));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

#ifdef FOSVOS_WG_STAMP
unsigned long long *g_wg_stamps = nullptr;
// wave 0 only: phase p's clocks are added to sum[p]; the stamp's own lgkmcnt(0) keeps s_memtime ordered with LDS reads
#define FOSVOS_WG_STAMP_AT(p_)                                                                   \
    {                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        unsigned long long now_;                                                                 \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");             \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        st_sum[p_] += now_ - st_last;                                                            \
        st_last = now_;                                                                          \
    }
#else
#define FOSVOS_WG_STAMP_AT(p_)
#endif

#ifndef FOSVOS_WG_LPR
#define FOSVOS_WG_LPR 1
#endif
#ifndef FOSVOS_WG_BIAS_IN_ROWS
#define FOSVOS_WG_BIAS_IN_ROWS 0
#endif

template <int WCO, int WCI>
struct Cfg {
    static constexpr int BCO = 32 * WCO, BCIW = 32 * WCI;   // workgroup tile
    static constexpr int CHY = 4 * WCO, CHX = 4 * WCI;       // 16-byte pieces per pixel
    // half stride = pixels x 64 B + 64: the two halves a staging store hits (8 lanes = one pixel's 128 B) then sit on
    // different banks
    static constexpr int Y_HALF = TPIX * 64 + 64, X_HALF = NPH * 64 + 64;
    static constexpr int Y_BYTES = WCO * Y_HALF, X_BYTES = WCI * X_HALF, BUF_BYTES = Y_BYTES + X_BYTES;
    static constexpr int NT = 64 * WCO * WCI;                       // threads: one wave per 32 x 32 output block
    static constexpr int Y_IT = TPIX * CHY / NT;                   // staging pieces per thread
    static constexpr int X_IT = (NPH * CHX + NT - 1) / NT;
    static constexpr int LDS_BYTES = 2 * BUF_BYTES;                // two tile images (double buffer)
    static_assert(WCO * WCI == 4 || WCO * WCI == 8, "4 or 8 waves per workgroup");
    static_assert(TPIX * CHY % NT == 0, "dy pieces divide evenly");
};

template <int WCO, int WCI>
__global__ __launch_bounds__(64 * WCO * WCI) __attribute__((amdgpu_waves_per_eu(WCO * WCI / 4, 2))) void k_wgrad3x3(const WgArgs a) {
    using C = Cfg<WCO, WCI>;
    extern __shared__ __attribute__((aligned(16))) char smem_w[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave / WCI, wi = wave % WCI;
    const int split = blockIdx.x;
    const int ci0 = blockIdx.y * C::BCIW;
    const int co0 = blockIdx.z * C::BCO;
    const int H = a.H, W = a.W;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // transposed-read address of this lane inside a half image: 16-lane group g reads channel block g&1 of the half,
    // pixels 8(g>>1) + q (+4 for the second read); lane (q, p) of the group supplies row q, channels 4p..4p+3
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int lane_off = ((g >> 1) * 8 + q) * 64 + (g & 1) * 32 + p * 8;
    const int rd_y = wc * C::Y_HALF + lane_off;                 // + row * 16 * 64
    const int rd_x = C::Y_BYTES + wi * C::X_HALF + lane_off;    // + (R * 18 + kx) * 64

    // the bias gradient (column sums of dy) is the same for every ci block of a split: the ci blocks share it out by tile, so
    // no workgroup carries all of it (all workgroups of a launch end together: the slowest sets the kernel time)
    const bool do_bias = a.bias_part != nullptr;
    float bsum[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bsum[e] = 0.f;

    // ---- tile-independent staging plan of this thread: piece i = it * 256 + tid -> (pixel, 16-byte chunk)
    // y_goff / x_goff: BYTE offsets inside the tile, from the tile's dy origin / from the x halo origin (one row and one
    // pixel in front of the tile).  The loads are buffer loads: descriptor + this 32-bit offset + a scalar tile offset, and
    // an out-of-image piece gets the offset ~0, which the descriptor's range check answers with zeros (no zero page, no
    // 64-bit address arithmetic per piece)
    unsigned y_goff[C::Y_IT], x_goff[C::X_IT];
    int y_lds[C::Y_IT], y_rc[C::Y_IT];
#pragma unroll
    for (int it = 0; it < C::Y_IT; ++it) {
        const int i = it * C::NT + tid, pix = i / C::CHY, c = i % C::CHY;
        const int ty = pix >> 4, tx = pix & 15;
        y_rc[it] = (ty << 16) | tx;
        y_goff[it] = (unsigned)(((ty * W + tx) * a.Cy + co0 + c * 8) * 2);
        y_lds[it] = (c >> 2) * C::Y_HALF + pix * 64 + (c & 3) * 16;
    }
    int x_lds[C::X_IT], x_rc[C::X_IT];
#pragma unroll
    for (int it = 0; it < C::X_IT; ++it) {
        const int i = it * C::NT + tid, pix = i / C::CHX, c = i % C::CHX;
        const int hy = pix / HALO_W, hx = pix - hy * HALO_W;
        x_rc[it] = pix < NPH ? ((hy << 16) | hx) : (0x7fff << 16);  // slots past the halo: never in the image, never stored
        x_goff[it] = pix < NPH ? (unsigned)(((hy * W + hx) * a.Ci + ci0 + c * 8) * 2) : ~0u;  // slots past the halo: zeros
        x_lds[it] = C::Y_BYTES + (c >> 2) * C::X_HALF + pix * 64 + (c & 3) * 16;
    }
    // x descriptor: based one row and one pixel in FRONT of the tensor, so that halo offsets are non-negative (the bytes in
    // front of the tensor belong to out-of-image halo pieces, which are never requested)
    const int64_t halo_shift = (int64_t)(W + 1) * a.Ci;
    const unsigned y_total = (unsigned)((int64_t)a.N * H * W * a.Cy * 2), x_total = (unsigned)(((int64_t)a.N * H * W * a.Ci + halo_shift) * 2);
    const auto y_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(a.dy), 0, y_total, 0x00020000);
    const auto x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(a.x) - halo_shift, 0, x_total, 0x00020000);

    const int t_begin = split * a.tiles_per_split;
    const int t_end = min(t_begin + a.tiles_per_split, a.n_tiles);
    int lt_x = t_begin % a.tiles_x, lt_y = (t_begin / a.tiles_x) % a.tiles_y, lt_n = t_begin / (a.tiles_x * a.tiles_y);

    // Register-staged tile pipeline: tile t+1 is loaded into registers at the top of tile t (latency hidden by the MFMA
    // loop) and written to the other LDS image behind it; one barrier per tile.
    // The staging registers are NAMED scalars (macro-unrolled), not arrays: hipcc keeps the array form in scratch memory
    // (store after load, reload before the LDS write), which serialises the pipeline.
    static_assert(C::Y_IT <= 4 && C::X_IT <= 12, "staging register file");
    uint4 py0, py1, py2, py3;
    uint4 px0, px1, px2, px3, px4, px5, px6, px7, px8, px9, px10, px11;
    py0 = py1 = py2 = py3 = px0 = px1 = px2 = px3 = px4 = px5 = px6 = px7 = px8 = px9 = px10 = px11 = make_uint4(0, 0, 0, 0);
#define FOSVOS_WG_LDY(i_)                                                                               \
    if constexpr ((i_) < C::Y_IT) {                                                                     \
        unsigned v_ = y_goff[i_];                                                                       \
        if (!interior_) v_ = ((y_rc[i_] >> 16) < vrows_ && (y_rc[i_] & 0xffff) < vcols_ && live_) ? v_ : ~0u; \
        py##i_ = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(y_rsrc, v_, ysoff_, 0)); \
    }
#define FOSVOS_WG_LDX(i_)                                                                               \
    if constexpr ((i_) < C::X_IT) {                                                                     \
        unsigned v_ = x_goff[i_];                                                                       \
        if (!interior_) {                                                                               \
            const int hy_ = x_rc[i_] >> 16, hx_ = x_rc[i_] & 0xffff;                                    \
            v_ = (hy_ >= 1 - y0_ && hy_ <= vrows_ && hx_ >= 1 - x0_ && hx_ <= vcols_ && live_) ? v_ : ~0u; \
        }                                                                                               \
        px##i_ = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, v_, xsoff_, 0)); \
    }
    // tile-level scalars of the tile the NEXT loads fetch: (lt_x, lt_y, lt_n)
#define FOSVOS_WG_TILE_SCALARS()                                                                        \
    const int y0_ = lt_y * TH, x0_ = lt_x * 16;                                                         \
    const int vrows_ = H - y0_, vcols_ = W - x0_;                                                       \
    const int64_t org_ = ((int64_t)lt_n * H + y0_) * W + x0_;                                           \
    const unsigned ysoff_ = (unsigned)(org_ * a.Cy * 2), xsoff_ = (unsigned)(org_ * a.Ci * 2);          \
    /* interior: the whole halo lies inside the image and the loads are wanted - no per-piece checks */ \
    const bool interior_ = y0_ >= 1 && y0_ + TH < H && x0_ >= 1 && x0_ + 16 < W && live_;
#define FOSVOS_WG_ADVANCE()                                                                             \
    if (++lt_x == a.tiles_x) {                                                                          \
        lt_x = 0;                                                                                       \
        if (++lt_y == a.tiles_y) { lt_y = 0; ++lt_n; }                                                  \
    }
    // piece p of the Y_IT + X_IT staging loads of a tile (dy pieces first)
#define FOSVOS_WG_LDP(p_)                                                                               \
    if constexpr ((p_) < C::Y_IT) { FOSVOS_WG_LDY_((p_)) }                                              \
    else if constexpr ((p_) < C::Y_IT + C::X_IT) { FOSVOS_WG_LDX_((p_) - C::Y_IT) }
#define FOSVOS_WG_LDY_(i_)                                                                              \
    if constexpr ((i_) == 0) { FOSVOS_WG_LDY(0) } else if constexpr ((i_) == 1) { FOSVOS_WG_LDY(1) }    \
    else if constexpr ((i_) == 2) { FOSVOS_WG_LDY(2) } else if constexpr ((i_) == 3) { FOSVOS_WG_LDY(3) }
#define FOSVOS_WG_LDX_(i_)                                                                              \
    if constexpr ((i_) == 0) { FOSVOS_WG_LDX(0) } else if constexpr ((i_) == 1) { FOSVOS_WG_LDX(1) }    \
    else if constexpr ((i_) == 2) { FOSVOS_WG_LDX(2) } else if constexpr ((i_) == 3) { FOSVOS_WG_LDX(3) } \
    else if constexpr ((i_) == 4) { FOSVOS_WG_LDX(4) } else if constexpr ((i_) == 5) { FOSVOS_WG_LDX(5) } \
    else if constexpr ((i_) == 6) { FOSVOS_WG_LDX(6) } else if constexpr ((i_) == 7) { FOSVOS_WG_LDX(7) } \
    else if constexpr ((i_) == 8) { FOSVOS_WG_LDX(8) } else if constexpr ((i_) == 9) { FOSVOS_WG_LDX(9) } \
    else if constexpr ((i_) == 10) { FOSVOS_WG_LDX(10) } else if constexpr ((i_) == 11) { FOSVOS_WG_LDX(11) }
#define FOSVOS_WG_LOAD_TILE()                                                                           \
    {                                                                                                   \
        FOSVOS_WG_TILE_SCALARS()                                                                        \
        FOSVOS_WG_LDY(0) FOSVOS_WG_LDY(1) FOSVOS_WG_LDY(2) FOSVOS_WG_LDY(3) \
        FOSVOS_WG_LDX(0) FOSVOS_WG_LDX(1) FOSVOS_WG_LDX(2) FOSVOS_WG_LDX(3) FOSVOS_WG_LDX(4) FOSVOS_WG_LDX(5) FOSVOS_WG_LDX(6) FOSVOS_WG_LDX(7) FOSVOS_WG_LDX(8) FOSVOS_WG_LDX(9) FOSVOS_WG_LDX(10) FOSVOS_WG_LDX(11) \
        FOSVOS_WG_ADVANCE()                                                                             \
    }
#define FOSVOS_WG_STY(i_, img_) \
    if constexpr ((i_) < C::Y_IT) *reinterpret_cast<uint4 *>((img_) + y_lds[i_]) = py##i_;
#define FOSVOS_WG_STX(i_, img_)        \
    if constexpr ((i_) < C::X_IT) {    \
        if ((i_) * C::NT + tid < NPH * C::CHX) *reinterpret_cast<uint4 *>((img_) + x_lds[i_]) = px##i_; \
    }
#define FOSVOS_WG_STORE_TILE(img_) \
    { FOSVOS_WG_STY(0, img_) FOSVOS_WG_STY(1, img_) FOSVOS_WG_STY(2, img_) FOSVOS_WG_STY(3, img_) FOSVOS_WG_STX(0, img_) FOSVOS_WG_STX(1, img_) FOSVOS_WG_STX(2, img_) FOSVOS_WG_STX(3, img_) FOSVOS_WG_STX(4, img_) FOSVOS_WG_STX(5, img_) FOSVOS_WG_STX(6, img_) FOSVOS_WG_STX(7, img_) FOSVOS_WG_STX(8, img_) FOSVOS_WG_STX(9, img_) FOSVOS_WG_STX(10, img_) FOSVOS_WG_STX(11, img_) }

#ifdef FOSVOS_WG_STAMP
    unsigned long long st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last)::"memory");
#endif
    if (t_begin < t_end) {
        const bool live_ = !(a.lab & 2);
        FOSVOS_WG_LOAD_TILE()
        FOSVOS_WG_STORE_TILE(smem_w)
    }
    __syncthreads();
    FOSVOS_WG_STAMP_AT(0)
    for (int tile = t_begin; tile < t_end; ++tile) {
        char *cur = smem_w + ((tile - t_begin) & 1) * C::BUF_BYTES;
        char *nxt = smem_w + (((tile - t_begin) & 1) ^ 1) * C::BUF_BYTES;
        // The next tile's staging loads are NOT issued in one burst: a CU takes in ~17 B/clock from L2, so the 40 KB of a
        // tile need as long as its 72 MFMAs per wave, and a wave that issues all of them up front sits in the load
        // instructions (the memory pipe accepts them only as data returns) instead of starting its matrix work - measured
        // with tools/wgrad_stamp_lab.hip: 2270 clocks per tile in the issue block.  One or two pieces ride in every
        // halo row of the MFMA loop instead.  Without a next tile the pieces read the zero page (never stored).
        const bool has_next = tile + 1 < t_end;
        const bool live_ = has_next && !(a.lab & 2);
        FOSVOS_WG_TILE_SCALARS()
        FOSVOS_WG_STAMP_AT(1)
        if (!FOSVOS_WG_BIAS_IN_ROWS && do_bias && (tile % (int)gridDim.y) == (int)blockIdx.y) {  // thread t keeps chunk t % CHY
#pragma unroll
            for (int k = 0; k < C::Y_IT; ++k) {
                float f[8];
                unpack8(*reinterpret_cast<const uint4 *>(cur + y_lds[k]), f);
#pragma unroll
                for (int e = 0; e < 8; ++e) bsum[e] += f[e];
            }
        }
        FOSVOS_WG_STAMP_AT(2)
        // ---- MFMA loop over the 10 halo rows: x fragments of row R meet dy rows R (ky 0), R-1 (ky 1), R-2 (ky 2).
        // The transposed reads of row R+1 are placed in front of the MFMAs of row R (a compiler memory fence per row keeps
        // them - and the row's staging loads - in their row), so a read has a row of matrix work to land.
        const char *yb = cur + rd_y, *xb = cur + rd_x;
        bf16x8 a0, a1, a2, an, b0, b1, b2, bn0, bn1, bn2;
        a1 = a2 = an = bf16x8{};
        bn0 = bn1 = bn2 = bf16x8{};
        a0 = tr_pair(yb);
        b0 = tr_pair(xb + 0 * 64);
        b1 = tr_pair(xb + 1 * 64);
        b2 = tr_pair(xb + 2 * 64);
        constexpr int NP = C::Y_IT + C::X_IT;  // staging pieces per tile, dealt over the TH + 2 rows
#define FOSVOS_WG_ROW(R)                                                                                \
        {                                                                                                   \
            /* FOSVOS_WG_LPR staging pieces per row from the first row on */                                 \
            FOSVOS_WG_LDP(FOSVOS_WG_LPR * (R))                                                              \
            if constexpr (FOSVOS_WG_LPR > 1) { FOSVOS_WG_LDP(FOSVOS_WG_LPR * (R) + 1) }                     \
            if constexpr (NP > (TH + 2) && FOSVOS_WG_LPR == 1) { FOSVOS_WG_LDP((TH + 2) + (R)) }            \
            /* bias: column sums of the dy tile, one 16-byte piece per row (thread t keeps chunk t % CHY) */ \
            if constexpr (FOSVOS_WG_BIAS_IN_ROWS && (R) < C::Y_IT) {                                        \
                if (do_bias) {                                                                              \
                    float f_[8];                                                                            \
                    unpack8(*reinterpret_cast<const uint4 *>(cur + y_lds[R]), f_);                          \
                    _Pragma("unroll") for (int e = 0; e < 8; ++e) bsum[e] += f_[e];                         \
                }                                                                                           \
            }                                                                                               \
            if constexpr ((R) + 1 < TH + 2) {                                                               \
                bn0 = tr_pair(xb + (((R) + 1) * HALO_W + 0) * 64);                                          \
                bn1 = tr_pair(xb + (((R) + 1) * HALO_W + 1) * 64);                                          \
                bn2 = tr_pair(xb + (((R) + 1) * HALO_W + 2) * 64);                                          \
                if constexpr ((R) + 1 < TH) an = tr_pair(yb + ((R) + 1) * 16 * 64);                         \
            }                                                                                               \
            {                                                                                               \
                if constexpr ((R) < TH) {                                                                   \
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0], 0, 0, 0);              \
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[1], 0, 0, 0);              \
                    acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2, acc[2], 0, 0, 0);              \
                }                                                                                           \
                if constexpr ((R) >= 1 && (R) <= TH) {                                                      \
                    acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[3], 0, 0, 0);              \
                    acc[4] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[4], 0, 0, 0);              \
                    acc[5] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc[5], 0, 0, 0);              \
                }                                                                                           \
                if constexpr ((R) >= 2) {                                                                   \
                    acc[6] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0, acc[6], 0, 0, 0);              \
                    acc[7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, acc[7], 0, 0, 0);              \
                    acc[8] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc[8], 0, 0, 0);              \
                }                                                                                           \
            }                                                                                               \
            asm volatile("" ::: "memory");                                                                  \
            a2 = a1; a1 = a0; a0 = an; b0 = bn0; b1 = bn1; b2 = bn2;                                        \
        }
        static_assert(TH + 2 == 10 && NP <= 2 * (TH + 2), "row macro instances / pieces per row");
        FOSVOS_WG_ROW(0) FOSVOS_WG_ROW(1) FOSVOS_WG_ROW(2) FOSVOS_WG_ROW(3) FOSVOS_WG_ROW(4)
        FOSVOS_WG_ROW(5) FOSVOS_WG_ROW(6) FOSVOS_WG_ROW(7) FOSVOS_WG_ROW(8) FOSVOS_WG_ROW(9)
        if (has_next) FOSVOS_WG_ADVANCE()
        FOSVOS_WG_STAMP_AT(3)
        // image `nxt` was last read during tile-1 (every wave has passed that tile's barrier)
        if (has_next) FOSVOS_WG_STORE_TILE(nxt)
        FOSVOS_WG_STAMP_AT(4)
        __syncthreads();  // tile+1 is in place and image `cur` is retired
        FOSVOS_WG_STAMP_AT(5)
    }
    if (do_bias) {
        // 256 x 8 partials -> BCO channel sums in a fixed order: channel ch = 8 c + e lives in the threads t = c (mod CHY)
        float *sb = reinterpret_cast<float *>(smem_w);  // [NT][9] floats
#pragma unroll
        for (int e = 0; e < 8; ++e) sb[tid * 9 + e] = bsum[e];
        __syncthreads();
        if (tid < C::BCO && co0 + tid < a.Cor) {
            const int c = tid >> 3, e = tid & 7;
            float acc_b = 0.f;
            for (int t2 = c; t2 < C::NT; t2 += C::CHY) acc_b += sb[t2 * 9 + e];
            a.bias_part[((int64_t)split * gridDim.y + blockIdx.y) * a.Cor + co0 + tid] = acc_b;
        }
    }
    // ---- slab write, laid out like dw (OIHW): accumulator register r of lane l is
    //   co = co0 + 32 wc + (r&3) + 8 (r>>2) + 4 (l>>5),  ci = ci0 + 32 wi + (l&31),  9 taps contiguous (36 B);
    // a wave instruction covers 32 consecutive ci of one co row
    float *slab = a.slabs + (int64_t)split * a.Cor * a.Ci * 9;
    const int ci = ci0 + wi * 32 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = co0 + wc * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (co >= a.Cor || (a.lab & 1)) continue;  // side_prep: rows 16..31 of the fragment are padding (registers 8..15)
        float *d = slab + ((int64_t)co * a.Ci + ci) * 9;
#pragma unroll
        for (int t = 0; t < 9; ++t) d[t] = acc[t][r];
    }
#ifdef FOSVOS_WG_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    FOSVOS_WG_STAMP_AT(6)
    if (a.stamps && tid == 0) {
        unsigned long long *o = a.stamps + (((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8;
        for (int i = 0; i < 8; ++i) o[i] = st_sum[i];
    }
#endif
}

// =====================================================================================================================
// Round 3 form of the MFMA kernel for the backbone layers (Co % 64 == 0): same tile (8 x 16 pixels of dy, 10 x 18 halo of
// x), same grid, same slabs - rebuilt around three measurements of the form above (1 wave per SIMD, 280 registers, 33-35 % of
// the MFMA cycles when alone on the chip, 25 % of its wave cycles in s_waitcnt / s_barrier):
//   * staging is LDS-DMA (buffer_load ... lds): the next tile goes from L2 straight into the other LDS image while the
//     MFMA loop runs - no staging registers (-40), no ds_write, nothing to wait for behind the loop but the DMA itself;
//     pieces are 8 pixels x 128 B, i.e. whole 128-byte lines of the NHWC tensors;
//   * 8 waves per workgroup, two per SIMD, each 32 co x 16 ci x 9 taps on v_mfma_f32_16x16x32_bf16 (18 accumulators of 4
//     registers): ~150 registers per wave, so one of these workgroups and one 192-register igemm workgroup still share a CU;
//   * the bias gradient (column sums of dy) is one extra MFMA against a vector of ones, taken in turn by the waves.
// LDS image of a tile: [pixel][128 B = 64 channels], 16-byte chunk c of pixel px stored at chunk slot c ^ 2((px >> 1) & 3).
// An LDS-DMA instruction writes lane i's 16 bytes at base + 16 i, so the swizzle is applied to the SOURCE address; the
// transposed reads apply the same XOR.  Row strides (16 px for dy, 24 px for the x halo: 18 used) are multiples of 8 px, so
// the XOR term of a lane's address does not change from row to row.  With it a 32-lane half of a ds_read_b64_tr_b16 covers
// 8 consecutive pixels x 32 B on 64 distinct banks for every tap shift (simulated with the bank rule of the guide).
// k index of an MFMA (32 pixels): lane group g, element j  <->  tile row r + 4 (g >> 1), pixel 4 (g & 1) + (j & 3) + 8 (j >> 2):
// any bijection does as long as both operands use the same one - this one makes the halves of a read contiguous.
// The loop walks R = 0..5: the x fragments of halo rows (R, R + 4) meet the dy fragments of tile rows (R - ky, R + 4 - ky).
#ifndef FOSVOS_W2_PPR
#define FOSVOS_W2_PPR 2  // staging pieces per row of the MFMA loop (lab: 3 = all six in the first two rows)
#endif
namespace v2 {
constexpr int XROW = 24;                          // pixels per halo row in LDS (18 used)
constexpr int Y_BYTES = TPIX * 128;               // 16 KB
constexpr int X_BYTES = (TH + 2) * XROW * 128;    // 30 KB
constexpr int BUF_BYTES = Y_BYTES + X_BYTES;
constexpr int LDS_BYTES = 2 * BUF_BYTES;          // 92 KB: two tile images
constexpr int NT = 512;
constexpr int Y_PIECES = TPIX / 8, X_PIECES = (TH + 2) * XROW / 8;   // 16 + 30 pieces of 1 KB per tile
constexpr int N_IT = (Y_PIECES + X_PIECES + 7) / 8;                   // pieces per wave (6; waves 6, 7 issue 5)
static_assert(Y_PIECES == 16 && X_PIECES == 30 && N_IT == 6, "piece schedule below");

typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ bf16x8 tr_pair16(const char *p) {  // 8 pixels of one channel: p .. +3, then 8 pixels on (+1024 B)
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p + 1024));
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

__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_wgrad3x3_v2(const WgArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_w[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave >> 2, wi = wave & 3;          // 32-co block, 16-ci block of this wave
    // Workgroup -> (pixel split, ci block, co block).  The n_ci * n_co workgroups of a split walk the same tiles: each x slice
    // is read by n_co of them, each dy slice by n_ci.  The dispatcher deals consecutive workgroup ids round-robin over the 8
    // XCDs (each with its own L2), so with a plain 3-D grid the workgroups that share operands never meet in an L2 and every
    // slice is fetched from beyond L2 once per reader (measured: the kernel moved 2x its algorithmic bytes and did not get
    // faster when its staging stalls were removed).  Remapped - a speed choice, never correctness: the ids that share an XCD
    // (id % 8) take one contiguous eighth of the (split, ci, co) space, co fastest, and start together.
    int bid = blockIdx.x;
    if (a.xcd_order) {
        const int n_wg = gridDim.x, q8 = n_wg >> 3, r8 = n_wg & 7, xcd = bid & 7;
        bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);  // bijective for any n_wg
    }
    const int co_blk = bid % a.n_co, ci_blk = (bid / a.n_co) % a.n_ci, split = bid / (a.n_co * a.n_ci);
    const int ci0 = ci_blk * 64, co0 = co_blk * 64;
    const int H = a.H, W = a.W;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void *)smem_w);

    f32x4 acc[9][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) acc[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum[2] = {0.f, 0.f};  // bias gradient: this lane's share (its 8 pixels of a k-step) of sum over pixels of dy[., co]

    // ---- transposed-read addresses of this lane (bytes inside a tile image)
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int pxl = 4 * (g & 1) + q;                                   // pixel inside the row, first read
    auto swz = [](int px, int c) { return (c ^ (2 * ((px >> 1) & 3))) * 16; };
    int rd_y[2], rd_x[3];
#pragma unroll
    for (int u = 0; u < 2; ++u)
        rd_y[u] = ((4 * (g >> 1)) * 16 + pxl) * 128 + swz(pxl, 2 * (2 * wc + u) + (p >> 1)) + (p & 1) * 8;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
        rd_x[kx] = Y_BYTES + ((4 * (g >> 1)) * XROW + pxl + kx) * 128 + swz(pxl + kx, 2 * wi + (p >> 1)) + (p & 1) * 8;

    // ---- staging plan: piece pc = 8 it + wave; pieces 0..15 are dy, 16..45 x.  Lane i fetches the 16 bytes that belong at
    // LDS offset 1024 pc' + 16 i: pixel 8 pc' + (i >> 3), chunk slot i & 7 -> source chunk (i & 7) ^ swizzle(pixel).
    // The two dy pieces of a wave are 4 tile rows apart (same columns): one offset register serves both.
    const int ppx = lane >> 3, slot = lane & 7;
    unsigned goff_y, goff_x[N_IT - 2];
    {
        const int pix = wave * 8 + ppx, ty = pix >> 4, tx = pix & 15;
        goff_y = (unsigned)(((ty * W + tx) * a.Cy + co0 + (slot ^ (2 * ((tx >> 1) & 3))) * 8) * 2);
    }
#pragma unroll
    for (int it = 2; it < N_IT; ++it) {
        const int pix = ((it - 2) * 8 + wave) * 8 + ppx, hy = pix / XROW, hx = pix - hy * XROW;
        const bool used = (it - 2) * 8 + wave < X_PIECES && hx < HALO_W;
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
            const int pix_ = ((it_) * 8 + wave) * 8 + ppx;                                                      \
            v_ = ((pix_ >> 4) < vrows_ && (pix_ & 15) < vcols_) ? v_ : ~0u;                                     \
        }                                                                                                       \
        dma16(y_rsrc, (img_) + ((it_) * 8 + wave) * 1024, v_, ysoff_);                                          \
    }
#define FOSVOS_W2_PIECE_X(it_, img_)                                                                            \
    if ((it_) < N_IT - 1 || wave < 6) {                                                                         \
        unsigned v_ = goff_x[(it_) - 2];                                                                        \
        if (!interior_) {                                                                                       \
            const int pix_ = (((it_) - 2) * 8 + wave) * 8 + ppx, hy_ = pix_ / XROW, hx_ = pix_ - hy_ * XROW;    \
            v_ = (hy_ >= 1 - y0_ && hy_ <= vrows_ && hx_ >= 1 - x0_ && hx_ <= vcols_) ? v_ : ~0u;               \
        }                                                                                                       \
        dma16(x_rsrc, (img_) + Y_BYTES + (((it_) - 2) * 8 + wave) * 1024, v_, xsoff_);                          \
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
#define FOSVOS_W2_STAGE(img_)                                                                                   \
    {                                                                                                           \
        FOSVOS_W2_TILE_SCALARS()                                                                                \
        FOSVOS_W2_PIECE_Y(0, img_) FOSVOS_W2_PIECE_Y(1, img_)                                                   \
        FOSVOS_W2_PIECE_X(2, img_) FOSVOS_W2_PIECE_X(3, img_) FOSVOS_W2_PIECE_X(4, img_) FOSVOS_W2_PIECE_X(5, img_) \
        FOSVOS_W2_ADVANCE()                                                                                     \
    }
    // piece `it_` of the next tile, issued from inside the MFMA loop (one per row: see there)
#define FOSVOS_W2_PIECE(it_, img_)                                                                              \
    if (has_next) {                                                                                             \
        if constexpr ((it_) < 2) { FOSVOS_W2_PIECE_Y(it_, img_) } else { FOSVOS_W2_PIECE_X(it_, img_) }         \
    }

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
        // may start now.  Its six pieces are NOT issued in one burst: an LDS-DMA instruction holds its wave for ~60-180
        // clocks while the memory pipe takes it in, and eight waves bursting together left the matrix pipe idle for the
        // first ~1000 clocks of every tile.  Two pieces ride in front of each of the first three rows; the last ones then
        // have the other three rows of matrix work (half the tile) to land under before the wave waits for them.
        const bool has_next = tile + 1 < t_end;
        const unsigned nxt_ = lds0 + (par ^ 1) * BUF_BYTES;
        FOSVOS_W2_TILE_SCALARS()
        // the bias sums of a tile are taken by one ci block of the grid and, inside it, by one of the four ci waves in turn
        const bool my_bias = do_bias && (tile % a.n_ci) == ci_blk && ((tile / a.n_ci) & 3) == wi;

        const char *yb0 = cur + rd_y[0], *yb1 = cur + rd_y[1];
        const char *xb0 = cur + rd_x[0], *xb1 = cur + rd_x[1], *xb2 = cur + rd_x[2];
        bf16x8 a0[2], a1[2], a2[2], b[3], bn[3];
        a0[0] = a0[1] = a1[0] = a1[1] = a2[0] = a2[1] = bn[0] = bn[1] = bn[2] = bf16x8{};
        b[0] = tr_pair16(xb0);
        b[1] = tr_pair16(xb1);
        b[2] = tr_pair16(xb2);
        // Row R: the dy fragments of row R are requested in front of the row's MFMAs and used by its LAST six (ky = 0); the x
        // fragments of row R + 1 are requested behind the first six (ky = 2), whose dy registers they may then take, and have
        // the other twelve to land under: every request has matrix work of this very wave to cover it (the two waves of a SIMD run in lockstep - same program, one
        // barrier per tile - and do not cover each other's LDS latency).  A compiler fence per row keeps the reads in their row.
#define FOSVOS_W2_ROW(R)                                                                                        \
        {                                                                                                           \
            a2[0] = a1[0]; a2[1] = a1[1]; a1[0] = a0[0]; a1[1] = a0[1];                                             \
            if constexpr (FOSVOS_W2_PPR == 2 && (R) < 3) { FOSVOS_W2_PIECE(2 * (R), nxt_) FOSVOS_W2_PIECE(2 * (R) + 1, nxt_) } \
            if constexpr (FOSVOS_W2_PPR == 3 && (R) < 2) {                                                          \
                FOSVOS_W2_PIECE(3 * (R), nxt_) FOSVOS_W2_PIECE(3 * (R) + 1, nxt_) FOSVOS_W2_PIECE(3 * (R) + 2, nxt_) } \
            if constexpr ((R) < 4) {                                                                                \
                a0[0] = tr_pair16(yb0 + (R) * 16 * 128);                                                            \
                a0[1] = tr_pair16(yb1 + (R) * 16 * 128);                                                            \
            }                                                                                                       \
            if constexpr ((R) >= 2) {                                                                               \
                _Pragma("unroll") for (int kx = 0; kx < 3; ++kx)                                                    \
                _Pragma("unroll") for (int u = 0; u < 2; ++u)                                                       \
                    acc[6 + kx][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[u], b[kx], acc[6 + kx][u], 0, 0, 0); \
                __builtin_amdgcn_sched_barrier(0);  /* the registers of a2 are free from here: the next reads may take them */ \
            }                                                                                                       \
            if constexpr ((R) + 1 < 6) {                                                                            \
                bn[0] = tr_pair16(xb0 + ((R) + 1) * XROW * 128);                                                    \
                bn[1] = tr_pair16(xb1 + ((R) + 1) * XROW * 128);                                                    \
                bn[2] = tr_pair16(xb2 + ((R) + 1) * XROW * 128);                                                    \
            }                                                                                                       \
            if constexpr ((R) >= 1 && (R) <= 4) {                                                                   \
                _Pragma("unroll") for (int kx = 0; kx < 3; ++kx)                                                    \
                _Pragma("unroll") for (int u = 0; u < 2; ++u)                                                       \
                    acc[3 + kx][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[u], b[kx], acc[3 + kx][u], 0, 0, 0); \
            }                                                                                                       \
            if constexpr ((R) < 4) {                                                                                \
                _Pragma("unroll") for (int kx = 0; kx < 3; ++kx)                                                    \
                _Pragma("unroll") for (int u = 0; u < 2; ++u)                                                       \
                    acc[kx][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[u], b[kx], acc[kx][u], 0, 0, 0);        \
                if (my_bias) {                                                                                      \
                    bsum[0] = add8(a0[0], bsum[0]);                                                                 \
                    bsum[1] = add8(a0[1], bsum[1]);                                                                 \
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
        float *sb = reinterpret_cast<float *>(smem_w);  // [4 wi][4 g][64 co]
#pragma unroll
        for (int u = 0; u < 2; ++u) sb[(wi * 4 + g) * 64 + wc * 32 + u * 16 + (lane & 15)] = bsum[u];
        __syncthreads();
        if (tid < 64 && co0 + tid < a.Cor) {
            float acc_b = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) acc_b += sb[k * 64 + tid];
            a.bias_part[((int64_t)split * a.n_ci + ci_blk) * a.Cor + co0 + tid] = acc_b;
        }
    }
    // ---- slab write, laid out like dw (OIHW): register r of accumulator (t, u) of lane l is
    //   co = co0 + 32 wc + 16 u + 4 (l >> 4) + r,  ci = ci0 + 16 wi + (l & 15),  tap t: the 9 taps of an element are contiguous
    float *slab = a.slabs + (int64_t)split * a.Cor * a.Ci * 9;
    const int ci = ci0 + wi * 16 + (lane & 15);
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = co0 + wc * 32 + u * 16 + 4 * g + r;
            if (co >= a.Cor || (a.lab & 1)) continue;
            float *d = slab + ((int64_t)co * a.Ci + ci) * 9;
#pragma unroll
            for (int t = 0; t < 9; ++t) d[t] = acc[t][u][r];
        }
#undef FOSVOS_W2_ROW
#undef FOSVOS_W2_PIECE
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
    if (local == 0 && q.db) {  // the layer's bias gradient: S partials per channel, fixed order
        // 16 partials in flight per thread (a rolled loop paid one L2 round trip per split: 65 us at 250 splits)
        for (int co = threadIdx.x; co < q.Co; co += 256) {
            float acc_b = 0.f;
            for (int s0 = 0; s0 < q.S_bias; s0 += 16) {
                float v[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) v[j] = q.bias_part[(int64_t)min(s0 + j, q.S_bias - 1) * q.Cor + co];
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    if (s0 + j < q.S_bias) acc_b += v[j];
            }
            q.db[co] = q.accumulate ? q.db[co] + acc_b : acc_b;
        }
    }
}

int target_blocks() {  // FOSVOS_WGRAD_BLOCKS: lab switch, read once
    static const int v = [] {
        const char *e = getenv("FOSVOS_WGRAD_BLOCKS");
        const int n = e ? atoi(e) : 0;
        // The weight-gradient kernels run BESIDE the data-gradient chain (vgg_net.hip), and a CU that hosts one of these
        // workgroups has room for one igemm workgroup instead of two; slab bytes grow with the workgroup count.  Measured on
        // the fine-tune step with three frames per pass: 128 -> 927, 192 -> 952, 256 -> 945 frames/s
        return n > 0 ? n : 192;
    }();
    return v;
}

int wgrad_waves() {  // FOSVOS_WGRAD_WAVES=8: 128 co x 64 ci eight-wave workgroups where Co allows (lab switch, read once)
    static const int v = [] {
        const char *e = getenv("FOSVOS_WGRAD_WAVES");
        return e && atoi(e) == 8 ? 8 : 4;
    }();
    return v;
}

bool wgrad_v2() {  // FOSVOS_WGRAD_V2=0: the round-2 kernel (lab switch, read at every call so that one process can A/B)
    const char *e = getenv("FOSVOS_WGRAD_V2");
    return !(e && atoi(e) == 0);
}

struct Plan {
    int Cor, Cy, side, wide, tiles_x, tiles_y, n_tiles, tps, S, n_ci;
    size_t slab_bytes, bias_bytes;
};

// alone: the launch will find the chip idle (the cycle's last backward pass has run out of data-gradient kernels by the
// time stages 1-2 get their weight gradients): 256 workgroups, one per CU, instead of the shared-chip default
Plan make_plan(int N, int H, int W, int Ci, int Co, bool alone = false) {
    Plan p;
    p.side = Co % 64 != 0;            // side_prep: 16 outputs in a 32-wide dy image, <1,4> workgroups
    p.Cor = p.side ? Co : roundup(Co, 64);  // slab / bias-partial rows: side_prep keeps its 16 real channels only
    p.Cy = roundup(Co, 32);
    p.tiles_x = (int)cdiv(W, 16);
    p.tiles_y = (int)cdiv(H, TH);
    p.n_tiles = p.tiles_x * p.tiles_y * N;
    // wide (opt-in): 8-wave workgroups, 128 co x 64 ci, two waves per SIMD.  Measured: 1.55x the per-CU rate of the 4-wave
    // form alone on the chip, but it owns its CUs (112 KB of LDS, the whole register file) and the data-gradient kernels
    // beside it lose more than the weight-gradient stream gains: 705 vs 738 frames/s on the fine-tune step
    p.wide = !p.side && Co % 128 == 0 && wgrad_waves() == 8;
    const int out_blocks = p.side ? Ci / 128 : (p.Cor / (p.wide ? 128 : 64)) * (Ci / 64);
    int S = (int)cdiv(alone ? std::max(256, target_blocks()) : target_blocks(), out_blocks);
    if (S > p.n_tiles) S = p.n_tiles;
    if (S < 1) S = 1;
    p.tps = (int)cdiv(p.n_tiles, S);
    p.S = (int)cdiv(p.n_tiles, p.tps);
    p.slab_bytes = (size_t)p.S * 9 * p.Cor * Ci * sizeof(float);
    p.n_ci = p.side ? Ci / 128 : Ci / 64;  // ci blocks (gridDim.y) share the bias sums of a split
    p.bias_bytes = (size_t)p.S * p.n_ci * p.Cor * sizeof(float);
    return p;
}

int check_shape(int N, int H, int W, int Ci, int Co, const char *who) {
    FOSVOS_REQUIRE(N > 0 && H > 0 && W > 0 && Ci > 0 && Co > 0, FOSVOS_E_SHAPE, "%s: bad shape", who);
    FOSVOS_REQUIRE(Ci % BCI == 0, FOSVOS_E_SHAPE, "%s: Ci=%d must be a multiple of %d", who, Ci, BCI);
    FOSVOS_REQUIRE(Co % 64 == 0 || (Co == 16 && Ci % 128 == 0), FOSVOS_E_SHAPE,
                   "%s: Co=%d must be a multiple of 64, or 16 with Ci a multiple of 128", who, Co);
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
    a.S = p.S; a.n_ci = a.n_co = 0; a.xcd_order = 0;
    {
        static const int lab = getenv("FOSVOS_WGRAD_LAB") ? atoi(getenv("FOSVOS_WGRAD_LAB")) : 0;
        a.lab = lab;
    }
#ifdef FOSVOS_WG_STAMP
    a.stamps = g_wg_stamps;
#endif
    static bool once[64][4];  // per device: opt in to the dynamic LDS size
    const double flops = 2.0 * N * H * W * 9.0 * Ci * Co;
    if (p.wide) {
        using C = Cfg<4, 2>;
        if (device >= 0 && device < 64 && !once[device][2]) {
            FOSVOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_wgrad3x3<4, 2>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
            once[device][2] = true;
        }
        const dim3 grid((unsigned)p.S, (unsigned)(Ci / C::BCIW), (unsigned)(p.Cor / C::BCO));
        FOSVOS_PROF("k_wgrad3x3<4, 2>", st, flops);
        hipLaunchKernelGGL((k_wgrad3x3<4, 2>), grid, dim3(C::NT), C::LDS_BYTES, st, a);
    } else if (!p.side && wgrad_v2()) {
        if (device >= 0 && device < 64 && !once[device][3]) {
            FOSVOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(v2::k_wgrad3x3_v2),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, v2::LDS_BYTES));
            once[device][3] = true;
        }
        a.S = p.S; a.n_ci = Ci / 64; a.n_co = p.Cor / 64;
        {
            const char *e = getenv("FOSVOS_WGRAD_XCD");  // lab switch: 0 = plain (split fastest) workgroup order
            a.xcd_order = !(e && atoi(e) == 0);
        }
        const dim3 grid((unsigned)(p.S * a.n_ci * a.n_co));
        FOSVOS_PROF("k_wgrad3x3_v2", st, flops);
        hipLaunchKernelGGL(v2::k_wgrad3x3_v2, grid, dim3(v2::NT), v2::LDS_BYTES, st, a);
    } else if (!p.side) {
        using C = Cfg<2, 2>;
        if (device >= 0 && device < 64 && !once[device][0]) {
            FOSVOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_wgrad3x3<2, 2>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
            once[device][0] = true;
        }
        const dim3 grid((unsigned)p.S, (unsigned)(Ci / C::BCIW), (unsigned)(p.Cor / C::BCO));
        FOSVOS_PROF("k_wgrad3x3<2, 2>", st, flops);
        hipLaunchKernelGGL((k_wgrad3x3<2, 2>), grid, dim3(256), C::LDS_BYTES, st, a);
    } else {
        using C = Cfg<1, 4>;
        if (device >= 0 && device < 64 && !once[device][1]) {
            FOSVOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_wgrad3x3<1, 4>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
            once[device][1] = true;
        }
        const dim3 grid((unsigned)p.S, (unsigned)(Ci / C::BCIW), 1);
        FOSVOS_PROF("k_wgrad3x3<1, 4>", st, flops);
        hipLaunchKernelGGL((k_wgrad3x3<1, 4>), grid, dim3(256), C::LDS_BYTES, st, a);
    }
    FOSVOS_LAUNCH_CHECK();
    return finish_or_queue(p, a.slabs, a.bias_part, dw, db, Ci, Co, accumulate, reduce, device, st);
}
