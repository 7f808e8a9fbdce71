// 3x3 / pad 1 / stride 1 convolution on bf16 NHWC as an implicit GEMM on the CDNA4 matrix cores
// (v_mfma_f32_16x16x32_bf16, fp32 accumulate).  One kernel serves
//   forward  : Conv2d(3x3)+bias[+ReLU]       (src/networks/osvos_vgg.py:42,92-93)
//   dgrad    : the same contraction on the rotated/transposed filter image, with the ReLU
//              backward of the producing layer and the "other consumer" gradient add fused into
//              the epilogue.
//
// Decomposition (MFMA roofline; GEMM view M = pixels, N = out channels, K = 9 * in channels):
//   workgroup = 256 threads = 4 waves, output tile = TH x TW pixels x BN channels;
//   K loop over chunks of 32 input channels: the (TH+2) x (TW+2) input halo tile is staged into
//   LDS ONCE per chunk and re-read for all 9 taps (im2col exists only as LDS addressing, never in
//   HBM); the 9 x 32 x BN weight tile comes from the pre-packed image in contiguous 16-byte runs.
//   An M fragment = 16 consecutive pixels of one tile row, so a tap shift is a pixel-index offset.
//
// LDS images (16-byte slots = 8 bf16 channels):
//   A  [4 k-groups][NPIX_PAD pixels]      slot = kg * NPIX_PAD + P      (NPIX_PAD % 16 == 0)
//   B  [9 taps][4 k-groups][BN channels]  slot = (tap*4 + kg) * BN + co
// A ds_read_b128 fragment read (lane l: row l&15, k-group l>>4) then touches, per 16-lane service
// group, 16 distinct slots of the 256-byte bank row for ANY base pixel: conflict-free for all taps.
#include <stdlib.h>

#include "common.hpp"

using namespace fosvos;

namespace {

// internal epilogue flag (never crosses the ABI): ReLU after the addend has been added - the residual form
constexpr unsigned kReluAfterAdd = 0x100u;
// ... and: keep only the even (row, column) pixels, stored as [N,(H-1)/2+1,(W-1)/2+1,Cout] - a stride-2 conv computed at
// stride 1 (4x the arithmetic of a strided kernel, still several times faster than the vector-ALU conv at these widths)
constexpr unsigned kSubsample2 = 0x200u;
// ... and: one launch per layer, no split-K.  The inference chain of the ResNet path pays ~5 us per dependent launch, more
// than the slabs + epilogue kernel win on its small maps (measured at 1080p: -7 % per frame at exponent 1, -3 % at 2)
constexpr unsigned kNoSplitK = 0x400u;

struct ConvArgs {
    const uint16_t *x;         // [N,H,W,Cin] bf16, Cin % 32 == 0
    const uint16_t *w;         // packed [Cin/32][9][4][Co_pad][8]
    const float *bias;         // [Cout] or null
    const uint16_t *relu_src;  // [N,H,W,Cout] or null
    const uint8_t *relu_bits;  // the same mask as one BIT per element ([N,H,W,Cout/8] bytes, bit e of a byte = channel 8 g + e > 0) or null
    const uint16_t *addend;    // [N,H,W,Cout] or null
    void *y;                   // [N,H,W,Cout] bf16 (or fp32)
    uint16_t *y_pool;          // optional [N,ceil(H/2),ceil(W/2),Cout] bf16: 2x2 ceil-mode max pool of y, same launch
    const uint16_t *unpool_g;  // UNPOOL kernels: [N,ceil(H/2),ceil(W/2),Cout] bf16, the gradient wrt the 2x2 max pool of relu_src
    int N, H, W, Cin, Cout, Co_pad;
    int tiles_x, tiles_y;
    unsigned flags;
    int swizzle;       // 0: blockIdx = (tile, channel block); 1 / 2: 1-D grid remapped per XCD (see the kernel), channel block
                       // fastest / slowest
    int k_splits;      // > 1: blockIdx.z owns a range of K chunks and writes raw fp32 partials
    int chunks_per_split;
    float *partial;    // [k_splits][N*H*W][Cout] fp32 when k_splits > 1
    // Reciprocal multipliers for the workgroup -> (tile, channel block) arithmetic (see fast_div): four runtime divisions cost
    // the prologue of every workgroup a reciprocal on the vector ALU and a round trip to the scalar unit each.  0 = divide.
    unsigned m_ncb, m_nt, m_tx, m_ty;
    int n_t;           // pixel tiles of the launch (swizzled grids)
#ifdef FOSVOS_STAMP
    unsigned long long *stamps;  // diagnostic build only (tools/igemm_lab.hip): 16 slots per workgroup
#endif
};

#ifdef FOSVOS_STAMP
unsigned long long *g_stamps = nullptr;
#define FOSVOS_STAMP_AT(i)                                                                               \
    if (a.stamps && threadIdx.x == 0)                                                                    \
        a.stamps[((((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) << 4) + (i)] = \
            __builtin_amdgcn_s_memtime();
#define FOSVOS_STAMP_RT(i)                                                                               \
    if (a.stamps && threadIdx.x == 0)                                                                    \
        a.stamps[((((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) << 4) + (i)] = \
            __builtin_amdgcn_s_memrealtime();
#else
#define FOSVOS_STAMP_AT(i)
#define FOSVOS_STAMP_RT(i)
#endif

// n / d for n * d < 2^32 with m = floor(2^32 / d) + 1 (host side: recip_mul; exact: the error n (m d - 2^32) / (d 2^32) stays
// below 1 / d); m == 0 means "no multiplier" (d == 1, or a launch too large for the bound): plain division.
__device__ __forceinline__ unsigned fast_div(unsigned n, unsigned d, unsigned m) { return m ? __umulhi(n, m) : n / d; }
unsigned recip_mul(int64_t n_max, int64_t d) {
    return (d >= 2 && n_max * d < (int64_t(1) << 32)) ? (unsigned)((int64_t(1) << 32) / d + 1) : 0u;
}

template <int TH_, int TW_, int BN_, int WM_, int WN_>
struct Tile {
    static constexpr int TH = TH_, TW = TW_, BN = BN_, WM = WM_, WN = WN_;
    static constexpr int BM = TH * TW;
    static constexpr int HALO_W = TW + 2, HALO_H = TH + 2;
    static constexpr int NPIX = HALO_W * HALO_H;
    static constexpr int NPIX8 = (NPIX + 7) / 8 * 8;
    static constexpr int NPIX_PAD = (NPIX + 15) / 16 * 16;
    static constexpr int WAVE_M = BM / WM, WAVE_N = BN / WN;
    static constexpr int MF = WAVE_M / 16, NF = WAVE_N / 16;
    static constexpr int A_SLOTS = 4 * NPIX_PAD;
    static constexpr int B_SLOTS = 36 * BN;
    static constexpr int NT = 64 * WM * WN;  // threads per workgroup
    static constexpr int A_IT = (NPIX8 * 4 + NT - 1) / NT;
    static constexpr int B_IT = (B_SLOTS + NT - 1) / NT;
    static constexpr int OUT_LD = BN + 8;  // bf16 elements per staged output row (16-B aligned rows)
    static constexpr int LDS_MAIN = (A_SLOTS + B_SLOTS) * 16;
    static constexpr int LDS_OUT = BM * OUT_LD * 2;
    static constexpr int LDS_BYTES = LDS_MAIN > LDS_OUT ? LDS_MAIN : LDS_OUT;
#ifndef FOSVOS_MID_WPE
#define FOSVOS_MID_WPE 3
#endif
    // resident workgroups per CU (= waves per SIMD): what 160 KB of LDS allows, at most 2 for the 256-px tiles
    static constexpr int WPE = (BM >= 256 || BN < 64) ? 2 : FOSVOS_MID_WPE;  // (an 8-wave workgroup fills two by itself)
    static_assert(WM * WN == 4 || WM * WN == 8, "4 or 8 waves per workgroup");
    static_assert(TW % 16 == 0 && WAVE_M % 16 == 0 && WAVE_N % 16 == 0, "fragment alignment");
    static_assert(BM % WM == 0 && BN % WN == 0, "wave split");
};

// Scheduling of one tap of the MFMA loop (row-major over the MF pixel fragments).  After row i the register of
// pixel fragment i is dead, so the same row of the NEXT tap is read into it, and the next tap's weight fragments
// (second register set) are read behind the first rows; the prefetch loads of the next chunk ride in the last rows.
//   row i : NF MFMAs | 1 + nb(i) LDS reads | (1 global load)
// sched_group_barrier masks (LLVM AMDGPU): 0x008 MFMA, 0x020 VMEM read, 0x100 DS read.
template <int MF, int NF>
struct TapPlan {
    static constexpr int rows_b = MF / 2 > 0 ? MF / 2 : 1;      // rows that carry weight-fragment reads
    static constexpr int per_row = (NF + rows_b - 1) / rows_b;  // weight fragments read behind one such row
    static constexpr int nb(int i) {
        const int left = NF - i * per_row;
        return left <= 0 ? 0 : (left < per_row ? left : per_row);
    }
};
template <int MF, int NF, int I, bool NEXT, int V>
__device__ __forceinline__ void sched_row() {
    if constexpr (I < MF) {
        __builtin_amdgcn_sched_group_barrier(0x008, NF, 0);
        if constexpr (NEXT) __builtin_amdgcn_sched_group_barrier(0x100, 1 + TapPlan<MF, NF>::nb(I), 0);
        constexpr int rows_v = V < MF ? V : MF;  // the last rows_v rows carry the V prefetch loads
        constexpr int nv = I < MF - rows_v ? 0 : (I == MF - 1 ? V - (rows_v - 1) : 1);
        if constexpr (nv > 0) __builtin_amdgcn_sched_group_barrier(0x020, nv, 0);
    }
}
template <int MF, int NF, bool NEXT, int V>
__device__ __forceinline__ void sched_tap() {
    static_assert(MF <= 8, "sched_tap rows");
    sched_row<MF, NF, 0, NEXT, V>(); sched_row<MF, NF, 1, NEXT, V>(); sched_row<MF, NF, 2, NEXT, V>();
    sched_row<MF, NF, 3, NEXT, V>(); sched_row<MF, NF, 4, NEXT, V>(); sched_row<MF, NF, 5, NEXT, V>();
    sched_row<MF, NF, 6, NEXT, V>(); sched_row<MF, NF, 7, NEXT, V>();
}

// waves_per_eu(2,2): two workgroups per CU, up to 256 registers each - without the cap hipcc spills the
// prefetched chunk to scratch to reach an occupancy the LDS image would not allow anyway
// UNPOOL (the side_prep data gradient of the training step): y = [relu_src > 0] * conv + maxpool2x2_ceil_bwd(relu_src, unpool_g) -
// the pool backward of the stage output runs in THIS epilogue instead of as its own pass whose result would be read back as
// the addend (per stage output of P pooled bytes: 9 P of pool pass + 12 P here become 9 P).  See the epilogue.
template <class T, bool OUT_F32, bool UNPOOL = false>
__global__ __launch_bounds__(T::NT) __attribute__((amdgpu_waves_per_eu(2, T::WPE))) void k_conv3x3_igemm(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
    uint4 *sA = smem;
    uint4 *sB = smem + T::A_SLOTS;
#ifdef FOSVOS_IGEMM_PRIO
    __builtin_amdgcn_s_setprio(FOSVOS_IGEMM_PRIO);  // lab build: these waves win the SIMD's arbitration against the weight-gradient waves beside them
#endif

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / T::WN, wn = wave % T::WN;
    // Workgroup -> (pixel tile, channel block).  The dispatcher deals consecutive workgroup ids round-robin over the 8 XCDs
    // (each with its own L2), so with the plain mapping neighbouring tiles - which share halo rows - and the channel blocks
    // of one tile - which read the SAME input tile - never meet in an L2.  Remapped (a speed choice, never correctness):
    // the ids that share an XCD (id % 8) walk one contiguous eighth of the (tile, channel block) space in order, with the
    // channel blocks of a tile next to each other (mode 1, the default), or tile by tile inside one channel block (mode 2).
    int t = blockIdx.x;
    int cb = blockIdx.y;
    if (a.swizzle) {
        const int n_wg = gridDim.x, n_cb = a.Cout / T::BN, n_t = a.n_t;
        const int q8 = n_wg >> 3, r8 = n_wg & 7, xcd = t & 7;
        const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (t >> 3);  // bijective for any n_wg
        if (a.swizzle == 1) { t = fast_div(swz, n_cb, a.m_ncb); cb = swz - t * n_cb; }
        else { cb = fast_div(swz, n_t, a.m_nt); t = swz - cb * n_t; }
    }
    const int t_row = fast_div(t, a.tiles_x, a.m_tx);   // (frame, tile row)
    const int tx_i = t - t_row * a.tiles_x;
    const int n = fast_div(t_row, a.tiles_y, a.m_ty);
    const int ty_i = t_row - n * a.tiles_y;
    const int y0 = ty_i * T::TH, x0 = tx_i * T::TW;
    const int n0 = cb * T::BN;
    const int H = a.H, W = a.W, Cin = a.Cin;
    const int n_chunks = Cin >> 5;
    const uint16_t *xn = a.x + (int64_t)n * H * W * Cin;

    // ---- A staging plan: 8 consecutive lanes = 8 consecutive pixels of one k-group (conflict-free
    // ds_write_b128), the 4 k-groups of those pixels in the next 3 octets (same 64-byte global runs)
    // Loads are UNCONDITIONAL (a load under a branch makes hipcc drain vmcnt(0) inside the prefetch block); an element of the
    // zero padding (or beyond the halo) gets the byte offset ~0, which the descriptor's range check turns into zeros - no
    // select when the registers go to LDS (that was 24 of the ~40 vector-ALU instructions of a chunk).
    int a_goff[T::A_IT];
    int a_slot[T::A_IT];
#pragma unroll
    for (int it = 0; it < T::A_IT; ++it) {
        const int idx = it * T::NT + tid;
        const int oct = idx >> 5, within = idx & 31;
        const int kg = within >> 3;
        const int P = oct * 8 + (within & 7);
        a_slot[it] = -1;
        a_goff[it] = -1;
        if (P < T::NPIX) {
            const int hy = P / T::HALO_W, hx = P - hy * T::HALO_W;
            const int gy = y0 + hy - 1, gx = x0 + hx - 1;
            a_slot[it] = kg * T::NPIX_PAD + P;
            if (gy >= 0 && gy < H && gx >= 0 && gx < W) a_goff[it] = ((gy * W + gx) * Cin + kg * 8) * 2;  // inside image n
        }
    }

    f32x4 acc[T::MF][T::NF];
#pragma unroll
    for (int i = 0; i < T::MF; ++i)
#pragma unroll
        for (int j = 0; j < T::NF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment base addresses (slots)
    int a_base[T::MF];
#pragma unroll
    for (int i = 0; i < T::MF; ++i) {
        const int pb = wm * T::WAVE_M + i * 16;
        const int ty = pb / T::TW, tx = pb % T::TW;
        a_base[i] = (lane >> 4) * T::NPIX_PAD + ty * T::HALO_W + tx + (lane & 15);
    }
    const int b_base = (lane >> 4) * T::BN + wn * T::WAVE_N + (lane & 15);

    const int c_begin = a.k_splits > 1 ? blockIdx.z * a.chunks_per_split : 0;
    const int c_end = a.k_splits > 1 ? min(c_begin + a.chunks_per_split, n_chunks) : n_chunks;

    // Software pipeline (register prefetch, one LDS image): the global loads of chunk cc+1 are issued right
    // before the MFMA phase of chunk cc and only waited for when they are written to LDS after it, so the
    // L2/HBM latency hides under the matrix work (and under the co-resident workgroup's).
    // The prefetch registers are NAMED scalars (macro-unrolled), not arrays: hipcc demoted the array form to
    // scratch memory (store-after-load + reload), which serialises the pipeline.
    static_assert(T::A_IT <= 6 && T::B_IT <= 9, "prefetch register file");
    uint4 pa0, pa1, pa2, pa3, pa4, pa5, pb0, pb1, pb2, pb3, pb4, pb5, pb6, pb7, pb8;
    [[maybe_unused]] uint4 pa6, pa7, pa8;  // named by the tap macros, never live (A_IT <= 6)
    pa0 = pa1 = pa2 = pa3 = pa4 = pa5 = pb0 = pb1 = pb2 = pb3 = pb4 = pb5 = pb6 = pb7 = pb8 = make_uint4(0, 0, 0, 0);
    // buffer loads: one 32-bit byte offset per lane (+ a wave-uniform SGPR offset per chunk / slice) instead of
    // a 64-bit address per load, and the descriptor's range check covers the weight-slice overrun of the
    // 16-channel tiles (those lanes are never stored to LDS)
    // The tap loop prefetches unconditionally; behind the last chunk the descriptors have zero records, so the
    // range check drops those loads in the texture unit (no branch, no second copy of the MFMA loop).
    const int x_bytes = H * W * Cin * 2, w_bytes = (n_chunks * 36 * a.Co_pad - n0) * 16;
    auto x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(xn), 0, x_bytes, 0x00020000);
    auto w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(a.w) + (int64_t)n0 * 8, 0, w_bytes,
                                                    0x00020000);
    static_assert(T::NT % T::BN == 0, "weight staging assumes BN divides the workgroup size");
    const int b_voff = ((tid / T::BN) * a.Co_pad + tid % T::BN) * 16;
    const int b_row_bytes = a.Co_pad * 16;
#define FOSVOS_LD_A(i)                                                                                  \
    if constexpr (i < T::A_IT)                                                                          \
        pa##i = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, a_goff[i], cc_ * 64, 0));
#define FOSVOS_LD_B(i)                                                                                  \
    if constexpr (i < T::B_IT)                                                                          \
        pb##i = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(                        \
            w_rsrc, b_voff, (cc_ * 36 + i * (T::NT / T::BN)) * b_row_bytes, 0));
#define FOSVOS_LOAD_CHUNK(cc_expr)                                                                      \
    {                                                                                                   \
        const int cc_ = (cc_expr);                                                                      \
        FOSVOS_LD_A(0) FOSVOS_LD_A(1) FOSVOS_LD_A(2) FOSVOS_LD_A(3) FOSVOS_LD_A(4) FOSVOS_LD_A(5)       \
        FOSVOS_LD_B(0) FOSVOS_LD_B(1) FOSVOS_LD_B(2) FOSVOS_LD_B(3) FOSVOS_LD_B(4) FOSVOS_LD_B(5)       \
        FOSVOS_LD_B(6) FOSVOS_LD_B(7) FOSVOS_LD_B(8)                                                    \
    }
#define FOSVOS_ST_A(i)                                                                                  \
    if constexpr (i < T::A_IT) {                                                                        \
        if (a_slot[i] >= 0) sA[a_slot[i]] = pa##i;                                                      \
    }
#define FOSVOS_ST_B(i)                                                                                  \
    if constexpr (i < T::B_IT) {                                                                        \
        if (i * T::NT + tid < T::B_SLOTS) sB[i * T::NT + tid] = pb##i;                                  \
    }
#define FOSVOS_STORE_CHUNK()                                                                            \
    {                                                                                                   \
        FOSVOS_ST_A(0) FOSVOS_ST_A(1) FOSVOS_ST_A(2) FOSVOS_ST_A(3) FOSVOS_ST_A(4) FOSVOS_ST_A(5)       \
        FOSVOS_ST_B(0) FOSVOS_ST_B(1) FOSVOS_ST_B(2) FOSVOS_ST_B(3) FOSVOS_ST_B(4) FOSVOS_ST_B(5)       \
        FOSVOS_ST_B(6) FOSVOS_ST_B(7) FOSVOS_ST_B(8)                                                    \
    }

    // Tap loop, software-pipelined by hand (see TapPlan): no LDS round trip and no texture-address stall
    // (64 B/clk: 15 loads x 8 waves is ~1.7k clocks when issued in one burst) in front of the matrix pipe.
#define FOSVOS_RD_A(tap_, i_) \
    af[i_] = __builtin_bit_cast(bf16x8, sA[a_base[i_] + ((tap_) / 3) * T::HALO_W + (tap_) % 3]);
#define FOSVOS_RD_B(tap_, buf_, j_) \
    bfr[buf_][j_] = __builtin_bit_cast(bf16x8, sB[b_base + (tap_) * 4 * T::BN + (j_) * 16]);
#define FOSVOS_ROW(tap_, i_)                                                                            \
    if constexpr ((i_) < T::MF) {                                                                       \
        _Pragma("unroll") for (int j = 0; j < T::NF; ++j) acc[i_][j] =                                  \
            __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[(tap_) & 1][j], af[i_], acc[i_][j], 0, 0, 0);   \
        if constexpr ((tap_) < 8) {                                                                     \
            FOSVOS_RD_A((tap_) + 1, i_)                                                                 \
            _Pragma("unroll") for (int j = 0; j < TP::nb(i_); ++j)                                      \
                FOSVOS_RD_B((tap_) + 1, ((tap_) + 1) & 1, (i_) * TP::per_row + j)                       \
        }                                                                                               \
    }
#define FOSVOS_TAP(tap_)                                                                                \
    {                                                                                                   \
        FOSVOS_ROW(tap_, 0) FOSVOS_ROW(tap_, 1) FOSVOS_ROW(tap_, 2) FOSVOS_ROW(tap_, 3)                 \
        FOSVOS_ROW(tap_, 4) FOSVOS_ROW(tap_, 5) FOSVOS_ROW(tap_, 6) FOSVOS_ROW(tap_, 7)                 \
        {                                                                                               \
            const int cc_ = cc + 1;                                                                     \
            FOSVOS_LD_B(tap_)                                                                           \
            FOSVOS_LD_A(tap_)                                                                           \
        }                                                                                               \
        sched_tap<T::MF, T::NF, ((tap_) < 8), 1 + ((tap_) < T::A_IT ? 1 : 0)>();                        \
    }

    FOSVOS_STAMP_RT(11)
    FOSVOS_STAMP_AT(0)
    if (c_begin < c_end) FOSVOS_LOAD_CHUNK(c_begin)
    FOSVOS_STAMP_AT(1)
    using TP = TapPlan<T::MF, T::NF>;
    static_assert(T::MF <= 8, "row macros cover up to 8 pixel fragments per wave");
    bf16x8 af[T::MF], bfr[2][T::NF];
    for (int cc = c_begin; cc < c_end; ++cc) {
        if (cc > c_begin) __syncthreads();  // every wave finished reading the previous chunk
        if (cc == c_end - 1) { FOSVOS_STAMP_AT(6) }
        FOSVOS_STORE_CHUNK()
        if (cc == c_begin) { FOSVOS_STAMP_AT(2) }
        if (cc == c_end - 1) { FOSVOS_STAMP_AT(7) }
        __syncthreads();
        if (cc == c_begin) { FOSVOS_STAMP_AT(3) }
#pragma unroll
        for (int i = 0; i < T::MF; ++i) FOSVOS_RD_A(0, i)
#pragma unroll
        for (int j = 0; j < T::NF; ++j) FOSVOS_RD_B(0, 0, j)
        __builtin_amdgcn_sched_group_barrier(0x100, T::MF + T::NF, 0);
        if (cc + 1 == c_end) {  // nothing left to prefetch: zero-record descriptors
            x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(xn), 0, 0, 0x00020000);
            w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(a.w) + (int64_t)n0 * 8, 0, 0, 0x00020000);
        }
        FOSVOS_TAP(0) FOSVOS_TAP(1) FOSVOS_TAP(2) FOSVOS_TAP(3) FOSVOS_TAP(4) FOSVOS_TAP(5) FOSVOS_TAP(6)
        FOSVOS_TAP(7) FOSVOS_TAP(8)
        if (cc == c_begin) { FOSVOS_STAMP_AT(5) }
    }
    FOSVOS_STAMP_AT(8)

    // ---- epilogue.  The MFMAs ran with the weight fragment as the row operand, so
    //   acc[i][j][r] = out[pixel wm*WAVE_M + 16 i + (lane&15)][channel wn*WAVE_N + 16 j + 4 (lane>>4) + r]:
    // a lane owns 4 consecutive channels of one pixel (one 8-byte bf16 / 16-byte fp32 store).
    const int q = lane >> 4, cl = lane & 15;
    if (a.k_splits > 1) {  // split-K: raw fp32 partial sums, finished by k_splitk_epilogue
        float *po = a.partial + (int64_t)blockIdx.z * a.N * H * W * a.Cout;
#pragma unroll
        for (int i = 0; i < T::MF; ++i) {
            const int pix = wm * T::WAVE_M + i * 16 + cl;
            const int gy = y0 + pix / T::TW, gx = x0 + pix % T::TW;
            if (gy < H && gx < W) {
                float *row = po + (((int64_t)n * H + gy) * W + gx) * a.Cout + n0 + wn * T::WAVE_N + q * 4;
#pragma unroll
                for (int j = 0; j < T::NF; ++j)
                    *reinterpret_cast<float4 *>(row + j * 16) =
                        make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            }
        }
        return;
    }
    if (OUT_F32) {
        float *yo = reinterpret_cast<float *>(a.y);
#pragma unroll
        for (int j = 0; j < T::NF; ++j) {
            const int co = n0 + wn * T::WAVE_N + j * 16 + q * 4;
            float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a.bias) b = *reinterpret_cast<const float4 *>(a.bias + co);
#pragma unroll
            for (int i = 0; i < T::MF; ++i) {
                const int pix = wm * T::WAVE_M + i * 16 + cl;
                const int gy = y0 + pix / T::TW, gx = x0 + pix % T::TW;
                if (gy < H && gx < W) {
                    float4 v = make_float4(acc[i][j][0] + b.x, acc[i][j][1] + b.y, acc[i][j][2] + b.z,
                                           acc[i][j][3] + b.w);
                    if (a.flags & FOSVOS_CONV_RELU)
                        v = make_float4(relu_f(v.x), relu_f(v.y), relu_f(v.z), relu_f(v.w));
                    *reinterpret_cast<float4 *>(yo + (((int64_t)n * H + gy) * W + gx) * a.Cout + co) = v;
                }
            }
        }
        return;
    }
    constexpr int VEC_PER_PIX = T::BN / 8;
    constexpr int OUT_N = T::BM * VEC_PER_PIX;      // 16-byte output vectors of the tile
    constexpr int OUT_IT = (OUT_N + T::NT - 1) / T::NT;  // ... per thread
    // A thread's vectors are PIX_STEP pixels = ROW_STEP whole tile rows apart: same column and channel group, so every byte
    // offset is the first one plus a wave-uniform row stride (one add per vector instead of a 64-bit multiply chain), and all
    // global accesses of the epilogue are buffer operations on the image's own descriptor: 32-bit offsets, out-of-image
    // vectors get the offset ~0 and are dropped (stores) or come back as zeros (loads) by the range check.
    constexpr int PIX_STEP = T::NT / VEC_PER_PIX;
    static_assert(T::NT % VEC_PER_PIX == 0 && PIX_STEP % T::TW == 0, "a thread's output vectors must share a column");
    constexpr int ROW_STEP = PIX_STEP / T::TW;
    const int o_cg = tid % VEC_PER_PIX, o_pix = tid / VEC_PER_PIX;
    const int o_ly = o_pix / T::TW, o_gx = x0 + o_pix % T::TW;
    const int img_bytes = H * W * a.Cout * 2;
    uint16_t *yo = reinterpret_cast<uint16_t *>(a.y);
    // Everything the epilogue reads from memory is requested HERE, in front of the barrier and the staging of the tile:
    // the dgrad operands (ReLU mask of the producing layer, the other consumer's gradient: all of a thread's vectors at
    // once) and the bias.  Loaded where they are used, each was an exposed round trip per tile (the bias: one per 16 channels).
    int voff[OUT_IT];
#pragma unroll
    for (int it = 0; it < OUT_IT; ++it) {
        const int gy = y0 + o_ly + it * ROW_STEP;
        const bool ok = it * T::NT + tid < OUT_N && gy < H && o_gx < W;
        voff[it] = ok ? ((gy * W + o_gx) * a.Cout + n0 + o_cg * 8) * 2 : -1;
    }
    const bool masked = (a.relu_src || a.relu_bits || a.addend) && !(a.flags & kSubsample2);
    uint4 mk[OUT_IT], ad[OUT_IT];
#pragma unroll
    for (int it = 0; it < OUT_IT; ++it) mk[it] = ad[it] = make_uint4(0, 0, 0, 0);
    if constexpr (UNPOOL) {
        // own pixel of the pooled map's input (it is also the ReLU mask) and the pooled gradient of its window
        const int OH = (H + 1) >> 1, OW = (W + 1) >> 1;
        auto m_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(a.relu_src + (int64_t)n * H * W * a.Cout), 0,
                                                        img_bytes, 0x00020000);
        auto g_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(a.unpool_g + (int64_t)n * OH * OW * a.Cout), 0,
                                                        OH * OW * a.Cout * 2, 0x00020000);
#pragma unroll
        for (int it = 0; it < OUT_IT; ++it)
            mk[it] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(m_rsrc, voff[it], 0, 0));
#pragma unroll
        for (int it = 0; it < OUT_IT; ++it) {
            const int gy = y0 + o_ly + it * ROW_STEP;
            const int goff = voff[it] < 0 ? -1 : (((gy >> 1) * OW + (o_gx >> 1)) * a.Cout + n0 + o_cg * 8) * 2;
            ad[it] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(g_rsrc, goff, 0, 0));
        }
    } else if (masked && a.relu_bits) {
        // the ReLU mask as bits: one byte per 16-byte vector of the bf16 image, expanded to the 0 / 1 halfwords that
        // keep_where_pos_bf16x8 tests (an HBM-bound data gradient - conv1_2's - reads 8 instead of 128 bytes per pixel)
        auto b_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(a.relu_bits + (int64_t)n * H * W * (a.Cout >> 3)), 0,
                                                        img_bytes >> 4, 0x00020000);
#pragma unroll
        for (int it = 0; it < OUT_IT; ++it) {
            const unsigned b = __builtin_amdgcn_raw_buffer_load_b8(b_rsrc, voff[it] < 0 ? -1 : voff[it] >> 4, 0, 0);
            mk[it] = make_uint4((b & 1u) | ((b & 2u) << 15), ((b >> 2) & 1u) | ((b & 8u) << 13), ((b >> 4) & 1u) | ((b & 32u) << 11),
                                ((b >> 6) & 1u) | ((b & 128u) << 9));
        }
        if (a.addend) {
            auto ad_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(a.addend + (int64_t)n * H * W * a.Cout), 0,
                                                             img_bytes, 0x00020000);
#pragma unroll
            for (int it = 0; it < OUT_IT; ++it)
                ad[it] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(ad_rsrc, voff[it], 0, 0));
        }
    } else if (masked) {
        auto m_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint16_t *>(a.relu_src ? a.relu_src + (int64_t)n * H * W * a.Cout : yo), 0,
            a.relu_src ? img_bytes : 0, 0x00020000);
        auto ad_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint16_t *>(a.addend ? a.addend + (int64_t)n * H * W * a.Cout : yo), 0,
            a.addend ? img_bytes : 0, 0x00020000);
#pragma unroll
        for (int it = 0; it < OUT_IT; ++it)
            mk[it] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(m_rsrc, voff[it], 0, 0));
#pragma unroll
        for (int it = 0; it < OUT_IT; ++it)
            ad[it] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(ad_rsrc, voff[it], 0, 0));
    }
    float4 bias_r[T::NF];
#pragma unroll
    for (int j = 0; j < T::NF; ++j)
        bias_r[j] = a.bias ? *reinterpret_cast<const float4 *>(a.bias + n0 + wn * T::WAVE_N + j * 16 + q * 4)
                           : make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();  // every wave is done with sA/sB: reuse the memory as the bf16 output tile
    uint16_t *sO = reinterpret_cast<uint16_t *>(smem);
#pragma unroll
    for (int j = 0; j < T::NF; ++j) {
        const int col = wn * T::WAVE_N + j * 16 + q * 4;
        const float4 b = bias_r[j];
#pragma unroll
        for (int i = 0; i < T::MF; ++i) {
            const int pix = wm * T::WAVE_M + i * 16 + cl;
            float v0 = acc[i][j][0] + b.x, v1 = acc[i][j][1] + b.y, v2 = acc[i][j][2] + b.z, v3 = acc[i][j][3] + b.w;
            if (a.flags & FOSVOS_CONV_RELU) {
                v0 = relu_f(v0); v1 = relu_f(v1); v2 = relu_f(v2); v3 = relu_f(v3);
            }
            *reinterpret_cast<uint2 *>(sO + pix * T::OUT_LD + col) = make_uint2(pack2bf(v0, v1), pack2bf(v2, v3));
        }
    }
    if constexpr (UNPOOL) {  // the tile of relu_src next to the output tile (out-of-image pixels: zeros, by the range check)
        uint16_t *sX = sO + T::BM * T::OUT_LD;
#pragma unroll
        for (int it = 0; it < OUT_IT; ++it)
            if (it * T::NT + tid < OUT_N)
                *reinterpret_cast<uint4 *>(sX + (o_pix + it * PIX_STEP) * T::OUT_LD + o_cg * 8) = mk[it];
    }
    __syncthreads();
    FOSVOS_STAMP_AT(9)
    const uint16_t *s_vec = sO + o_pix * T::OUT_LD + o_cg * 8;  // this thread's first staged vector; the next is PIX_STEP rows of sO on
    auto y_rsrc = __builtin_amdgcn_make_buffer_rsrc(yo + (int64_t)n * H * W * a.Cout, 0, img_bytes, 0x00020000);
    if constexpr (UNPOOL) {
        // MaxPool2d(2, 2, ceil_mode=True) backward of this tile (its origin and sizes are even: no window straddles two tiles).
        // A pixel at position t = 2 (row & 1) + (column & 1) of its window takes the window's gradient where it is the FIRST
        // maximum in scan order and > 0 (k_pool_bwd's rule, its ReLU mask included): own > x for the earlier positions, own >=
        // x for the later ones.  The values are post-ReLU bf16 - they order like 16-bit integers below 0x8000 - so
        // "own > x" is the sign of x - own and "own >= x" the sign of x - own - 1, two elements per v_pk_sub_i16; missing
        // window pixels (ragged last row / column) are zeros in sX and never beat a positive own.
        static_assert((T::TW & (T::TW - 1)) == 0 && T::TH % 2 == 0, "window partners are pix ^ 1 and pix ^ TW");
        const uint16_t *sX = sO + T::BM * T::OUT_LD;
        const fosvos_i16x2 one2 = {1, 1}, zero2 = {0, 0};
        const fosvos_i16x2 late_x = (o_pix & 1) ? zero2 : one2;  // the column partner comes later in the scan
#pragma unroll
        for (int it = 0; it < OUT_IT; ++it) {
            const int pix = o_pix + it * PIX_STEP;
            if (it * T::NT + tid >= OUT_N) continue;
            const fosvos_i16x2 late_y = (pix & T::TW) ? zero2 : one2;  // the partners in the other row (the diagonal one too)
            const uint4 cv = keep_where_pos_bf16x8(*reinterpret_cast<const uint4 *>(s_vec + it * PIX_STEP * T::OUT_LD), mk[it]);
            const uint4 x1 = *reinterpret_cast<const uint4 *>(sX + (pix ^ 1) * T::OUT_LD + o_cg * 8);
            const uint4 x2 = *reinterpret_cast<const uint4 *>(sX + (pix ^ T::TW) * T::OUT_LD + o_cg * 8);
            const uint4 x3 = *reinterpret_cast<const uint4 *>(sX + (pix ^ T::TW ^ 1) * T::OUT_LD + o_cg * 8);
            auto win = [&](unsigned own_, unsigned p1, unsigned p2, unsigned p3) {
                const fosvos_i16x2 own = __builtin_bit_cast(fosvos_i16x2, own_);
                const fosvos_i16x2 s = (zero2 - own) & (__builtin_bit_cast(fosvos_i16x2, p1) - own - late_x) &
                                       (__builtin_bit_cast(fosvos_i16x2, p2) - own - late_y) &
                                       (__builtin_bit_cast(fosvos_i16x2, p3) - own - late_y);
                return __builtin_bit_cast(unsigned, s >> 15);  // 0xffff where every sign is set
            };
            const uint4 routed = make_uint4(ad[it].x & win(mk[it].x, x1.x, x2.x, x3.x), ad[it].y & win(mk[it].y, x1.y, x2.y, x3.y),
                                            ad[it].z & win(mk[it].z, x1.z, x2.z, x3.z), ad[it].w & win(mk[it].w, x1.w, x2.w, x3.w));
            float f[8], r[8];
            unpack8(cv, f);
            unpack8(routed, r);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] += r[e];
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, pack8(f)), y_rsrc, voff[it], 0, 0);
        }
    } else if (!(a.flags & kSubsample2)) {
        if (masked) {
#pragma unroll
            for (int it = 0; it < OUT_IT; ++it) {
                uint4 v = *reinterpret_cast<const uint4 *>(s_vec + it * PIX_STEP * T::OUT_LD);
                if (a.relu_src || a.relu_bits) v = keep_where_pos_bf16x8(v, mk[it]);  // the ReLU mask, on the packed pairs
                if (a.addend || (a.flags & kReluAfterAdd)) {
                    float f[8];
                    unpack8(v, f);
                    if (a.addend) {
                        float av[8];
                        unpack8(ad[it], av);
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] += av[e];
                    }
                    if (a.flags & kReluAfterAdd) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] = relu_f(f[e]);
                    }
                    v = pack8(f);
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), y_rsrc, voff[it], 0, 0);
            }
        } else {
#pragma unroll
            for (int it = 0; it < OUT_IT; ++it)
                __builtin_amdgcn_raw_buffer_store_b128(
                    __builtin_bit_cast(u32x4, *reinterpret_cast<const uint4 *>(s_vec + it * PIX_STEP * T::OUT_LD)), y_rsrc,
                    voff[it], 0, 0);
        }
    } else {
        // stride-2 form (ResNet path): only the even (row, column) pixels are kept, in a half-size map
#pragma unroll
        for (int it = 0; it < OUT_IT; ++it) {
            const int gy = y0 + o_ly + it * ROW_STEP;
            if (it * T::NT + tid >= OUT_N || gy >= H || o_gx >= W || ((gy | o_gx) & 1)) continue;
            const int64_t off =
                (((int64_t)n * ((H + 1) >> 1) + (gy >> 1)) * ((W + 1) >> 1) + (o_gx >> 1)) * a.Cout + n0 + o_cg * 8;
            *reinterpret_cast<uint4 *>(yo + off) = *reinterpret_cast<const uint4 *>(s_vec + it * PIX_STEP * T::OUT_LD);
        }
    }
    if (a.y_pool) {
        // fused MaxPool2d(2,2,ceil_mode=True) of this tile (tile origins and sizes are even, so no pooling window
        // straddles two tiles): the next stage's input is written from the staged tile instead of re-reading y
        static_assert(T::TH % 2 == 0 && T::TW % 2 == 0, "pooling windows must not straddle tiles");
        constexpr int PW = T::TW / 2, PN = (T::TH / 2) * PW;
        const int OH = (H + 1) >> 1, OW = (W + 1) >> 1;
        auto p_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.y_pool + (int64_t)n * OH * OW * a.Cout, 0, OH * OW * a.Cout * 2,
                                                        0x00020000);
        for (int idx = tid; idx < PN * VEC_PER_PIX; idx += T::NT) {
            const int pp = idx / VEC_PER_PIX, cg = idx % VEC_PER_PIX;
            const int py = pp / PW, px = pp % PW;
            const int oy = (y0 >> 1) + py, ox = (x0 >> 1) + px;
            if (oy >= OH || ox >= OW) continue;
            const int poff = ((oy * OW + ox) * a.Cout + n0 + cg * 8) * 2;
            if (a.flags & FOSVOS_CONV_RELU) {  // the staged values are >= 0: the maximum is an integer max on the packed pairs
                uint4 mx = make_uint4(0, 0, 0, 0);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int ly = 2 * py + (t >> 1), lx = 2 * px + (t & 1);
                    if (y0 + ly < H && x0 + lx < W)
                        mx = max_nonneg_bf16x8(mx, *reinterpret_cast<const uint4 *>(sO + (ly * T::TW + lx) * T::OUT_LD + cg * 8));
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, mx), p_rsrc, poff, 0, 0);
                continue;
            }
            float m[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) m[j] = -INFINITY;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int ly = 2 * py + (t >> 1), lx = 2 * px + (t & 1);
                if (y0 + ly < H && x0 + lx < W) {
                    float f[8];
                    unpack8(*reinterpret_cast<const uint4 *>(sO + (ly * T::TW + lx) * T::OUT_LD + cg * 8), f);
#pragma unroll
                    for (int j = 0; j < 8; ++j) m[j] = f[j] > m[j] ? f[j] : m[j];
                }
            }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, pack8(m)), p_rsrc, poff, 0, 0);
        }
    }
    FOSVOS_STAMP_AT(10)
    FOSVOS_STAMP_RT(12)
}

// Finish a split-K convolution: out = epilogue(sum_z partial[z] + bias).  One thread = 8 channels of a pixel.
__global__ __launch_bounds__(256) void k_splitk_epilogue(const ConvArgs a, int64_t total8) {
    const int groups = a.Cout >> 3;
    const int64_t plane = (int64_t)a.N * a.H * a.W * a.Cout;
    for (int64_t i = blockIdx.x * 256LL + threadIdx.x; i < total8; i += (int64_t)gridDim.x * 256) {
        const int g = (int)(i % groups);
        int64_t off = i * 8;  // where the sums are read; the result goes to i * 8
        if (a.flags & kSubsample2) {
            const int Ho = (a.H + 1) >> 1, Wo = (a.W + 1) >> 1;
            int64_t pq = i / groups;
            const int ox = (int)(pq % Wo);
            pq /= Wo;
            const int oy = (int)(pq % Ho), n = (int)(pq / Ho);
            off = (((int64_t)n * a.H + 2 * oy) * a.W + 2 * ox) * a.Cout + g * 8;
        }
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = 0.f;
        // the partial planes are requested four at a time before they are added (in increasing z): a rolled loop
        // waited out one memory round trip per plane
        for (int z0 = 0; z0 < a.k_splits; z0 += 4) {
            float4 v0[4], v1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int z = min(z0 + j, a.k_splits - 1);
                v0[j] = *reinterpret_cast<const float4 *>(a.partial + z * plane + off);
                v1[j] = *reinterpret_cast<const float4 *>(a.partial + z * plane + off + 4);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (z0 + j < a.k_splits) {
                    f[0] += v0[j].x; f[1] += v0[j].y; f[2] += v0[j].z; f[3] += v0[j].w;
                    f[4] += v1[j].x; f[5] += v1[j].y; f[6] += v1[j].z; f[7] += v1[j].w;
                }
            }
        }
        if (a.bias) {
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] += a.bias[g * 8 + e];
        }
        if (a.flags & FOSVOS_CONV_RELU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = relu_f(f[e]);
        }
        if (a.flags & FOSVOS_CONV_OUT_F32) {
            float *yo = reinterpret_cast<float *>(a.y) + off;
            *reinterpret_cast<float4 *>(yo) = make_float4(f[0], f[1], f[2], f[3]);
            *reinterpret_cast<float4 *>(yo + 4) = make_float4(f[4], f[5], f[6], f[7]);
            continue;
        }
        if (a.relu_src || a.relu_bits || a.addend) {
            // same rounding sequence as the fused epilogue: round, mask, add, round
            uint4 v = pack8(f);
            unpack8(v, f);
            if (a.relu_bits) {
                const unsigned b = a.relu_bits[off >> 3];
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = ((b >> e) & 1u) ? f[e] : 0.f;
            }
            if (a.relu_src) {
                float m[8];
                unpack8(*reinterpret_cast<const uint4 *>(a.relu_src + off), m);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = m[e] > 0.f ? f[e] : 0.f;
            }
            if (a.addend) {
                float ad[8];
                unpack8(*reinterpret_cast<const uint4 *>(a.addend + off), ad);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] += ad[e];
            }
            if (a.flags & kReluAfterAdd) {
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = relu_f(f[e]);
            }
        }
        *reinterpret_cast<uint4 *>(reinterpret_cast<uint16_t *>(a.y) + i * 8) = pack8(f);
    }
}

// tile configurations
// (Tile<16, 32, 64, 8, 1>, 512 px x 64 ch on eight waves, loads a chunk's weight tile once per 512 pixels: -35 % staging bytes,
// and 5-15 % SLOWER on every layer at three frames per launch (895 vs 966 frames/s on the step): the template takes it as is)
using TileBig = Tile<8, 32, 64, 4, 1>;    // 256 px x 64 ch
using TileSquare = Tile<16, 16, 64, 4, 1>; // 256 px x 64 ch as 16 x 16: the same kernel where 8 x 32 tiles overhang the map more
using TileMid = Tile<8, 16, 64, 2, 2>;    // 128 px x 64 ch
using TileSmall = Tile<4, 16, 64, 2, 2>;  //  64 px x 64 ch
using TileSide = Tile<8, 32, 16, 4, 1>;   // 256 px x 16 ch: side_prep at large maps
using TileSideS = Tile<4, 16, 16, 4, 1>;  //  64 px x 16 ch: side_prep at small maps
using TileHalf = Tile<8, 32, 32, 4, 1>;   // 256 px x 32 ch: the 32-channel layers of the thinned ResNets
using TileHalfS = Tile<4, 16, 32, 4, 1>;  //  64 px x 32 ch: ... at small maps

enum TileId { kBig, kMid, kSmall, kSide, kSideS, kHalf, kHalfS, kSquare };

struct ConvPlan {
    TileId tile;
    int k_splits, chunks_per_split;
    size_t workspace_bytes;
};

// Pick the tile and the K split: prefer the largest pixel tile that still yields >= kMinBlocks workgroups
// (256 CUs x 2 resident workgroups); when even the smallest tile cannot, split the K chunks over blockIdx.z
// so that weights are streamed once per pixel tile and the chip is filled.
constexpr int kMinBlocks = 512;

ConvPlan make_plan(int N, int H, int W, int in_ch, int out_ch) {
    ConvPlan p{};
    const int n_chunks = roundup(in_ch, 32) / 32;
    const int64_t pixels = (int64_t)N * H * W;
    auto blocks = [&](int th, int tw, int bn) { return cdiv(W, tw) * cdiv(H, th) * N * (int64_t)(out_ch / bn); };
    int64_t nb;
    if (out_ch % 64 == 0) {
        // 256-pixel tiles as soon as there is one per CU (measured: 420 workgroups of 8x32 beat 840 of 8x16 by 8 % at
        // 120x214x256 alone; at 60x107x512, 256 of them beat 448 of 8x16 by 1 % of the whole step beside the wgrad
        // stream, which fills the second slot of each CU); below that the 128-pixel tile, 3 per CU
        bool big = blocks(8, 32, 64) >= kMinBlocks / 2;
        if (big) {
            // ... as 8 x 32 or as 16 x 16 pixels, whichever overhangs the map less (a tile computes all of its 256 pixels:
            // 60 x 107 is 64 x 128 = 8192 pixels of work in 8 x 32 tiles, 64 x 112 = 7168 in 16 x 16 ones)
            const int64_t area_wide = cdiv(H, 8) * 8 * cdiv(W, 32) * 32, area_sq = cdiv(H, 16) * 16 * cdiv(W, 16) * 16;
            static const bool allow_sq = lab_env_int("FOSVOS_NO_SQUARE_TILE", 0) == 0;
            if (allow_sq && area_sq < area_wide) { p.tile = kSquare; nb = blocks(16, 16, 64); }
            else { p.tile = kBig; nb = blocks(8, 32, 64); }
            // ... unless the 128-pixel tiles fit in ONE round (3 per CU = 768) and the 256-pixel tiles chosen are not even one
            // per CU (one frame's 60x107x512 layers: 224 tiles of 16 x 16, which would go on to split K).  Until round 4 the
            // condition was "fewer than 384 of the 8 x 32 tiles" (lab switch FOSVOS_MID_ONE_ROUND=1), which also took the
            // five-frame 30x54x512 data gradients - 320 big workgroups, 62 % of the 512 slots, 16 chunks each, or 640 small
            // ones: alone the small ones win, 53 against 60 us per launch; beside the weight-gradient stream, where these
            // launches run, they lose: the step +0.5 % with the big tile (profiles/r04_lab_step_ab_tunables.txt).
            static const bool one_round = lab_env_int("FOSVOS_MID_ONE_ROUND", 0) != 0;
            if (blocks(8, 16, 64) <= 768 && (one_round ? blocks(8, 32, 64) < kMinBlocks * 3 / 4 : nb < kMinBlocks / 2)) big = false;
        }
        if (!big) { p.tile = kMid; nb = blocks(8, 16, 64); }
    } else if (out_ch == 32) {
        if (pixels >= 256 * 256) { p.tile = kHalf; nb = blocks(8, 32, 32); }
        else { p.tile = kHalfS; nb = blocks(4, 16, 32); }
    } else {
        if (pixels >= 256 * 256) { p.tile = kSide; nb = blocks(8, 32, 16); }
        else { p.tile = kSideS; nb = blocks(4, 16, 16); }
    }
    if (out_ch % 64 == 0) {
        // lab switch: 0 = 8x32, 1 = 8x16, 2 = 4x16 pixel tiles of 64 channels; 5 = 8x32 pixels x 32 channels
        static const char *force = lab_env("FOSVOS_FORCE_TILE");
        if (force && ((atoi(force) >= 0 && atoi(force) <= 2) || atoi(force) == (int)kHalf)) {
            p.tile = (TileId)atoi(force);
            nb = p.tile == kBig ? blocks(8, 32, 64) : p.tile == kMid ? blocks(8, 16, 64)
                 : p.tile == kHalf ? blocks(8, 32, 32) : blocks(4, 16, 64);
        }
    }
    int ks = 1;
    if (nb < kMinBlocks * 3 / 4) {
        ks = (int)cdiv(kMinBlocks / 2, nb);  // one workgroup per CU is enough (measured at 30x54x512: 2 splits beat 4)
        if (ks < 2) ks = 1;
        if (ks > n_chunks) ks = n_chunks;
        if (ks > 16) ks = 16;
    }
#ifdef FOSVOS_STAMP
    if (const char *e = lab_env("FOSVOS_FORCE_KS")) ks = atoi(e);
#endif
    p.chunks_per_split = (int)cdiv(n_chunks, ks);
    p.k_splits = (int)cdiv(n_chunks, p.chunks_per_split);
    p.workspace_bytes = p.k_splits > 1 ? (size_t)p.k_splits * pixels * out_ch * sizeof(float) : 0;
    return p;
}

template <class T, bool OUT_F32, bool UNPOOL = false>
int launch(const ConvArgs &a0, const ConvPlan &plan, hipStream_t st, int in_ch) {
    // (UNPOOL: the tile of the pooled map's input sits behind the output tile)
    constexpr int LDS = UNPOOL && 2 * T::LDS_OUT > T::LDS_MAIN ? 2 * T::LDS_OUT : T::LDS_BYTES;
    ConvArgs a = a0;
    a.tiles_x = (int)cdiv(a.W, T::TW);
    a.tiles_y = (int)cdiv(a.H, T::TH);
    a.k_splits = plan.k_splits;
    a.chunks_per_split = plan.chunks_per_split;
#ifdef FOSVOS_STAMP
    a.stamps = g_stamps;
#endif
    const int64_t tiles = (int64_t)a.tiles_x * a.tiles_y * a.N;
    FOSVOS_REQUIRE(tiles <= 0x7fffffff && a.Cout % T::BN == 0, FOSVOS_E_SHAPE, "conv3x3: tile grid");
    if constexpr (LDS > 64 * 1024) {  // opt in to more than 64 KB of dynamic LDS, once per device
        static bool once[64];
        int dev = 0;
        FOSVOS_HIP_CHECK(hipGetDevice(&dev));
        if (dev >= 0 && dev < 64 && !once[dev]) {
            FOSVOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_conv3x3_igemm<T, OUT_F32, UNPOOL>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
            once[dev] = true;
        }
    }
    if (g_prof_on) {  // the name rocprofv3 prints for this instantiation
        static char name[80];
        snprintf(name, sizeof(name), "k_conv3x3_igemm<Tile<%d, %d, %d, %d, %d>, %s, %s>", T::TH, T::TW, T::BN, T::WM, T::WN,
                 OUT_F32 ? "true" : "false", UNPOOL ? "true" : "false");
        prof_begin(name, st, 2.0 * a.N * a.H * a.W * 9.0 * in_ch * a.Cout);
    }
    {
        static const int swz_env = lab_env_int("FOSVOS_IGEMM_SWIZZLE", -1);  // lab switch
        const int n_cb = a.Cout / T::BN;
        // Measured at five 480x854 frames per launch, every layer alone: mode 1 -2.6 % forward / -1.6 % data gradient over the
        // plain mapping (conv3 -3..-4.5 %, conv2_1 forward -6 %), also for the 512-channel layers whose 4.7 MB of weights do
        // not fit an L2 beside the input tiles; mode 2 -1 %.  On the step +0.6 %.
        a.swizzle = swz_env >= 0 ? swz_env : (tiles * n_cb >= 64 ? 1 : 0);
        if (tiles * n_cb > 0x7fffffff) a.swizzle = 0;
    }
    {
        const int64_t n_cb = a.Cout / T::BN, n_wg = tiles * n_cb;
        a.n_t = (int)tiles;
        a.m_ncb = recip_mul(n_wg, n_cb);
        a.m_nt = recip_mul(n_wg, tiles);
        a.m_tx = recip_mul(tiles, a.tiles_x);
        a.m_ty = recip_mul(tiles, a.tiles_y);
    }
    const dim3 grid = a.swizzle ? dim3((unsigned)(tiles * (a.Cout / T::BN)), 1u, (unsigned)plan.k_splits)
                                : dim3((unsigned)tiles, (unsigned)(a.Cout / T::BN), (unsigned)plan.k_splits);
    hipLaunchKernelGGL((k_conv3x3_igemm<T, OUT_F32, UNPOOL>), grid, dim3(T::NT), LDS, st, a);
    FOSVOS_LAUNCH_CHECK();
    if (plan.k_splits > 1) {
        const int64_t total8 = (a.flags & kSubsample2) ? (int64_t)a.N * ((a.H + 1) >> 1) * ((a.W + 1) >> 1) * (a.Cout / 8)
                                                       : (int64_t)a.N * a.H * a.W * (a.Cout / 8);
        int64_t g = cdiv(total8, 256);
        if (g > 4096) g = 4096;
        FOSVOS_PROF("k_splitk_epilogue", st, 0.0);
        hipLaunchKernelGGL(k_splitk_epilogue, dim3((unsigned)g), dim3(256), 0, st, a, total8);
        FOSVOS_LAUNCH_CHECK();
    }
    return FOSVOS_OK;
}

int dispatch(ConvArgs a, int in_ch, void *workspace, size_t workspace_bytes, hipStream_t st) {
    const bool f32 = (a.flags & FOSVOS_CONV_OUT_F32) != 0;
    if (a.Cout == 32) {
        FOSVOS_REQUIRE(!f32, FOSVOS_E_ARG, "conv3x3: fp32 output is only built for 16-channel outputs");
    } else if (a.Cout % 64 != 0) {
        FOSVOS_REQUIRE(a.Cout == 16, FOSVOS_E_SHAPE, "conv3x3: output channels must be 16, 32 or a multiple of 64 (got %d)",
                       a.Cout);
        FOSVOS_REQUIRE(a.relu_src == nullptr && a.addend == nullptr, FOSVOS_E_ARG,
                       "conv3x3: 16-channel output has no mask/add epilogue");
    } else {
        FOSVOS_REQUIRE(!f32, FOSVOS_E_ARG, "conv3x3: fp32 output is only built for 16-channel outputs");
    }
    ConvPlan plan = make_plan(a.N, a.H, a.W, in_ch, a.Cout);
    if (a.flags & kNoSplitK) {
        plan.k_splits = 1;
        plan.chunks_per_split = roundup(in_ch, 32) / 32;
        plan.workspace_bytes = 0;
    }
    if (plan.k_splits > 1) {
        FOSVOS_REQUIRE(workspace && workspace_bytes >= plan.workspace_bytes, FOSVOS_E_WORKSPACE,
                       "conv3x3: split-K needs %zu workspace bytes, got %zu", plan.workspace_bytes, workspace_bytes);
        a.partial = reinterpret_cast<float *>(workspace);
    }
    if (a.unpool_g) {  // (fosvos_conv3x3_dgrad_unpool has checked that the plan is one of these, unsplit)
        switch (plan.tile) {
            case kBig: return launch<TileBig, false, true>(a, plan, st, in_ch);
            case kSquare: return launch<TileSquare, false, true>(a, plan, st, in_ch);
            case kMid: return launch<TileMid, false, true>(a, plan, st, in_ch);
            default: return fail(FOSVOS_E_ARG, "conv3x3: no fused pool backward for this tile");
        }
    }
    switch (plan.tile) {
        case kBig: return launch<TileBig, false>(a, plan, st, in_ch);
        case kSquare: return launch<TileSquare, false>(a, plan, st, in_ch);
        case kMid: return launch<TileMid, false>(a, plan, st, in_ch);
        case kSmall: return launch<TileSmall, false>(a, plan, st, in_ch);
        case kSide: return f32 ? launch<TileSide, true>(a, plan, st, in_ch) : launch<TileSide, false>(a, plan, st, in_ch);
        case kSideS: return f32 ? launch<TileSideS, true>(a, plan, st, in_ch) : launch<TileSideS, false>(a, plan, st, in_ch);
        case kHalf: return launch<TileHalf, false>(a, plan, st, in_ch);
        case kHalfS: return launch<TileHalfS, false>(a, plan, st, in_ch);
    }
    return fail(FOSVOS_E_ARG, "conv3x3: bad plan");
}

// Forward launches with bf16 output and no addend take the persistent eight-wave kernel (conv_pp.hip) where its tiles fill
// the chip.  Lab builds: FOSVOS_PP=0 never, 1 whenever the shape fits.
bool forward_takes_pp(int N, int H, int W, int in_ch, int out_ch, unsigned flags) {
    if (flags & ~FOSVOS_CONV_RELU) return false;
    static const int mode = lab_env_int("FOSVOS_PP", -1);
    if (mode == 0) return false;
    return conv_pp_applicable(N, H, W, roundup(in_ch, 32), out_ch);
}

int check_common(const void *x, const void *w, const void *y, int N, int H, int W, int in_ch, int out_ch,
                 const char *who) {
    FOSVOS_REQUIRE(x && w && y, FOSVOS_E_ARG, "%s: null pointer", who);
    FOSVOS_REQUIRE(N > 0 && H > 0 && W > 0 && in_ch > 0 && out_ch > 0, FOSVOS_E_SHAPE, "%s: bad shape N=%d H=%d W=%d in=%d out=%d",
                   who, N, H, W, in_ch, out_ch);
    FOSVOS_REQUIRE(out_ch % 16 == 0, FOSVOS_E_SHAPE, "%s: output channels %d not a multiple of 16", who, out_ch);
    // (the kernels address one image with 32-bit BYTE offsets through a buffer descriptor)
    FOSVOS_REQUIRE((int64_t)H * W * roundup(in_ch, 32) * 2 < 0x7fffffffLL && (int64_t)H * W * out_ch * 4 < 0x7fffffffLL,
                   FOSVOS_E_SHAPE, "%s: one image exceeds 2^31 bytes", who);
    return FOSVOS_OK;
}
}  // namespace

extern "C" size_t fosvos_conv3x3_workspace_bytes(int N, int H, int W, int in_ch, int out_ch) {
    if (N <= 0 || H <= 0 || W <= 0 || in_ch <= 0 || out_ch <= 0 || (out_ch % 64 != 0 && out_ch != 16)) return 0;
    return make_plan(N, H, W, in_ch, out_ch).workspace_bytes;
}

extern "C" int fosvos_conv3x3_plan(int N, int H, int W, int in_ch, int out_ch, fosvos_conv3x3_plan_info *out) {
    FOSVOS_REQUIRE(out, FOSVOS_E_ARG, "conv3x3_plan: null output");
    FOSVOS_REQUIRE(N > 0 && H > 0 && W > 0 && in_ch > 0 && out_ch > 0 && (out_ch % 64 == 0 || out_ch == 32 || out_ch == 16),
                   FOSVOS_E_SHAPE, "conv3x3_plan: bad shape N=%d H=%d W=%d in=%d out=%d", N, H, W, in_ch, out_ch);
    const ConvPlan p = make_plan(N, H, W, in_ch, out_ch);
    int th = 0, tw = 0, bn = 0;
    switch (p.tile) {
        case kBig: th = TileBig::TH; tw = TileBig::TW; bn = TileBig::BN; break;
        case kSquare: th = TileSquare::TH; tw = TileSquare::TW; bn = TileSquare::BN; break;
        case kMid: th = TileMid::TH; tw = TileMid::TW; bn = TileMid::BN; break;
        case kSmall: th = TileSmall::TH; tw = TileSmall::TW; bn = TileSmall::BN; break;
        case kSide: th = TileSide::TH; tw = TileSide::TW; bn = TileSide::BN; break;
        case kSideS: th = TileSideS::TH; tw = TileSideS::TW; bn = TileSideS::BN; break;
        case kHalf: th = TileHalf::TH; tw = TileHalf::TW; bn = TileHalf::BN; break;
        case kHalfS: th = TileHalfS::TH; tw = TileHalfS::TW; bn = TileHalfS::BN; break;
    }
    out->tile_h = th; out->tile_w = tw; out->tile_co = bn;
    out->k_splits = p.k_splits;
    out->workgroups = (int)(cdiv(W, tw) * cdiv(H, th) * N * (out_ch / bn));
    out->persistent = 0;
    return FOSVOS_OK;
}

extern "C" int fosvos_conv3x3_fwd_plan(int N, int H, int W, int in_ch, int out_ch, unsigned flags,
                                       fosvos_conv3x3_plan_info *out) {
    if (int rc = fosvos_conv3x3_plan(N, H, W, in_ch, out_ch, out)) return rc;
    if (out_ch % 64 == 0 && forward_takes_pp(N, H, W, in_ch, out_ch, flags)) {
        out->tile_h = 8; out->tile_w = 32; out->tile_co = 64;
        out->k_splits = 1;
        out->workgroups = conv_pp_workgroups();
        out->persistent = 1;
    }
    return FOSVOS_OK;
}

extern "C" int fosvos_conv3x3_fwd(const uint16_t *x, const uint16_t *w_packed, const float *bias, void *y, int N, int H,
                                  int W, int Ci, int Co, unsigned flags, void *workspace, size_t workspace_bytes,
                                  int device, void *stream) {
    if (int rc = check_common(x, w_packed, y, N, H, W, Ci, Co, "conv3x3_fwd")) return rc;
    FOSVOS_REQUIRE((flags & ~(FOSVOS_CONV_RELU | FOSVOS_CONV_OUT_F32)) == 0, FOSVOS_E_ARG, "conv3x3_fwd: unknown flags 0x%x", flags);
    FOSVOS_ENTER(device);
    if (Co % 64 == 0 && forward_takes_pp(N, H, W, Ci, Co, flags))
        return conv_pp_forward(x, w_packed, bias, (uint16_t *)y, nullptr, N, H, W, roundup(Ci, 32), Co, roundup(Co, 16),
                               (flags & FOSVOS_CONV_RELU) != 0, (hipStream_t)stream);
    ConvArgs a{};
    a.x = x; a.w = w_packed; a.bias = bias; a.relu_src = nullptr; a.addend = nullptr; a.y = y;
    a.N = N; a.H = H; a.W = W; a.Cin = roundup(Ci, 32); a.Cout = Co; a.Co_pad = roundup(Co, 16); a.flags = flags;
    return dispatch(a, Ci, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int fosvos_conv3x3_fwd_add(const uint16_t *x, const uint16_t *w_packed, const float *bias,
                                      const uint16_t *addend, void *y, int N, int H, int W, int Ci, int Co,
                                      unsigned flags, void *workspace, size_t workspace_bytes, int device,
                                      void *stream) {
    if (int rc = check_common(x, w_packed, y, N, H, W, Ci, Co, "conv3x3_fwd_add")) return rc;
    FOSVOS_REQUIRE((flags & ~(FOSVOS_CONV_RELU | FOSVOS_CONV_OUT_F32)) == 0, FOSVOS_E_ARG, "conv3x3_fwd_add: unknown flags 0x%x",
                   flags);
    FOSVOS_REQUIRE(Co % 64 == 0 || Co == 32 || (Co == 16 && !addend), FOSVOS_E_SHAPE,
                   "conv3x3_fwd_add: Co=%d must be 32 or a multiple of 64 (16 without an addend)", Co);
    FOSVOS_REQUIRE(!(flags & FOSVOS_CONV_OUT_F32) || (Co == 16 && !addend), FOSVOS_E_ARG,
                   "conv3x3_fwd_add: fp32 output is the 16-channel side_prep form");
    FOSVOS_ENTER(device);
    ConvArgs a{};
    a.x = x; a.w = w_packed; a.bias = bias; a.relu_src = nullptr; a.addend = addend; a.y = y;
    a.N = N; a.H = H; a.W = W; a.Cin = roundup(Ci, 32); a.Cout = Co; a.Co_pad = roundup(Co, 16);
    a.flags = kNoSplitK | (addend ? ((flags & FOSVOS_CONV_RELU) ? kReluAfterAdd : 0u) : flags);
    return dispatch(a, Ci, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int fosvos_conv3x3_s2_fwd(const uint16_t *x, const uint16_t *w_packed, const float *bias, uint16_t *y, int N,
                                     int H, int W, int Ci, int Co, unsigned flags, void *workspace, size_t workspace_bytes,
                                     int device, void *stream) {
    if (int rc = check_common(x, w_packed, y, N, H, W, Ci, Co, "conv3x3_s2_fwd")) return rc;
    FOSVOS_REQUIRE((flags & ~FOSVOS_CONV_RELU) == 0, FOSVOS_E_ARG, "conv3x3_s2_fwd: unknown flags 0x%x", flags);
    FOSVOS_REQUIRE(Co % 64 == 0 || Co == 32, FOSVOS_E_SHAPE, "conv3x3_s2_fwd: Co=%d must be 32 or a multiple of 64", Co);
    FOSVOS_ENTER(device);
    ConvArgs a{};
    a.x = x; a.w = w_packed; a.bias = bias; a.relu_src = nullptr; a.addend = nullptr; a.y = y;
    a.N = N; a.H = H; a.W = W; a.Cin = roundup(Ci, 32); a.Cout = Co; a.Co_pad = roundup(Co, 16);
    a.flags = flags | kSubsample2 | kNoSplitK;
    return dispatch(a, Ci, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int fosvos_conv3x3_fwd_pool(const uint16_t *x, const uint16_t *w_packed, const float *bias, uint16_t *y,
                                       uint16_t *y_pool, int N, int H, int W, int Ci, int Co, unsigned flags,
                                       void *workspace, size_t workspace_bytes, int device, void *stream) {
    if (int rc = check_common(x, w_packed, y, N, H, W, Ci, Co, "conv3x3_fwd_pool")) return rc;
    FOSVOS_REQUIRE(y_pool, FOSVOS_E_ARG, "conv3x3_fwd_pool: null pooled output");
    FOSVOS_REQUIRE((flags & ~FOSVOS_CONV_RELU) == 0, FOSVOS_E_ARG, "conv3x3_fwd_pool: unknown flags 0x%x", flags);
    FOSVOS_REQUIRE(Co % 64 == 0, FOSVOS_E_SHAPE, "conv3x3_fwd_pool: Co=%d must be a multiple of 64", Co);
    if ((flags & FOSVOS_CONV_RELU) && forward_takes_pp(N, H, W, Ci, Co, flags)) {  // (the fused pool takes post-ReLU values)
        FOSVOS_ENTER(device);
        return conv_pp_forward(x, w_packed, bias, y, y_pool, N, H, W, roundup(Ci, 32), Co, roundup(Co, 16), true,
                               (hipStream_t)stream);
    }
    if (make_plan(N, H, W, Ci, Co).k_splits > 1) {  // small maps (split-K): the epilogue kernel has no tile to pool
        if (int rc = fosvos_conv3x3_fwd(x, w_packed, bias, y, N, H, W, Ci, Co, flags, workspace, workspace_bytes, device, stream))
            return rc;
        return fosvos_maxpool2x2_ceil_fwd(y, y_pool, N, H, W, Co, device, stream);
    }
    FOSVOS_ENTER(device);
    ConvArgs a{};
    a.x = x; a.w = w_packed; a.bias = bias; a.relu_src = nullptr; a.addend = nullptr; a.y = y; a.y_pool = y_pool;
    a.N = N; a.H = H; a.W = W; a.Cin = roundup(Ci, 32); a.Cout = Co; a.Co_pad = roundup(Co, 16); a.flags = flags;
    return dispatch(a, Ci, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int fosvos_conv3x3_dgrad_bits(const uint16_t *dy, const uint16_t *w_dgrad_packed, const uint8_t *relu_bits,
                                         const uint16_t *addend, uint16_t *dx, int N, int H, int W, int Ci, int Co,
                                         void *workspace, size_t workspace_bytes, int device, void *stream) {
    if (int rc = check_common(dy, w_dgrad_packed, dx, N, H, W, Co, Ci, "conv3x3_dgrad_bits")) return rc;
    FOSVOS_REQUIRE(relu_bits && Ci % 8 == 0, FOSVOS_E_ARG, "conv3x3_dgrad_bits: null mask or Ci=%d not a multiple of 8", Ci);
    FOSVOS_ENTER(device);
    // the persistent kernel (conv_pp.hip) where the forward pass takes it too - as a data gradient it is the same conv on the
    // rotated filter image, its epilogue the bit mask (lab switch FOSVOS_PP_DGRAD=0: the igemm)
    if (!addend && lab_env_int("FOSVOS_PP_DGRAD", 1) != 0 && lab_env_int("FOSVOS_PP", -1) != 0 &&
        conv_pp_applicable(N, H, W, roundup(Co, 32), Ci))
        return conv_pp_forward(dy, w_dgrad_packed, nullptr, dx, nullptr, N, H, W, roundup(Co, 32), Ci, roundup(Ci, 16), false,
                               (hipStream_t)stream, relu_bits);
    ConvArgs a{};
    a.x = dy; a.w = w_dgrad_packed; a.bias = nullptr; a.relu_src = nullptr; a.relu_bits = relu_bits; a.addend = addend; a.y = dx;
    a.N = N; a.H = H; a.W = W; a.Cin = roundup(Co, 32); a.Cout = Ci; a.Co_pad = roundup(Ci, 16); a.flags = 0;
    return dispatch(a, Co, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int fosvos_conv3x3_dgrad_unpool(const uint16_t *dy, const uint16_t *w_dgrad_packed, const uint16_t *x,
                                           const uint16_t *d_pooled, uint16_t *dx, int N, int H, int W, int Ci, int Co,
                                           void *workspace, size_t workspace_bytes, int device, void *stream) {
    if (int rc = check_common(dy, w_dgrad_packed, dx, N, H, W, Co, Ci, "conv3x3_dgrad_unpool")) return rc;
    FOSVOS_REQUIRE(x && d_pooled && x != dx, FOSVOS_E_ARG, "conv3x3_dgrad_unpool: null or aliased pooled-map operands");
    FOSVOS_ENTER(device);
    const ConvPlan plan = make_plan(N, H, W, Co, Ci);
    const bool fused = Ci % 64 == 0 && plan.k_splits == 1 && (plan.tile == kBig || plan.tile == kSquare || plan.tile == kMid) &&
                       lab_env_int("FOSVOS_UNPOOL_FUSED", 1) != 0;  // lab switch: 0 = the two passes below
    if (!fused) {  // the same sum as two passes: pool backward into dx, then the data gradient with dx as its addend
        if (int rc = fosvos_maxpool2x2_ceil_bwd(x, d_pooled, dx, N, H, W, Ci, 1, device, stream)) return rc;
        return fosvos_conv3x3_dgrad(dy, w_dgrad_packed, x, dx, dx, N, H, W, Ci, Co, workspace, workspace_bytes, device, stream);
    }
    ConvArgs a{};
    a.x = dy; a.w = w_dgrad_packed; a.relu_src = x; a.unpool_g = d_pooled; a.y = dx;
    a.N = N; a.H = H; a.W = W; a.Cin = roundup(Co, 32); a.Cout = Ci; a.Co_pad = roundup(Ci, 16); a.flags = 0;
    return dispatch(a, Co, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int fosvos_conv3x3_dgrad(const uint16_t *dy, const uint16_t *w_dgrad_packed, const uint16_t *relu_src,
                                    const uint16_t *addend, uint16_t *dx, int N, int H, int W, int Ci, int Co,
                                    void *workspace, size_t workspace_bytes, int device, void *stream) {
    // contraction over the forward op's output channels (Co), result has its input channels (Ci)
    if (int rc = check_common(dy, w_dgrad_packed, dx, N, H, W, Co, Ci, "conv3x3_dgrad")) return rc;
    FOSVOS_ENTER(device);
    ConvArgs a{};
    a.x = dy; a.w = w_dgrad_packed; a.bias = nullptr; a.relu_src = relu_src; a.addend = addend; a.y = dx;
    a.N = N; a.H = H; a.W = W; a.Cin = roundup(Co, 32); a.Cout = Ci; a.Co_pad = roundup(Ci, 16); a.flags = 0;
    return dispatch(a, Co, workspace, workspace_bytes, (hipStream_t)stream);
}
