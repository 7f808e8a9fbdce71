// Forward 3x3 / pad 1 / stride 1 convolution (+ bias + ReLU [+ 2x2 ceil-mode max pool]) on bf16 NHWC: the PERSISTENT,
// eight-wave "ping-pong" form of the implicit GEMM (src/networks/osvos_vgg.py:90-93 - Conv2d(3x3) + ReLU, and the MaxPool2d
// that follows the last conv of a stage).
//
// Why a second kernel beside k_conv3x3_igemm (conv_igemm.hip): that kernel's workgroup lives 19 k clocks on a 64-channel
// layer for 4.6 k clocks of matrix work (launch -> offsets -> first tile in LDS -> epilogue round trips), so two of them per
// CU keep the matrix pipe 48 % busy (70 % at 256 channels).  Here ONE workgroup owns the CU for the whole launch:
//   * 8 waves = two GROUPS of four (waves 0-3 / 4-7; wave w and w + 4 share a SIMD).  Each group owns a pixel tile
//     (8 x 32 pixels x 64 output channels, 64 accumulator registers per lane) of its own.
//   * time is cut into STEPS of one K chunk (32 input channels x 9 taps = 144 MFMAs per wave, ~2.3 k clocks): in step s
//     group s & 1 runs the MFMAs of its chunk, the other group is in its MEMORY phase - it requests the operands of later
//     chunks, and when its tile is finished it runs the epilogue (bias, ReLU, bf16, pool, global stores) straight from its
//     accumulators.  One s_barrier per step.  The matrix pipe of every SIMD always has exactly one wave that wants it, and
//     the wave beside it is doing memory work: the complementary pairing (matrix beside memory).
//   * every operand byte goes global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds): no staging registers, no ds_write.
//     Pixel tiles: [halo pixel][4 x 16 B] per chunk, the 16-byte k-group kq of pixel P at slot kq ^ ((P >> 1) & 2) - the
//     swizzle sits on the SOURCE address, the LDS image is lane-linear - which makes every ds_read_b128 fragment read
//     conflict-free for every tap shift (the read's 16-lane service groups mix two k-groups: lanes 0-3, 12-15, 20-27).
//     Weights: the packed image's [tap][kq][64 channels][16 B] chunk, one copy shared by BOTH groups (group 1 runs one step
//     behind group 0 on the same chunk index) and double-buffered; with <= 64 input channels the two chunks are loaded once
//     and stay (weights stationary).
//   * the workgroups are persistent: 256 of them (one per CU), each walks its share of the tiles; the ids that share an XCD
//     take one contiguous eighth of the tile space, neighbouring tiles at the same time (shared halo rows meet in that L2).
//   * channel order inside a 64-channel block is permuted in the LDS weight image so that a lane's accumulators hold 8
//     CONSECUTIVE channels of a pixel: the epilogue stores 16 bytes per lane and instruction without an LDS round trip.
// LDS: 2 x 36,864 (weights) + 2 groups x 2 x 21,760 (pixel tiles) + 256 (bias) = 161,024 of the CU's 163,840 bytes.
#include <stdlib.h>

#include "common.hpp"

using namespace fosvos;

namespace fosvos {
namespace pp {

constexpr int TH = 8, TW = 32, BN = 64;
constexpr int HALO_W = TW + 2, HALO_H = TH + 2, NPIX = HALO_W * HALO_H;  // 34 x 10 = 340 halo pixels
constexpr int A_BYTES = NPIX * 64;                                       // one 32-channel chunk of a halo tile
constexpr int A_PIECES = (NPIX * 4 + 63) / 64;                           // 1-KB DMA pieces of it: 22 (the last: 16 lanes)
constexpr int W_BYTES = 36 * BN * 16;                                    // one chunk of the weight tile: 36 rows of 1 KB
constexpr int LDS_W = 0, LDS_A = 2 * W_BYTES, LDS_BIAS = LDS_A + 4 * A_BYTES, LDS_BYTES = LDS_BIAS + BN * 4;
constexpr int NT = 512;
static_assert(LDS_BYTES <= 160 * 1024, "one workgroup per CU");

struct Args {
    const uint16_t *x;   // [N,H,W,Cin] bf16, Cin % 32 == 0
    const uint16_t *w;   // packed [Cin/32][9][4][Co_pad][8]
    const float *bias;   // [Cout] or null
    uint16_t *y;         // [N,H,W,Cout] bf16
    uint16_t *y_pool;    // optional [N,ceil(H/2),ceil(W/2),Cout]
    int N, H, W, Cin, Cout, Co_pad;
    int tiles_x, tiles_y, n_tiles;  // pixel tiles per row / per image / of the launch
    int n_cb;                       // 64-channel blocks
    unsigned long long *stamps;  // lab builds only
    int lab;  // timing-only switches of lab builds (FOSVOS_PP_LAB; wrong results): 1 no MFMAs, 2 no epilogue, 4 no pixel DMA,
              // 8 no weight DMA; always 0 in the shipped library
};
#ifdef FOSVOS_LAB_BUILD
#define PP_LAB(bit) (a.lab & (bit))
// phase stamps of lab builds (tools/pp_stamp_lab.py): lane 0 of the first wave of each group of the first 8 workgroups writes
// s_memtime at slot (workgroup, group, index); 512 slots per (workgroup, group)
#define PP_STAMP(idx_)                                                                                            \
    if (a.stamps && blockIdx.x < 8 && wm == 0 && lane == 0 && (idx_) < 512)                                       \
        a.stamps[(blockIdx.x * 2 + g) * 512 + (idx_)] = __builtin_amdgcn_s_memtime();
#else
#define PP_LAB(bit) 0
#define PP_STAMP(idx_)
#endif

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
// One LDS-DMA piece: lane i's 16 bytes land at lds_dst + 16 i (M0 carries the wave-uniform destination; written in the same
// statement that uses it).  s_nop 4: the descriptor / offset SGPRs may come straight from a v_readfirstlane.
__device__ __forceinline__ void dma16(const __amdgpu_buffer_rsrc_t rsrc, unsigned lds_dst, unsigned voff, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds_dst), "v"(voff), "s"(rsrc), "s"(soff) : "memory", "m0");
}
__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void wait_all() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }
// the step barrier: a bare s_barrier (no vmcnt / lgkmcnt drain: requests stay in flight across it), fenced against the
// compiler moving LDS accesses over it
__device__ __forceinline__ void pp_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
#pragma clang diagnostic pop

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) const uint4 *lds_u4_ptr;

// A group's walk over its tiles: tile index idx = first + i * stride, kept as (image, tile row, tile column) and advanced by
// the precomputed decomposition of the stride: wave-uniform integer adds and compares, no division after the first tile.
struct TileIt {
    int idx, n, ty, tx;       // linear index in the launch; image; tile row; tile column
    int d_n, d_ty, d_tx, stride;
    int tiles_x, tiles_y, end;
    __device__ __forceinline__ void init(int first, int stride_, int tiles_x_, int tiles_y_, int end_) {
        tiles_x = tiles_x_; tiles_y = tiles_y_; end = end_; stride = stride_;
        idx = first;
        const int per_img = tiles_x * tiles_y;
        n = first / per_img;
        const int r = first - n * per_img;
        ty = r / tiles_x;
        tx = r - ty * tiles_x;
        d_n = stride / per_img;
        const int rs = stride - d_n * per_img;
        d_ty = rs / tiles_x;
        d_tx = rs - d_ty * tiles_x;
    }
    __device__ __forceinline__ void next() {
        idx += stride;
        tx += d_tx;
        const int cx = tx >= tiles_x;
        tx -= cx ? tiles_x : 0;
        ty += d_ty + cx;
        const int cy = ty >= tiles_y;
        ty -= cy ? tiles_y : 0;
        n += d_n + cy;
    }
    __device__ __forceinline__ bool valid() const { return idx < end; }
    __device__ __forceinline__ int y0() const { return ty * TH; }
    __device__ __forceinline__ int x0() const { return tx * TW; }
};

template <bool RELU, bool POOL>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_conv3x3_pp(const Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem_pp[];
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void *)smem_pp);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wave >> 2, wm = wave & 3;  // group, wave inside the group
    const int cl = lane & 15, kq = lane >> 4;
    const int H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout;
    const int NC = Cin >> 5;

    // ---- workgroup -> (channel block, tile slots).  b & 7 labels the workgroups that share an XCD: they take one eighth of the
    // tiles, and inside it slot 2 jj + g of every round of 2 S consecutive tiles (S = workgroups per XCD and channel block).
    const int b = blockIdx.x, n_wg8 = gridDim.x >> 3;
    const int xcd = b & 7, j8 = b >> 3;
    const int cb = j8 % a.n_cb, jj = j8 / a.n_cb, S = n_wg8 / a.n_cb;
    const int t_begin = (int)(((int64_t)xcd * a.n_tiles) >> 3), t_end = (int)(((int64_t)(xcd + 1) * a.n_tiles) >> 3);
    const int span = t_end - t_begin;
    const int n_it = span > 2 * jj ? (span - 2 * jj + 2 * S - 1) / (2 * S) : 0;  // tiles of group 0 (group 1: the same or one less)
    if (n_it == 0) return;
    const int K = n_it * NC;  // super-steps: every group runs K chunks
    const int n0 = cb * BN;
    // The vector ALU of a SIMD is shared by its two waves (an MFMA holds its issue for 8 of 16 clocks): what the memory phases
    // issue on it comes straight out of the matrix wave's time.  Everything wave-uniform below (tile walk, descriptors, DMA
    // offsets of interior tiles) is therefore kept on the scalar unit.
    TileIt cur, ldt;  // the tile being accumulated / finished; the tile whose chunks are being requested
    cur.init(t_begin + 2 * jj + g, 2 * S, a.tiles_x, a.tiles_y, t_end);
    ldt = cur;

    // ---- weight pieces (group 1 requests them; the prologue: all waves): row r = tap * 4 + kq of a chunk is 64 channels x
    // 16 B; LDS slot p = 16 j + m of the row holds channel perm(p) = 32 (j >> 1) + 8 (m >> 2) + 4 (j & 1) + (m & 3), so that
    // MFMA block j, output row m = 4 q + r of lane q is channel 32 (j >> 1) + 8 q + 4 (j & 1) + r: 8 consecutive channels
    // per lane over the block pair (2 jp, 2 jp + 1).
    const int pj = lane >> 4, pm = lane & 15;
    const unsigned w_voff = (unsigned)(32 * (pj >> 1) + 8 * (pm >> 2) + 4 * (pj & 1) + (pm & 3)) * 16u;
    const int w_row_bytes = a.Co_pad * 16;
    const auto w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(a.w), 0, NC * 36 * w_row_bytes, 0x00020000);
    auto issue_w = [&](int chunk, int buf, int first, int stride) {
        if (PP_LAB(8)) return;
        for (int r = first; r < 36; r += stride)
            dma16(w_rsrc, lds0 + LDS_W + buf * W_BYTES + r * 1024, w_voff, (unsigned)((chunk * 36 + r) * w_row_bytes + n0 * 16));
    };

    // ---- pixel-tile pieces of this wave: piece pc = wm + 4 i covers halo pixels 16 pc .. 16 pc + 15, lane -> (pixel, slot).
    // a_rel: the lane's byte offset from the halo's first pixel (row y0 - 1, column x0 - 1), its k-group swizzled.
    constexpr int A_IT = (A_PIECES + 3) / 4;  // 6
    int a_rel[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int pc = wm + 4 * i;
        const int P = pc * 16 + (lane >> 2), s4 = lane & 3;
        const int hy = P / HALO_W, hx = P - hy * HALO_W;
        const int kq_src = s4 ^ ((P >> 1) & 2);
        a_rel[i] = (hy * W + hx) * Cin * 2 + kq_src * 16;
    }
    // the load stream of this group: super-step ld_k = (tile ldt, chunk ld_c), one chunk per call
    int ld_k = 0, ld_c = 0;
    int ld_base = 0, ld_edge = 0;
    __amdgpu_buffer_rsrc_t ld_rsrc;
    auto ld_open_tile = [&]() {
        if (!ldt.valid()) return;
        const int y0 = ldt.y0(), x0 = ldt.x0();
        ld_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(a.x) + (int64_t)ldt.n * H * W * Cin, 0,
                                                    H * W * Cin * 2, 0x00020000);
        ld_base = ((y0 - 1) * W + (x0 - 1)) * Cin * 2;  // the halo's first pixel (negative on the top / left edge)
        // an interior tile: every halo pixel is inside the image - the tile's offset rides in the scalar offset of the loads
        // (not range-checked, and nothing to check), the lanes keep their constant offsets
        ld_edge = !(y0 > 0 && x0 > 0 && y0 + TH < H && x0 + TW < W);
    };
    auto ld_issue = [&]() {  // request chunk ld_c of tile ldt into this group's buffer ld_k & 1, advance
        if (ld_k < K && ldt.valid() && !PP_LAB(4)) {
            const unsigned dst = lds0 + LDS_A + (g * 2 + (ld_k & 1)) * A_BYTES;
            constexpr int LAST_LANES = NPIX * 4 - (A_PIECES - 1) * 64;  // 16: the rest of the last piece would land past the image
            if (!ld_edge) {
                const unsigned soff = (unsigned)(ld_base + ld_c * 64);
#pragma unroll
                for (int i = 0; i < A_IT; ++i) {
                    const int pc = wm + 4 * i;  // (wave-uniform)
                    if (pc < A_PIECES - 1) dma16(ld_rsrc, dst + pc * 1024, (unsigned)a_rel[i], soff);
                    else if (pc == A_PIECES - 1 && lane < LAST_LANES) dma16(ld_rsrc, dst + pc * 1024, (unsigned)a_rel[i], soff);
                }
            } else {
                // a tile on the image border: halo pixels outside the image get an offset beyond the descriptor's range and
                // come back as zeros (the zero padding of the convolution)
                const int y0 = ldt.y0(), x0 = ldt.x0();
#pragma unroll
                for (int i = 0; i < A_IT; ++i) {
                    const int pc = wm + 4 * i;
                    const int P = pc * 16 + (lane >> 2);
                    const int hy = P / HALO_W, hx = P - hy * HALO_W;
                    const int gy = y0 + hy - 1, gx = x0 + hx - 1;
                    const bool ok = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
                    const unsigned voff = ok ? (unsigned)(ld_base + a_rel[i]) : 0x80000000u;
                    if (pc < A_PIECES - 1) dma16(ld_rsrc, dst + pc * 1024, voff, (unsigned)(ld_c * 64));
                    else if (pc == A_PIECES - 1 && lane < LAST_LANES) dma16(ld_rsrc, dst + pc * 1024, voff, (unsigned)(ld_c * 64));
                }
            }
        }
        ++ld_k;
        if (++ld_c == NC) {
            ld_c = 0;
            ldt.next();
            if (ld_k < K) ld_open_tile();
        }
    };

    // ---- fragment addresses.  Wave wm owns tile rows 2 wm and 2 wm + 1: fragment i = (row 2 wm + (i >> 1), columns
    // 16 (i & 1) .. + 15); lane (cl, kq) reads pixel cl of the fragment, k-group kq.  A tap (ky, kx) moves the fragment to
    // halo row 2 wm + (i >> 1) + ky, column + kx: four halo rows x three column shifts = 12 swizzled offsets per lane, the
    // same for every tile and chunk (kept in registers; the second column half is + 16 pixels = + 1024 bytes, which leaves the
    // swizzle bit alone).
    unsigned a_sw[4][3];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int P = (2 * wm + rr) * HALO_W + kx + cl;
            a_sw[rr][kx] = (unsigned)(P * 64 + ((kq ^ ((P >> 1) & 2)) << 4));
        }
    const unsigned w_frag0 = lds0 + LDS_W + kq * 1024 + cl * 16;

    // bias: 64 floats in LDS; the accumulators START from it (one fp32 addition per output value less in the epilogue, whose
    // vector instructions compete with the other group's MFMAs).  Lane (kq) owns channels 8 kq .. 8 kq + 7 of both halves.
    if (tid < BN) *(__attribute__((address_space(3))) float *)(size_t)(lds0 + LDS_BIAS + tid * 4) = a.bias ? a.bias[n0 + tid] : 0.f;
    const unsigned bias_at = lds0 + LDS_BIAS + kq * 32;

    f32x4 acc[4][4];
    int cp_c = 0;  // the chunk of the NEXT MFMA phase of this group
    // The first tap's fragments of a group's next MFMA phase are read at the END of its memory phase, in front of the step
    // barrier (its pixel chunk was published a step earlier; group 1's weight chunk too - group 0's is only published by that
    // very barrier, so group 0 reads its first weight fragments behind it): the MFMA phase starts on operands in registers.
    bf16x8 af[4], wf[2][4];
#define PP_RA(k_, tap_, i_)                                                                                   \
    af[i_] = __builtin_bit_cast(bf16x8, *(lds_u4_ptr)(size_t)(lds0 + LDS_A + (g * 2 + ((k_) & 1)) * A_BYTES +  \
                                                              a_sw[((i_) >> 1) + (tap_) / 3][(tap_) % 3] + ((i_) & 1) * 1024));
#define PP_RW(k_, tap_, s_, j_) \
    wf[s_][j_] = __builtin_bit_cast(bf16x8, *(lds_u4_ptr)(size_t)(w_frag0 + ((k_) & 1) * W_BYTES + (tap_) * 4096 + (j_) * 256));
    // a tile's first MFMAs take the bias as their C operand (block j of this lane: channels 32 (j >> 1) + 8 kq + 4 (j & 1) .. + 3):
    // no accumulator initialisation pass
    f32x4 bj[4];
#define PP_RBIAS() \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) bj[j] =                                                      \
        *(__attribute__((address_space(3))) const f32x4 *)(size_t)(bias_at + (j >> 1) * 128 + (j & 1) * 16);
    auto preload = [&](int k) {  // k: the super-step of this group's next MFMA phase
        if (k >= K || !cur.valid()) return;
        if (cp_c == 0) { PP_RBIAS() }
        PP_RA(k, 0, 0) PP_RA(k, 0, 1) PP_RA(k, 0, 2) PP_RA(k, 0, 3)
        if (g == 1) { PP_RW(k, 0, 0, 0) PP_RW(k, 0, 0, 1) PP_RW(k, 0, 0, 2) PP_RW(k, 0, 0, 3) }
    };

    // ---- prologue: weights of chunk 0 (all waves), this group's first chunk(s)
    issue_w(0, 0, wave, 8);
    ld_open_tile();
    ld_issue();
    if (g == 0) ld_issue();  // group 0 keeps two chunks in flight (it computes first)
    wait_all();  // (the DMA pieces and the bias words this wave wrote to LDS)
    pp_barrier();
    int first_phase = 1;  // group 0's first MFMA phase has no memory phase in front of it: it reads its first fragments itself

    auto mfma_phase = [&](int k) {
        if (cur.valid() && !PP_LAB(1)) {
            // Software pipeline, pinned with sched_group_barrier (masks: 0x008 MFMA, 0x100 DS read, 0x002 VALU): the four
            // MFMAs of pixel fragment i are followed by the read of the NEXT tap's fragment i (its register is dead by then)
            // and, behind the first two rows, by the reads of the next tap's four weight fragments (second register set): every
            // LDS read has at least eight MFMAs to land.  Left to itself hipcc reads one fragment, waits lgkmcnt(0), issues four
            // MFMAs, and repeats.  Measured (stamps, 144 MFMAs per phase, one wave per SIMD): reads in clumps behind the rows
            // 3.0-3.1 k clocks; ONE read in every MFMA gap 4.2 k (a lone LDS instruction between two MFMAs costs far more than
            // its share of a clump); the floor of this instruction mix with one wave per SIMD is 0.78 of the MFMA rate
            // (tools/mfma_lds_lab.hip), 2.95 k.
#define PP_MM(tap_, i_, j_) \
    acc[i_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[(tap_) & 1][j_], af[i_], acc[i_][j_], 0, 0, 0);
#define PP_MM0(i_, j_) acc[i_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][j_], af[i_], bj[j_], 0, 0, 0);
#define PP_ROW_READS(n_, i_)                                                                                  \
    PP_RA(k, n_, i_)                                                                                          \
    if constexpr ((i_) < 2) {                                                                                 \
        PP_RW(k, n_, (n_) & 1, 2 * (i_))                                                                      \
        PP_RW(k, n_, (n_) & 1, 2 * (i_) + 1)                                                                  \
    }                                                                                                         \
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                                        \
    __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);                                                        \
    __builtin_amdgcn_sched_group_barrier(0x100, (i_) < 2 ? 3 : 1, 0);
#define PP_ROW(tap_, i_)                                                                                      \
    PP_MM(tap_, i_, 0) PP_MM(tap_, i_, 1) PP_MM(tap_, i_, 2) PP_MM(tap_, i_, 3)                               \
    PP_ROW_READS((tap_) + 1, i_)
#define PP_ROW0(i_)  /* tap 0 of a tile's first chunk: C = the bias vectors */                                \
    PP_MM0(i_, 0) PP_MM0(i_, 1) PP_MM0(i_, 2) PP_MM0(i_, 3)                                                   \
    PP_ROW_READS(1, i_)
#define PP_TAP(tap_) { PP_ROW(tap_, 0) PP_ROW(tap_, 1) PP_ROW(tap_, 2) PP_ROW(tap_, 3) }
#define PP_TAP0_BIAS() { PP_ROW0(0) PP_ROW0(1) PP_ROW0(2) PP_ROW0(3) }
#define PP_TAP_LAST()                                                                                         \
    {                                                                                                         \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[i][j] = \
            __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][j], af[i], acc[i][j], 0, 0, 0);                      \
        __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);                                                   \
    }
            if (g == 0) {
                if (first_phase) { PP_RBIAS() PP_RA(k, 0, 0) PP_RA(k, 0, 1) PP_RA(k, 0, 2) PP_RA(k, 0, 3) }
                PP_RW(k, 0, 0, 0) PP_RW(k, 0, 0, 1) PP_RW(k, 0, 0, 2) PP_RW(k, 0, 0, 3)
            }
            first_phase = 0;
            if (!PP_LAB(16)) __builtin_amdgcn_s_setprio(1);
            if (cp_c == 0) PP_TAP0_BIAS()
            else PP_TAP(0)
            PP_TAP(1) PP_TAP(2) PP_TAP(3) PP_TAP(4) PP_TAP(5) PP_TAP(6) PP_TAP(7) PP_TAP_LAST()
            __builtin_amdgcn_s_setprio(0);
#undef PP_MM
#undef PP_MM0
#undef PP_ROW_READS
#undef PP_ROW
#undef PP_ROW0
#undef PP_TAP
#undef PP_TAP0_BIAS
#undef PP_TAP_LAST
        }
        if (++cp_c == NC) cp_c = 0;
    };

    // The epilogue of the tile this group has just finished, straight from the accumulators, in TWO parts so that no memory
    // phase is much longer than the MFMA phase beside it (what is slow are the stores: ~300 clocks of issue each):
    //   part 1 (the memory phase right behind the tile's last chunk): bf16 rounding, ReLU -> packed registers (the
    //           accumulators are free again), the stores of tile row 2 wm (fragments 0, 1) and the pooled map;
    //   part 2 (this group's NEXT memory phase, two steps later): the stores of row 2 wm + 1 (fragments 2, 3: 16 registers
    //           kept across one MFMA phase).
    uint4 pk2[2][2];                     // fragments 2 and 3, packed, between the two parts
    int dn_n = 0, dn_y0 = 0, dn_x0 = 0;  // the tile they belong to
    int pk_pending = 0;
    auto store_frag = [&](const __amdgpu_buffer_rsrc_t y_rsrc, int i, const uint4 &v0, const uint4 &v1) {
        const int gy = dn_y0 + 2 * wm + (i >> 1), gx = dn_x0 + (i & 1) * 16 + cl;
        const unsigned off = (gy < H && gx < W) ? (unsigned)(((gy * W + gx) * Cout + n0 + kq * 8) * 2) : 0x80000000u;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v0), y_rsrc, off, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v1), y_rsrc, off + 64, 0, 0);
    };
    auto pack_frag = [&](int i, int jp) {
        unsigned u[4];
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {  // channels 8 q + 2 e2, + 1 of the 32-channel half jp
            const int e = 2 * e2;
            u[e2] = pack2bf(acc[i][2 * jp + (e >> 2)][e & 3], acc[i][2 * jp + (e >> 2)][(e & 3) + 1]);
            // ReLU on the rounded pair: a bf16 is < 0 exactly when its bits, read as int16, are (and -0 -> +0)
            if constexpr (RELU) {
                const fosvos_i16x2 zero = {0, 0};
                u[e2] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(fosvos_i16x2, u[e2]), zero));
            }
        }
        return make_uint4(u[0], u[1], u[2], u[3]);
    };
    auto epilogue_part1 = [&]() {
        if (!cur.valid() || PP_LAB(2)) return;
        dn_n = cur.n; dn_y0 = cur.y0(); dn_x0 = cur.x0();
        pk_pending = 1;
        const auto y_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.y + (int64_t)dn_n * H * W * Cout, 0, H * W * Cout * 2, 0x00020000);
        uint4 pk01[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            pk01[i][0] = pack_frag(i, 0);
            pk01[i][1] = pack_frag(i, 1);
            store_frag(y_rsrc, i, pk01[i][0], pk01[i][1]);
            pk2[i][0] = pack_frag(i + 2, 0);
            pk2[i][1] = pack_frag(i + 2, 1);
        }
        if constexpr (POOL) {
            // MaxPool2d(2, 2, ceil_mode=True) of the tile (its origin is even, so no window straddles two tiles): rows 2 wm and
            // 2 wm + 1 are fragments i and i + 2 of the SAME lane, columns 2 c and 2 c + 1 are neighbouring lanes.  Post-ReLU
            // values are >= 0: the maximum is an unsigned max on the packed pairs, and pixels outside the image count as zero
            // (tiles on the right / bottom edge only), which makes the ragged last row / column windows come out right.
            const int OH = (H + 1) >> 1, OW = (W + 1) >> 1;
            const bool edge = dn_y0 + TH > H || dn_x0 + TW > W;  // wave-uniform
            const auto p_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.y_pool + (int64_t)dn_n * OH * OW * Cout, 0,
                                                                  OH * OW * Cout * 2, 0x00020000);
            const int oy = (dn_y0 >> 1) + wm;
#pragma unroll
            for (int ih = 0; ih < 2; ++ih) {
                const int ox = (dn_x0 >> 1) + ih * 8 + (cl >> 1);
                const bool ok = !(cl & 1) && oy < OH && ox < OW;
                const unsigned off = ok ? (unsigned)(((oy * OW + ox) * Cout + n0 + kq * 8) * 2) : 0x80000000u;
#pragma unroll
                for (int jp = 0; jp < 2; ++jp) {
                    uint4 top = pk01[ih][jp], bot = pk2[ih][jp];
                    if (edge) {
                        const int gy = dn_y0 + 2 * wm, gx = dn_x0 + ih * 16 + cl;
                        if (!(gy < H && gx < W)) top = make_uint4(0, 0, 0, 0);
                        if (!(gy + 1 < H && gx < W)) bot = make_uint4(0, 0, 0, 0);
                    }
                    const uint4 m = max_nonneg_bf16x8(top, bot);
                    uint4 o;  // the neighbouring lane's column (quad_perm [1, 0, 3, 2])
                    o.x = (unsigned)__builtin_amdgcn_update_dpp(0, (int)m.x, 0xB1, 0xF, 0xF, true);
                    o.y = (unsigned)__builtin_amdgcn_update_dpp(0, (int)m.y, 0xB1, 0xF, 0xF, true);
                    o.z = (unsigned)__builtin_amdgcn_update_dpp(0, (int)m.z, 0xB1, 0xF, 0xF, true);
                    o.w = (unsigned)__builtin_amdgcn_update_dpp(0, (int)m.w, 0xB1, 0xF, 0xF, true);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, max_nonneg_bf16x8(m, o)), p_rsrc,
                                                           off + jp * 64, 0, 0);
                }
            }
        }
    };
    auto epilogue_part2 = [&]() {
        if (!pk_pending) return;
        pk_pending = 0;
        const auto y_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.y + (int64_t)dn_n * H * W * Cout, 0, H * W * Cout * 2, 0x00020000);
        store_frag(y_rsrc, 2, pk2[0][0], pk2[0][1]);
        store_frag(y_rsrc, 3, pk2[1][0], pk2[1][1]);
    };

    // what a group does in the step after its MFMA phase on super-step k: request later operands, finish a tile
    auto memory_phase = [&](int k_done) {  // k_done = -1: group 1's first memory phase (nothing computed yet)
        if (g == 1 && k_done + 2 < K && (NC > 2 || k_done + 2 < 2))
            issue_w((k_done + 2) % NC, (k_done + 2) & 1, wm, 4);  // the chunk both groups read in the NEXT super-step
        ld_issue();
        epilogue_part2();                // (of the tile finished one memory phase ago)
        if (k_done >= 0 && cp_c == 0) {  // the phase before closed a tile
            epilogue_part1();
            cur.next();
        }
        preload(k_done + 1);
    };

    for (int k = 0; k < K; ++k) {
        // step 2 k: group 0 computes super-step k, group 1 is behind its super-step k - 1
        PP_STAMP(4 * k)
        if (g == 0) {
            mfma_phase(k);
            PP_STAMP(256 + k)
            wait_vm0();  // this group's pixel chunk requested in the step before: published by the barrier below
        } else {
            memory_phase(k - 1);
        }
        PP_STAMP(4 * k + 1)
        pp_barrier();
        // step 2 k + 1
        PP_STAMP(4 * k + 2)
        if (g == 1) {
            mfma_phase(k);
            PP_STAMP(256 + k)
            wait_vm0();  // the weights (and this group's pixels) requested in the step before: published by the barrier below
        } else {
            memory_phase(k);
        }
        PP_STAMP(4 * k + 3)
        pp_barrier();
    }
    PP_STAMP(4 * K)
    if (g == 1) memory_phase(K - 1);
    epilogue_part2();  // both groups: the second half of their last tile
#undef PP_RA
#undef PP_RW
#undef PP_RBIAS
}

}  // namespace pp
}  // namespace fosvos

namespace fosvos {
int conv_pp_workgroups() { return 256; }
#ifdef FOSVOS_LAB_BUILD
static unsigned long long *g_pp_stamps = nullptr;
extern "C" void fosvos_lab_set_pp_stamps(void *p) { g_pp_stamps = (unsigned long long *)p; }
#endif

// Which forward launches take the persistent kernel: 64-channel output blocks that divide the 32 workgroups of an XCD, enough
// 8 x 32 tiles that every group of every workgroup has one, and
//   * at most 64 input channels (both weight chunks stay in LDS, the layers where a tile's life outside its MFMAs weighs most:
//     conv1_2 at five 480x854 frames 156 against 217 us of the igemm), or
//   * a tile count that fills whole rounds of the 512 group slots to 90 %: a tile is the unit of work of a group, so 2100 tile
//     units (conv3 at five frames: 4.1 rounds) cost five rounds, where the igemm's independent workgroups backfill.
bool conv_pp_applicable(int N, int H, int W, int in_ch, int out_ch) {
    if (out_ch % 64 != 0 || in_ch % 32 != 0) return false;
    const int n_cb = out_ch / 64;
    if (n_cb != 1 && n_cb != 2 && n_cb != 4 && n_cb != 8) return false;
    const int64_t tiles = cdiv(W, pp::TW) * cdiv(H, pp::TH) * N;
    if (tiles > 0x3fffffff) return false;
    const int64_t units = tiles * n_cb, slots = 2 * conv_pp_workgroups();
    if (units < slots) return false;
    static const int mode = lab_env_int("FOSVOS_PP", -1);  // lab builds: 1 = whenever the shape fits
    if (mode == 1 || in_ch <= 64) return true;
    return units * 10 >= cdiv(units, slots) * slots * 9;
}



int conv_pp_forward(const uint16_t *x, const uint16_t *w_packed, const float *bias, uint16_t *y, uint16_t *y_pool, int N,
                    int H, int W, int Cin_pad, int Cout, int Co_pad, bool relu, hipStream_t st) {
    FOSVOS_REQUIRE(!y_pool || relu, FOSVOS_E_ARG, "conv3x3 (persistent): the fused pool takes post-ReLU values");
    pp::Args a{};
    a.x = x; a.w = w_packed; a.bias = bias; a.y = y; a.y_pool = y_pool;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin_pad; a.Cout = Cout; a.Co_pad = Co_pad;
    a.tiles_x = (int)cdiv(W, pp::TW);
    a.tiles_y = (int)cdiv(H, pp::TH);
    a.n_tiles = a.tiles_x * a.tiles_y * N;
    a.n_cb = Cout / 64;
    a.lab = lab_env_int("FOSVOS_PP_LAB", 0);
#ifdef FOSVOS_LAB_BUILD
    a.stamps = g_pp_stamps;
#endif
    void (*kern)(const pp::Args) = y_pool ? pp::k_conv3x3_pp<true, true>
                                   : relu ? pp::k_conv3x3_pp<true, false> : pp::k_conv3x3_pp<false, false>;
    const int which = y_pool ? 2 : relu ? 1 : 0;
    static bool once[64][3];
    int dev = 0;
    FOSVOS_HIP_CHECK(hipGetDevice(&dev));
    if (dev >= 0 && dev < 64 && !once[dev][which]) {  // opt in to the whole LDS of a CU, once per device and instantiation
        FOSVOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             pp::LDS_BYTES));
        once[dev][which] = true;
    }
    FOSVOS_PROF(y_pool ? "k_conv3x3_pp<true, true>" : relu ? "k_conv3x3_pp<true, false>" : "k_conv3x3_pp<false, false>", st,
                2.0 * N * H * W * 9.0 * Cin_pad * Cout);
    hipLaunchKernelGGL(kern, dim3(conv_pp_workgroups()), dim3(pp::NT), pp::LDS_BYTES, st, a);
    FOSVOS_LAUNCH_CHECK();
    return FOSVOS_OK;
}

}  // namespace fosvos
