// Forward 3x3 / pad 1 / stride 1 convolution (+ bias + ReLU [+ 2x2 ceil-mode max pool]) on bf16 NHWC: the PERSISTENT
// eight-wave form of the implicit GEMM (src/networks/osvos_vgg.py:90-93 - Conv2d(3x3) + ReLU, and the MaxPool2d that follows
// the last conv of a stage).
//
// Why a second kernel beside k_conv3x3_igemm (conv_igemm.hip): that kernel's workgroup lives 19 k clocks on a 64-channel
// layer for 4.6 k clocks of matrix work (launch -> offsets -> first tile in LDS -> epilogue round trips), so two of them per
// CU keep the matrix pipe 48 % busy (70 % at 256 channels).  Here ONE workgroup owns the CU for the whole launch:
//   * 8 waves = two GROUPS of four (waves 0-3 / 4-7; wave w and w + 4 share a SIMD).  Each group owns a pixel tile
//     (8 x 32 pixels x 64 output channels, 64 accumulator registers per lane) of its own.
//   * time is cut into SUPER-STEPS of one K chunk (32 input channels x 9 taps = 144 MFMAs per wave): both groups run the
//     chunk's MFMAs on their own tiles in the same super-step (two waves per SIMD feed the matrix pipe), ONE bare s_barrier per
//     super-step.  The memory work rides BETWEEN the taps: the requests for the next chunk (group 0 in front of tap 0,
//     group 1 behind tap 4) and the stores of the tile finished in the previous super-step (group 1 behind taps 0-2, group 0
//     behind taps 3-8), so that at any time at most one of the two waves of a SIMD is away from its MFMAs.
//   * every operand byte goes global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds): no staging registers, no ds_write.
//     Pixel tiles: [halo pixel][4 x 16 B] per chunk, the 16-byte k-group kq of pixel P at slot kq ^ ((P >> 1) & 2) - the
//     swizzle sits on the SOURCE address, the LDS image is lane-linear - which makes every ds_read_b128 fragment read
//     conflict-free for every tap shift (the read's 16-lane service groups mix two k-groups: lanes 0-3, 12-15, 20-27;
//     SQ_LDS_BANK_CONFLICT = 0 in profiles/r04_pmc_sq.txt).
//     Weights: the packed image's [tap][kq][64 channels][16 B] chunk, one copy shared by BOTH groups, double-buffered; with
//     <= 64 input channels the two chunks are loaded once and stay (weights stationary).
//   * the epilogue runs straight from the accumulators: they START from the bias, are rounded to bf16 and clamped (ReLU on the
//     packed pairs) into 32 registers when the tile's last chunk is done, and those are stored during the next super-step; the
//     2 x 2 pool is a max over two fragments of the same lane and one DPP lane swap.
//   * the workgroups are persistent: 256 of them (one per CU), each walks its share of the tiles; the ids that share an XCD
//     take one contiguous eighth of the tile space, neighbouring tiles at the same time (shared halo rows meet in that L2).
//   * channel order inside a 64-channel block is permuted in the LDS weight image so that a lane's accumulators hold 8
//     CONSECUTIVE channels of a pixel: the epilogue stores 16 bytes per lane and instruction without an LDS round trip.
// LDS: 2 x 36,864 (weights) + 2 groups x 2 x 21,760 (pixel tiles) + 256 (bias) = 161,024 of the CU's 163,840 bytes.
//
// Measured on the way (profiles/r04_lab_pp_*.txt; in-kernel stamps of lab builds, tools/pp_stamp_lab.py):
//   * the first schedule ALTERNATED the groups ("ping-pong": in step s group s & 1 runs its MFMAs, the other one its memory
//     phase, a barrier per step).  Its MFMA phase took 3.0-3.2 k clocks for 2.3 k of matrix work whatever the other group did
//     (also with every memory operation switched off): ONE wave per SIMD issuing 16 MFMAs + 8 ds_read_b128 per tap reaches
//     ~75 % of the pipe (tools/mfma_lds_lab.hip: 0.78); and a memory phase is bound by its wave's own in-order issue (one
//     instruction per ~4 clocks: 600 scalar + 400 vector instructions = 4 k clocks until the tile walk, the descriptors and
//     the DMA offsets of interior tiles were moved to a handful of scalar instructions), by ~170 clocks per LDS-DMA piece,
//     and by ~300 clocks per 1-KB store.  conv3_2 at five frames: 148-152 us (igemm 140), conv1_2 156-178 (igemm 217).
//   * this (lockstep) schedule: conv3_2 140, conv4_2 156 (alternating: 171), conv1_2 178; in the fine-tune step the two are
//     equal (1253 vs 1252 frames/s; 1234 without the persistent kernel).  A super-step still takes ~7.0 k clocks for 4.6 k
//     of matrix work: 10 DMA pieces per wave (1.7 k clocks during which its SIMD partner computes alone) and the stores.
//   * LDS reads dealt out one (or two) per MFMA gap instead of in clumps behind four MFMAs: 4.2 k (3.5 k) clocks per phase
//     instead of 3.1 k - a lone LDS instruction between two MFMAs costs far more than its share of a clump.
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
    const uint8_t *mask_bits;  // MASK: [N,H,W,Cout/8] bytes, bit e of byte g set = keep channel 8 g + e (a data gradient's ReLU mask)
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

template <bool RELU, bool POOL, bool MASK = false>
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
    int cp_c = 0;  // the chunk of this super-step inside its tile
    bf16x8 af[4], wf[2][4];
#define PP_RA(k_, tap_, i_)                                                                                   \
    af[i_] = __builtin_bit_cast(bf16x8, *(lds_u4_ptr)(size_t)(lds0 + LDS_A + (g * 2 + ((k_) & 1)) * A_BYTES +  \
                                                              a_sw[((i_) >> 1) + (tap_) / 3][(tap_) % 3] + ((i_) & 1) * 1024));
#define PP_RW(k_, tap_, s_, j_) \
    wf[s_][j_] = __builtin_bit_cast(bf16x8, *(lds_u4_ptr)(size_t)(w_frag0 + ((k_) & 1) * W_BYTES + (tap_) * 4096 + (j_) * 256));
    // a tile's accumulators start from the bias (block j of this lane: channels 32 (j >> 1) + 8 kq + 4 (j & 1) .. + 3), read
    // from LDS straight into the accumulator registers
#define PP_RBIAS()                                                                                            \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[i][j] =    \
        *(__attribute__((address_space(3))) const f32x4 *)(size_t)(bias_at + (j >> 1) * 128 + (j & 1) * 16);

    // ---- the finished tile's output: packed at the end of its last chunk, STORED during the next super-step, a few
    // instructions at a time between the taps (a burst of twelve stores would hold this wave - and, the groups running in
    // lockstep, its SIMD partner too - for thousands of clocks with no MFMA issued)
    uint4 pk[4][2];
    unsigned mb[MASK ? 4 : 1][2];  // MASK: the mask bytes of the packed vectors, requested when the tile is packed
    int dn_n = 0, dn_y0 = 0, dn_x0 = 0;  // the tile they belong to
    int st_pending = 0;
    auto pack_frag = [&](int i, int jp) {
        unsigned u[4];
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {
            const int e = 2 * e2;
            u[e2] = pack2bf(acc[i][2 * jp + (e >> 2)][e & 3], acc[i][2 * jp + (e >> 2)][(e & 3) + 1]);
            if constexpr (RELU) {
                const fosvos_i16x2 zero = {0, 0};
                u[e2] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(fosvos_i16x2, u[e2]), zero));
            }
        }
        return make_uint4(u[0], u[1], u[2], u[3]);
    };
    auto pack_tile = [&]() {  // accumulators -> packed registers (+ the pooled vectors); the accumulators are free afterwards
        if (!cur.valid() || PP_LAB(2)) return;
        dn_n = cur.n; dn_y0 = cur.y0(); dn_x0 = cur.x0();
        st_pending = 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            pk[i][0] = pack_frag(i, 0);
            pk[i][1] = pack_frag(i, 1);
        }
        if constexpr (MASK) {  // one byte per 16-byte vector; they have the next super-step's first taps to arrive under
            const auto m_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(a.mask_bits) + (int64_t)dn_n * H * W * (Cout >> 3),
                                                                  0, H * W * (Cout >> 3), 0x00020000);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int gy = dn_y0 + 2 * wm + (i >> 1), gx = dn_x0 + (i & 1) * 16 + cl;
                const unsigned off = (gy < H && gx < W) ? (unsigned)((gy * W + gx) * (Cout >> 3) + (n0 >> 3) + kq) : 0x80000000u;
                mb[i][0] = __builtin_amdgcn_raw_buffer_load_b8(m_rsrc, off, 0, 0);
                mb[i][1] = __builtin_amdgcn_raw_buffer_load_b8(m_rsrc, off + 4, 0, 0);
            }
        }
    };
    auto mask_bits_of = [](unsigned b) {  // eight bits -> the 0 / 1 halfwords keep_where_pos_bf16x8 tests
        return make_uint4((b & 1u) | ((b & 2u) << 15), ((b >> 2) & 1u) | ((b & 8u) << 13), ((b >> 4) & 1u) | ((b & 32u) << 11),
                          ((b >> 6) & 1u) | ((b & 128u) << 9));
    };
    // slot 0..3: fragment i's two 16-byte stores; slots 4, 5 (POOL): the pooled vectors of column half ih = slot - 4
    auto store_slot = [&](int slot) {
        if (!st_pending) return;
        if (slot < 4) {
            const auto y_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.y + (int64_t)dn_n * H * W * Cout, 0, H * W * Cout * 2, 0x00020000);
            const int gy = dn_y0 + 2 * wm + (slot >> 1), gx = dn_x0 + (slot & 1) * 16 + cl;
            const unsigned off = (gy < H && gx < W) ? (unsigned)(((gy * W + gx) * Cout + n0 + kq * 8) * 2) : 0x80000000u;
            uint4 v0 = pk[slot][0], v1 = pk[slot][1];
            if constexpr (MASK) {
                v0 = keep_where_pos_bf16x8(v0, mask_bits_of(mb[slot][0]));
                v1 = keep_where_pos_bf16x8(v1, mask_bits_of(mb[slot][1]));
            }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v0), y_rsrc, off, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v1), y_rsrc, off + 64, 0, 0);
        } else if constexpr (POOL) {
            // MaxPool2d(2, 2, ceil_mode=True) of the tile (its origin is even, so no window straddles two tiles): rows 2 wm and
            // 2 wm + 1 are fragments ih and ih + 2 of the SAME lane, columns 2 c and 2 c + 1 are neighbouring lanes.  Post-ReLU
            // values are >= 0: the maximum is an unsigned max on the packed pairs, and pixels outside the image count as zero
            // (tiles on the right / bottom edge only), which makes the ragged last row / column windows come out right.
            const int ih = slot - 4;
            const int OH = (H + 1) >> 1, OW = (W + 1) >> 1;
            const bool edge = dn_y0 + TH > H || dn_x0 + TW > W;  // wave-uniform
            const auto p_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.y_pool + (int64_t)dn_n * OH * OW * Cout, 0,
                                                                  OH * OW * Cout * 2, 0x00020000);
            const int oy = (dn_y0 >> 1) + wm, ox = (dn_x0 >> 1) + ih * 8 + (cl >> 1);
            const bool ok = !(cl & 1) && oy < OH && ox < OW;
            const unsigned off = ok ? (unsigned)(((oy * OW + ox) * Cout + n0 + kq * 8) * 2) : 0x80000000u;
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
                uint4 top = pk[ih][jp], bot = pk[ih + 2][jp];
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
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, max_nonneg_bf16x8(m, o)), p_rsrc, off + jp * 64, 0, 0);
            }
        }
    };
    constexpr int N_SLOTS = POOL ? 6 : 4;

    // the requests of super-step k + 1 (this group's pixel chunk; this wave's share of the weight chunk both groups use)
    auto request_next = [&](int k) {
        if (k + 1 < K && (NC > 2 || k + 1 < 2)) issue_w((k + 1) % NC, (k + 1) & 1, wave, 8);
        ld_issue();
    };

    // ---- prologue: weights of chunk 0 (all waves), this group's first chunk
    issue_w(0, 0, wave, 8);
    ld_open_tile();
    ld_issue();
    wait_all();  // (the DMA pieces and the bias words this wave wrote to LDS)
    pp_barrier();

    // Software pipeline of a tap, pinned with sched_group_barrier (masks: 0x008 MFMA, 0x100 DS read, 0x002 VALU): the four
    // MFMAs of pixel fragment i, then the read of the NEXT tap's fragment i (its register is dead by then) and, behind the
    // first two rows, the reads of the next tap's four weight fragments (second register set).
#define PP_MM(tap_, i_, j_) \
    acc[i_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[(tap_) & 1][j_], af[i_], acc[i_][j_], 0, 0, 0);
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
#define PP_TAP(tap_) { PP_ROW(tap_, 0) PP_ROW(tap_, 1) PP_ROW(tap_, 2) PP_ROW(tap_, 3) }
#define PP_TAP_LAST()                                                                                         \
    {                                                                                                         \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[i][j] = \
            __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][j], af[i], acc[i][j], 0, 0, 0);                      \
        __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);                                                   \
    }
// what rides between the taps of a super-step: group 0 requests the next chunks in FRONT of tap 0 and stores the finished tile
// behind taps 3-8; group 1 stores behind taps 0-2 (two slots each) and requests behind tap 4 - at any time at most one of the
// two waves of a SIMD is away from its MFMAs
#define PP_BETWEEN(tap_)                                                                                      \
    if (g == 0) {                                                                                             \
        if constexpr ((tap_) >= 3 && (tap_) - 3 < N_SLOTS) store_slot((tap_) - 3);                            \
    } else {                                                                                                  \
        if constexpr ((tap_) < 3) {                                                                           \
            store_slot(2 * (tap_));                                                                           \
            if constexpr (2 * (tap_) + 1 < N_SLOTS) store_slot(2 * (tap_) + 1);                               \
        }                                                                                                     \
        if constexpr ((tap_) == 4) request_next(k);                                                           \
    }

    for (int k = 0; k < K; ++k) {
        PP_STAMP(4 * k)
        if (g == 0) request_next(k);
        PP_STAMP(4 * k + 1)
        if (cur.valid() && !PP_LAB(1)) {
            if (cp_c == 0) { PP_RBIAS() }
            PP_RA(k, 0, 0) PP_RA(k, 0, 1) PP_RA(k, 0, 2) PP_RA(k, 0, 3)
            PP_RW(k, 0, 0, 0) PP_RW(k, 0, 0, 1) PP_RW(k, 0, 0, 2) PP_RW(k, 0, 0, 3)
            PP_TAP(0) PP_BETWEEN(0) PP_TAP(1) PP_BETWEEN(1) PP_TAP(2) PP_BETWEEN(2) PP_TAP(3) PP_BETWEEN(3) PP_TAP(4) PP_BETWEEN(4)
            PP_TAP(5) PP_BETWEEN(5) PP_TAP(6) PP_BETWEEN(6) PP_TAP(7) PP_BETWEEN(7) PP_TAP_LAST() PP_BETWEEN(8)
        } else {  // a group without a tile in its last round: the memory work only
            PP_BETWEEN(0) PP_BETWEEN(1) PP_BETWEEN(2) PP_BETWEEN(3) PP_BETWEEN(4) PP_BETWEEN(5) PP_BETWEEN(6) PP_BETWEEN(7) PP_BETWEEN(8)
        }
        st_pending = 0;  // (every slot of the previous tile has been stored by now)
        if (++cp_c == NC) {
            cp_c = 0;
            pack_tile();
            cur.next();
        }
        PP_STAMP(4 * k + 2)
        wait_vm0();  // this wave's requests of the super-step (and its stores): published by the barrier below
        PP_STAMP(4 * k + 3)
        pp_barrier();
    }
    PP_STAMP(4 * K)
#pragma unroll
    for (int slot = 0; slot < N_SLOTS; ++slot) store_slot(slot);  // the last tile
#undef PP_MM
#undef PP_ROW_READS
#undef PP_ROW
#undef PP_TAP
#undef PP_TAP_LAST
#undef PP_BETWEEN
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
                    int H, int W, int Cin_pad, int Cout, int Co_pad, bool relu, hipStream_t st, const uint8_t *mask_bits) {
    FOSVOS_REQUIRE(!y_pool || relu, FOSVOS_E_ARG, "conv3x3 (persistent): the fused pool takes post-ReLU values");
    FOSVOS_REQUIRE(!mask_bits || (!relu && !y_pool && !bias), FOSVOS_E_ARG, "conv3x3 (persistent): the bit mask is the data gradient's epilogue");
    pp::Args a{};
    a.x = x; a.w = w_packed; a.bias = bias; a.y = y; a.y_pool = y_pool; a.mask_bits = mask_bits;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin_pad; a.Cout = Cout; a.Co_pad = Co_pad;
    a.tiles_x = (int)cdiv(W, pp::TW);
    a.tiles_y = (int)cdiv(H, pp::TH);
    a.n_tiles = a.tiles_x * a.tiles_y * N;
    a.n_cb = Cout / 64;
    a.lab = lab_env_int("FOSVOS_PP_LAB", 0);
#ifdef FOSVOS_LAB_BUILD
    a.stamps = g_pp_stamps;
#endif
    void (*kern)(const pp::Args) = mask_bits ? pp::k_conv3x3_pp<false, false, true>
                                   : y_pool  ? pp::k_conv3x3_pp<true, true>
                                   : relu    ? pp::k_conv3x3_pp<true, false> : pp::k_conv3x3_pp<false, false>;
    const int which = mask_bits ? 3 : y_pool ? 2 : relu ? 1 : 0;
    static bool once[64][4];
    int dev = 0;
    FOSVOS_HIP_CHECK(hipGetDevice(&dev));
    if (dev >= 0 && dev < 64 && !once[dev][which]) {  // opt in to the whole LDS of a CU, once per device and instantiation
        FOSVOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             pp::LDS_BYTES));
        once[dev][which] = true;
    }
    FOSVOS_PROF(mask_bits ? "k_conv3x3_pp<false, false, true>" : y_pool ? "k_conv3x3_pp<true, true, false>"
                : relu    ? "k_conv3x3_pp<true, false, false>" : "k_conv3x3_pp<false, false, false>", st,
                2.0 * N * H * W * 9.0 * Cin_pad * Cout);
    hipLaunchKernelGGL(kern, dim3(conv_pp_workgroups()), dim3(pp::NT), pp::LDS_BYTES, st, a);
    FOSVOS_LAUNCH_CHECK();
    return FOSVOS_OK;
}

}  // namespace fosvos
