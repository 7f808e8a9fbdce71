// conv1_1 forward: Conv2d(3 -> 64, 3x3, pad 1) + bias + ReLU straight from the fp32 NCHW frame
// (reference: stages[0][0..1], src/networks/osvos_vgg.py:92-93).
//
// The layer is 0.7 GMAC per 480x854 frame and writes 52 MB of bf16 NHWC activation: HBM-write-bound if the arithmetic
// is cheap.  On the vector ALU it is not (round 1's kernel: 864 FMAs per lane and 32-pixel segment, 45 % of the fp32 VALU
// rate, 23 us per frame, and every one of those instructions competes with the MFMA waves of the other stream), so the
// arithmetic runs on the matrix pipe without giving up fp32 inputs: frame and weights are each split into two bf16 terms
// (x = xh + xl, xh = bf16(x), xl = bf16(x - xh); likewise w) and a product is the sum of the four term products,
// accumulated in fp32 by v_mfma_f32_16x16x32_bf16 - what is lost is the second rounding of the low terms, ~2^-18 relative
// (a plain bf16 frame, 2^-9, shifts the logits measurably: DESIGN.md section 4).
//
// GEMM view per 16 pixels: out[64 co][16 px] = W[64][K] * X[K][16], K = (ky, kx, ci) laid out so that the operand of a
// lane is 16 contiguous bytes of the staged tile: a halo row is stored pixel-major with 4 channels per pixel (3 real + a
// zero), 8 bytes per pixel and term; k-block b, lane group kg, element e <-> ky = 2 b + (kg >> 1), kx = 2 (kg & 1) + (e >> 2),
// ci = e & 3.  ky = 3, kx = 3 and ci = 3 are padding of K: their WEIGHTS are zero, the data read there is finite
// (a neighbouring pixel, a zeroed pad column or the zero channel).  Two k-blocks of 32 cover the 27 taps.
// The weight rows are permuted so that a lane ends up with 16 CONSECUTIVE output channels of its pixel (row r of
// co-block j is channel 16 (r >> 2) + 4 j + (r & 3)): two 16-byte stores per lane and pixel.
//
// Its weight gradient reduces over 410 k pixels and lives in conv_wgrad_first.hip.  No dgrad: the image needs none.
#include "common.hpp"

using namespace fosvos;

namespace {
constexpr int CO = 64;              // the only instantiation the model needs (checked at the entry point)
constexpr int TH = 8, TW = 32;      // output pixels of a tile: 4 waves x 2 rows x 2 steps of 16 pixels
constexpr int HR = TH + 2;          // halo rows
constexpr int HC = TW + 4;          // halo columns: 34 real + 2 zero columns the kx = 3 padding taps may read
constexpr int IMG = HR * HC;        // staged pixels per tile
constexpr int IMG_BYTES = IMG * 8;  // ... of one term (hi or lo): 4 bf16 per pixel
constexpr int BUF_BYTES = 2 * IMG_BYTES;
constexpr int S_IT = (IMG + 255) / 256;  // staged pixels per thread

struct Split {
    uint16_t hi, lo;
};
struct alignas(16) Pair8 {  // two 8-byte halves of one 16-byte MFMA operand
    uint2 a, b;
};
__device__ __forceinline__ Split split_bf16(float v) {
    const uint16_t h = f2bf(v);
    return Split{h, f2bf(v - bf2f(h))};
}

// relu_bits (may be null): the ReLU mask of y as one bit per element, [N,H,W,8] bytes - bit e of byte g = channel 8 g + e
// is > 0 - for the data gradient of the NEXT conv (fosvos_conv3x3_dgrad_bits), which at 480x854 is bound by HBM traffic and
// otherwise reads all 128 bytes of every pixel of y only to test their signs.
__global__ __launch_bounds__(256) void k_first_fwd(const float *__restrict__ frame, const float *__restrict__ w,
                                                    const float *__restrict__ bias, uint16_t *__restrict__ y,
                                                    uint8_t *__restrict__ relu_bits, int N, int H, int W, int tiles_x,
                                                    int tiles_y) {
    __shared__ __attribute__((aligned(16))) char s_x[2 * BUF_BYTES];  // two tiles (double buffer) x {hi, lo} images
    __shared__ float s_w[CO * 27];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p = lane & 15, kg = lane >> 4;
    const int n_tiles = tiles_x * tiles_y * N;
    const int64_t plane = (int64_t)H * W;

    // ---- weights -> the MFMA row operands of this lane, both terms, kept in registers for every tile of the workgroup
    for (int i = tid; i < CO * 27; i += 256) s_w[i] = w[i];
    __syncthreads();
    bf16x8 wh[2][4], wl[2][4];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ch = 16 * (p >> 2) + 4 * j + (p & 3);  // MFMA row p of co-block j
            uint16_t hv[8], lv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int ky = 2 * b + (kg >> 1), kx = 2 * (kg & 1) + (e >> 2), ci = e & 3;
                float v = 0.f;
                if (ky < 3 && kx < 3 && ci < 3) v = s_w[ch * 27 + ci * 9 + ky * 3 + kx];
                const Split s = split_bf16(v);
                hv[e] = s.hi;
                lv[e] = s.lo;
            }
            const Pair8 hh{make_uint2(hv[0] | ((uint32_t)hv[1] << 16), hv[2] | ((uint32_t)hv[3] << 16)),
                           make_uint2(hv[4] | ((uint32_t)hv[5] << 16), hv[6] | ((uint32_t)hv[7] << 16))};
            const Pair8 ll{make_uint2(lv[0] | ((uint32_t)lv[1] << 16), lv[2] | ((uint32_t)lv[3] << 16)),
                           make_uint2(lv[4] | ((uint32_t)lv[5] << 16), lv[6] | ((uint32_t)lv[7] << 16))};
            wh[b][j] = __builtin_bit_cast(bf16x8, hh);
            wl[b][j] = __builtin_bit_cast(bf16x8, ll);
        }
    // bias of this lane's 16 output channels
    float4 bv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bv[j] = *reinterpret_cast<const float4 *>(bias + 16 * kg + 4 * j);

    // ---- staging: thread t owns staged pixels t, t + 256 of a tile (row-major over HR x HC)
    float st[S_IT][3];
    auto load_tile = [&](int tile) {
        const int tx = tile % tiles_x, r = tile / tiles_x;
        const int ty = r % tiles_y, n = r / tiles_y;
        const float *fn = frame + (int64_t)n * 3 * plane;
#pragma unroll
        for (int it = 0; it < S_IT; ++it) {
            const int idx = it * 256 + tid;
            const int hy = idx / HC, hx = idx - hy * HC;
            const int gy = ty * TH + hy - 1, gx = tx * TW + hx - 1;
            const bool ok = idx < IMG && hx < TW + 2 && gy >= 0 && gy < H && gx >= 0 && gx < W;
            const int64_t off = ok ? (int64_t)gy * W + gx : 0;  // unconditional loads from a valid address, masked below
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float v = fn[c * plane + off];
                st[it][c] = ok ? v : 0.f;
            }
        }
    };
    auto store_tile = [&](char *buf) {
#pragma unroll
        for (int it = 0; it < S_IT; ++it) {
            const int idx = it * 256 + tid;
            if (idx < IMG) {
                const Split a = split_bf16(st[it][0]), b = split_bf16(st[it][1]), c = split_bf16(st[it][2]);
                *reinterpret_cast<uint2 *>(buf + idx * 8) = make_uint2(a.hi | ((uint32_t)b.hi << 16), c.hi);
                *reinterpret_cast<uint2 *>(buf + IMG_BYTES + idx * 8) = make_uint2(a.lo | ((uint32_t)b.lo << 16), c.lo);
            }
        }
    };

    int tile = blockIdx.x;
    if (tile < n_tiles) {
        load_tile(tile);
        store_tile(s_x);
    }
    __syncthreads();
    for (int it_tile = 0; tile < n_tiles; tile += gridDim.x, ++it_tile) {
        const char *cur = s_x + (it_tile & 1) * BUF_BYTES;
        char *nxt = s_x + ((it_tile & 1) ^ 1) * BUF_BYTES;
        const int next = tile + (int)gridDim.x;
        if (next < n_tiles) load_tile(next);  // in flight under this tile's matrix work

        const int tx = tile % tiles_x, r = tile / tiles_x;
        const int ty = r % tiles_y, n = r / tiles_y;
        // operand address of this lane inside a halo row: pixel p + 2 (kg & 1) of the step, rows ky = kg >> 1 (k-block 0)
        // and 2 (k-block 1; its upper lane groups are all padding and re-read row 2)
        const int lane_px = (p + 2 * (kg & 1)) * 8;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int row = 2 * wave + (s >> 1), xb = (s & 1) * 16;
            const int gy = ty * TH + row, gx = tx * TW + xb + p;
            f32x4 acc[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = f32x4{bv[j].x, bv[j].y, bv[j].z, bv[j].w};
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int hy = row + (b == 0 ? (kg >> 1) : 2);
                const char *src = cur + (hy * HC + xb) * 8 + lane_px;
                // 16 bytes at an 8-byte-aligned address: two 8-byte LDS reads (one ds_read2_b64) per term
                const Pair8 h{*reinterpret_cast<const uint2 *>(src), *reinterpret_cast<const uint2 *>(src + 8)};
                const Pair8 l{*reinterpret_cast<const uint2 *>(src + IMG_BYTES),
                              *reinterpret_cast<const uint2 *>(src + IMG_BYTES + 8)};
                const bf16x8 xh = __builtin_bit_cast(bf16x8, h), xl = __builtin_bit_cast(bf16x8, l);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[b][j], xl, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[b][j], xh, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[b][j], xl, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[b][j], xh, acc[j], 0, 0, 0);
                }
            }
            // acc[j][i] = channel 16 kg + 4 j + i of pixel p: ReLU, pack, 32 contiguous bytes per lane
            if (gy < H && gx < W) {
                uint4 o0, o1;
                o0.x = pack2bf(relu_f(acc[0][0]), relu_f(acc[0][1]));
                o0.y = pack2bf(relu_f(acc[0][2]), relu_f(acc[0][3]));
                o0.z = pack2bf(relu_f(acc[1][0]), relu_f(acc[1][1]));
                o0.w = pack2bf(relu_f(acc[1][2]), relu_f(acc[1][3]));
                o1.x = pack2bf(relu_f(acc[2][0]), relu_f(acc[2][1]));
                o1.y = pack2bf(relu_f(acc[2][2]), relu_f(acc[2][3]));
                o1.z = pack2bf(relu_f(acc[3][0]), relu_f(acc[3][1]));
                o1.w = pack2bf(relu_f(acc[3][2]), relu_f(acc[3][3]));
                uint16_t *dst = y + (((int64_t)n * H + gy) * W + gx) * CO + 16 * kg;
                *reinterpret_cast<uint4 *>(dst) = o0;
                *reinterpret_cast<uint4 *>(dst + 8) = o1;
                if (relu_bits) {  // from the STORED values (post-ReLU, so > 0 is != 0): this lane's 16 channels = two bytes
                    const unsigned wds[8] = {o0.x, o0.y, o0.z, o0.w, o1.x, o1.y, o1.z, o1.w};
                    unsigned m = 0;
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        m |= ((wds[e] & 0xffffu) ? 1u : 0u) << (2 * e) | ((wds[e] >> 16) ? 1u : 0u) << (2 * e + 1);
                    *reinterpret_cast<uint16_t *>(relu_bits + (((int64_t)n * H + gy) * W + gx) * (CO / 8) + 2 * kg) = (uint16_t)m;
                }
            }
        }
        if (next < n_tiles) store_tile(nxt);  // image `nxt` was last read in the previous iteration (barrier below)
        __syncthreads();
    }
}

}  // namespace

namespace {
constexpr int kFirstMaxWorkgroups = 1024;  // persistent workgroups (the weight operands are built once per workgroup): four per CU
}
extern "C" int fosvos_conv3x3_first_plan(int N, int H, int W, int *tiles, int *workgroups) {
    FOSVOS_REQUIRE(tiles && workgroups, FOSVOS_E_ARG, "conv3x3_first_plan: null output");
    FOSVOS_REQUIRE(N > 0 && H > 0 && W > 0, FOSVOS_E_SHAPE, "conv3x3_first_plan: bad shape N=%d H=%d W=%d", N, H, W);
    const int64_t t = cdiv(W, TW) * cdiv(H, TH) * N;
    FOSVOS_REQUIRE(t < 0x7fffffffLL, FOSVOS_E_SHAPE, "conv3x3_first_plan: too many tiles");
    *tiles = (int)t;
    *workgroups = (int)std::min<int64_t>(t, kFirstMaxWorkgroups);
    return FOSVOS_OK;
}

extern "C" int fosvos_conv3x3_first_fwd(const float *frame, const float *w, const float *bias, uint16_t *y, int N,
                                        int H, int W, int Co, int device, void *stream) {
    return fosvos_conv3x3_first_fwd_bits(frame, w, bias, y, nullptr, N, H, W, Co, device, stream);
}

extern "C" int fosvos_conv3x3_first_fwd_bits(const float *frame, const float *w, const float *bias, uint16_t *y,
                                             uint8_t *relu_bits, int N, int H, int W, int Co, int device, void *stream) {
    FOSVOS_REQUIRE(frame && w && bias && y, FOSVOS_E_ARG, "conv3x3_first_fwd: null pointer");
    FOSVOS_REQUIRE(Co == CO, FOSVOS_E_SHAPE, "conv3x3_first_fwd: Co=%d, only %d is built", Co, CO);
    FOSVOS_REQUIRE(N > 0 && H > 0 && W > 0 && N <= 65535, FOSVOS_E_SHAPE, "conv3x3_first_fwd: bad shape N=%d H=%d W=%d",
                   N, H, W);
    FOSVOS_ENTER(device);
    const int tiles_x = (int)cdiv(W, TW), tiles_y = (int)cdiv(H, TH);
    const int64_t tiles = (int64_t)tiles_x * tiles_y * N;
    FOSVOS_REQUIRE(tiles < 0x7fffffffLL, FOSVOS_E_SHAPE, "conv3x3_first_fwd: too many tiles");
    const unsigned grid = (unsigned)std::min<int64_t>(tiles, kFirstMaxWorkgroups);
    FOSVOS_PROF("k_first_fwd", stream, 2.0 * N * H * W * 27 * Co);
    hipLaunchKernelGGL(k_first_fwd, dim3(grid), dim3(256), 0, (hipStream_t)stream, frame, w, bias, y, relu_bits, N, H, W,
                       tiles_x, tiles_y);
    FOSVOS_LAUNCH_CHECK();
    return FOSVOS_OK;
}
