// 2x2 stride-2 ceil-mode max pool on bf16 NHWC (reference: MaxPool2d(2,2,ceil_mode=True),
// src/networks/osvos_vgg.py:90) and its backward with the producer's ReLU backward fused.
//
// HBM-bound.  One thread owns one output pixel x 8 channels (16-byte vectors).  Forward moves
// 2 B/in-element + 2 B/out-element; backward reads x and dy and writes dx.  Windows never
// overlap, so backward is a gather per window: no atomics, every dx element written exactly once.
#include "common.hpp"

using namespace fosvos;

namespace {
__global__ __launch_bounds__(256) void k_pool_fwd(const uint16_t *__restrict__ x, uint16_t *__restrict__ y, int H,
                                                   int W, int C, int OH, int OW, int64_t total) {
    const int groups = C >> 3;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int g = (int)(i % groups);
        int64_t r = i / groups;
        const int ox = (int)(r % OW);
        r /= OW;
        const int oy = (int)(r % OH);
        const int64_t n = r / OH;
        const int iy = oy * 2, ix = ox * 2;
        float m[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = -INFINITY;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                if (iy + dy < H && ix + dx < W) {
                    const uint4 v = *reinterpret_cast<const uint4 *>(
                        x + (((n * H + iy + dy) * W + ix + dx) * C + g * 8));
                    float f[8];
                    unpack8(v, f);
#pragma unroll
                    for (int j = 0; j < 8; ++j) m[j] = f[j] > m[j] ? f[j] : m[j];
                }
            }
        *reinterpret_cast<uint4 *>(y + (((n * OH + oy) * OW + ox) * C + g * 8)) = pack8(m);
    }
}

// Backward: grid = (column blocks, pooled rows, images), a thread = one pooled pixel x 8 channels.  All index arithmetic is
// 32-bit inside an image (the 64-bit division chain of a flat index cost more vector-ALU issue than the comparisons: the
// kernel runs beside the weight-gradient MFMA stream and every instruction it issues is taken from that stream's SIMDs).
// The four window pixels and the gradient are unpacked to fp32 only for the comparisons; what is stored is either the
// gradient's own bf16 bits or zero, so the results are put back together with one byte permute per pair (no converts).
__global__ __launch_bounds__(256) void k_pool_bwd(const uint16_t *__restrict__ x, const uint16_t *__restrict__ dy,
                                                   uint16_t *__restrict__ dx, int H, int W, int C, int OH, int OW,
                                                   int relu_mask) {
    const int groups = C >> 3;
    const int col = blockIdx.x * 256 + threadIdx.x;  // (ox, channel group)
    if (col >= OW * groups) return;
    const int ox = col / groups, g = col - ox * groups;
    const int oy = blockIdx.y;
    const int64_t n = blockIdx.z;
    const uint16_t *xn = x + n * H * W * C;
    uint16_t *dxn = dx + n * H * W * C;
    const int iy = oy * 2, ix = ox * 2;
    const bool vx = ix + 1 < W, vy = iy + 1 < H;  // (iy, ix) itself is always inside
    const int o00 = (iy * W + ix) * C + g * 8;
    const int o01 = o00 + C, o10 = o00 + W * C, o11 = o10 + C;
    // unconditional loads from clamped addresses, the out-of-image ones are masked below
    const uint4 v0 = *reinterpret_cast<const uint4 *>(xn + o00);
    const uint4 v1 = *reinterpret_cast<const uint4 *>(xn + (vx ? o01 : o00));
    const uint4 v2 = *reinterpret_cast<const uint4 *>(xn + (vy ? o10 : o00));
    const uint4 v3 = *reinterpret_cast<const uint4 *>(xn + (vx && vy ? o11 : o00));
    const uint4 gv = *reinterpret_cast<const uint4 *>(dy + ((n * OH + oy) * OW + ox) * C + g * 8);
    float x0[8], x1[8], x2[8], x3[8], gy[8];
    unpack8(v0, x0);
    unpack8(v1, x1);
    unpack8(v2, x2);
    unpack8(v3, x3);
    unpack8(gv, gy);
    float r0[8], r1[8], r2[8], r3[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float a1 = vx ? x1[j] : -INFINITY, a2 = vy ? x2[j] : -INFINITY, a3 = (vx && vy) ? x3[j] : -INFINITY;
        // first maximum in (row, column) scan order: a later pixel wins only with a strictly larger value
        const bool w1 = a1 > x0[j];
        const float m1 = w1 ? a1 : x0[j];
        const bool w2 = a2 > m1;
        const float m2 = w2 ? a2 : m1;
        const bool w3 = a3 > m2;
        const float m = w3 ? a3 : m2;
        const float gval = (relu_mask && !(m > 0.f)) ? 0.f : gy[j];
        r0[j] = (w1 || w2 || w3) ? 0.f : gval;
        r1[j] = (w1 && !w2 && !w3) ? gval : 0.f;
        r2[j] = (w2 && !w3) ? gval : 0.f;
        r3[j] = w3 ? gval : 0.f;
    }
    *reinterpret_cast<uint4 *>(dxn + o00) = repack8(r0);
    if (vx) *reinterpret_cast<uint4 *>(dxn + o01) = repack8(r1);
    if (vy) *reinterpret_cast<uint4 *>(dxn + o10) = repack8(r2);
    if (vx && vy) *reinterpret_cast<uint4 *>(dxn + o11) = repack8(r3);
}

inline int grid_for(int64_t total) {
    int64_t g = cdiv(total, 256);
    if (g > 16384) g = 16384;
    if (g < 1) g = 1;
    return (int)g;
}
}  // namespace

extern "C" int fosvos_maxpool2x2_ceil_fwd(const uint16_t *x, uint16_t *y, int N, int H, int W, int C, int device,
                                          void *stream) {
    FOSVOS_REQUIRE(x && y, FOSVOS_E_ARG, "maxpool_fwd: null pointer");
    FOSVOS_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, FOSVOS_E_SHAPE,
                   "maxpool_fwd: bad shape N=%d H=%d W=%d C=%d (C %% 8 must be 0)", N, H, W, C);
    FOSVOS_ENTER(device);
    const int OH = (H + 1) / 2, OW = (W + 1) / 2;
    const int64_t total = (int64_t)N * OH * OW * (C / 8);
    FOSVOS_PROF("k_pool_fwd", stream, 0.0);
    hipLaunchKernelGGL(k_pool_fwd, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, y, H, W, C, OH, OW, total);
    FOSVOS_LAUNCH_CHECK();
    return FOSVOS_OK;
}

extern "C" int fosvos_maxpool2x2_ceil_bwd(const uint16_t *x, const uint16_t *dy, uint16_t *dx, int N, int H, int W,
                                          int C, int relu_mask, int device, void *stream) {
    FOSVOS_REQUIRE(x && dy && dx, FOSVOS_E_ARG, "maxpool_bwd: null pointer");
    FOSVOS_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, FOSVOS_E_SHAPE,
                   "maxpool_bwd: bad shape N=%d H=%d W=%d C=%d (C %% 8 must be 0)", N, H, W, C);
    FOSVOS_ENTER(device);
    const int OH = (H + 1) / 2, OW = (W + 1) / 2;
    FOSVOS_REQUIRE((int64_t)H * W * C < 0x7fffffffLL && OH <= 65535 && N <= 65535, FOSVOS_E_SHAPE,
                   "maxpool_bwd: image too large (N=%d H=%d W=%d C=%d)", N, H, W, C);
    FOSVOS_PROF("k_pool_bwd", stream, 0.0);
    hipLaunchKernelGGL(k_pool_bwd, dim3((unsigned)cdiv((int64_t)OW * (C / 8), 256), (unsigned)OH, (unsigned)N), dim3(256), 0,
                       (hipStream_t)stream, x, dy, dx, H, W, C, OH, OW, relu_mask);
    FOSVOS_LAUNCH_CHECK();
    return FOSVOS_OK;
}
