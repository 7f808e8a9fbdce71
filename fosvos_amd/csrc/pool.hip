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

__global__ __launch_bounds__(256) void k_pool_bwd(const uint16_t *__restrict__ x, const uint16_t *__restrict__ dy,
                                                   uint16_t *__restrict__ dx, int H, int W, int C, int OH, int OW,
                                                   int relu_mask, int64_t total) {
    const int groups = C >> 3;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int g = (int)(i % groups);
        int64_t r = i / groups;
        const int ox = (int)(r % OW);
        r /= OW;
        const int oy = (int)(r % OH);
        const int64_t n = r / OH;
        const int iy = oy * 2, ix = ox * 2;
        float xin[4][8];
        bool valid[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int yy = iy + (t >> 1), xx = ix + (t & 1);
            valid[t] = yy < H && xx < W;
            if (valid[t]) {
                const uint4 v = *reinterpret_cast<const uint4 *>(x + (((n * H + yy) * W + xx) * C + g * 8));
                unpack8(v, xin[t]);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) xin[t][j] = -INFINITY;
            }
        }
        const uint4 gv = *reinterpret_cast<const uint4 *>(dy + (((n * OH + oy) * OW + ox) * C + g * 8));
        float gy[8];
        unpack8(gv, gy);
        float out[4][8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            // first maximum in (row, column) scan order: strict '>' keeps the earliest
            int arg = 0;
            float m = xin[0][j];
#pragma unroll
            for (int t = 1; t < 4; ++t)
                if (xin[t][j] > m) {
                    m = xin[t][j];
                    arg = t;
                }
            const float gval = (relu_mask && !(m > 0.f)) ? 0.f : gy[j];
#pragma unroll
            for (int t = 0; t < 4; ++t) out[t][j] = (t == arg) ? gval : 0.f;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (valid[t]) {
                const int yy = iy + (t >> 1), xx = ix + (t & 1);
                *reinterpret_cast<uint4 *>(dx + (((n * H + yy) * W + xx) * C + g * 8)) = pack8(out[t]);
            }
        }
    }
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
    const int64_t total = (int64_t)N * OH * OW * (C / 8);
    FOSVOS_PROF("k_pool_bwd", stream, 0.0);
    hipLaunchKernelGGL(k_pool_bwd, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, H, W, C, OH, OW,
                       relu_mask, total);
    FOSVOS_LAUNCH_CHECK();
    return FOSVOS_OK;
}
