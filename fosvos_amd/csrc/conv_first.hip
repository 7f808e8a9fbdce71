// conv1_1: Conv2d(3 -> Co, 3x3, pad 1) + bias + ReLU straight from the fp32 NCHW frame, and its
// weight/bias gradient (reference: stages[0][0..1], src/networks/osvos_vgg.py:92-93).
//
// K = 27 is far too small for MFMA to pay; both kernels are fp32 VALU and bandwidth-leaning:
//   fwd   reads 12 B/pixel of frame, writes 2*Co B/pixel of bf16 NHWC activation
//   wgrad reads 2*Co B/pixel of dy (+ the frame through the scalar cache), writes Co*28 floats/block
// No dgrad: the image needs no gradient.
#include "common.hpp"

using namespace fosvos;

namespace {
constexpr int CO = 64;     // the only instantiation the model needs (checked at the entry point)
constexpr int PX = 4;      // consecutive pixels per thread
constexpr int TW = 8 * PX; // pixels per wave (one image-row segment)
constexpr int ROWS = 4;    // waves per block = image rows per block

// ---------------------------------------------------------------------------------------- forward
// Block = 4 waves = 4 image rows x 32 pixels.  Lane l of a wave owns pixels 4*(l/8)..+3 of the
// segment and output channels 8*(l%8)..+7, so the 8 lanes of a pixel store one full 128-byte
// NHWC pixel.  Weights live in LDS as [27][64] fp32 (each lane reads its 8 channels as 2 x 16 B:
// the 8 channel groups cover one 256-byte bank row, conflict-free); the 6 x 66 input halo rows
// per channel live in LDS too and are read as broadcasts within a pixel group.
__global__ __launch_bounds__(256) void k_first_fwd(const float *__restrict__ frame, const float *__restrict__ w,
                                                    const float *__restrict__ bias, uint16_t *__restrict__ y, int H,
                                                    int W) {
    __shared__ __attribute__((aligned(16))) float s_w[27][CO];
    __shared__ __attribute__((aligned(16))) float s_in[3][ROWS + 2][TW + 8];  // x index 0 <-> image x0-4 (16-B aligned)
    const int n = blockIdx.z;
    const int y0 = blockIdx.y * ROWS;
    const int x0 = blockIdx.x * TW;
    const int tid = threadIdx.x;
    for (int i = tid; i < 27 * CO; i += 256) {
        const int co = i % CO, k = i / CO;  // k = ci*9 + tap
        s_w[k][co] = w[co * 27 + k];
    }
    const int64_t plane = (int64_t)H * W;
    for (int i = tid; i < 3 * (ROWS + 2) * (TW + 8); i += 256) {
        const int xx = i % (TW + 8);
        const int r = (i / (TW + 8)) % (ROWS + 2);
        const int c = i / ((TW + 8) * (ROWS + 2));
        const int gy = y0 + r - 1, gx = x0 + xx - 4;
        float v = 0.f;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = frame[((int64_t)n * 3 + c) * plane + (int64_t)gy * W + gx];
        s_in[c][r][xx] = v;
    }
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63;
    const int pg = lane >> 3, cg = lane & 7;
    const int gy = y0 + wave;
    if (gy >= H) return;
    float acc[PX][8];
    {
        const float4 b0 = *reinterpret_cast<const float4 *>(bias + cg * 8);
        const float4 b1 = *reinterpret_cast<const float4 *>(bias + cg * 8 + 4);
#pragma unroll
        for (int p = 0; p < PX; ++p) {
            acc[p][0] = b0.x; acc[p][1] = b0.y; acc[p][2] = b0.z; acc[p][3] = b0.w;
            acc[p][4] = b1.x; acc[p][5] = b1.y; acc[p][6] = b1.z; acc[p][7] = b1.w;
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            // inputs x = 4*pg-1 .. 4*pg+4 of halo row (wave + ky): LDS x index = 4*pg + 3 .. 4*pg + 8
            float in[PX + 2];
            const float *row = &s_in[c][wave + ky][pg * 4];
            const float4 a = *reinterpret_cast<const float4 *>(row);       // idx 0..3
            const float4 b = *reinterpret_cast<const float4 *>(row + 4);   // idx 4..7
            const float e = row[8];
            in[0] = a.w; in[1] = b.x; in[2] = b.y; in[3] = b.z; in[4] = b.w; in[5] = e;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int k = c * 9 + ky * 3 + kx;
                const float4 w0 = *reinterpret_cast<const float4 *>(&s_w[k][cg * 8]);
                const float4 w1 = *reinterpret_cast<const float4 *>(&s_w[k][cg * 8 + 4]);
#pragma unroll
                for (int p = 0; p < PX; ++p) {
                    const float v = in[p + kx];
                    acc[p][0] += v * w0.x; acc[p][1] += v * w0.y; acc[p][2] += v * w0.z; acc[p][3] += v * w0.w;
                    acc[p][4] += v * w1.x; acc[p][5] += v * w1.y; acc[p][6] += v * w1.z; acc[p][7] += v * w1.w;
                }
            }
        }
    }
#pragma unroll
    for (int p = 0; p < PX; ++p) {
        const int gx = x0 + pg * PX + p;
        if (gx < W) {
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = fmaxf(acc[p][j], 0.f);
            *reinterpret_cast<uint4 *>(y + (((int64_t)n * H + gy) * W + gx) * CO + cg * 8) = pack8(o);
        }
    }
}

// ---------------------------------------------------------------------------------------- wgrad
// dw[co][ci][tap] = sum_px dy[px][co] * in[ci][px + tap];  db[co] = sum_px dy[px][co].
// Block = 4 waves = 8 image rows x 256 pixels; wave q sweeps its own 64 columns.  Lane = output
// channel, so dy loads are 128 contiguous bytes per pixel; the frame tile sits in LDS and the input
// values of a 4-pixel group are wave-uniform broadcast reads (27 ds_read_b128-class reads per 108 FMAs).
// Each block writes one slab of Co*28 floats; k_first_reduce sums the slabs in index order.
constexpr int WG_PIX = 256;
constexpr int WG_ROWS = 8;  // image rows swept by one block

__global__ __launch_bounds__(256) void k_first_wgrad(const float *__restrict__ frame, const uint16_t *__restrict__ dy,
                                                      float *__restrict__ slabs, int H, int W) {
    // frame tile with a one-pixel halo, zero outside the image; LDS x index 0 <-> image x0-4 so that the
    // 4-pixel groups below are 16-byte aligned
    __shared__ __attribute__((aligned(16))) float s_in[3][WG_ROWS + 2][WG_PIX + 8];
    __shared__ float s_red[4][28][CO];
    const int n = blockIdx.z;
    const int q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int co = threadIdx.x & 63;
    const int x0 = blockIdx.x * WG_PIX, y0 = blockIdx.y * WG_ROWS;
    const int64_t plane = (int64_t)H * W;
    const float *fr = frame + (int64_t)n * 3 * plane;
    for (int i = threadIdx.x; i < 3 * (WG_ROWS + 2) * (WG_PIX + 8); i += 256) {
        const int xx = i % (WG_PIX + 8);
        const int r = (i / (WG_PIX + 8)) % (WG_ROWS + 2);
        const int c = i / ((WG_PIX + 8) * (WG_ROWS + 2));
        const int gy = y0 + r - 1, gx = x0 + xx - 4;
        float v = 0.f;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = fr[c * plane + (int64_t)gy * W + gx];
        s_in[c][r][xx] = v;
    }
    __syncthreads();
    float acc[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) acc[k] = 0.f;
    float accb = 0.f;
    const int rows = min(WG_ROWS, H - y0);
    for (int r = 0; r < rows; ++r) {
        const uint16_t *dyr = dy + (((int64_t)n * H + y0 + r) * W) * CO + co;
        for (int xb = 0; xb < 16; ++xb) {
            const int xl = q * 64 + xb * 4;  // tile-local x of the first of 4 pixels
            const int gx = x0 + xl;
            if (gx >= W) break;
            float g[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) g[p] = (gx + p < W) ? bf2f(dyr[(int64_t)(gx + p) * CO]) : 0.f;
            accb += (g[0] + g[1]) + (g[2] + g[3]);
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    // wave-uniform address: LDS broadcast reads of inputs x-1 .. x+4
                    const float *row = &s_in[c][r + ky][xl];
                    const float4 a = *reinterpret_cast<const float4 *>(row);
                    const float4 b = *reinterpret_cast<const float4 *>(row + 4);
                    const float e = row[8];
                    const float in[6] = {a.w, b.x, b.y, b.z, b.w, e};
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        float t = acc[c * 9 + ky * 3 + kx];
#pragma unroll
                        for (int p = 0; p < 4; ++p) t += g[p] * in[p + kx];
                        acc[c * 9 + ky * 3 + kx] = t;
                    }
                }
        }
    }
    // combine the 4 waves through LDS in wave order (fixed order => deterministic)
#pragma unroll
    for (int k = 0; k < 27; ++k) s_red[q][k][co] = acc[k];
    s_red[q][27][co] = accb;
    __syncthreads();
    const int blk = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    for (int i = threadIdx.x; i < 28 * CO; i += 256) {
        const int k = i / CO, c = i % CO;
        slabs[(int64_t)blk * 28 * CO + i] = (s_red[0][k][c] + s_red[1][k][c]) + (s_red[2][k][c] + s_red[3][k][c]);
    }
}

// out[co*27 + k] = sum over slabs of slab[k][co]; bias = row 27.  Block = 64 outputs x 4 slab groups; group
// g sums slabs g, g+4, ... in order, then the 4 group sums are added in order (deterministic).
__global__ __launch_bounds__(256) void k_first_reduce(const float *__restrict__ slabs, int n_slabs,
                                                       float *__restrict__ dw, float *__restrict__ db) {
    __shared__ float red[4][64];
    const int il = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + il;  // 28*CO outputs, CO == 64: one block per k
    float a = 0.f;
    for (int s = g; s < n_slabs; s += 4) a += slabs[(int64_t)s * 28 * CO + i];
    red[g][il] = a;
    __syncthreads();
    if (g != 0) return;
    const float v = (red[0][il] + red[1][il]) + (red[2][il] + red[3][il]);
    const int k = i / CO, c = i % CO;
    if (k < 27)
        dw[c * 27 + k] = v;
    else if (db)
        db[c] = v;
}
}  // namespace

extern "C" int fosvos_conv3x3_first_fwd(const float *frame, const float *w, const float *bias, uint16_t *y, int N,
                                        int H, int W, int Co, int device, void *stream) {
    FOSVOS_REQUIRE(frame && w && bias && y, FOSVOS_E_ARG, "conv3x3_first_fwd: null pointer");
    FOSVOS_REQUIRE(Co == CO, FOSVOS_E_SHAPE, "conv3x3_first_fwd: Co=%d, only %d is built", Co, CO);
    FOSVOS_REQUIRE(N > 0 && H > 0 && W > 0 && N <= 65535, FOSVOS_E_SHAPE, "conv3x3_first_fwd: bad shape N=%d H=%d W=%d",
                   N, H, W);
    FOSVOS_ENTER(device);
    dim3 grid((unsigned)cdiv(W, TW), (unsigned)cdiv(H, ROWS), (unsigned)N);
    hipLaunchKernelGGL(k_first_fwd, grid, dim3(256), 0, (hipStream_t)stream, frame, w, bias, y, H, W);
    FOSVOS_LAUNCH_CHECK();
    return FOSVOS_OK;
}

extern "C" size_t fosvos_conv3x3_first_wgrad_workspace_bytes(int N, int H, int W, int Co) {
    return (size_t)N * cdiv(H, WG_ROWS) * cdiv(W, WG_PIX) * 28 * (size_t)Co * sizeof(float);
}

extern "C" int fosvos_conv3x3_first_wgrad(const float *frame, const uint16_t *dy, float *dw, float *db, int N, int H,
                                          int W, int Co, void *workspace, size_t workspace_bytes, int device,
                                          void *stream) {
    FOSVOS_REQUIRE(frame && dy && dw && workspace, FOSVOS_E_ARG, "conv3x3_first_wgrad: null pointer");
    FOSVOS_REQUIRE(Co == CO, FOSVOS_E_SHAPE, "conv3x3_first_wgrad: Co=%d, only %d is built", Co, CO);
    FOSVOS_REQUIRE(N > 0 && H > 0 && W > 0 && N <= 65535 && H <= 65535, FOSVOS_E_SHAPE,
                   "conv3x3_first_wgrad: bad shape N=%d H=%d W=%d", N, H, W);
    const size_t need = fosvos_conv3x3_first_wgrad_workspace_bytes(N, H, W, Co);
    FOSVOS_REQUIRE(workspace_bytes >= need, FOSVOS_E_WORKSPACE, "conv3x3_first_wgrad: workspace %zu < %zu",
                   workspace_bytes, need);
    FOSVOS_ENTER(device);
    dim3 grid((unsigned)cdiv(W, WG_PIX), (unsigned)cdiv(H, WG_ROWS), (unsigned)N);
    float *slabs = reinterpret_cast<float *>(workspace);
    hipLaunchKernelGGL(k_first_wgrad, grid, dim3(256), 0, (hipStream_t)stream, frame, dy, slabs, H, W);
    FOSVOS_LAUNCH_CHECK();
    const int n_slabs = (int)(grid.x * grid.y * grid.z);
    hipLaunchKernelGGL(k_first_reduce, dim3(28), dim3(256), 0, (hipStream_t)stream, slabs,
                       n_slabs, dw, db);
    FOSVOS_LAUNCH_CHECK();
    return FOSVOS_OK;
}
