// conv1_1 forward: Conv2d(3 -> Co, 3x3, pad 1) + bias + ReLU straight from the fp32 NCHW frame
// (reference: stages[0][0..1], src/networks/osvos_vgg.py:92-93).
//
// K = 27 is far too small for MFMA to pay in the forward direction: fp32 VALU, bandwidth-leaning (reads
// 12 B/pixel of frame, writes 2*Co B/pixel of bf16 NHWC activation).  Its weight gradient reduces over 410 k
// pixels and lives in conv_wgrad.hip (MFMA, 16-channel padded image).  No dgrad: the image needs none.
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
    // staging: all of a thread's global loads are issued before its first LDS store (unconditional loads from
    // clamped addresses); rolled, each element waited out its own memory round trip (7 + 3 in a row per thread)
    {
        constexpr int WN = 27 * CO, WIT = (WN + 255) / 256, IN = 3 * (ROWS + 2) * (TW + 8), IIT = (IN + 255) / 256;
        float tw[WIT], ti[IIT];
        bool oki[IIT];
        const int64_t plane = (int64_t)H * W;
#pragma unroll
        for (int it = 0; it < WIT; ++it) {
            const int i = min(it * 256 + tid, WN - 1);
            tw[it] = w[(i % CO) * 27 + i / CO];  // s_w[k = ci*9 + tap][co]
        }
#pragma unroll
        for (int it = 0; it < IIT; ++it) {
            const int i = min(it * 256 + tid, IN - 1);
            const int xx = i % (TW + 8);
            const int r = (i / (TW + 8)) % (ROWS + 2);
            const int c = i / ((TW + 8) * (ROWS + 2));
            const int gy = y0 + r - 1, gx = x0 + xx - 4;
            oki[it] = gy >= 0 && gy < H && gx >= 0 && gx < W;
            ti[it] = frame[((int64_t)n * 3 + c) * plane + (int64_t)min(max(gy, 0), H - 1) * W + min(max(gx, 0), W - 1)];
        }
#pragma unroll
        for (int it = 0; it < WIT; ++it) {
            const int i = it * 256 + tid;
            if (i < WN) s_w[i / CO][i % CO] = tw[it];
        }
#pragma unroll
        for (int it = 0; it < IIT; ++it) {
            const int i = it * 256 + tid;
            if (i < IN) (&s_in[0][0][0])[i] = oki[it] ? ti[it] : 0.f;
        }
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
    // the (c, ky) loops stay rolled: fully unrolled, hipcc hoists all 27 weight rows and every input row into
    // registers (256 VGPRs + AGPR spills, one wave per SIMD)
#pragma unroll 1
    for (int c = 0; c < 3; ++c) {
#pragma unroll 1
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
            for (int j = 0; j < 8; ++j) o[j] = relu_f(acc[p][j]);
            *reinterpret_cast<uint4 *>(y + (((int64_t)n * H + gy) * W + gx) * CO + cg * 8) = pack8(o);
        }
    }
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
    FOSVOS_PROF("k_first_fwd", stream, 2.0 * N * H * W * 27 * Co);
    hipLaunchKernelGGL(k_first_fwd, grid, dim3(256), 0, (hipStream_t)stream, frame, w, bias, y, H, W);
    FOSVOS_LAUNCH_CHECK();
    return FOSVOS_OK;
}

