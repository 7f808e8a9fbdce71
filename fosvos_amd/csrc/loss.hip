// Class-balanced BCE-with-logits: loss + gradient, fused (reference: src/layers/osvos_layers.py:17-44).
//
// HBM-bound.  Algorithmic bytes per pixel: pass 1 reads the label (4 B); pass 2 reads logit and
// label (8 B) and writes the gradient (4 B) = 16 B/pixel.  Three launches:
//   k_count    per-block positive counts            -> ws.count[block]
//   k_loss     every block re-sums the counts in index order (so all agree bit-for-bit), then
//              writes grad and per-block fp64 partial sums of the positive / negative losses
//   k_finish   one wave sums the partials in index order and writes the fp32 loss
// All sums are fixed-order: results are bitwise reproducible run to run.
#include "common.hpp"

using namespace fosvos;

namespace {
constexpr int kBlock = 256;
constexpr int kMaxBlocks = 1024;
constexpr int kPerThread = 4;

struct Ws {
    unsigned long long count[kMaxBlocks];
    double pos[kMaxBlocks];
    double neg[kMaxBlocks];
};

__device__ __forceinline__ void load4(const float *__restrict__ p, int64_t i, int64_t n, float (&v)[4], float fill) {
    if (i + 3 < n) {
        const float4 t = *reinterpret_cast<const float4 *>(p + i);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (i + j < n) ? p[i + j] : fill;
    }
}

// blockIdx.y = frame: every frame of a batch is its own loss (own class counts, own workspace record, own output)
__global__ __launch_bounds__(kBlock) void k_count(const float *__restrict__ label, int64_t n, Ws *ws) {
    label += (int64_t)blockIdx.y * n;
    ws += blockIdx.y;
    unsigned cnt = 0;
    const int64_t stride = (int64_t)gridDim.x * kBlock * kPerThread;
    for (int64_t i = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * kPerThread; i < n; i += stride) {
        float y[4];
        load4(label, i, n, y, 0.f);
#pragma unroll
        for (int j = 0; j < 4; ++j) cnt += (y[j] >= 0.5f) ? 1u : 0u;
    }
    __shared__ unsigned s[kBlock / 64];
    unsigned w = cnt;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) w += __shfl_xor(w, o, 64);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0) ws->count[blockIdx.x] = (unsigned long long)s[0] + s[1] + s[2] + s[3];
}

__global__ __launch_bounds__(kBlock) void k_loss(const float *__restrict__ x, const float *__restrict__ label,
                                                  int64_t n, int size_average, float grad_scale,
                                                  float *__restrict__ grad, Ws *ws, int n_count_blocks,
                                                  const double *__restrict__ ext_counts) {
    x += (int64_t)blockIdx.y * n;
    label += (int64_t)blockIdx.y * n;
    if (grad) grad += (int64_t)blockIdx.y * n;
    ws += blockIdx.y;
    __shared__ double s_pos[kBlock / 64], s_neg[kBlock / 64];
    __shared__ unsigned long long s_np;
    if (!ext_counts) {   // integer sum of the per-block counts: order-independent, so every block agrees exactly
        __shared__ unsigned long long s_c[kBlock / 64];
        unsigned long long c = 0;
        for (int b = threadIdx.x; b < n_count_blocks; b += kBlock) c += ws->count[b];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
        if ((threadIdx.x & 63) == 0) s_c[threadIdx.x >> 6] = c;
        __syncthreads();
        if (threadIdx.x == 0) s_np = s_c[0] + s_c[1] + s_c[2] + s_c[3];
        __syncthreads();
    }
    // ext_counts (data-parallel batches): {positives, pixels} of the WHOLE batch, counted over all ranks
    const double n_tot = ext_counts ? ext_counts[1] : (double)n;
    const double n_pos = ext_counts ? ext_counts[0] : (double)s_np;
    const double n_neg = n_tot - n_pos;
    double gscale = (double)grad_scale;
    if (size_average) gscale /= n_tot;
    const float w_pos = (float)(n_neg / n_tot * gscale);  // weight of a positive pixel
    const float w_neg = (float)(n_pos / n_tot * gscale);

    double pos = 0.0, neg = 0.0;
    const int64_t stride = (int64_t)gridDim.x * kBlock * kPerThread;
    for (int64_t i = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * kPerThread; i < n; i += stride) {
        float xv[4], yv[4], g[4];
        load4(x, i, n, xv, 0.f);
        load4(label, i, n, yv, 0.f);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool y = yv[j] >= 0.5f;
            const float xx = xv[j];
            const float e = expf(-fabsf(xx));          // in (0,1]
            const float l = fmaxf(xx, 0.f) - (y ? xx : 0.f) + log1pf(e);
            const float inv = 1.f / (1.f + e);
            const float sig = xx >= 0.f ? inv : e * inv;
            g[j] = y ? w_pos * (sig - 1.f) : w_neg * sig;
            if (i + j < n) {
                if (y) pos += (double)l; else neg += (double)l;
            }
        }
        if (grad) {
            if (i + 3 < n) {
                *reinterpret_cast<float4 *>(grad + i) = make_float4(g[0], g[1], g[2], g[3]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (i + j < n) grad[i + j] = g[j];
            }
        }
    }
    pos = wave_sum(pos);
    neg = wave_sum(neg);
    if ((threadIdx.x & 63) == 0) {
        s_pos[threadIdx.x >> 6] = pos;
        s_neg[threadIdx.x >> 6] = neg;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        ws->pos[blockIdx.x] = (s_pos[0] + s_pos[1]) + (s_pos[2] + s_pos[3]);
        ws->neg[blockIdx.x] = (s_neg[0] + s_neg[1]) + (s_neg[2] + s_neg[3]);
    }
}

__global__ __launch_bounds__(64) void k_finish(int64_t n, int size_average, const Ws *ws, int n_blocks,
                                                float *__restrict__ loss_out, const double *__restrict__ ext_counts) {
    ws += blockIdx.x;
    loss_out += blockIdx.x;
    // lane t sums entries t, t+64, ... then a fixed butterfly: the order never changes run to run
    unsigned long long np = 0;
    double pos = 0.0, neg = 0.0;
    for (int b0 = threadIdx.x; b0 < n_blocks; b0 += 64 * 8) {  // 8 entries in flight (same order of additions)
        unsigned long long c[8];
        double p[8], q[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int b = min(b0 + 64 * j, n_blocks - 1);
            c[j] = ext_counts ? 0ull : ws->count[b];
            p[j] = ws->pos[b];
            q[j] = ws->neg[b];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (b0 + 64 * j < n_blocks) {
                np += c[j];
                pos += p[j];
                neg += q[j];
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) np += __shfl_xor(np, o, 64);
    pos = wave_sum(pos);
    neg = wave_sum(neg);
    if (threadIdx.x != 0) return;
    const double n_tot = ext_counts ? ext_counts[1] : (double)n;
    const double n_pos = ext_counts ? ext_counts[0] : (double)np, n_neg = n_tot - n_pos;
    double loss = n_neg / n_tot * pos + n_pos / n_tot * neg;
    if (size_average) loss /= n_tot;
    *loss_out = (float)loss;
}
}  // namespace

extern "C" size_t fosvos_cbce_workspace_bytes(int64_t) { return sizeof(Ws); }

namespace {
// parts: which of the three launches run (FOSVOS_CBCE_COUNT | _LOSS | _FINISH; the one-call entry points pass all three)
int cbce_impl(const float *logits, const float *label, int64_t numel, int n_frames, int size_average, float grad_scale,
              const double *batch_counts, float *loss_out, float *grad, void *workspace, size_t workspace_bytes,
              int device, void *stream, int parts = FOSVOS_CBCE_COUNT | FOSVOS_CBCE_LOSS | FOSVOS_CBCE_FINISH) {
    FOSVOS_REQUIRE(parts > 0 && parts <= 7, FOSVOS_E_ARG, "cbce_loss: parts=%d", parts);
    FOSVOS_REQUIRE(workspace && (label || !(parts & (FOSVOS_CBCE_COUNT | FOSVOS_CBCE_LOSS))) &&
                       (logits || !(parts & FOSVOS_CBCE_LOSS)) && (loss_out || !(parts & FOSVOS_CBCE_FINISH)),
                   FOSVOS_E_ARG, "cbce_loss: null pointer");
    FOSVOS_REQUIRE(numel > 0, FOSVOS_E_SHAPE, "cbce_loss: numel=%lld", (long long)numel);
    FOSVOS_REQUIRE(n_frames >= 1 && n_frames <= 65535, FOSVOS_E_SHAPE, "cbce_loss: n_frames=%d", n_frames);
    FOSVOS_REQUIRE(n_frames == 1 || numel % 4 == 0, FOSVOS_E_SHAPE,
                   "cbce_loss_frames: %lld elements per frame - frames after the first would start off a 16-byte boundary",
                   (long long)numel);
    FOSVOS_REQUIRE(workspace_bytes >= n_frames * sizeof(Ws), FOSVOS_E_WORKSPACE, "cbce_loss: workspace %zu < %zu",
                   workspace_bytes, n_frames * sizeof(Ws));
    FOSVOS_REQUIRE((!logits || (uintptr_t)logits % 16 == 0) && (!label || (uintptr_t)label % 16 == 0) &&
                       (!grad || (uintptr_t)grad % 16 == 0),
                   FOSVOS_E_ARG, "cbce_loss: pointers must be 16-byte aligned");
    FOSVOS_ENTER(device);
    int blocks = (int)cdiv(numel, (int64_t)kBlock * kPerThread);
    if (blocks > kMaxBlocks) blocks = kMaxBlocks;
    Ws *ws = reinterpret_cast<Ws *>(workspace);
    hipStream_t s = (hipStream_t)stream;
    if (!batch_counts && (parts & FOSVOS_CBCE_COUNT)) {
        FOSVOS_PROF("k_count", s, 0.0);
        hipLaunchKernelGGL(k_count, dim3(blocks, n_frames), dim3(kBlock), 0, s, label, numel, ws);
        FOSVOS_LAUNCH_CHECK();
    }
    if (parts & FOSVOS_CBCE_LOSS) {
        FOSVOS_PROF("k_loss", s, 0.0);
        hipLaunchKernelGGL(k_loss, dim3(blocks, n_frames), dim3(kBlock), 0, s, logits, label, numel, size_average, grad_scale,
                           grad, ws, blocks, batch_counts);
        FOSVOS_LAUNCH_CHECK();
    }
    if (parts & FOSVOS_CBCE_FINISH) {
        FOSVOS_PROF("k_finish", s, 0.0);
        hipLaunchKernelGGL(k_finish, dim3(n_frames), dim3(64), 0, s, numel, size_average, ws, blocks, loss_out, batch_counts);
        FOSVOS_LAUNCH_CHECK();
    }
    return FOSVOS_OK;
}
}  // namespace

extern "C" int fosvos_cbce_loss(const float *logits, const float *label, int64_t numel, int size_average,
                                float grad_scale, float *loss_out, float *grad, void *workspace,
                                size_t workspace_bytes, int device, void *stream) {
    return cbce_impl(logits, label, numel, 1, size_average, grad_scale, nullptr, loss_out, grad, workspace, workspace_bytes,
                     device, stream);
}

extern "C" int fosvos_cbce_loss_frames(const float *logits, const float *label, int64_t frame_numel, int n_frames,
                                       int size_average, float grad_scale, float *loss_out, float *grad, void *workspace,
                                       size_t workspace_bytes, int device, void *stream) {
    return cbce_impl(logits, label, frame_numel, n_frames, size_average, grad_scale, nullptr, loss_out, grad, workspace,
                     workspace_bytes, device, stream);
}

extern "C" int fosvos_cbce_loss_frames_parts(const float *logits, const float *label, int64_t frame_numel, int n_frames,
                                             int size_average, float grad_scale, float *loss_out, float *grad,
                                             void *workspace, size_t workspace_bytes, int parts, int device, void *stream) {
    return cbce_impl(logits, label, frame_numel, n_frames, size_average, grad_scale, nullptr, loss_out, grad, workspace,
                     workspace_bytes, device, stream, parts);
}

extern "C" int fosvos_cbce_loss_batch_counts(const float *logits, const float *label, int64_t numel, int size_average,
                                             float grad_scale, const double *batch_counts, float *loss_out,
                                             float *grad, void *workspace, size_t workspace_bytes, int device,
                                             void *stream) {
    FOSVOS_REQUIRE(batch_counts, FOSVOS_E_ARG, "cbce_loss_batch_counts: null batch_counts");
    return cbce_impl(logits, label, numel, 1, size_average, grad_scale, batch_counts, loss_out, grad, workspace,
                     workspace_bytes, device, stream);
}
