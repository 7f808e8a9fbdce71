// Multi-tensor SGD with momentum, torch.optim.SGD semantics (reference: optim.SGD built with the
// param groups of src/util/network_provider.py:144-159 / 98-125).
//
// HBM-bound: per element read p, g, buf and write p, buf = 20 B.  One launch covers every tensor:
// blockIdx.y picks the tensor (its record is read through the scalar path), blockIdx.x grid-strides
// over its elements with 16-byte vectors.
#include "common.hpp"

using namespace fosvos;

namespace {
__global__ __launch_bounds__(256) void k_sgd(const fosvos_sgd_entry *__restrict__ table, float momentum,
                                              int flags) {
    const bool first_step = flags & FOSVOS_SGD_FIRST_STEP, zero = flags & FOSVOS_SGD_ZERO_GRAD;
    const fosvos_sgd_entry e = table[blockIdx.y];
    if (e.grad == nullptr || e.numel <= 0) return;
    float *__restrict__ p = e.param;
    float *__restrict__ g = const_cast<float *>(e.grad);
    float *__restrict__ m = e.momentum_buf;
    const float lr = e.lr, wd = e.weight_decay;
    const int64_t n = e.numel;
    const bool vec = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m) & 15) == 0;
    const int64_t n4 = vec ? (n >> 2) : 0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 pv = reinterpret_cast<float4 *>(p)[i];
        const float4 gv = reinterpret_cast<const float4 *>(g)[i];
        float4 mv = first_step ? make_float4(0, 0, 0, 0) : reinterpret_cast<float4 *>(m)[i];
        float pp[4] = {pv.x, pv.y, pv.z, pv.w}, gg[4] = {gv.x, gv.y, gv.z, gv.w}, mm[4] = {mv.x, mv.y, mv.z, mv.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float d = wd != 0.f ? gg[j] + wd * pp[j] : gg[j];
            mm[j] = first_step ? d : momentum * mm[j] + d;
            pp[j] = pp[j] - lr * mm[j];
        }
        reinterpret_cast<float4 *>(p)[i] = make_float4(pp[0], pp[1], pp[2], pp[3]);
        reinterpret_cast<float4 *>(m)[i] = make_float4(mm[0], mm[1], mm[2], mm[3]);
        if (zero) reinterpret_cast<float4 *>(g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int64_t i = n4 * 4 + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const float d = wd != 0.f ? g[i] + wd * p[i] : g[i];
        const float mm = first_step ? d : momentum * m[i] + d;
        m[i] = mm;
        p[i] = p[i] - lr * mm;
        if (zero) g[i] = 0.f;
    }
}
}  // namespace

extern "C" int fosvos_sgd_momentum_step(const fosvos_sgd_entry *table, int n_tensors, int64_t max_numel, float momentum,
                                        int first_step, int device, void *stream) {
    FOSVOS_REQUIRE(table, FOSVOS_E_ARG, "sgd_momentum_step: null table");
    FOSVOS_REQUIRE(n_tensors > 0 && n_tensors <= 65535 && max_numel > 0, FOSVOS_E_SHAPE,
                   "sgd_momentum_step: n_tensors=%d max_numel=%lld", n_tensors, (long long)max_numel);
    FOSVOS_ENTER(device);
    int64_t gx = cdiv(max_numel, 256 * 4);
    if (gx > 512) gx = 512;
    FOSVOS_PROF("k_sgd", stream, 0.0);
    hipLaunchKernelGGL(k_sgd, dim3((unsigned)gx, (unsigned)n_tensors), dim3(256), 0, (hipStream_t)stream, table,
                       momentum, first_step);
    FOSVOS_LAUNCH_CHECK();
    return FOSVOS_OK;
}
