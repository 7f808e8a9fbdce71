// Shared host/device helpers for libfosvos_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <algorithm>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/fosvos_hip.h"

namespace fosvos {

// ------------------------------------------------------------------------------------------ errors
extern thread_local char g_err[512];

inline int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
inline int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define FOSVOS_HIP_CHECK(expr)                                                                         \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return ::fosvos::fail(FOSVOS_E_HIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                                  __LINE__);                                                           \
    } while (0)

#define FOSVOS_REQUIRE(cond, code, ...)                    \
    do {                                                   \
        if (!(cond)) return ::fosvos::fail(code, __VA_ARGS__); \
    } while (0)

// Select the device for this call (backward runs on an autograd thread with its own current device).
#define FOSVOS_ENTER(device) FOSVOS_HIP_CHECK(hipSetDevice(device))
// Launch profiler (util.hip; fosvos_profile_start / _stop): when on, every kernel launch of the library is bracketed
// by a pair of timing events ON THE STREAM IT IS LAUNCHED ON.  FOSVOS_PROF(name, stream, flops) goes in front of a
// launch, FOSVOS_LAUNCH_CHECK() behind it closes the bracket.  Off (the default): one predictable branch per launch.
extern bool g_prof_on;
void prof_begin(const char *name, hipStream_t st, double flops);
void prof_end();
#define FOSVOS_PROF(name, st, flops)                                        \
    do {                                                                    \
        if (::fosvos::g_prof_on) ::fosvos::prof_begin(name, (hipStream_t)(st), flops); \
    } while (0)
#define FOSVOS_LAUNCH_CHECK()                            \
    do {                                                 \
        if (::fosvos::g_prof_on) ::fosvos::prof_end();   \
        FOSVOS_HIP_CHECK(hipGetLastError());             \
    } while (0)

// Weight-gradient reduction queue (conv_wgrad.hip): the MFMA kernel of a layer writes one fp32 slab per pixel split, laid
// out like dw itself; summing them into dw/db can run right away (reduce == nullptr) or be queued here and run for many
// layers in one launch (wgrad_reduce_all).  A queued layer's workspace must stay untouched until then.
struct WgradReduceEntry {
    const float *slabs;      // S slabs of E_pad floats ([Cor][Ci][9]); the first E_real of each are dw's elements
    float *dw, *db;          // db may be null
    const float *bias_part;  // [S][Cor]
    int64_t E_real, E_pad;
    int S, S_bias, Co, Cor, accumulate;  // S slabs, S_bias bias partials per channel
    int block_begin, n_blocks;  // block range inside the batched final launch (n_blocks cover the layer's elements once)
    int fold_begin, fold_blocks;  // ... inside the batched fold launch (0 blocks: no fold stage, S <= 8)
};
constexpr int kWgradReduceMax = 20;
struct WgradReduceTable {
    int n;
    WgradReduceEntry e[kWgradReduceMax];
};
int wgrad_impl(const uint16_t *x, const uint16_t *dy, float *dw, float *db, int N, int H, int W, int Ci, int Co,
               int accumulate, void *workspace, size_t workspace_bytes, int device, void *stream,
               WgradReduceTable *reduce, bool alone = false);
int first_wgrad_impl(const float *frame, const uint16_t *dy, float *dw, float *db, int N, int H, int W, int Co,
                     int accumulate, void *workspace, size_t workspace_bytes, int device, void *stream,
                     WgradReduceTable *reduce);
int wgrad_reduce_all(WgradReduceTable *reduce, int device, void *stream);
// queue (or, reduce == nullptr, run) the sum over S slabs of E_pad floats each into dw's E_real floats and db's Co
int wgrad_queue_reduce(const float *slabs, const float *bias_part, float *dw, float *db, int S, int S_bias, int64_t E_real,
                       int64_t E_pad, int Co, int Cor, int accumulate, WgradReduceTable *reduce, int device, void *stream);
// Head backward in pieces, so that vgg_net.hip can put the scales the data-gradient chain does not need yet on the
// auxiliary stream: check once, then one call per scale (any order, any stream) and the finish pass (after all four).
struct HeadBwdArgs {
    const float *const *side;
    const int *hs, *ws;
    const float *const *filt, *const *filt1;
    const float *dsn_w, *fuse_w, *d_fused;
    const float *const *d_side_out;
    uint16_t *const *d_side;
    float *d_fuse_w, *d_fuse_b, *d_dsn_w, *d_dsn_b;
    int N, H, W, accumulate;
    void *workspace;
    size_t workspace_bytes;
    int device;
    int filt_uniform;  // bit s: the 16 per-channel filters of scale s are identical (fosvos_head_fwd)
};
int head_bwd_check(const HeadBwdArgs &a);
int head_bwd_scale(const HeadBwdArgs &a, int s, void *stream);
int head_bwd_finish(const HeadBwdArgs &a, void *stream);

}  // namespace fosvos
// The caller-owned execution context of include/fosvos_hip.h: every event the multi-stream network calls order their
// streams with.  Created whole by fosvos_ctx_create (util.hip), never grown later.
constexpr int kFosvosVggEvents = 25, kFosvosResnetEvents = 12;
struct fosvos_ctx {
    uint32_t magic;                       // 'FCTX' while alive
    int device;
    hipEvent_t vgg_ev[kFosvosVggEvents];
    hipEvent_t resnet_ev[kFosvosResnetEvents];
    bool buckets_recorded;                // the last fosvos_vgg_backward on this context published its gradient buckets
};
constexpr uint32_t kFosvosCtxMagic = 0x58544346u;
namespace fosvos {
inline int ctx_check(const fosvos_ctx *ctx, const char *who) {
    FOSVOS_REQUIRE(ctx != nullptr, FOSVOS_E_ARG, "%s: null context (fosvos_ctx_create)", who);
    FOSVOS_REQUIRE(ctx->magic == kFosvosCtxMagic, FOSVOS_E_ARG, "%s: not a live fosvos_ctx", who);
    return FOSVOS_OK;
}

// Lab switches: FOSVOS_* environment variables steer A/B experiments in LAB builds only (make LAB=1 -> -DFOSVOS_LAB_BUILD).
// The shipped library reads no environment: every site below sees "unset" and takes its measured default, so the C ABI has
// no hidden inputs (no global mutable state).
#ifdef FOSVOS_LAB_BUILD
inline const char *lab_env(const char *name) { return getenv(name); }
#else
inline const char *lab_env(const char *) { return nullptr; }
#endif
inline int lab_env_int(const char *name, int dflt) {
    const char *e = lab_env(name);
    return e ? atoi(e) : dflt;
}

// Persistent eight-wave forward conv (conv_pp.hip): which launches take it, and the launch itself.
bool conv_pp_applicable(int N, int H, int W, int in_ch, int out_ch);
int conv_pp_workgroups();
int conv_pp_forward(const uint16_t *x, const uint16_t *w_packed, const float *bias, uint16_t *y, uint16_t *y_pool, int N,
                    int H, int W, int Cin_pad, int Cout, int Co_pad, bool relu, hipStream_t st,
                    const uint8_t *mask_bits = nullptr);

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline int roundup(int a, int b) { return (a + b - 1) / b * b; }

// ------------------------------------------------------------------------------------------ bf16
typedef uint16_t bf16_t;  // storage type
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef unsigned int u32x4 __attribute__((__vector_size__(16)));  // data operand of the buffer store builtins

__device__ __forceinline__ uint16_t f2bf(float f) {  // round-to-nearest-even, NaN stays NaN
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(uint16_t, b);
}
__device__ __forceinline__ float bf2f(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// eight fp32 values that ARE bf16 values (low 16 mantissa bits zero: unpacked bf16, or zero) back into packed pairs:
// one byte permute per pair, no rounding involved
__device__ __forceinline__ uint4 repack8(const float (&f)[8]) {
    uint4 v;
    v.x = __builtin_amdgcn_perm(__float_as_uint(f[1]), __float_as_uint(f[0]), 0x07060302u);
    v.y = __builtin_amdgcn_perm(__float_as_uint(f[3]), __float_as_uint(f[2]), 0x07060302u);
    v.z = __builtin_amdgcn_perm(__float_as_uint(f[5]), __float_as_uint(f[4]), 0x07060302u);
    v.w = __builtin_amdgcn_perm(__float_as_uint(f[7]), __float_as_uint(f[6]), 0x07060302u);
    return v;
}
// two floats -> two bf16 in one register: ONE v_cvt_pk_bf16_f32 (the scalar form costs two converts + shift + or)
typedef float fosvos_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 fosvos_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
    const fosvos_f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, fosvos_bf16x2));
}
// max(v, 0) as ONE instruction: a signed-integer max on the float's bits (negative floats are negative integers; -0 -> +0).
// fmaxf costs a canonicalising v_max per operand in front of the max itself.  A NaN with the sign bit clear passes through
// (as torch.relu would have it), one with the sign bit set becomes 0.
__device__ __forceinline__ float relu_f(float v) { return __int_as_float(max(__float_as_int(v), 0)); }
__device__ __forceinline__ void unpack8(const uint4 &v, float (&f)[8]) {
    f[0] = __uint_as_float(v.x << 16);
    f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16);
    f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16);
    f[5] = __uint_as_float(v.z & 0xffff0000u);
    f[6] = __uint_as_float(v.w << 16);
    f[7] = __uint_as_float(v.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
    uint4 v;
    v.x = pack2bf(f[0], f[1]);
    v.y = pack2bf(f[2], f[3]);
    v.z = pack2bf(f[4], f[5]);
    v.w = pack2bf(f[6], f[7]);
    return v;
}

// ---- packed bf16 pairs handled as 16-bit integers (v_pk_*_i16/u16: one instruction per two elements, no unpack / repack)
typedef short fosvos_i16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short fosvos_u16x2 __attribute__((ext_vector_type(2)));
// per element: f where m > 0, else +0.  A bf16 is > 0 exactly when its bits, read as int16, are > 0 (NaNs aside).
__device__ __forceinline__ uint32_t keep_where_pos_bf16x2(uint32_t f, uint32_t m) {
    const fosvos_i16x2 zero = {0, 0}, one = {1, 1};
    const fosvos_i16x2 t = __builtin_elementwise_min(
        __builtin_elementwise_max(__builtin_bit_cast(fosvos_i16x2, m), zero), one);  // 1 where m > 0
    const fosvos_u16x2 all = {0xffff, 0xffff};
    return f & __builtin_bit_cast(uint32_t, __builtin_bit_cast(fosvos_u16x2, t) * all);
}
__device__ __forceinline__ uint4 keep_where_pos_bf16x8(const uint4 &f, const uint4 &m) {
    return make_uint4(keep_where_pos_bf16x2(f.x, m.x), keep_where_pos_bf16x2(f.y, m.y),
                      keep_where_pos_bf16x2(f.z, m.z), keep_where_pos_bf16x2(f.w, m.w));
}
// per element max of NON-NEGATIVE bf16 values (post-ReLU maps): their bit patterns order like unsigned integers
__device__ __forceinline__ uint32_t max_nonneg_bf16x2(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(fosvos_u16x2, a),
                                                                 __builtin_bit_cast(fosvos_u16x2, b)));
}
__device__ __forceinline__ uint4 max_nonneg_bf16x8(const uint4 &a, const uint4 &b) {
    return make_uint4(max_nonneg_bf16x2(a.x, b.x), max_nonneg_bf16x2(a.y, b.y), max_nonneg_bf16x2(a.z, b.z),
                      max_nonneg_bf16x2(a.w, b.w));
}

// wave64 all-lanes sum
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

}  // namespace fosvos
