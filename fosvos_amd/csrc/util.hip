// Layout conversion and weight packing kernels + library-wide state.
#include <stdarg.h>

#include "common.hpp"

namespace fosvos {
thread_local char g_err[512] = "";
}
using namespace fosvos;

extern "C" int fosvos_abi_version(void) { return FOSVOS_ABI_VERSION; }

// ------------------------------------------------------------------------------------------ launch profiler
// Timing events around every kernel launch, on the stream the kernel is launched on (the two-stream backward has two).
// Events exist only between fosvos_profile_start and fosvos_profile_stop; nothing is allocated otherwise.
#include <mutex>
#include <new>
#include <vector>
namespace fosvos {
bool g_prof_on = false;
namespace {
struct ProfLaunch {
    const char *name;
    double flops;
    hipStream_t stream;
};
std::mutex g_prof_mutex;
std::vector<hipEvent_t> g_prof_events;   // 2 per launch slot
std::vector<ProfLaunch> g_prof_launches;
int g_prof_device = -1;
thread_local int t_prof_open = -1;       // slot whose stop event is still to be recorded by this thread
thread_local hipStream_t t_prof_stream = nullptr;
}  // namespace

void prof_begin(const char *name, hipStream_t st, double flops) {
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    t_prof_open = -1;
    const size_t slot = g_prof_launches.size();
    if (!g_prof_on || 2 * slot + 1 >= g_prof_events.size()) return;  // capacity reached: later launches go untimed
    g_prof_launches.push_back({name, flops, st});
    if (hipEventRecord(g_prof_events[2 * slot], st) != hipSuccess) {
        g_prof_launches.pop_back();
        return;
    }
    t_prof_open = (int)slot;
    t_prof_stream = st;
}

void prof_end() {
    if (t_prof_open < 0) return;
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    if (g_prof_on && 2 * (size_t)t_prof_open + 1 < g_prof_events.size())
        (void)hipEventRecord(g_prof_events[2 * t_prof_open + 1], t_prof_stream);
    t_prof_open = -1;
}
}  // namespace fosvos

extern "C" int fosvos_profile_start(int device, int max_launches) {
    FOSVOS_REQUIRE(max_launches > 0 && max_launches <= (1 << 20), FOSVOS_E_ARG, "profile_start: max_launches=%d", max_launches);
    FOSVOS_ENTER(device);
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    FOSVOS_REQUIRE(!g_prof_on && g_prof_events.empty(), FOSVOS_E_ARG, "profile_start: a profile is already running");
    g_prof_events.resize(2 * (size_t)max_launches);
    for (size_t i = 0; i < g_prof_events.size(); ++i) {
        if (hipEventCreate(&g_prof_events[i]) != hipSuccess) {
            for (size_t j = 0; j < i; ++j) (void)hipEventDestroy(g_prof_events[j]);
            g_prof_events.clear();
            return fail(FOSVOS_E_HIP, "profile_start: hipEventCreate failed at event %zu", i);
        }
    }
    g_prof_launches.clear();
    g_prof_launches.reserve(max_launches);
    g_prof_device = device;
    g_prof_on = true;
    return FOSVOS_OK;
}

extern "C" int fosvos_profile_stop(int device, fosvos_profile_record *out, int capacity, int *n_out) {
    FOSVOS_REQUIRE(out && n_out && capacity > 0, FOSVOS_E_ARG, "profile_stop: null output");
    FOSVOS_ENTER(device);
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    FOSVOS_REQUIRE(g_prof_on && device == g_prof_device, FOSVOS_E_ARG, "profile_stop: no profile running on device %d", device);
    g_prof_on = false;
    int rc = FOSVOS_OK;
    if (hipDeviceSynchronize() != hipSuccess) rc = fail(FOSVOS_E_HIP, "profile_stop: hipDeviceSynchronize failed");
    int n = 0;
    // lab builds: FOSVOS_PROF_TIMELINE=<file> also writes every launch's start (after the first launch's) and duration
    FILE *timeline = nullptr;
    if (const char *path = lab_env("FOSVOS_PROF_TIMELINE")) timeline = fopen(path, "w");
    if (timeline) fprintf(timeline, "index,name,stream,start_us,dur_us\n");
    for (size_t i = 0; rc == FOSVOS_OK && i < g_prof_launches.size(); ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, g_prof_events[2 * i], g_prof_events[2 * i + 1]) != hipSuccess) continue;
        if (timeline) {
            float t0 = 0.f;
            if (hipEventElapsedTime(&t0, g_prof_events[0], g_prof_events[2 * i]) == hipSuccess)
                fprintf(timeline, "%zu,%s,%p,%.2f,%.2f\n", i, g_prof_launches[i].name, (void *)g_prof_launches[i].stream,
                        t0 * 1e3, ms * 1e3);
        }
        int k = 0;
        while (k < n && strncmp(out[k].name, g_prof_launches[i].name, sizeof(out[k].name) - 1) != 0) ++k;
        if (k == n) {
            if (n == capacity) continue;  // more distinct kernels than the caller has room for
            memset(&out[n], 0, sizeof(out[n]));
            strncpy(out[n].name, g_prof_launches[i].name, sizeof(out[n].name) - 1);
            ++n;
        }
        out[k].launches += 1;
        out[k].ms += ms;
        out[k].flops += g_prof_launches[i].flops;
    }
    if (timeline) fclose(timeline);
    for (hipEvent_t e : g_prof_events) (void)hipEventDestroy(e);
    g_prof_events.clear();
    g_prof_launches.clear();
    g_prof_device = -1;
    *n_out = n;
    return rc;
}
// ------------------------------------------------------------------------------------------ execution context
extern "C" int fosvos_ctx_create(int device, fosvos_ctx **ctx_out) {
    FOSVOS_REQUIRE(ctx_out, FOSVOS_E_ARG, "ctx_create: null output");
    *ctx_out = nullptr;
    FOSVOS_REQUIRE(device >= 0, FOSVOS_E_ARG, "ctx_create: device %d", device);
    FOSVOS_ENTER(device);
    fosvos_ctx *c = new (std::nothrow) fosvos_ctx();
    FOSVOS_REQUIRE(c, FOSVOS_E_ARG, "ctx_create: out of host memory");
    c->device = device;
    c->buckets_recorded = false;
    hipEvent_t *all[2] = {c->vgg_ev, c->resnet_ev};
    const int count[2] = {kFosvosVggEvents, kFosvosResnetEvents};
    for (int k = 0; k < 2; ++k)
        for (int i = 0; i < count[k]; ++i)
            if (hipEventCreateWithFlags(&all[k][i], hipEventDisableTiming) != hipSuccess) {
                for (int k2 = 0; k2 <= k; ++k2)
                    for (int j = 0; j < (k2 < k ? count[k2] : i); ++j) (void)hipEventDestroy(all[k2][j]);
                delete c;
                return fail(FOSVOS_E_HIP, "ctx_create: hipEventCreateWithFlags failed on device %d", device);
            }
    c->magic = kFosvosCtxMagic;
    *ctx_out = c;
    return FOSVOS_OK;
}

extern "C" int fosvos_ctx_destroy(fosvos_ctx *ctx) {
    if (!ctx) return FOSVOS_OK;
    if (int rc = ctx_check(ctx, "ctx_destroy")) return rc;
    FOSVOS_ENTER(ctx->device);
    ctx->magic = 0;
    for (int i = 0; i < kFosvosVggEvents; ++i) (void)hipEventDestroy(ctx->vgg_ev[i]);
    for (int i = 0; i < kFosvosResnetEvents; ++i) (void)hipEventDestroy(ctx->resnet_ev[i]);
    delete ctx;
    return FOSVOS_OK;
}

extern "C" int fosvos_ctx_device(const fosvos_ctx *ctx) {
    return (ctx && ctx->magic == kFosvosCtxMagic) ? ctx->device : -1;
}

extern "C" const char *fosvos_last_error(void) { return fosvos::g_err; }
extern "C" const char *fosvos_build_arch(void) { return "gfx950"; }

// ------------------------------------------------------------------------------ layout conversion
// One thread = one pixel x 8 output channels.  Reads are coalesced per channel plane (consecutive
// lanes = consecutive w), the 16-byte store is coalesced across the 8-channel groups of a pixel row.
__global__ void k_nchw_f32_to_nhwc_bf16(const float *__restrict__ src, uint16_t *__restrict__ dst, int C, int HW,
                                         int Cpad, int64_t total /* N*HW*(Cpad/8) */) {
    const int groups = Cpad >> 3;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        // pixel fastest inside a group so that plane reads coalesce
        const int64_t pix = i % HW;
        const int64_t r = i / HW;
        const int g = (int)(r % groups);
        const int64_t n = r / groups;
        float f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = g * 8 + j;
            f[j] = c < C ? src[(n * C + c) * HW + pix] : 0.f;
        }
        *reinterpret_cast<uint4 *>(dst + ((n * HW + pix) * Cpad + g * 8)) = pack8(f);
    }
}

__global__ void k_nhwc_bf16_to_nchw_f32(const uint16_t *__restrict__ src, float *__restrict__ dst, int C, int HW,
                                         int Cpad, int64_t total /* N*C*HW */) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pix = i % HW;
        const int64_t r = i / HW;
        const int c = (int)(r % C);
        const int64_t n = r / C;
        dst[i] = bf2f(src[(n * HW + pix) * Cpad + c]);
    }
}

__global__ void k_nhwc_f32_to_nchw_f32(const float *__restrict__ src, float *__restrict__ dst, int C, int HW,
                                        int64_t total) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pix = i % HW;
        const int64_t r = i / HW;
        const int c = (int)(r % C);
        const int64_t n = r / C;
        dst[i] = src[(n * HW + pix) * C + c];
    }
}

__global__ void k_nchw_f32_to_nhwc_f32(const float *__restrict__ src, float *__restrict__ dst, int C, int HW,
                                        int64_t total) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pix = i % HW;
        const int64_t r = i / HW;
        const int c = (int)(r % C);
        const int64_t n = r / C;
        dst[(n * HW + pix) * C + c] = src[i];
    }
}

static inline int grid_for(int64_t total, int block = 256, int cap = 8192) {
    int64_t g = cdiv(total, block);
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

extern "C" int fosvos_nchw_f32_to_nhwc_bf16(const float *src, uint16_t *dst, int N, int C, int H, int W, int Cpad,
                                            int device, void *stream) {
    FOSVOS_REQUIRE(src && dst, FOSVOS_E_ARG, "nchw_f32_to_nhwc_bf16: null pointer");
    FOSVOS_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && Cpad >= C && Cpad % 8 == 0, FOSVOS_E_SHAPE,
                   "nchw_f32_to_nhwc_bf16: bad shape N=%d C=%d H=%d W=%d Cpad=%d", N, C, H, W, Cpad);
    FOSVOS_ENTER(device);
    const int64_t total = (int64_t)N * H * W * (Cpad / 8);
    FOSVOS_PROF("k_nchw_f32_to_nhwc_bf16", stream, 0.0);
    hipLaunchKernelGGL(k_nchw_f32_to_nhwc_bf16, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, src, dst, C,
                       H * W, Cpad, total);
    FOSVOS_LAUNCH_CHECK();
    return FOSVOS_OK;
}

extern "C" int fosvos_nhwc_bf16_to_nchw_f32(const uint16_t *src, float *dst, int N, int C, int H, int W, int Cpad,
                                            int device, void *stream) {
    FOSVOS_REQUIRE(src && dst, FOSVOS_E_ARG, "nhwc_bf16_to_nchw_f32: null pointer");
    FOSVOS_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && Cpad >= C, FOSVOS_E_SHAPE, "nhwc_bf16_to_nchw_f32: bad shape");
    FOSVOS_ENTER(device);
    const int64_t total = (int64_t)N * C * H * W;
    hipLaunchKernelGGL(k_nhwc_bf16_to_nchw_f32, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, src, dst, C,
                       H * W, Cpad, total);
    FOSVOS_LAUNCH_CHECK();
    return FOSVOS_OK;
}

extern "C" int fosvos_nhwc_f32_to_nchw_f32(const float *src, float *dst, int N, int C, int H, int W, int device,
                                           void *stream) {
    FOSVOS_REQUIRE(src && dst, FOSVOS_E_ARG, "nhwc_f32_to_nchw_f32: null pointer");
    FOSVOS_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0, FOSVOS_E_SHAPE, "nhwc_f32_to_nchw_f32: bad shape");
    FOSVOS_ENTER(device);
    const int64_t total = (int64_t)N * C * H * W;
    hipLaunchKernelGGL(k_nhwc_f32_to_nchw_f32, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, src, dst, C,
                       H * W, total);
    FOSVOS_LAUNCH_CHECK();
    return FOSVOS_OK;
}

extern "C" int fosvos_nchw_f32_to_nhwc_f32(const float *src, float *dst, int N, int C, int H, int W, int device,
                                           void *stream) {
    FOSVOS_REQUIRE(src && dst, FOSVOS_E_ARG, "nchw_f32_to_nhwc_f32: null pointer");
    FOSVOS_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0, FOSVOS_E_SHAPE, "nchw_f32_to_nhwc_f32: bad shape");
    FOSVOS_ENTER(device);
    const int64_t total = (int64_t)N * C * H * W;
    hipLaunchKernelGGL(k_nchw_f32_to_nhwc_f32, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, src, dst, C,
                       H * W, total);
    FOSVOS_LAUNCH_CHECK();
    return FOSVOS_OK;
}

// ------------------------------------------------------------------------------ weight packing
// Packed image: [K32 = ceil(in/32)][tap 9][kc 4][out_pad][8] bf16, where for the forward image
// (in, out) = (Ci, Co) and element = w[co][ci][tap]; for the dgrad image (in, out) = (Co, Ci) and
// element = w[co][ci][8 - tap].  One thread writes one 16-byte group of 8 contraction channels.
__global__ void k_pack_w(const float *__restrict__ w, uint16_t *__restrict__ dst, int Co, int Ci, int in_ch,
                         int out_ch, int out_pad, int transpose, int64_t total_groups) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total_groups;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int o = (int)(i % out_pad);
        int64_t r = i / out_pad;
        const int kc = (int)(r & 3);
        r >>= 2;
        const int tap = (int)(r % 9);
        const int k32 = (int)(r / 9);
        float f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int ic = k32 * 32 + kc * 8 + j;
            float v = 0.f;
            if (ic < in_ch && o < out_ch) {
                if (!transpose)
                    v = w[((int64_t)o * Ci + ic) * 9 + tap];  // co = o, ci = ic
                else
                    v = w[((int64_t)ic * Ci + o) * 9 + (8 - tap)];  // co = ic, ci = o
            }
            f[j] = v;
        }
        *reinterpret_cast<uint4 *>(dst + i * 8) = pack8(f);
    }
}

// Both images of MANY layers in one launch (the step after an optimizer update repacks all 17 layers): a block owns
// 32 co x 32 ci x 9 taps of one layer, reads its 32 rows of 288 contiguous floats, and writes the two images from LDS
// in 512-byte runs.  (k_pack_w above gathers 8 floats 36 bytes apart per thread: ~0.5 TB/s.)
struct PackTable {
    int n;
    int co_tile;  // output channels per block: 32, or 8 when the launch would otherwise be a few dozen blocks
    fosvos_pack_entry e[24];
    int block_begin[25];
};

__global__ __launch_bounds__(256) void k_pack_w_tiled(const PackTable t) {
    __shared__ float tile[32][289];
    int e = 0;
    while (e + 1 < t.n && (int)blockIdx.x >= t.block_begin[e + 1]) ++e;
    const fosvos_pack_entry &q = t.e[e];
    const int local = blockIdx.x - t.block_begin[e];
    const int nkb = q.Ci / 32, co_t = t.co_tile;
    const int cb = local / nkb, kb = local % nkb;
    const int co0 = cb * co_t, ci0 = kb * 32;
    // a row is 288 contiguous floats, 16-byte aligned when w is (ci0 * 9 floats = a multiple of 1152 bytes)
    if (((uintptr_t)q.w & 15) == 0) {
        for (int i = threadIdx.x; i < co_t * 72; i += 256) {
            const int r = i / 72, c = (i - r * 72) * 4;
            const float4 v = (co0 + r < q.Co) ? *reinterpret_cast<const float4 *>(q.w + ((int64_t)(co0 + r) * q.Ci + ci0) * 9 + c)
                                              : make_float4(0.f, 0.f, 0.f, 0.f);
            tile[r][c] = v.x; tile[r][c + 1] = v.y; tile[r][c + 2] = v.z; tile[r][c + 3] = v.w;
        }
    } else {
        for (int i = threadIdx.x; i < co_t * 288; i += 256) {
            const int r = i / 288, c = i - r * 288;
            tile[r][c] = (co0 + r < q.Co) ? q.w[((int64_t)(co0 + r) * q.Ci + ci0) * 9 + c] : 0.f;
        }
    }
    __syncthreads();
    const int co_pad = (q.Co + 15) / 16 * 16, ci_pad = (q.Ci + 15) / 16 * 16;
    if (q.w_fwd) {  // contraction over ci: k-chunk kb, group kc = 8 ci; output channel co0 + l
        for (int g = threadIdx.x; g < 9 * 4 * co_t; g += 256) {
            const int l = g % co_t, kc = (g / co_t) & 3, tap = g / (4 * co_t);
            if (co0 + l >= co_pad) continue;
            float f[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = tile[l][(kc * 8 + j) * 9 + tap];
            *reinterpret_cast<uint4 *>(q.w_fwd + ((((int64_t)kb * 9 + tap) * 4 + kc) * co_pad + co0 + l) * 8) = pack8(f);
        }
    }
    if (q.w_dgrad) {  // contraction over co: k-chunk (co0 + 8 kc) / 32, group = its 8 co; output channel ci0 + l; rotated taps
        const int n_kc = co_t / 8;
        for (int g = threadIdx.x; g < 9 * n_kc * 32; g += 256) {
            const int l = g & 31, kc = (g >> 5) % n_kc, tap = (g >> 5) / n_kc;
            float f[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = tile[kc * 8 + j][l * 9 + tap];
            const int co = co0 + kc * 8;
            *reinterpret_cast<uint4 *>(q.w_dgrad + ((((int64_t)(co >> 5) * 9 + (8 - tap)) * 4 + ((co >> 3) & 3)) * ci_pad + ci0 + l) * 8) =
                pack8(f);
        }
    }
}

extern "C" int fosvos_pack_conv3x3_weights_multi(const fosvos_pack_entry *entries, int n, int device, void *stream) {
    FOSVOS_REQUIRE(entries && n > 0 && n <= 24, FOSVOS_E_ARG, "pack_conv3x3_weights_multi: 1..24 entries, got %d", n);
    PackTable t;
    t.n = n;
    for (int i = 0; i < n; ++i) {
        const fosvos_pack_entry &q = entries[i];
        FOSVOS_REQUIRE(q.w && (q.w_fwd || q.w_dgrad), FOSVOS_E_ARG, "pack_conv3x3_weights_multi: null pointer in entry %d", i);
        FOSVOS_REQUIRE(q.Co > 0 && q.Ci > 0 && q.Ci % 32 == 0, FOSVOS_E_SHAPE,
                       "pack_conv3x3_weights_multi: entry %d has Co=%d Ci=%d (Ci must be a multiple of 32)", i, q.Co, q.Ci);
        t.e[i] = q;
    }
    // (the rows of a 32-channel chunk of the dgrad image are all written, zeros where co >= Co: blocks cover roundup(Co, 32))
    auto count = [&](int co_tile) {
        int blocks = 0;
        for (int i = 0; i < n; ++i) {
            t.block_begin[i] = blocks;
            blocks += roundup(entries[i].Co, 32) / co_tile * (entries[i].Ci / 32);
        }
        t.block_begin[n] = blocks;
        return blocks;
    };
    // the layers the late half of a split optimizer step repacks (conv1_2 .. conv2_2, side_prep[0]) are 32 blocks of 32
    // channels: a launch bound by one block's latency, in front of the first kernel of the next cycle
    t.co_tile = 32;
    int blocks = count(32);
    if (blocks < 256) { t.co_tile = 8; blocks = count(8); }
    FOSVOS_ENTER(device);
    FOSVOS_PROF("k_pack_w_tiled", stream, 0.0);
    hipLaunchKernelGGL(k_pack_w_tiled, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, t);
    FOSVOS_LAUNCH_CHECK();
    return FOSVOS_OK;
}

extern "C" size_t fosvos_packed_weight_elems(int out_ch, int in_ch) {
    return (size_t)roundup(in_ch, 32) * 9 * (size_t)roundup(out_ch, 16);
}

extern "C" int fosvos_pack_conv3x3_weights(const float *w, int Co, int Ci, uint16_t *w_fwd, uint16_t *w_dgrad,
                                           int device, void *stream) {
    FOSVOS_REQUIRE(w, FOSVOS_E_ARG, "pack_conv3x3_weights: null weight pointer");
    FOSVOS_REQUIRE(Co > 0 && Ci > 0, FOSVOS_E_SHAPE, "pack_conv3x3_weights: bad shape Co=%d Ci=%d", Co, Ci);
    FOSVOS_ENTER(device);
    if (w_fwd) {
        const int out_pad = roundup(Co, 16);
        const int64_t groups = (int64_t)(roundup(Ci, 32) / 32) * 9 * 4 * out_pad;
        hipLaunchKernelGGL(k_pack_w, dim3(grid_for(groups)), dim3(256), 0, (hipStream_t)stream, w, w_fwd, Co, Ci, Ci, Co,
                           out_pad, 0, groups);
        FOSVOS_LAUNCH_CHECK();
    }
    if (w_dgrad) {
        const int out_pad = roundup(Ci, 16);
        const int64_t groups = (int64_t)(roundup(Co, 32) / 32) * 9 * 4 * out_pad;
        hipLaunchKernelGGL(k_pack_w, dim3(grid_for(groups)), dim3(256), 0, (hipStream_t)stream, w, w_dgrad, Co, Ci, Co, Ci,
                           out_pad, 1, groups);
        FOSVOS_LAUNCH_CHECK();
    }
    return FOSVOS_OK;
}
