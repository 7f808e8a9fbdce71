// Diagnostic build of the implicit-GEMM conv kernel with in-kernel clock stamps (never shipped):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DFOSVOS_STAMP tools/igemm_lab.hip -o build/igemm_lab
//   build/igemm_lab H W Ci Co [reps] [frames per launch]
// Times fosvos_conv3x3_fwd on random data with HIP events, then prints per-phase medians (shader
// clocks, wave 0 of each workgroup) and the distribution of workgroup start/end times.
#include "../fosvos_amd/csrc/conv_igemm.hip"

#include <algorithm>
#include <cstdlib>
#include <vector>

namespace fosvos {
thread_local char g_err[512] = "";
bool g_prof_on = false;
void prof_begin(const char *, hipStream_t, double) {}
void prof_end() {}
}  // namespace fosvos
// the pooled-output fallback of fosvos_conv3x3_fwd_pool lives in another translation unit; the lab never takes it
extern "C" int fosvos_maxpool2x2_ceil_fwd(const uint16_t *, uint16_t *, int, int, int, int, int, void *) { return -1; }

#define CK(x)                                                                            \
    do {                                                                                 \
        hipError_t e = (x);                                                              \
        if (e != hipSuccess) {                                                           \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                       \
            return 1;                                                                    \
        }                                                                                \
    } while (0)

static uint16_t rnd_bf16(uint32_t &s, float scale) {
    s = s * 1664525u + 1013904223u;
    float f = ((int)(s >> 8) % 2001 - 1000) * 0.001f * scale;
    uint32_t u;
    memcpy(&u, &f, 4);
    return (uint16_t)((u + 0x7fff + ((u >> 16) & 1)) >> 16);
}

int main(int argc, char **argv) {
    if (argc < 5) {
        fprintf(stderr, "usage: igemm_lab H W Ci Co [reps] [frames]\n");
        return 2;
    }
    const int H = atoi(argv[1]), W = atoi(argv[2]), Ci = atoi(argv[3]), Co = atoi(argv[4]);
    const int reps = argc > 5 ? atoi(argv[5]) : 20;
    const int N = argc > 6 ? atoi(argv[6]) : 1;  // frames per launch
    const size_t nx = (size_t)N * H * W * Ci, ny = (size_t)N * H * W * Co;
    const size_t nw = (size_t)(Ci / 32) * 36 * Co * 8;
    std::vector<uint16_t> hx(nx), hw(nw);
    uint32_t seed = 12345;
    for (auto &v : hx) v = rnd_bf16(seed, 1.0f);
    for (auto &v : hw) v = rnd_bf16(seed, 0.05f);
    std::vector<float> hb(Co, 0.1f);
    uint16_t *dx, *dw, *dy;
    float *db;
    CK(hipMalloc(&dx, nx * 2));
    CK(hipMalloc(&dw, nw * 2));
    CK(hipMalloc(&dy, ny * 2));
    CK(hipMalloc(&db, Co * 4));
    CK(hipMemcpy(dx, hx.data(), nx * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw, hw.data(), nw * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), Co * 4, hipMemcpyHostToDevice));
    const size_t wsb = fosvos_conv3x3_workspace_bytes(N, H, W, Ci, Co);
    void *ws = nullptr;
    if (wsb) CK(hipMalloc(&ws, wsb));
    const ConvPlan plan = make_plan(N, H, W, Ci, Co);
    const int th = plan.tile == kSmall ? 4 : plan.tile == kSquare ? 16 : 8, tw = plan.tile == kBig ? 32 : 16;
    const int64_t nwg = (int64_t)N * cdiv(W, tw) * cdiv(H, th) * (Co / 64) * plan.k_splits;
    printf("conv %dx%d Ci=%d Co=%d  tile=%d k_splits=%d  workgroups=%lld\n", H, W, Ci, Co, (int)plan.tile,
           plan.k_splits, (long long)nwg);
    unsigned long long *dst;
    CK(hipMalloc(&dst, nwg * 16 * 8));
    CK(hipMemset(dst, 0, nwg * 16 * 8));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i)
        if (fosvos_conv3x3_fwd(dx, dw, db, dy, N, H, W, Ci, Co, FOSVOS_CONV_RELU, ws, wsb, 0, st)) return 1;
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i)
        if (fosvos_conv3x3_fwd(dx, dw, db, dy, N, H, W, Ci, Co, FOSVOS_CONV_RELU, ws, wsb, 0, st)) return 1;
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps, tf = 2.0 * N * H * W * 9.0 * Ci * Co / us * 1e-6;
    printf("no stamps: %.1f us/launch  %.1f TFLOP/s\n", us, tf);
    g_stamps = dst;
    if (fosvos_conv3x3_fwd(dx, dw, db, dy, N, H, W, Ci, Co, FOSVOS_CONV_RELU, ws, wsb, 0, st)) return 1;
    CK(hipStreamSynchronize(st));
    std::vector<unsigned long long> hs(nwg * 16);
    CK(hipMemcpy(hs.data(), dst, nwg * 16 * 8, hipMemcpyDeviceToHost));
    const char *names[] = {"issue first loads        (0->1)", "first loads land + store (1->2)",
                           "barrier                  (2->3)", "issue next loads         (3->4)",
                           "MFMA phase chunk 0       (3->5)", "middle chunks            (5->6)",
                           "last store               (6->7)", "last barrier+MFMA        (7->8)",
                           "epilogue to LDS          (8->9)", "epilogue global stores  (9->10)"};
    for (int p = 0; p < 10; ++p) {
        std::vector<long long> d;
        for (int64_t g = 0; g < nwg; ++g) {
            const unsigned long long a = hs[g * 16 + (p == 4 ? 3 : p)], b = hs[g * 16 + p + 1];
            if (a && b) d.push_back((long long)(b - a));
        }
        if (d.empty()) continue;
        std::sort(d.begin(), d.end());
        printf("  %-34s median %8lld  p10 %8lld  p90 %8lld clk\n", names[p], d[d.size() / 2], d[d.size() / 10],
               d[d.size() * 9 / 10]);
    }
    {
        std::vector<long long> tot, st0, en;
        unsigned long long t0 = ~0ull;
        for (int64_t g = 0; g < nwg; ++g) t0 = std::min(t0, hs[g * 16 + 11]);
        for (int64_t g = 0; g < nwg; ++g) {
            tot.push_back((long long)(hs[g * 16 + 10] - hs[g * 16 + 0]));
            st0.push_back((long long)(hs[g * 16 + 11] - t0));
            en.push_back((long long)(hs[g * 16 + 12] - t0));
        }
        std::sort(tot.begin(), tot.end());
        std::sort(st0.begin(), st0.end());
        std::sort(en.begin(), en.end());
        printf("  workgroup lifetime: median %lld clk, p10 %lld, p90 %lld\n", tot[tot.size() / 2], tot[tot.size() / 10],
               tot[tot.size() * 9 / 10]);
        printf("  start times (us after first start, 100 MHz realtime): p25 %.2f p50 %.2f p75 %.2f max %.2f\n",
               st0[st0.size() / 4] * 0.01, st0[st0.size() / 2] * 0.01, st0[st0.size() * 3 / 4] * 0.01, st0.back() * 0.01);
        printf("  end   times: p25 %.2f p50 %.2f p75 %.2f max %.2f us\n", en[en.size() / 4] * 0.01,
               en[en.size() / 2] * 0.01, en[en.size() * 3 / 4] * 0.01, en.back() * 0.01);
        // in-kernel clock: shader clocks per realtime tick over each workgroup's life
        double clk = 0;
        int n = 0;
        for (int64_t g = 0; g < nwg; ++g) {
            const double rt = (double)(hs[g * 16 + 12] - hs[g * 16 + 11]);
            if (rt > 50) { clk += (double)(hs[g * 16 + 10] - hs[g * 16 + 0]) / rt * 100.0; ++n; }
        }
        if (n) printf("  s_memtime clock: %.0f MHz (over %d workgroups)\n", clk / n, n);
    }
    return 0;
}
