// Diagnostic build of the weight-gradient kernel with in-kernel clock stamps (never shipped):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DFOSVOS_WG_STAMP tools/wgrad_stamp_lab.hip -o build/wgrad_stamp_lab
//   build/wgrad_stamp_lab H W Ci Co [reps]
// Prints, per phase of the tile loop, the median over workgroups of the clocks wave 0 spent there (sum over its tiles):
//   0 prologue (first tile staged)  1 issue next tile's loads  2 bias sums  3 MFMA loop  4 wait + LDS writes  5 barrier
//   6 epilogue (slab stores landed)
#include "../fosvos_amd/csrc/conv_wgrad.hip"

#include <algorithm>
#include <cstdlib>
#include <vector>

namespace fosvos {
thread_local char g_err[512] = "";
bool g_prof_on = false;
void prof_begin(const char *, hipStream_t, double) {}
void prof_end() {}
}  // namespace fosvos

#define CK(x)                                                      \
    do {                                                           \
        hipError_t e = (x);                                        \
        if (e != hipSuccess) {                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); \
            return 1;                                              \
        }                                                          \
    } while (0)

int main(int argc, char **argv) {
    if (argc < 5) {
        fprintf(stderr, "usage: wgrad_stamp_lab H W Ci Co [reps]\n");
        return 2;
    }
    const int H = atoi(argv[1]), W = atoi(argv[2]), Ci = atoi(argv[3]), Co = atoi(argv[4]);
    const int reps = argc > 5 ? atoi(argv[5]) : 10;
    const int Cy = (Co + 31) / 32 * 32;
    const size_t nx = (size_t)H * W * Ci, ny = (size_t)H * W * Cy;
    std::vector<uint16_t> hx(nx), hy(ny);
    uint32_t s = 12345;
    auto rnd = [&]() {
        s = s * 1664525u + 1013904223u;
        const float f = ((int)(s >> 8) % 2001 - 1000) * 0.001f;
        uint32_t u;
        memcpy(&u, &f, 4);
        return (uint16_t)((u + 0x7fff + ((u >> 16) & 1)) >> 16);
    };
    for (auto &v : hx) v = rnd();
    for (auto &v : hy) v = rnd();
    uint16_t *dx, *dy;
    float *dw, *db;
    CK(hipMalloc(&dx, nx * 2));
    CK(hipMalloc(&dy, ny * 2));
    CK(hipMalloc(&dw, (size_t)Co * Ci * 9 * 4));
    CK(hipMalloc(&db, Co * 4));
    CK(hipMemcpy(dx, hx.data(), nx * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dy, hy.data(), ny * 2, hipMemcpyHostToDevice));
    const size_t wsb = fosvos_conv3x3_wgrad_workspace_bytes(1, H, W, Ci, Co);
    void *ws;
    CK(hipMalloc(&ws, wsb));
    const int max_wg = 4096;
    unsigned long long *stamps;
    CK(hipMalloc(&stamps, max_wg * 8 * sizeof(unsigned long long)));
    CK(hipMemset(stamps, 0, max_wg * 8 * sizeof(unsigned long long)));
    g_wg_stamps = stamps;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i)
        if (fosvos_conv3x3_wgrad_slabs(dx, dy, 1, 1, H, W, Ci, Co, ws, wsb, 0, nullptr)) {
            fprintf(stderr, "%s\n", fosvos::g_err);
            return 1;
        }
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) fosvos_conv3x3_wgrad_slabs(dx, dy, 1, 1, H, W, Ci, Co, ws, wsb, 0, nullptr);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps;
    printf("%dx%d %d->%d: %.1f us/launch (stamped build), %.1f TFLOP/s\n", H, W, Ci, Co, us,
           2.0 * H * W * 9.0 * Ci * Co / us / 1e6);
    std::vector<unsigned long long> h(max_wg * 8);
    CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
    int n_wg = 0;
    while (n_wg < max_wg && h[n_wg * 8 + 0]) ++n_wg;
    const char *names[7] = {"prologue", "issue loads", "bias", "mfma loop", "wait+lds write", "barrier", "epilogue"};
    unsigned long long total = 0;
    for (int p = 0; p < 7; ++p) {
        std::vector<unsigned long long> v;
        for (int w = 0; w < n_wg; ++w) v.push_back(h[w * 8 + p]);
        std::sort(v.begin(), v.end());
        printf("  %-16s median %8llu  min %8llu  max %8llu clocks\n", names[p], v[v.size() / 2], v.front(), v.back());
        total += v[v.size() / 2];
    }
    printf("  workgroups %d, sum of medians %llu clocks\n", n_wg, total);
    return 0;
}
