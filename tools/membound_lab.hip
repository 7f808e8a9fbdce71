// How fast can the memory system move conv1_2's traffic pattern, with no arithmetic at all?  Each workgroup reads the
// (TH+2) x (TW+2) pixel halo of its tile (64 bf16 channels = 128 B per pixel, NHWC) and writes TH x TW pixels x 128 B.
//   hipcc -O3 --offload-arch=gfx950 tools/membound_lab.hip -o build/membound_lab ; build/membound_lab [H W TH TW wgs_per_cu]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(256) void k_halo_copy(const uint4 *__restrict__ x, uint4 *__restrict__ y, int H, int W, int TH,
                                                    int TW, int tiles_x, int tiles) {
    for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
        const int y0 = (t / tiles_x) * TH, x0 = (t % tiles_x) * TW;
        const int hw = TW + 2, hn = (TH + 2) * hw * 8;  // 8 sixteen-byte vectors per pixel
        uint4 acc = make_uint4(0, 0, 0, 0);
        for (int i = threadIdx.x; i < hn; i += 256) {
            const int p = i >> 3, v = i & 7;
            const int gy = min(max(y0 + p / hw - 1, 0), H - 1), gx = min(max(x0 + p % hw - 1, 0), W - 1);
            const uint4 r = x[((int64_t)gy * W + gx) * 8 + v];
            acc.x ^= r.x; acc.y ^= r.y; acc.z ^= r.z; acc.w ^= r.w;
        }
        for (int i = threadIdx.x; i < TH * TW * 8; i += 256) {
            const int p = i >> 3, v = i & 7;
            const int gy = y0 + p / TW, gx = x0 + p % TW;
            if (gy < H && gx < W) y[((int64_t)gy * W + gx) * 8 + v] = acc;
        }
    }
}

int main(int argc, char **argv) {
    const int H = argc > 1 ? atoi(argv[1]) : 480, W = argc > 2 ? atoi(argv[2]) : 854;
    const int TH = argc > 3 ? atoi(argv[3]) : 8, TW = argc > 4 ? atoi(argv[4]) : 32, per_cu = argc > 5 ? atoi(argv[5]) : 2;
    const size_t bytes = (size_t)H * W * 128;
    uint4 *x, *y;
    hipMalloc(&x, bytes);
    hipMalloc(&y, bytes);
    hipMemset(x, 1, bytes);
    const int tiles_x = (W + TW - 1) / TW, tiles = tiles_x * ((H + TH - 1) / TH);
    const int grid = tiles < 256 * per_cu ? tiles : 256 * per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k_halo_copy, dim3(grid), dim3(256), 0, 0, x, y, H, W, TH, TW, tiles_x, tiles);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k_halo_copy, dim3(grid), dim3(256), 0, 0, x, y, H, W, TH, TW, tiles_x, tiles);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%dx%d tiles %dx%d grid %d: %.1f us/launch, %.2f TB/s algorithmic (read+write %zu MB)\n", H, W, TH, TW, grid,
           ms * 1e3 / 50, 2.0 * bytes / (ms / 50 * 1e-3) * 1e-12, 2 * bytes >> 20);
    return 0;
}
