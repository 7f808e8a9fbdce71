// Lab tool (never shipped): how many bytes per clock can ONE CU pull from its XCD's L2 through the vector-memory path?
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/cu_ingest_lab.hip -o build/cu_ingest_lab && build/cu_ingest_lab
// Every workgroup of an XCD (blockIdx % 8) walks the same 2-MB window (L2-resident after the first pass, far larger than the
// 32-KB L1) from its own start offset, with 16-byte-per-lane loads: to registers (buffer_load_dwordx4), to LDS
// (buffer_load_dwordx4 ... lds) or 4 bytes per lane (buffer_load_dword).  Reported per CU: bytes per shader clock
// (s_memtime over each workgroup's life) at 4 / 8 / 16 waves per CU, and the chip-wide rate.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                            \
    do {                                                                                 \
        hipError_t e = (x);                                                              \
        if (e != hipSuccess) {                                                           \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                       \
            exit(1);                                                                     \
        }                                                                                \
    } while (0)

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr unsigned kWindow = 2u << 20;  // bytes per XCD

// MODE 0: dwordx4 -> registers, 1: dwordx4 -> LDS (DMA), 2: dword -> registers, 3: dwordx4 -> registers, every wave of the
// chip reads the SAME 36 KB (a weight tile: L1 hits after the first touch)
template <int MODE, int UNROLL>
__global__ __launch_bounds__(256) void k_ingest(const unsigned char *base, unsigned long long *stamps, unsigned *sink, int iters) {
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(base) + (size_t)xcd * kWindow, 0, kWindow, 0x00020000);
    const unsigned ldsb = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void_t *)smem);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    u32x4 acc = {0, 0, 0, 0};
    // a wave's load covers 1 KB (MODE 2: 256 B); the workgroup's waves read consecutive kilobytes
    unsigned off = (slot * 65536u + wave * 1024u) & (kWindow - 1);
    const unsigned span = MODE == 3 ? 36864u : kWindow;
    if (MODE == 3) off = wave * 1024u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if constexpr (MODE == 1) {
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(ldsb + (wave * UNROLL + u) * 1024u),
                             "v"((unsigned)(lane * 16)), "s"(rsrc), "s"(off)
                             : "memory", "m0");
            } else if constexpr (MODE == 2) {
                const unsigned v = __builtin_amdgcn_raw_buffer_load_b32(rsrc, lane * 4, off, 0);
                acc[0] ^= v;
            } else {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16, off, 0);
                acc ^= v;
            }
            off += MODE == 2 ? 4 * 256u : 4 * 1024u;
            if (off >= span) off -= span;
        }
        if constexpr (MODE == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) {
        stamps[2 * blockIdx.x] = t0;
        stamps[2 * blockIdx.x + 1] = t1;
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) sink[blockIdx.x * 256 + tid] = acc[0];
}

template <int MODE, int UNROLL>
void run(const char *name, int wg_per_cu, const unsigned char *base, unsigned long long *stamps, unsigned *sink) {
    const int grid = 256 * wg_per_cu, iters = 512 / UNROLL * 4;
    const int lds = std::max(160 * 1024 / wg_per_cu - 2048, 40 * 1024);
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ingest<MODE, UNROLL>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_ingest<MODE, UNROLL>), dim3(grid), dim3(256), lds, 0, base, stamps, sink, iters);  // warms the L2
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_ingest<MODE, UNROLL>), dim3(grid), dim3(256), lds, 0, base, stamps, sink, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> hs(2 * grid);
    CK(hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost));
    const double per_wg = (double)iters * UNROLL * 4 * (MODE == 2 ? 256.0 : 1024.0);
    std::vector<double> rate;
    for (int g = 0; g < grid; ++g) rate.push_back(per_wg / (double)(hs[2 * g + 1] - hs[2 * g]) * wg_per_cu);
    std::sort(rate.begin(), rate.end());
    printf("%-28s %2d waves/CU  %6.1f B/clk/CU (p10 %5.1f p90 %5.1f)  chip %6.2f TB/s  %.3f ms\n", name, 4 * wg_per_cu, rate[grid / 2], rate[grid / 10],
           rate[grid * 9 / 10], per_wg * grid / (ms * 1e-3) / 1e12, ms);
}

int main() {
    unsigned char *base;
    unsigned long long *stamps;
    unsigned *sink;
    CK(hipMalloc(&base, 8 * (size_t)kWindow));
    CK(hipMemset(base, 1, 8 * (size_t)kWindow));
    CK(hipMalloc(&stamps, 2 * 4096 * 8));
    CK(hipMalloc(&sink, 4096 * 256 * 4));
    for (int w : {1, 2, 4}) {
        if (w == 1) {
            run<0, 8>("dwordx4 -> regs", 1, base, stamps, sink);
            run<1, 8>("dwordx4 -> LDS (DMA)", 1, base, stamps, sink);
            run<2, 8>("dword -> regs", 1, base, stamps, sink);
            run<3, 8>("dwordx4, same 36 KB (L1)", 1, base, stamps, sink);
        } else if (w == 2) {
            run<0, 8>("dwordx4 -> regs", 2, base, stamps, sink);
            run<1, 8>("dwordx4 -> LDS (DMA)", 2, base, stamps, sink);
            run<2, 8>("dword -> regs", 2, base, stamps, sink);
            run<3, 8>("dwordx4, same 36 KB (L1)", 2, base, stamps, sink);
        } else {
            run<0, 8>("dwordx4 -> regs", 4, base, stamps, sink);
            run<1, 8>("dwordx4 -> LDS (DMA)", 4, base, stamps, sink);
            run<2, 8>("dword -> regs", 4, base, stamps, sink);
            run<3, 8>("dwordx4, same 36 KB (L1)", 4, base, stamps, sink);
        }
    }
    return 0;
}
