// Lab tool (never shipped): the ceiling of an LDS-fed MFMA loop on gfx950, without any global traffic.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/mfma_lds_lab.hip -o build/mfma_lds_lab && build/mfma_lds_lab
// A workgroup of 4 waves keeps a pixel image (4 k-groups x NPIX x 16 B) and a weight image (9 taps x 4 k-groups x BN x 16 B) in
// LDS, as the implicit-GEMM conv does, and each wave multiplies its MF x NF fragments (16x16x32 bf16) for CHUNKS x 9 taps:
// MF + NF ds_read_b128 per MF * NF MFMAs.  Variants: wave tile, workgroups per CU (forced by the LDS request), fragment
// reads skipped (MFMA-only), MFMAs skipped (LDS-only).  Prints achieved TFLOP/s and the share of the 2.5 PFLOP/s peak.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define CK(x)                                                                            \
    do {                                                                                 \
        hipError_t e = (x);                                                              \
        if (e != hipSuccess) {                                                           \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                       \
            exit(1);                                                                     \
        }                                                                                \
    } while (0)

// MODE 0: reads + MFMAs, 1: MFMAs only (fragments read once), 2: reads only
template <int MF, int NF, int MODE, int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_loop(float *out, int chunks) {
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
    constexpr int TW = 32, HALO_W = TW + 2, NPIX_PAD = ((4 * MF * 16 / TW + 2) * HALO_W + 3 + 15) / 16 * 16;
    constexpr int BN = 16 * NF;
    uint4 *sA = smem, *sB = smem + 4 * NPIX_PAD;  // the weight image holds three taps (the lab measures the loop, not the footprint)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = tid; i < 4 * NPIX_PAD + 12 * BN; i += 256) {  // random mantissas and signs, |v| in [1, 2): realistic toggling for DVFS
        unsigned h = (i + 1) * 2654435761u;
        uint4 v;
        h ^= h >> 15; h *= 2246822519u; v.x = (h & 0x807f807fu) | 0x3f803f80u;
        h ^= h >> 13; h *= 3266489917u; v.y = (h & 0x807f807fu) | 0x3f803f80u;
        h ^= h >> 16; h *= 668265263u; v.z = (h & 0x807f807fu) | 0x3f803f80u;
        h ^= h >> 15; h *= 374761393u; v.w = (h & 0x807f807fu) | 0x3f803f80u;
        smem[i] = v;
    }
    __syncthreads();
    int a_base[MF];
#pragma unroll
    for (int i = 0; i < MF; ++i) {
        const int pb = (wave * MF + i) * 16;
        a_base[i] = (lane >> 4) * NPIX_PAD + (pb / TW) * HALO_W + pb % TW + (lane & 15);
    }
    const int b_base = (lane >> 4) * BN + (lane & 15);
    f32x4 acc[MF][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 af[MF], bfr[2][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i) af[i] = __builtin_bit_cast(bf16x8, sA[a_base[i]]);
#pragma unroll
    for (int j = 0; j < NF; ++j) bfr[0][j] = bfr[1][j] = __builtin_bit_cast(bf16x8, sB[b_base + j * 16]);
    for (int cc = 0; cc < chunks; ++cc) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int nt = (tap + 1) % 9;
#pragma unroll
            for (int i = 0; i < MF; ++i) {
                if constexpr (MODE != 2) {
#pragma unroll
                    for (int j = 0; j < NF; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[tap & 1][j], af[i], acc[i][j], 0, 0, 0);
                }
                if constexpr (MODE != 1) {
                    af[i] = __builtin_bit_cast(bf16x8, sA[a_base[i] + (nt / 3) * HALO_W + nt % 3]);
                    constexpr int per_row = (NF + MF - 1) / MF;
#pragma unroll
                    for (int j = i * per_row; j < (i + 1) * per_row && j < NF; ++j)
                        bfr[(tap + 1) & 1][j] = __builtin_bit_cast(bf16x8, sB[b_base + (nt % 3) * 4 * BN + j * 16]);
                    if constexpr (MODE == 2) {
#pragma unroll
                        for (int j = 0; j < NF; ++j) acc[i][j][0] += (float)af[i][0] + (float)bfr[(tap + 1) & 1][j][0];
                    }
                }
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (s == 12345.678f) out[blockIdx.x * 256 + tid] = s;
    if (blockIdx.x == 0 && tid == 0) {  // shader clocks per 100-MHz tick over this workgroup's life
        unsigned long long *st = reinterpret_cast<unsigned long long *>(out + (60 << 20) / 4);
        st[0] = __builtin_amdgcn_s_memtime() - c0;
        st[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

template <int MF, int NF, int MODE, int WPE>
void run(const char *name, int wg_per_cu, float *out, int chunks = 64) {
    const int grid = 256 * wg_per_cu * 4;
    const int lds = 160 * 1024 / wg_per_cu - 1024;  // forces wg_per_cu workgroups per CU
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_loop<MF, NF, MODE, WPE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_loop<MF, NF, MODE, WPE>), dim3(grid), dim3(256), lds, 0, out, chunks);
    CK(hipDeviceSynchronize());
    const int reps = 5;
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_loop<MF, NF, MODE, WPE>), dim3(grid), dim3(256), lds, 0, out, chunks);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double flops = (double)reps * grid * 4 * chunks * 9 * MF * NF * (16.0 * 16 * 32 * 2);
    const double lds_bytes = (double)reps * grid * 4 * chunks * 9 * (MF + NF) * 1024.0;
    const double tf = flops / (ms * 1e-3) / 1e12;
    unsigned long long st[2];
    CK(hipMemcpy(st, reinterpret_cast<char *>(out) + (60 << 20), 16, hipMemcpyDeviceToHost));
    printf("%-34s MF=%d NF=%d wg/cu=%d  %8.3f ms  %7.1f TFLOP/s (%.3f of 2500)  LDS reads %6.1f TB/s  clock %.0f MHz\n", name, MF, NF, wg_per_cu, ms / reps,
           MODE == 2 ? 0.0 : tf, MODE == 2 ? 0.0 : tf / 2500.0, MODE == 1 ? 0.0 : lds_bytes / (ms * 1e-3) / 1e12, 100.0 * st[0] / st[1]);
}

// The same 64 x 64 wave tile on 32x32x16 MFMAs (2 x 2 per k-half, 16 accumulator registers each): the operand bytes per
// flop read from the register file halve.  MODE as above.
template <int MODE, int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_loop32(float *out, int chunks) {
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
    constexpr int TW = 32, HALO_W = TW + 2, NPIX_PAD = ((4 * 4 * 16 / TW + 2) * HALO_W + 3 + 15) / 16 * 16, BN = 64;
    uint4 *sA = smem, *sB = smem + 4 * NPIX_PAD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = tid; i < 4 * NPIX_PAD + 12 * BN; i += 256) {
        unsigned h = (i + 1) * 2654435761u;
        uint4 v;
        h ^= h >> 15; h *= 2246822519u; v.x = (h & 0x807f807fu) | 0x3f803f80u;
        h ^= h >> 13; h *= 3266489917u; v.y = (h & 0x807f807fu) | 0x3f803f80u;
        h ^= h >> 16; h *= 668265263u; v.z = (h & 0x807f807fu) | 0x3f803f80u;
        h ^= h >> 15; h *= 374761393u; v.w = (h & 0x807f807fu) | 0x3f803f80u;
        smem[i] = v;
    }
    __syncthreads();
    // fragment (rb, h): 32 pixels (row block rb of the wave's 2 tile rows) x k-half h; lane -> pixel lane % 32, k-group 2 h + lane / 32
    int a_base[2], b_base[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) a_base[rb] = (lane >> 5) * NPIX_PAD + (wave * 2 + rb) * HALO_W + (lane & 31);
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) b_base[cb] = (lane >> 5) * BN + cb * 32 + (lane & 31);
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    bf16x8 af[2][2][2], bfr[2][2][2];  // [buffer][k-half][block]
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            af[0][h][b] = af[1][h][b] = __builtin_bit_cast(bf16x8, sA[a_base[b] + 2 * h * NPIX_PAD]);
            bfr[0][h][b] = bfr[1][h][b] = __builtin_bit_cast(bf16x8, sB[b_base[b] + 2 * h * BN]);
        }
    for (int cc = 0; cc < chunks; ++cc) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int nt = (tap + 1) % 9, cur = tap & 1, nxt = cur ^ 1;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if constexpr (MODE != 2) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[cur][h][j], af[cur][h][i], acc[i][j], 0, 0, 0);
                }
                if constexpr (MODE != 1) {
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        af[nxt][h][b] = __builtin_bit_cast(bf16x8, sA[a_base[b] + 2 * h * NPIX_PAD + (nt / 3) * HALO_W + nt % 3]);
                        bfr[nxt][h][b] = __builtin_bit_cast(bf16x8, sB[b_base[b] + (nt % 3) * 4 * BN + 2 * h * BN]);
                    }
                }
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    if (s == 12345.678f) out[blockIdx.x * 256 + tid] = s;
    if (blockIdx.x == 0 && tid == 0) {  // shader clocks per 100-MHz tick over this workgroup's life
        unsigned long long *st = reinterpret_cast<unsigned long long *>(out + (60 << 20) / 4);
        st[0] = __builtin_amdgcn_s_memtime() - c0;
        st[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

template <int MODE, int WPE>
void run32(const char *name, int wg_per_cu, float *out, int chunks = 64) {
    const int grid = 256 * wg_per_cu * 4;
    const int lds = 160 * 1024 / wg_per_cu - 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_loop32<MODE, WPE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_loop32<MODE, WPE>), dim3(grid), dim3(256), lds, 0, out, chunks);
    CK(hipDeviceSynchronize());
    const int reps = 5;
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_loop32<MODE, WPE>), dim3(grid), dim3(256), lds, 0, out, chunks);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double flops = (double)reps * grid * 4 * chunks * 9 * 8 * (32.0 * 32 * 16 * 2);
    const double tf = flops / (ms * 1e-3) / 1e12;
    unsigned long long st[2];
    CK(hipMemcpy(st, reinterpret_cast<char *>(out) + (60 << 20), 16, hipMemcpyDeviceToHost));
    printf("%-34s 32x32x16 wg/cu=%d  %8.3f ms  %7.1f TFLOP/s (%.3f of 2500)  clock %.0f MHz\n", name, wg_per_cu, ms / reps, MODE == 2 ? 0.0 : tf,
           MODE == 2 ? 0.0 : tf / 2500.0, 100.0 * st[0] / st[1]);
}

// ---- who issues the memory instructions?  The igemm-like loop again (64 x 64 per wave), but per chunk of nine taps each
// MFMA wave's share of the conv kernels' memory work has to be issued by SOMEONE: ND LDS-DMA loads of 1 KB (global -> a scratch
// LDS area nobody reads) and NS stores of 1 KB.  MEM = 0: nobody (the plain loop); 1: the MFMA waves themselves, spread
// between the taps - what k_conv3x3_igemm / k_conv3x3_pp do; 2: four extra PRODUCER waves per workgroup that do nothing else
// (512 threads: wave w + 4 issues for MFMA wave w).  One workgroup per CU in all three (the producer form needs the
// registers of the second wave slot), so MEM = 0 is the lone-wave-per-SIMD ceiling.
typedef unsigned int u32x4 __attribute__((__vector_size__(16)));
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void lab_dma16(const __amdgpu_buffer_rsrc_t rsrc, unsigned lds_dst, unsigned voff, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_dst), "v"(voff), "s"(rsrc), "s"(soff)
                 : "memory", "m0");
}
#pragma clang diagnostic pop

template <int MEM, int ND, int NS, bool SHARED_SRC = false, int WGS = 1>
__global__ __launch_bounds__(MEM == 2 ? 512 : 256) __attribute__((amdgpu_waves_per_eu(MEM == 2 ? 2 : WGS, MEM == 2 ? 2 : WGS)))
void k_mem(float *out, const uint4 *src, uint4 *dst, int chunks) {
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
    constexpr int MF = 4, NF = 4, TW = 32, HALO_W = TW + 2, NPIX_PAD = ((4 * MF * 16 / TW + 2) * HALO_W + 3 + 15) / 16 * 16, BN = 16 * NF;
    uint4 *sA = smem, *sB = smem + 4 * NPIX_PAD;
    constexpr int SCRATCH = 4 * NPIX_PAD + 12 * BN;  // uint4 index where the DMA scratch area starts (4 waves x ND KB)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = tid; i < SCRATCH; i += blockDim.x) {
        unsigned h = (i + 1) * 2654435761u;
        uint4 v;
        h ^= h >> 15; h *= 2246822519u; v.x = (h & 0x807f807fu) | 0x3f803f80u;
        h ^= h >> 13; h *= 3266489917u; v.y = (h & 0x807f807fu) | 0x3f803f80u;
        h ^= h >> 16; h *= 668265263u; v.z = (h & 0x807f807fu) | 0x3f803f80u;
        h ^= h >> 15; h *= 374761393u; v.w = (h & 0x807f807fu) | 0x3f803f80u;
        smem[i] = v;
    }
    __syncthreads();
    // this wave's slice of the global buffers: workgroup b, MFMA wave w -> 64 KB of src (walked round robin) and of dst
    const int mw = __builtin_amdgcn_readfirstlane(wave & 3);
    const auto s_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4 *>(src), 0, 0x7fffffff, 0x00020000);
    const auto d_rsrc = __builtin_amdgcn_make_buffer_rsrc(dst, 0, 0x7fffffff, 0x00020000);
    const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)((blockIdx.x * 4 + mw) * 65536u));
    const unsigned lds_scratch = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void *)(smem + SCRATCH) + mw * ND * 1024u);
    auto mem_op = [&](int cc, int slot) {  // slot 0 .. ND + NS - 1 of chunk cc
        const unsigned off = __builtin_amdgcn_readfirstlane(base + (unsigned)(((cc * (ND + NS) + slot) & 63) * 1024u));
        if (slot < ND) lab_dma16(s_rsrc, lds_scratch + slot * 1024u, lane * 16u, SHARED_SRC ? (off & 0xffffu) : off);
        else __builtin_amdgcn_raw_buffer_store_b128(u32x4{(unsigned)cc, (unsigned)slot, (unsigned)lane, 0u}, d_rsrc, lane * 16u, off, 0);
    };
    if (MEM == 2 && wave >= 4) {  // a producer wave: the memory instructions of MFMA wave (wave - 4), paced by chunk count only
        for (int cc = 0; cc < chunks; ++cc) {
#pragma unroll
            for (int slot = 0; slot < ND + NS; ++slot) mem_op(cc, slot);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // (stays a few instructions ahead, never unboundedly)
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    int a_base[MF];
#pragma unroll
    for (int i = 0; i < MF; ++i) {
        const int pb = (wave * MF + i) * 16;
        a_base[i] = (lane >> 4) * NPIX_PAD + (pb / TW) * HALO_W + pb % TW + (lane & 15);
    }
    const int b_base = (lane >> 4) * BN + (lane & 15);
    f32x4 acc[MF][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 af[MF], bfr[2][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i) af[i] = __builtin_bit_cast(bf16x8, sA[a_base[i]]);
#pragma unroll
    for (int j = 0; j < NF; ++j) bfr[0][j] = bfr[1][j] = __builtin_bit_cast(bf16x8, sB[b_base + j * 16]);
    for (int cc = 0; cc < chunks; ++cc) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int nt = (tap + 1) % 9;
#pragma unroll
            for (int i = 0; i < MF; ++i) {
#pragma unroll
                for (int j = 0; j < NF; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[tap & 1][j], af[i], acc[i][j], 0, 0, 0);
                af[i] = __builtin_bit_cast(bf16x8, sA[a_base[i] + (nt / 3) * HALO_W + nt % 3]);
                bfr[(tap + 1) & 1][i] = __builtin_bit_cast(bf16x8, sB[b_base + (nt % 3) * 4 * BN + i * 16]);
            }
            if (MEM == 1) {  // this wave's memory instructions, dealt out between the taps
#pragma unroll
                for (int slot = tap; slot < ND + NS; slot += 9) mem_op(cc, slot);
            }
        }
        if (MEM == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    if (MEM == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (s == 12345.678f) out[blockIdx.x * 256 + tid] = s;
    if (blockIdx.x == 0 && tid == 0) {
        unsigned long long *st = reinterpret_cast<unsigned long long *>(out + (60 << 20) / 4);
        st[0] = __builtin_amdgcn_s_memtime() - c0;
        st[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

template <int MEM, int ND, int NS, bool SHARED_SRC = false, int WGS = 1>
void run_mem(const char *name, float *out, const uint4 *src, uint4 *dst, int chunks = 1024) {
    const int grid = 256 * WGS, lds = 150 * 1024 / WGS;  // WGS workgroups per CU (by their LDS request)
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_mem<MEM, ND, NS, SHARED_SRC, WGS>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int threads = MEM == 2 ? 512 : 256;
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_mem<MEM, ND, NS, SHARED_SRC, WGS>), dim3(grid), dim3(threads), lds, 0, out, src, dst, chunks);
    CK(hipDeviceSynchronize());
    const int reps = 20;
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_mem<MEM, ND, NS, SHARED_SRC, WGS>), dim3(grid), dim3(threads), lds, 0, out, src, dst, chunks);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double flops = (double)reps * grid * 4 * chunks * 9 * 16 * (16.0 * 16 * 32 * 2);
    const double bytes = (double)reps * grid * 4 * chunks * (ND + NS) * 1024.0;
    unsigned long long st[2];
    CK(hipMemcpy(st, reinterpret_cast<char *>(out) + (60 << 20), 16, hipMemcpyDeviceToHost));
    // (workgroup 0's first MFMA wave stamps its own life: with producer waves the launch lasts as long as THEIR traffic takes)
    const double wave_ms = st[1] / 100.0 * 1e-3;
    printf("%-44s ND=%2d NS=%d  launch %7.3f ms  %7.1f TFLOP/s (%.3f)  memory instr. %5.2f TB/s  |  MFMA wave 0 alive %7.3f ms = %7.1f "
           "TFLOP/s at its rate  clock %.0f MHz\n", name, ND, NS, ms / reps, flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 1e12 / 2500.0,
           MEM ? bytes / (ms * 1e-3) / 1e12 : 0.0, wave_ms, flops / reps / (wave_ms * 1e-3) / 1e12, 100.0 * st[0] / st[1]);
}

// `mfma_lds_lab sustain [seconds]`: the igemm-like loop, then the MFMA-only loop, each for `seconds` (default 6) of
// back-to-back 10-ms launches, one line per second - run it under tools/smi_probe.sh to see what the card sustains at its
// power limit (the 50-ms measurements above end before the power controller has pulled the clocks down).
template <int MODE>
void sustain(const char *name, float *out, double seconds) {
    const int grid = 256 * 2 * 4, lds = 160 * 1024 / 2 - 1024, chunks = 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_loop<4, 4, MODE, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    double total_ms = 0;
    while (total_ms < seconds * 1e3) {
        const int reps = 100;
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_loop<4, 4, MODE, 2>), dim3(grid), dim3(256), lds, 0, out, chunks);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        total_ms += ms;
        const double flops = (double)reps * grid * 4 * chunks * 9 * 16 * (16.0 * 16 * 32 * 2);
        unsigned long long st[2];
        CK(hipMemcpy(st, reinterpret_cast<char *>(out) + (60 << 20), 16, hipMemcpyDeviceToHost));
        printf("%-18s t=%5.1f s  %7.1f TFLOP/s (%.3f of 2500)  in-kernel clock %.0f MHz\n", name, total_ms * 1e-3,
               flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 1e12 / 2500.0, 100.0 * st[0] / st[1]);
        fflush(stdout);
    }
}

int main(int argc, char **argv) {
    float *out;
    CK(hipMalloc(&out, 64 << 20));
    if (argc > 1 && !strcmp(argv[1], "mem")) {
        uint4 *src, *dst;
        CK(hipMalloc(&src, 512u * 4 * 65536));
        CK(hipMalloc(&dst, 512u * 4 * 65536));
        CK(hipMemset(src, 1, 512u * 4 * 65536));
        run_mem<0, 10, 6>("no memory instructions (1 wave / SIMD)", out, src, dst);
        run_mem<1, 10, 6>("MFMA waves issue them (as the conv kernels)", out, src, dst);
        run_mem<2, 10, 6>("four producer waves issue them", out, src, dst);
        run_mem<1, 5, 3>("MFMA waves issue them, half the traffic", out, src, dst);
        run_mem<2, 5, 3>("four producer waves, half the traffic", out, src, dst);
        run_mem<1, 10, 0, true>("MFMA waves: loads only, L2-resident source", out, src, dst);
        run_mem<2, 10, 0, true>("producer waves: loads only, L2-resident", out, src, dst);
        run_mem<1, 10, 3, true>("MFMA waves: L2 loads + 3 stores", out, src, dst);
        run_mem<2, 10, 3, true>("producer waves: L2 loads + 3 stores", out, src, dst);
        run_mem<0, 10, 3, true, 2>("2 workgroups / CU: no memory instructions", out, src, dst);
        run_mem<1, 10, 0, true, 2>("2 workgroups / CU: MFMA waves, L2 loads", out, src, dst);
        run_mem<1, 10, 3, true, 2>("2 workgroups / CU: MFMA waves, L2 loads + 3 stores", out, src, dst);
        run_mem<1, 10, 6, true, 2>("2 workgroups / CU: MFMA waves, L2 loads + 6 stores", out, src, dst);
        run_mem<1, 10, 0>("MFMA waves: loads only", out, src, dst);
        run_mem<1, 0, 6>("MFMA waves: stores only", out, src, dst);
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "sustain")) {
        const double seconds = argc > 2 ? atof(argv[2]) : 6.0;
        sustain<0>("igemm-like 64x64", out, seconds);
        sustain<1>("mfma only 64x64", out, seconds);
        return 0;
    }
    run<4, 4, 1, 2>("mfma only 64x64", 2, out);
    run<4, 4, 1, 1>("mfma only 64x64", 1, out);
    run<4, 4, 2, 2>("lds only 64x64", 2, out);
    run<4, 4, 2, 1>("lds only 64x64", 1, out);
    run<4, 4, 0, 2>("igemm-like 64x64", 2, out);
    run<4, 4, 0, 1>("igemm-like 64x64", 1, out);
    run<4, 4, 0, 3>("igemm-like 64x64", 3, out);
    run<8, 4, 0, 2>("128x64 per wave", 2, out);
    run<8, 4, 0, 1>("128x64 per wave", 1, out);
    run<4, 8, 0, 2>("64x128 per wave", 2, out);
    run<4, 8, 0, 1>("64x128 per wave", 1, out);
    run<8, 8, 0, 1>("128x128 per wave", 1, out);
    run<2, 4, 0, 2>("32x64 per wave", 2, out);
    run<2, 4, 0, 4>("32x64 per wave", 4, out);
    // steady-state clocks: ~50 ms per measurement
    printf("-- long runs (1024 chunks per workgroup)\n");
    run<4, 4, 1, 2>("mfma only 64x64", 2, out, 1024);
    run32<1, 2>("mfma only 64x64", 2, out, 1024);
    run<4, 4, 0, 2>("igemm-like 64x64", 2, out, 1024);
    run32<0, 2>("igemm-like 64x64", 2, out, 1024);
    run<2, 4, 0, 2>("32x64 per wave", 2, out, 1024);
    run<4, 4, 0, 1>("igemm-like 64x64", 1, out, 1024);
    run32<0, 1>("igemm-like 64x64", 1, out, 1024);
    return 0;
}
