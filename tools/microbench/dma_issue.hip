// What does an LDS-DMA instruction (global_load_lds_dwordx4: 64 lanes x 16 bytes straight into LDS) cost the wave that issues it, next
// to the MFMAs of the same wave?  The geometry of conv_h3d_kernel: 256-thread blocks, 2 per CU (64 KB of LDS each), per "stage" 180
// independent f16 MFMAs and 16 DMA instructions per wave (64 KB per block).  Modes:
//   0  MFMAs only                       1  DMAs only (burst, vmcnt(0) + barrier per stage)
//   2  burst of 16 DMAs, then the MFMAs 3  one DMA after every 11 MFMAs
// each with the source L2-resident (the same 64 KB per block every stage: "hot") or walking a 1 GiB buffer ("cold": HBM).
// Wall time from HIP events, cycles from s_memtime.  (tools/microbench: measurement only)
//   hipcc --offload-arch=gfx950 -O3 -o _dma_issue dma_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(const void* sbase, unsigned voff, unsigned lds_wave_base)
{
    const unsigned lds = __builtin_amdgcn_readfirstlane(lds_wave_base);
    const unsigned long long b = (unsigned long long)(__UINTPTR_TYPE__)sbase;
    const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)b), bhi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    const unsigned long long sb = ((unsigned long long)bhi << 32) | blo;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sb), "s"(lds) : "memory");
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void stage_loop(const char* src, size_t per_stage_stride, size_t wrap, int kmask, float* out,
                                                      unsigned long long* ticks, int stages, float seed)
{
    extern __shared__ char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned lds0 = (unsigned)(size_t)smem;
    h8 av, bv;
    unsigned x = tid * 2654435761u + blockIdx.x * 40503u + (unsigned)seed;
    for (int e = 0; e < 8; ++e) {
        x = x * 1664525u + 1013904223u; av[e] = (_Float16)(((int)(x >> 20) - 2048) * 0.001f);
        x = x * 1664525u + 1013904223u; bv[e] = (_Float16)(((int)(x >> 20) - 2048) * 0.001f);
    }
    f4 acc[12];
    for (int i = 0; i < 12; ++i) acc[i] = f4{0, 0, 0, 0};
    const unsigned voff = (unsigned)lane * 16u;
    size_t pos = (size_t)blockIdx.x * 65536;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int st = 0; st < stages; ++st) {
        const char* base = src + pos + (size_t)wave * 1024;
        if (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int k = 0; k < 16; ++k) dma16(base + (k & kmask) * 4096, voff, lds0 + (k * 4 + wave) * 1024);
        }
        if (MODE != 1) {
#pragma unroll
            for (int r = 0; r < 15; ++r) {
#pragma unroll
                for (int i = 0; i < 12; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, acc[i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (MODE == 3) {
                    dma16(base + (r & kmask) * 4096, voff, lds0 + (r * 4 + wave) * 1024);
                    if (r == 14) dma16(base + (15 & kmask) * 4096, voff, lds0 + (15 * 4 + wave) * 1024);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        pos += per_stage_stride;
        if (pos >= wrap) pos -= wrap;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
    for (int i = 0; i < 12; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    r += reinterpret_cast<float*>(smem)[tid];
    if (r == 12345.678f) out[0] = r;
    if (tid == 0) ticks[blockIdx.x] = t1 - t0;
}

int main(int argc, char** argv)
{
    // optional: <stages> <only: 0 MFMAs, 1 DMAs hot, 2 DMAs cold, 3 burst hot, 4 burst cold> - one long launch for tools/power_probe.sh
    const int arg_stages = argc > 1 ? atoi(argv[1]) : 0, only = argc > 2 ? atoi(argv[2]) : -1;
    const size_t GiB = 1ull << 30;
    char* src; float* out; unsigned long long* ticks;
    hipMalloc(&src, GiB + (1 << 20)); hipMemset(src, 0, GiB + (1 << 20));
    hipMalloc(&out, 64); hipMalloc(&ticks, 512 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 512, stages = arg_stages > 0 ? arg_stages : 200;
    unsigned long long h[512];
    printf("512 blocks x 256 threads, 2 per CU; per stage and wave 180 MFMAs (16x16x32 f16), 16 LDS-DMA instructions (64 KB per block)\n");
    auto run = [&](const char* what, auto kern, bool cold) {
        const size_t stride = cold ? (size_t)blocks * 65536 : 0, wrap = GiB;
        const int kmask = cold ? 15 : 3;      // hot: 16 KB per block, re-read by every stage
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 65536, 0, (const char*)src, stride, wrap, kmask, out, ticks, stages, 1.0f);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, ticks, blocks * 8, hipMemcpyDeviceToHost);
        double t = 0; for (int i = 0; i < blocks; ++i) t += (double)h[i]; t /= blocks;
        printf("%-44s %8.3f ms  %8.0f cycles / stage  (%.2f GHz)  DMA %6.2f TB/s chip, %5.1f B/clk/CU\n", what, ms, t / stages,
               t / (ms * 1e6), (double)blocks * stages * 65536 / (ms * 1e-3) / 1e12, 2.0 * 65536 / (t / stages));
    };
    hipFuncSetAttribute((const void*)stage_loop<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void*)stage_loop<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void*)stage_loop<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void*)stage_loop<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    if (only < 0 || only == 0) run("MFMAs only (DMA columns do not apply)", stage_loop<0>, false);
    if (only < 0 || only == 1) run("DMAs only, hot (L2)", stage_loop<1>, false);
    if (only < 0 || only == 2) run("DMAs only, cold (HBM)", stage_loop<1>, true);
    if (only < 0 || only == 3) run("burst of 16 DMAs + MFMAs, hot", stage_loop<2>, false);
    if (only < 0 || only == 4) run("burst of 16 DMAs + MFMAs, cold", stage_loop<2>, true);
    if (only < 0) run("one DMA per 11 MFMAs, hot", stage_loop<3>, false);
    if (only < 0) run("one DMA per 11 MFMAs, cold", stage_loop<3>, true);
    if (hipDeviceSynchronize() != hipSuccess) { printf("FAILED\n"); return 1; }
    return 0;
}
