// What does the chip clock at under (a) a dependent-FMA chain on every SIMD (light load), (b) a register-only f16 MFMA loop (the
// matrix pipe at full tilt), and how fast does s_memtime tick?  Each wave runs a chain of N dependent instructions whose issue-to-issue
// latency is known (v_fma_f32: 4 quad... measured as cycles/instr below is relative), so wall time / N gives ns per instruction;
// s_memtime deltas over the same chain give ticks per instruction; ticks / ns = the s_memtime rate.  Wall clock from HIP events.
// (tools/microbench: measurement only)   hipcc --offload-arch=gfx950 -O3 -o _clock_probe clock_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void fma_chain(float* out, unsigned long long* ticks, int iters)
{
    float a = threadIdx.x * 1e-3f, b = 1.000001f, c = 1e-7f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 64; ++k) a = __builtin_fmaf(a, b, c);        // 64 dependent FMAs
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (a == 12345.6789f) out[0] = a;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
// MODE 0: one dependent chain of MFMAs on ONE accumulator (latency-bound: issue-to-issue = the dependent latency);
// MODE 1: 12 independent accumulators (throughput-bound, the pipe full)
template <int MODE>
__global__ __launch_bounds__(256) void mfma_chain(float* out, unsigned long long* ticks, int iters, float seed)
{
    h8 av, bv;
    unsigned x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + (unsigned)seed;
    for (int e = 0; e < 8; ++e) {
        x = x * 1664525u + 1013904223u; av[e] = (_Float16)(((int)(x >> 20) - 2048) * 0.001f);
        x = x * 1664525u + 1013904223u; bv[e] = (_Float16)(((int)(x >> 20) - 2048) * 0.001f);
    }
    f4 acc[12];
    for (int i = 0; i < 12; ++i) acc[i] = f4{0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            if (MODE == 0) acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, acc[0], 0, 0, 0);
            else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, acc[i], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
    for (int i = 0; i < 12; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (r == 12345.678f) out[0] = r;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

int main()
{
    float* out; unsigned long long* ticks;
    hipMalloc(&out, 64); hipMalloc(&ticks, 8192 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    unsigned long long h[8192];
    auto report = [&](const char* what, float ms, int blocks, double instr_per_wave) {
        hipMemcpy(h, ticks, blocks * 8, hipMemcpyDeviceToHost);
        double t = 0; for (int i = 0; i < blocks; ++i) t += (double)h[i]; t /= blocks;
        printf("%-46s wall %8.3f ms  %7.3f ns/instr  s_memtime %12.0f ticks = %6.3f ticks/instr -> s_memtime rate %6.3f GHz\n", what, ms, ms * 1e6 / instr_per_wave,
               t, t / instr_per_wave, t / (ms * 1e6));
    };
    for (int rep = 0; rep < 2; ++rep) {
        for (int blocks : {256, 2048}) {     // one wave per SIMD / eight
            const int iters = 20000;
            fma_chain<<<blocks, 256>>>(out, ticks, 100);
            hipEventRecord(e0); fma_chain<<<blocks, 256>>>(out, ticks, iters); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            char nm[96]; snprintf(nm, sizeof nm, "fma chain, %d blocks (x64 dependent v_fma)", blocks);
            report(nm, ms, blocks > 8192 ? 8192 : blocks, 64.0 * iters * (blocks > 2048 ? 1 : 1));
        }
        {
            const int iters = 100000;
            mfma_chain<0><<<256, 256>>>(out, ticks, 100, 1.f);
            hipEventRecord(e0); mfma_chain<0><<<256, 256>>>(out, ticks, iters, 2.f); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            report("mfma 16x16x32 f16 dependent chain, 1 wave/SIMD", ms, 256, 12.0 * iters);
        }
        for (int blocks : {256, 512}) {
            const int iters = 100000;
            mfma_chain<1><<<blocks, 256>>>(out, ticks, 100, 1.f);
            hipEventRecord(e0); mfma_chain<1><<<blocks, 256>>>(out, ticks, iters, 2.f); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            char nm[96]; snprintf(nm, sizeof nm, "mfma 16x16x32 f16 x12 independent, %d blocks", blocks);
            report(nm, ms, blocks, 12.0 * iters);
            printf("    -> chip rate %.1f TFLOP/s f16 (%d waves/SIMD)\n", (double)blocks * 4 * 12.0 * iters * 16384.0 / (ms * 1e-3) / 1e12, blocks / 256);
        }
    }
    return 0;
}
