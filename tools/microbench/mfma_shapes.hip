// Sustained f16 MFMA rate of the whole chip for the two shapes (same FLOPs per instruction-cycle on paper): does the chip hold a
// higher clock on v_mfma_f32_32x32x16_f16 than on v_mfma_f32_16x16x32_f16?  (tools/microbench: measurement only, not part of the library)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters, float seed)
{
    // four different, noisy operand registers each (constant operands toggle nothing and flatter the power-limited clock)
    h8 av[4], bv[4];
    unsigned x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + (unsigned)seed;
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 8; ++e) {
            x = x * 1664525u + 1013904223u; av[i][e] = (_Float16)(((int)(x >> 20) - 2048) * 0.001f);
            x = x * 1664525u + 1013904223u; bv[i][e] = (_Float16)(((int)(x >> 20) - 2048) * 0.001f);
        }
    float r = 0.f;
    if (SHAPE == 16) {
        f4 acc[12];
        for (int i = 0; i < 12; ++i) acc[i] = f4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 12; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[i & 3], bv[(i >> 2) & 3], acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 12; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        constexpr int NA = SHAPE == 32 ? 3 : 6;          // SHAPE 36: six accumulators (96 registers) - the latency of the 32x32 shape covered
        f16v acc[NA];
        for (int i = 0; i < NA; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 6 / NA; ++rep)
#pragma unroll
                for (int i = 0; i < NA; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[(i + rep) & 3], bv[(i * 2 + rep) & 3], acc[i], 0, 0, 0);
        }
        for (int i = 0; i < NA; ++i) for (int e = 0; e < 16; ++e) r += acc[i][e];
    }
    if (r == 12345.678f) out[0] = r;
}

int main(int argc, char** argv)
{
    float* out; hipMalloc(&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 512, iters = argc > 1 ? atoi(argv[1]) : 20000;      // (2000000: launches of ~0.4 s, the clock settled at the power limit)
    for (int w = 0; w < 6; ++w) hipLaunchKernelGGL(k<16>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);     // warm the clocks
    for (int rep = 0; rep < 3; ++rep)
        for (int shape : {16, 32, 36}) {
            hipEventRecord(e0, 0);
            if (shape == 16) hipLaunchKernelGGL(k<16>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
            else if (shape == 32) hipLaunchKernelGGL(k<32>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
            else hipLaunchKernelGGL(k<36>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            // FLOPs: 16x16x32: 2*16*16*32 = 16384 per MFMA, 12 per iteration; 32x32x16: 2*32*32*16 = 32768, 6 per iteration
            const double fl = (double)blocks * 4 * iters * (shape == 16 ? 12 * 16384.0 : 6 * 32768.0);
            printf("shape %s: %.3f ms  %.0f TFLOP/s\n", shape == 16 ? "16x16x32, 12 accumulators" : shape == 32 ? "32x32x16, 3 accumulators" : "32x32x16, 6 accumulators", ms, fl / (ms * 1e-3) / 1e12);
        }
    return 0;
}
