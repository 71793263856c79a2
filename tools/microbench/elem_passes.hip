// The backward's two elementwise passes of one decoder layer (pass_a_kernel<A_UPH>, gz_split_h3_kernel in UPH mode), the library's own
// kernels on the layer's geometry, next to yardstick kernels that move the same bytes with nothing else in them - cold (a 1 GiB sweep
// between launches evicts the 256 MiB Infinity Cache) and warm (back to back).  Says how far the passes are from what the memory system
// gives THIS access pattern (two 100 MB streams read, one written), not a 1 GiB copy.
// (tools/microbench: measurement only)   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I<csrc> -I<include> -o _elem_passes elem_passes.hip
#include "wunet_h3_elem.h"
#include <cstdio>
#include <vector>

// yardsticks --------------------------------------------------------------------------------------------------------------------
// read two arrays (16 B per lane and array), one partial sum per block
template <int U>
__global__ __launch_bounds__(256) void y_read2(const wunet_f4* __restrict__ a, const wunet_f4* __restrict__ b, size_t n4, float* out)
{
    wunet_f4 acc = {0, 0, 0, 0};
    const size_t chunk = 256 * U;
    for (size_t c = blockIdx.x; (c + 1) * chunk <= n4; c += gridDim.x) {
        wunet_f4 v[U], w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { v[u] = a[c * chunk + u * 256 + threadIdx.x]; w[u] = b[c * chunk + u * 256 + threadIdx.x]; }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u] * w[u];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = acc[0];
}
// read two arrays, write one of the same size
template <int U>
__global__ __launch_bounds__(256) void y_read2_write1(const wunet_f4* __restrict__ a, const wunet_f4* __restrict__ b, wunet_f4* __restrict__ d, size_t n4)
{
    const size_t chunk = 256 * U;
    for (size_t c = blockIdx.x; (c + 1) * chunk <= n4; c += gridDim.x) {
        wunet_f4 v[U], w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { v[u] = a[c * chunk + u * 256 + threadIdx.x]; w[u] = b[c * chunk + u * 256 + threadIdx.x]; }
#pragma unroll
        for (int u = 0; u < U; ++u) d[c * chunk + u * 256 + threadIdx.x] = v[u] * 1.0001f + w[u];
    }
}
__global__ __launch_bounds__(256) void sweep(wunet_f4* p, size_t n4)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) p[i] = p[i] * 1.0001f;
}

int main(int argc, char** argv)
{
    const int B = 64, C = argc > 1 ? atoi(argv[1]) : 48, L = argc > 2 ? atoi(argv[2]) : 8192;
    int logL = 0; while ((1 << logL) < L) ++logL;
    const int C8 = (C + 7) / 8;
    const size_t n = (size_t)B * C * L, nh = (size_t)B * C8 * 8 * L;
    float *z, *gq, *dst, *small, *part, *pmax, *sp, *sc, *out;
    wunet_half *hi, *lo;
    wunet_f4* big;
    const size_t bigb = (size_t)1 << 30;
    hipMalloc(&z, n * 4); hipMalloc(&gq, n * 4); hipMalloc(&dst, n * 4); hipMalloc(&hi, nh * 2 + 64); hipMalloc(&lo, nh * 2 + 64);
    hipMalloc(&small, 64 * 1024 * 4); hipMalloc(&part, 1 << 22); hipMalloc(&pmax, 1 << 22); hipMalloc(&sp, 1 << 24); hipMalloc(&sc, 64); hipMalloc(&out, 64);
    hipMalloc(&big, bigb);
    hipMemset(big, 0, bigb); hipMemset(sp, 0, 1 << 24);
    {
        std::vector<float> h(n);
        unsigned s = 12345u;
        for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((int)(s >> 8) - (1 << 23)) * (1.0f / (1 << 23)); }
        hipMemcpy(z, h.data(), n * 4, hipMemcpyHostToDevice);
        for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((int)(s >> 8) - (1 << 23)) * (1.0f / (1 << 26)); }
        hipMemcpy(gq, h.data(), n * 4, hipMemcpyHostToDevice);
        std::vector<float> k(64 * 1024, 0.5f);
        hipMemcpy(small, k.data(), k.size() * 4, hipMemcpyHostToDevice);
    }
    // per-channel constant rows (padded to 64 floats each), all 0.5
    float* a_ = small; float* s_ = small + 1024; float* mean = small + 2048; float* rstd = small + 3072;
    float* k1 = small + 4096; float* k2 = small + 5120; float* k3 = small + 6144; float* bound = small + 7168;

    // the library's launch geometry (wunet_plan.cpp: a_split keeps a block at 4096 .. 8192 positions)
    int a_split = (int)(((size_t)B * L + 8191) / 8192);
    PassAArgs p{};
    p.z = z; p.a = a_; p.s = s_; p.mean = mean; p.rstd = rstd; p.gpre = nullptr; p.part = part; p.pmax = pmax;
    p.B = B; p.C = C; p.L = L; p.logL = logL; p.Lt = L; p.swap = 1;
    p.g0 = gq; p.sp = sp; p.ntiles = (int)(((size_t)B * 2 * L + 255) / 256); p.tpr = L >> 7;
    const dim3 ga(a_split, C);
    const size_t nt = (size_t)B * C8 * (L / 4);
    size_t hb = (nt + WUNET_THREADS - 1) / WUNET_THREADS; if (hb > 8192) hb = 8192;
    BnBwdArgs F{};
    GzHeadArgs H{}; H.gq = gq; H.a = a_; H.s = s_;

    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t n4 = n / 4;
#define TIME(NAME, BYTES, COLD, LAUNCH)                                                             \
    {                                                                                               \
        float best = 1e9f, tot = 0.f;                                                               \
        for (int r = 0; r < 12; ++r) {                                                              \
            if (COLD) sweep<<<4096, 256>>>(big, bigb / 16);                                         \
            hipEventRecord(e0); LAUNCH; hipEventRecord(e1); hipEventSynchronize(e1);                \
            float ms; hipEventElapsedTime(&ms, e0, e1);                                             \
            if (r >= 2) { tot += ms; if (ms < best) best = ms; }                                    \
        }                                                                                           \
        printf("%-58s %s  mean %7.1f us  best %7.1f us  = %5.0f GB/s\n", NAME, COLD ? "cold" : "warm", tot / 10 * 1e3, best * 1e3, (BYTES) / (tot / 10) / 1e6); \
    }
    setvbuf(stdout, nullptr, _IOLBF, 0);
    printf("B %d C %d L %d: %.1f MB per fp32 array\n", B, C, L, n * 4 / 1e6);
    for (int cold = 1; cold >= 0; --cold) {
        for (int g : {2048, 8192}) {
            char nm[96];
            snprintf(nm, sizeof nm, "yardstick read z, g (4 x 16 B per array), grid %d", g);
            TIME(nm, 8.0 * n, cold, (y_read2<4><<<g, 256>>>((const wunet_f4*)z, (const wunet_f4*)gq, n4, out)))
            snprintf(nm, sizeof nm, "yardstick read z, g (1 x 16 B per array), grid %d", g);
            TIME(nm, 8.0 * n, cold, (y_read2<1><<<g, 256>>>((const wunet_f4*)z, (const wunet_f4*)gq, n4, out)))
        }
        TIME("pass_a_kernel<A_UPH>", 8.0 * n, cold, (pass_a_kernel<A_UPH><<<ga, 256>>>(p)))
        for (int g : {2048, 8192}) {
            char nm[96];
            snprintf(nm, sizeof nm, "yardstick read z, g, write one (4 x 16 B), grid %d", g);
            TIME(nm, 12.0 * n, cold, (y_read2_write1<4><<<g, 256>>>((const wunet_f4*)z, (const wunet_f4*)gq, (wunet_f4*)dst, n4)))
            snprintf(nm, sizeof nm, "yardstick read z, g, write one (1 x 16 B), grid %d", g);
            TIME(nm, 12.0 * n, cold, (y_read2_write1<1><<<g, 256>>>((const wunet_f4*)z, (const wunet_f4*)gq, (wunet_f4*)dst, n4)))
        }
        TIME("gz_split_h3_kernel<false> UPH", 12.0 * n, cold, (gz_split_h3_kernel<false, GZ_UPH, false><<<dim3((unsigned)hb), 256>>>(nullptr, z, k1, k2, k3, bound, sc, hi, lo, B, C, C8, L, logL, L, F, H)))
    }
    hipDeviceSynchronize();
    printf("last error: %s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
