// Best achievable streaming rates on tensors that do NOT fit the 256 MiB Infinity Cache (1 GiB): read-only, write-only and copy, by
// loads in flight per thread (1 / 2 / 4 / 8 x 16 B), default vs non-temporal, and grid size - the yardstick for the HBM-bound
// elementwise kernels of the library (operand passes, gradient assembly), which move 4.1-5.0 TB/s of counter traffic.
// (tools/microbench: measurement only)   hipcc --offload-arch=gfx950 -O3 -o _stream_rate stream_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int U, bool NT>
__global__ __launch_bounds__(256) void rd(const f4* __restrict__ p, size_t n4, float* out)
{
    f4 acc = {0, 0, 0, 0};
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + u * stride) : p[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = acc[0];
}
template <int U, bool NT>
__global__ __launch_bounds__(256) void wr(f4* __restrict__ p, size_t n4, float v)
{
    const size_t stride = (size_t)gridDim.x * 256;
    const f4 x = {v, v, v, v};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + (U - 1) * stride < n4; i += U * stride) {
#pragma unroll
        for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(x, p + i + u * stride); else p[i + u * stride] = x; }
    }
}
template <int U, bool NT>
__global__ __launch_bounds__(256) void cp(const f4* __restrict__ s, f4* __restrict__ d, size_t n4)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + (U - 1) * stride < n4; i += U * stride) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(s + i + u * stride) : s[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) { const f4 y = v[u] * 1.0001f; if (NT) __builtin_nontemporal_store(y, d + i + u * stride); else d[i + u * stride] = y; }
    }
}
// block-contiguous variant: each block owns a contiguous 64 KiB chunk per iteration (instead of the chip-wide interleave above)
template <int U>
__global__ __launch_bounds__(256) void cp_chunk(const f4* __restrict__ s, f4* __restrict__ d, size_t n4)
{
    const size_t chunk = 256 * U;                       // float4 per block-iteration
    for (size_t c = blockIdx.x; (c + 1) * chunk <= n4; c += gridDim.x) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = s[c * chunk + u * 256 + threadIdx.x];
#pragma unroll
        for (int u = 0; u < U; ++u) d[c * chunk + u * 256 + threadIdx.x] = v[u] * 1.0001f;
    }
}

int main()
{
    const size_t bytes = (size_t)1 << 30, n4 = bytes / 16;
    f4 *a, *b; float* out;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&out, 64);
    hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
#define TIME(NAME, BYTES, LAUNCH)                                                                  \
    {                                                                                              \
        float best = 1e9f, tot = 0.f;                                                              \
        for (int r = 0; r < 7; ++r) {                                                              \
            hipEventRecord(e0); LAUNCH; hipEventRecord(e1); hipEventSynchronize(e1);               \
            float ms; hipEventElapsedTime(&ms, e0, e1);                                            \
            if (r >= 2) { tot += ms; if (ms < best) best = ms; }                                   \
        }                                                                                          \
        printf("%-44s grid %5d  mean %7.0f GB/s  best %7.0f GB/s\n", NAME, g, (BYTES) / (tot / 5) / 1e6, (BYTES) / best / 1e6); \
    }
    for (int g : {1024, 2048, 4096, 8192}) {
        TIME("read  1x16B", bytes, (rd<1, false><<<g, 256>>>(a, n4, out)))
        TIME("read  4x16B", bytes, (rd<4, false><<<g, 256>>>(a, n4, out)))
        TIME("read  8x16B", bytes, (rd<8, false><<<g, 256>>>(a, n4, out)))
        TIME("read  4x16B nt", bytes, (rd<4, true><<<g, 256>>>(a, n4, out)))
        TIME("write 1x16B", bytes, (wr<1, false><<<g, 256>>>(a, n4, 1.f)))
        TIME("write 4x16B", bytes, (wr<4, false><<<g, 256>>>(a, n4, 1.f)))
        TIME("write 4x16B nt", bytes, (wr<4, true><<<g, 256>>>(a, n4, 1.f)))
        TIME("copy  1x16B (r+w bytes)", 2.0 * bytes, (cp<1, false><<<g, 256>>>(a, b, n4)))
        TIME("copy  4x16B (r+w bytes)", 2.0 * bytes, (cp<4, false><<<g, 256>>>(a, b, n4)))
        TIME("copy  8x16B (r+w bytes)", 2.0 * bytes, (cp<8, false><<<g, 256>>>(a, b, n4)))
        TIME("copy  4x16B nt (r+w bytes)", 2.0 * bytes, (cp<4, true><<<g, 256>>>(a, b, n4)))
        TIME("copy  4x16B block-contiguous (r+w bytes)", 2.0 * bytes, (cp_chunk<4><<<g, 256>>>(a, b, n4)))
        TIME("copy  8x16B block-contiguous (r+w bytes)", 2.0 * bytes, (cp_chunk<8><<<g, 256>>>(a, b, n4)))
    }
    {
        int g = 0;
        TIME("hipMemcpyAsync D2D (r+w bytes)", 2.0 * bytes, (hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0)))
    }
    return 0;
}
