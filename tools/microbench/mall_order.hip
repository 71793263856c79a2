// Does the 256 MiB Infinity Cache (MALL) reward a consumer that reads a freshly written tensor in the REVERSE of the order it was
// written in?  Producer writes N bytes ascending; consumer reads them (a) ascending (the last-written part is still cached when the
// consumer starts, but it reaches it last - by then its own reads may have evicted it), (b) descending (newest first).
// Sizes around the cache: 64 MB .. 512 MB.  Also the copy-kernel bandwidth itself (read + write) as the 6.3 TB/s reference point.
// (tools/microbench: measurement only, not part of the library)   hipcc --offload-arch=gfx950 -O3 -o _mall_order mall_order.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void writer(f4* p, size_t n4, float v)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) p[i] = f4{v, v, v, v};
}
// grid-stride in chunks of 256 float4 (4 KiB): chunk order ascending or descending
template <bool REV>
__global__ __launch_bounds__(256) void reader(const f4* p, size_t n4, float* out)
{
    const size_t chunks = n4 / 256;
    f4 acc = {0, 0, 0, 0};
    for (size_t c = blockIdx.x; c < chunks; c += gridDim.x) {
        const size_t cc = REV ? chunks - 1 - c : c;
        const f4 v = p[cc * 256 + threadIdx.x];
        acc += v;
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = acc[0];
}
// streaming transform: reads src (REV or not), writes dst ascending in ITS OWN index space (same chunk index): the consumer also writes
template <bool REV>
__global__ __launch_bounds__(256) void copier(const f4* src, f4* dst, size_t n4)
{
    const size_t chunks = n4 / 256;
    for (size_t c = blockIdx.x; c < chunks; c += gridDim.x) {
        const size_t cc = REV ? chunks - 1 - c : c;
        dst[cc * 256 + threadIdx.x] = src[cc * 256 + threadIdx.x] * 1.0001f;
    }
}

int main()
{
    const size_t maxb = (size_t)1 << 30;
    f4 *a, *b; float* out;
    hipMalloc(&a, maxb); hipMalloc(&b, maxb); hipMalloc(&out, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * 8;
    printf("%8s %12s %12s %12s %12s %12s\n", "MB", "rd_asc GB/s", "rd_desc GB/s", "cp_asc GB/s", "cp_desc GB/s", "wr GB/s");
    for (size_t mb : {32, 64, 100, 128, 160, 200, 256, 302, 400, 512, 1024}) {
        const size_t bytes = mb << 20, n4 = bytes / 16;
        float t[5] = {0, 0, 0, 0, 0};
        const int reps = 10;
        for (int mode = 0; mode < 5; ++mode) {
            float tot = 0.f;
            for (int r = 0; r < reps + 2; ++r) {
                // flush-ish: touch the other buffer so the cache does not hold `a` from the previous repetition
                writer<<<grid, 256>>>(b, maxb / 16, 1.0f);
                hipEventRecord(e0);                     // (mode 4: the write itself is what is timed)
                writer<<<grid, 256>>>(a, n4, 2.0f);
                if (mode == 4) { hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (r >= 2) tot += ms; continue; }
                hipEventRecord(e0);
                if (mode == 0) reader<false><<<grid, 256>>>(a, n4, out);
                if (mode == 1) reader<true><<<grid, 256>>>(a, n4, out);
                if (mode == 2) copier<false><<<grid, 256>>>(a, b, n4);
                if (mode == 3) copier<true><<<grid, 256>>>(a, b, n4);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (r >= 2) tot += ms;
            }
            t[mode] = tot / reps;
        }
        printf("%8zu %12.0f %12.0f %12.0f %12.0f %12.0f\n", mb, bytes / t[0] / 1e6, bytes / t[1] / 1e6, 2.0 * bytes / t[2] / 1e6, 2.0 * bytes / t[3] / 1e6,
               bytes / t[4] / 1e6);
    }
    return 0;
}
