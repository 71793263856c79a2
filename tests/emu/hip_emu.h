// TEST INFRASTRUCTURE ONLY.  CPU emulation of the HIP execution model used by the kernels in
// wave-u-net-for-speech-enhancement_amd/csrc: one OS thread runs one 256-thread workgroup as 256
// cooperatively scheduled fibers (hand-rolled x86-64 context switch), with LDS (static / dynamic),
// s_barrier, 64-lane wave shuffles and the v_mfma_f32_16x16x4_f32 lane layout
// (cdna_hip_programming.md §3: lane l holds A[l&15][l>>4], B[l>>4][l&15]; D[row=(l>>4)*4+r][col=l&15]).
// It lets the GPU-less build container execute the exact kernel sources on tiny shapes.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float4 { float x, y, z, w; };
struct float2 { float x, y; };

namespace emu {
struct FiberState { dim3 tidx; int op_parity; };
struct BlockState {
    dim3 bidx, bdim, gdim;
    float* dyn_smem;
    float wave_a[4][2][64];
    float wave_b[4][2][64];
};
FiberState& cur_fiber();
BlockState& cur_block();
void block_barrier();
void wave_barrier();
void launch(dim3 grid, dim3 block, size_t smem_bytes, const std::function<void()>& body);
}  // namespace emu

#define threadIdx (emu::cur_fiber().tidx)
#define blockIdx (emu::cur_block().bidx)
#define blockDim (emu::cur_block().bdim)
#define gridDim (emu::cur_block().gdim)
#define WUNET_DYN_SMEM(name) float* name = emu::cur_block().dyn_smem

inline void __syncthreads() { emu::block_barrier(); }

struct wunet_f4 {
    float v[4];
    float& operator[](int i) { return v[i]; }
    const float& operator[](int i) const { return v[i]; }
};

inline wunet_f4 wunet_ld4(const float* p) { wunet_f4 r; std::memcpy(r.v, p, 16); return r; }
inline void wunet_st4(float* p, wunet_f4 v) { std::memcpy(p, v.v, 16); }
inline wunet_f4 wunet_sel4(bool ok, wunet_f4 v) { return ok ? v : wunet_f4{{0.f, 0.f, 0.f, 0.f}}; }

inline wunet_f4 wunet_mfma16(float a, float b, wunet_f4 c)
{
    emu::FiberState& f = emu::cur_fiber();
    emu::BlockState& blk = emu::cur_block();
    const int lane = f.tidx.x & 63, wave = f.tidx.x >> 6, par = f.op_parity;
    f.op_parity ^= 1;
    blk.wave_a[wave][par][lane] = a;
    blk.wave_b[wave][par][lane] = b;
    emu::wave_barrier();
    const int col = lane & 15;
    for (int r = 0; r < 4; ++r) {
        const int row = (lane >> 4) * 4 + r;
        float d = c[r];
        for (int k = 0; k < 4; ++k) d = fmaf(blk.wave_a[wave][par][k * 16 + row], blk.wave_b[wave][par][k * 16 + col], d);
        c[r] = d;
    }
    return c;
}

inline float wunet_shfl_xor(float v, int mask)
{
    emu::FiberState& f = emu::cur_fiber();
    emu::BlockState& blk = emu::cur_block();
    const int lane = f.tidx.x & 63, wave = f.tidx.x >> 6, par = f.op_parity;
    f.op_parity ^= 1;
    blk.wave_a[wave][par][lane] = v;
    emu::wave_barrier();
    return blk.wave_a[wave][par][lane ^ mask];
}

// ---- minimal host runtime
typedef void* hipStream_t;
typedef int hipError_t;
typedef void* hipEvent_t;
#define hipSuccess 0
inline hipError_t hipMalloc(void** p, size_t n) { *p = std::malloc(n ? n : 1); return *p ? 0 : 2; }
inline hipError_t hipFree(void* p) { std::free(p); return 0; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { std::memset(p, v, n); return 0; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { std::memcpy(d, s, n); return 0; }
#define hipMemcpyDeviceToDevice 3
inline hipError_t hipGetLastError() { return 0; }
#define hipStreamNonBlocking 1
#define hipEventDisableTiming 2
inline hipError_t hipGetDevice(int* d) { *d = 0; return 0; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (hipStream_t)1; return 0; }
inline hipError_t hipStreamDestroy(hipStream_t) { return 0; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (hipEvent_t)1; return 0; }
inline hipError_t hipEventDestroy(hipEvent_t) { return 0; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return 0; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
inline const char* hipGetErrorString(hipError_t) { return "emu"; }
