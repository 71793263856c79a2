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

#define EMU_MAX_WAVES 8      // workgroups of up to 512 threads
namespace emu {
struct FiberState { dim3 tidx; int op_parity; };
struct BlockState {
    dim3 bidx, bdim, gdim;
    float* dyn_smem;
    float wave_a[EMU_MAX_WAVES][2][64];
    float wave_b[EMU_MAX_WAVES][2][64];
    unsigned short wave_a8[EMU_MAX_WAVES][2][64][8];
    unsigned short wave_b8[EMU_MAX_WAVES][2][64][8];
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
inline wunet_f4 wunet_ld4u(const float* p) { return wunet_ld4(p); }      // (4-byte aligned 16-byte load)
inline void wunet_st4(float* p, wunet_f4 v) { std::memcpy(p, v.v, 16); }
inline wunet_f4 wunet_sel4(bool ok, wunet_f4 v) { return ok ? v : wunet_f4{{0.f, 0.f, 0.f, 0.f}}; }

// ---- fp16 emulation (round to nearest even, subnormals kept), storage as raw 16-bit words
typedef unsigned short wunet_half;
inline wunet_half wunet_f2h(float f)
{
    uint32_t x; std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return (wunet_half)(sign | 0x7c00u | (x > 0x7f800000u ? 0x200u : 0));   // inf / nan
    if (x >= 0x477ff000u) return (wunet_half)(sign | 0x7c00u);                                        // overflow -> inf
    if (x < 0x33000001u) return (wunet_half)sign;                                                     // underflow -> 0
    int e = (int)(x >> 23) - 127;
    uint32_t m = (x & 0x7fffffu) | 0x800000u;
    int shift = e < -14 ? (13 + (-14 - e)) : 13;           // bits to drop
    uint32_t half_m = m >> shift;
    const uint32_t rem = m & ((1u << shift) - 1), halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (half_m & 1))) half_m++;
    uint32_t h = e < -14 ? half_m : (((uint32_t)(e + 15) << 10) + (half_m - 0x400u));
    return (wunet_half)(sign | h);
}
inline float wunet_h2f(wunet_half h)
{
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1f, m = h & 0x3ffu, x;
    if (e == 0) {
        if (m == 0) x = sign;
        else { int sh = 0; while (!(m & 0x400u)) { m <<= 1; ++sh; } x = sign | ((uint32_t)(113 - sh) << 23) | ((m & 0x3ffu) << 13); }
    } else if (e == 31) x = sign | 0x7f800000u | (m << 13);
    else x = sign | ((e + 112) << 23) | (m << 13);
    float f; std::memcpy(&f, &x, 4); return f;
}
struct wunet_h8 { wunet_half v[8]; wunet_half& operator[](int i) { return v[i]; } const wunet_half& operator[](int i) const { return v[i]; } };
inline wunet_h8 wunet_ldh8(const wunet_half* p) { wunet_h8 r; std::memcpy(r.v, p, 16); return r; }
inline void wunet_sth8(wunet_half* p, wunet_h8 v) { std::memcpy(p, v.v, 16); }
inline void wunet_sth4(wunet_half* p, const wunet_half (&h)[4]) { std::memcpy(p, h, 8); }
inline void wunet_put_half(wunet_h8& v, int e, wunet_half h) { v.v[e] = h; }
inline unsigned wunet_fbits(float f) { unsigned u; std::memcpy(&u, &f, 4); return u; }
inline unsigned atomicMax(unsigned* p, unsigned v)
{
    unsigned old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
inline wunet_h8 wunet_selh8(bool ok, wunet_h8 v) { return ok ? v : wunet_h8{{0, 0, 0, 0, 0, 0, 0, 0}}; }

template <int O>
inline wunet_h8 wunet_funnel(const wunet_h8 (&p)[3])
{
    wunet_half all[24];
    for (int m = 0; m < 3; ++m) std::memcpy(all + 8 * m, p[m].v, 16);
    wunet_h8 r;
    for (int e = 0; e < 8; ++e) r.v[e] = all[O + e];
    return r;
}

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

// v_mfma_f32_16x16x32_f16 emulation: products of halfs are exact in fp32, accumulated in k order
inline wunet_f4 wunet_mfma16h(wunet_h8 a, wunet_h8 b, wunet_f4 c)
{
    emu::FiberState& f = emu::cur_fiber();
    emu::BlockState& blk = emu::cur_block();
    const int lane = f.tidx.x & 63, wave = f.tidx.x >> 6, par = f.op_parity;
    f.op_parity ^= 1;
    std::memcpy(&blk.wave_a8[wave][par][lane][0], a.v, 16);
    std::memcpy(&blk.wave_b8[wave][par][lane][0], b.v, 16);
    emu::wave_barrier();
    const int col = lane & 15;
    for (int r = 0; r < 4; ++r) {
        const int row = (lane >> 4) * 4 + r;
        float d = c[r];
        for (int q = 0; q < 4; ++q)
            for (int e = 0; e < 8; ++e)
                d += wunet_h2f(blk.wave_a8[wave][par][q * 16 + row][e]) * wunet_h2f(blk.wave_b8[wave][par][q * 16 + col][e]);
        c[r] = d;
    }
    return c;
}

// v_mfma_f32_16x16x32_bf16 emulation (the words are bf16 bit patterns)
inline wunet_f4 wunet_mfma16b(wunet_h8 a, wunet_h8 b, wunet_f4 c)
{
    emu::FiberState& f = emu::cur_fiber();
    emu::BlockState& blk = emu::cur_block();
    const int lane = f.tidx.x & 63, wave = f.tidx.x >> 6, par = f.op_parity;
    f.op_parity ^= 1;
    std::memcpy(&blk.wave_a8[wave][par][lane][0], a.v, 16);
    std::memcpy(&blk.wave_b8[wave][par][lane][0], b.v, 16);
    emu::wave_barrier();
    const int col = lane & 15;
    auto b2f = [](unsigned short h) { uint32_t x = (uint32_t)h << 16; float v; std::memcpy(&v, &x, 4); return v; };
    for (int r = 0; r < 4; ++r) {
        const int row = (lane >> 4) * 4 + r;
        float d = c[r];
        for (int q = 0; q < 4; ++q)
            for (int e = 0; e < 8; ++e)
                d += b2f(blk.wave_a8[wave][par][q * 16 + row][e]) * b2f(blk.wave_b8[wave][par][q * 16 + col][e]);
        c[r] = d;
    }
    return c;
}

// ds_read_b64_tr_b16 x2 (semantics measured on gfx950, see wunet_dev.h)
inline wunet_h8 wunet_ldtr8(const wunet_half* p0, const wunet_half* p1)
{
    emu::FiberState& f = emu::cur_fiber();
    emu::BlockState& blk = emu::cur_block();
    const int lane = f.tidx.x & 63, wave = f.tidx.x >> 6, par = f.op_parity;
    f.op_parity ^= 1;
    std::memcpy(&blk.wave_a8[wave][par][lane][0], p0, 8);
    std::memcpy(&blk.wave_a8[wave][par][lane][4], p1, 8);
    emu::wave_barrier();
    const int g = lane & ~15, i = lane & 15;
    wunet_h8 r;
    for (int j = 0; j < 4; ++j) {
        r.v[j] = blk.wave_a8[wave][par][g + 4 * j + (i >> 2)][i & 3];
        r.v[4 + j] = blk.wave_a8[wave][par][g + 4 * j + (i >> 2)][4 + (i & 3)];
    }
    return r;
}

#define wunet_setprio(N_) ((void)0)
inline void wunet_dma_wait() {}
// global -> LDS DMA model: immediate copy (the emulator cannot see a missing wait; the GPU parity tests do)
inline void wunet_dma16(const void* g, void* lds_wave_base)
{
    const int lane = emu::cur_fiber().tidx.x & 63;
    std::memcpy(static_cast<char*>(lds_wave_base) + 16 * lane, g, 16);
}

typedef char* wunet_lds_t;
inline wunet_lds_t wunet_lds_addr(const void* p) { return (char*)p; }
inline int wunet_uniform(int v) { return v; }
inline void wunet_dma16a(const void* g, wunet_lds_t lds_wave_base) { wunet_dma16(g, lds_wave_base); }
inline void wunet_dma16s(const void* sbase, unsigned voff, wunet_lds_t lds_wave_base) { wunet_dma16((const char*)sbase + voff, lds_wave_base); }
inline void wunet_dma16a_if(int pred, const void* g, wunet_lds_t lds_wave_base) { if (pred) wunet_dma16(g, lds_wave_base); }
inline void wunet_wait_lds_barrier_if(int pred) { if (pred) emu::block_barrier(); }
inline void wunet_wait_dma_barrier() { emu::block_barrier(); }
inline void wunet_wait_lds_barrier() { emu::block_barrier(); }
inline unsigned long long wunet_memtime() { return 0; }
inline void wunet_opaque(int&) {}
#define wunet_sched_fence() ((void)0)

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

inline double wunet_shfl_xor_d(double v, int mask)
{
    float h[2];
    std::memcpy(h, &v, 8);
    h[0] = wunet_shfl_xor(h[0], mask);
    h[1] = wunet_shfl_xor(h[1], mask);
    std::memcpy(&v, h, 8);
    return v;
}

inline float wunet_row16_sum(float v)
{
    for (int m = 1; m < 16; m <<= 1) v += wunet_shfl_xor(v, m);
    return v;
}

// lane i of a 16-lane row receives lane i - 1's (shr) / lane i + 1's (shl) value; the row's first / last lane receives 0
inline float wunet_row16_shr1(float v)
{
    emu::FiberState& f = emu::cur_fiber();
    emu::BlockState& blk = emu::cur_block();
    const int lane = f.tidx.x & 63, wave = f.tidx.x >> 6, par = f.op_parity;
    f.op_parity ^= 1;
    blk.wave_a[wave][par][lane] = v;
    emu::wave_barrier();
    return (lane & 15) ? blk.wave_a[wave][par][lane - 1] : 0.0f;
}
inline float wunet_row16_shl1(float v)
{
    emu::FiberState& f = emu::cur_fiber();
    emu::BlockState& blk = emu::cur_block();
    const int lane = f.tidx.x & 63, wave = f.tidx.x >> 6, par = f.op_parity;
    f.op_parity ^= 1;
    blk.wave_a[wave][par][lane] = v;
    emu::wave_barrier();
    return (lane & 15) != 15 ? blk.wave_a[wave][par][lane + 1] : 0.0f;
}

inline float wunet_lane_swap1(float v) { return wunet_shfl_xor(v, 1); }

inline float wunet_row16_max(float v)
{
    for (int m = 1; m < 16; m <<= 1) v = std::fmax(v, wunet_shfl_xor(v, m));
    return v;
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
inline hipError_t hipMemcpy2DAsync(void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t height, int, hipStream_t)
{
    for (size_t r = 0; r < height; ++r) std::memcpy(static_cast<char*>(d) + r * dpitch, static_cast<const char*>(s) + r * spitch, width);
    return 0;
}
inline hipError_t hipGetLastError() { return 0; }
#define hipStreamNonBlocking 1
#define hipEventDisableTiming 2
#define hipEventDisableSystemFence 0x20000000
inline hipError_t hipGetDevice(int* d) { *d = 0; return 0; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (hipStream_t)1; return 0; }
inline hipError_t hipStreamDestroy(hipStream_t) { return 0; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (hipEvent_t)1; return 0; }
inline hipError_t hipEventDestroy(hipEvent_t) { return 0; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return 0; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
inline const char* hipGetErrorString(hipError_t) { return "emu"; }
