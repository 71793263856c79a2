// Device-side vocabulary of the Wave-U-Net HIP kernels.
//
// Product build (hipcc --offload-arch=gfx950): plain HIP + CDNA4 builtins.
// Test build   (g++ -DWUNET_EMU, tests/emu/): the same kernel sources run on a CPU fiber
// emulator of a 256-thread workgroup (64-lane waves, LDS, s_barrier, the fp32 MFMA lane
// layout) so index math can be validated in the GPU-less build container.  The emulator is
// test infrastructure: the package never loads it.
#pragma once

#ifdef WUNET_EMU
#include "hip_emu.h"
#else
#include <hip/hip_runtime.h>
#define WUNET_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) float name[]
typedef float wunet_f4 __attribute__((ext_vector_type(4)));
// v_mfma_f32_16x16x4_f32: D(16x16) += A(16x4) * B(4x16), exact fp32 fma chain.
// lane l holds A[i=l&15][k=l>>4], B[k=l>>4][j=l&15]; D[row=(l>>4)*4+r][col=l&15] in reg r.
__device__ __forceinline__ wunet_f4 wunet_mfma16(float a, float b, wunet_f4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float wunet_shfl_xor(float v, int mask) { return __shfl_xor(v, mask, 64); }
__device__ __forceinline__ double wunet_shfl_xor_d(double v, int mask) { return __shfl_xor(v, mask, 64); }
// sum of v over the 16 lanes of a row (lanes 16k .. 16k+15), in every lane, by DPP moves (no LDS crossbar, no lgkmcnt wait):
// lane ^ 1, lane ^ 2 inside a quad, then the mirrored lane of the 8-group and of the row - after the quad steps all lanes of a
// quad hold one value and after the third all lanes of an 8-group, so the mirrored partner holds exactly what lane ^ 4 / lane ^ 8
// holds: bit-identical to the shuffle butterfly  v += shfl_xor(v, 1), 2, 4, 8.
__device__ __forceinline__ float wunet_row16_sum(float v)
{
#define WUNET_DPP_ADD(CTRL_) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL_, 0xf, 0xf, false))
    WUNET_DPP_ADD(0xB1);      // quad_perm [1,0,3,2]
    WUNET_DPP_ADD(0x4E);      // quad_perm [2,3,0,1]
    WUNET_DPP_ADD(0x141);     // row_half_mirror
    WUNET_DPP_ADD(0x140);     // row_mirror
#undef WUNET_DPP_ADD
    return v;
}
// maximum of v (>= 0) over the 16 lanes of a row, in every lane, by the same DPP moves
__device__ __forceinline__ float wunet_row16_max(float v)
{
#define WUNET_DPP_MAX(CTRL_) v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL_, 0xf, 0xf, false)))
    WUNET_DPP_MAX(0xB1);
    WUNET_DPP_MAX(0x4E);
    WUNET_DPP_MAX(0x141);
    WUNET_DPP_MAX(0x140);
#undef WUNET_DPP_MAX
    return v;
}
// lane i of a 16-lane row receives lane i - 1's (shr) / lane i + 1's (shl) value by one DPP move; the row's first / last lane receives 0
__device__ __forceinline__ float wunet_row16_shr1(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));      // row_shr:1
}
__device__ __forceinline__ float wunet_row16_shl1(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x101, 0xf, 0xf, true));      // row_shl:1
}
// the value of lane ^ 1 (DPP quad_perm [1, 0, 3, 2])
__device__ __forceinline__ float wunet_lane_swap1(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));
}
// 16-byte staging registers are the NATIVE vector type: arrays of HIP's float4 struct carried across loop
// iterations were demoted to scratch memory by hipcc (global_load -> vmcnt(0) -> scratch_store), which
// silently serialised the software pipeline.
__device__ __forceinline__ wunet_f4 wunet_ld4(const float* p) { return *reinterpret_cast<const wunet_f4*>(p); }
__device__ __forceinline__ void wunet_st4(float* p, wunet_f4 v) { *reinterpret_cast<wunet_f4*>(p) = v; }
// 16 bytes from a 4-byte aligned address (global_load_dwordx4 needs dword alignment only; the type tells the compiler not to assume more)
typedef float wunet_f4u __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ wunet_f4 wunet_ld4u(const float* p) { const wunet_f4u t = *reinterpret_cast<const wunet_f4u*>(p); return wunet_f4{t[0], t[1], t[2], t[3]}; }
__device__ __forceinline__ wunet_f4 wunet_sel4(bool ok, wunet_f4 v) { return ok ? v : wunet_f4{0.f, 0.f, 0.f, 0.f}; }

// ---- fp16 split ("h3") arithmetic: x = hi + lo with hi, lo fp16 (22 significant bits together); a product is
// hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_f16 with fp32 accumulation - fp32-grade accuracy at 3 passes of the
// 2.5 PF matrix pipe.  Halfs travel as raw 16-bit words.
typedef _Float16 wunet_h8 __attribute__((ext_vector_type(8)));     // 8 halfs = 16 bytes = one MFMA operand
typedef unsigned short wunet_half;                                  // storage type in global memory / LDS
__device__ __forceinline__ wunet_half wunet_f2h(float x) { return __builtin_bit_cast(unsigned short, (_Float16)x); }
__device__ __forceinline__ float wunet_h2f(wunet_half h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ wunet_h8 wunet_ldh8(const wunet_half* p) { return *reinterpret_cast<const wunet_h8*>(p); }
__device__ __forceinline__ void wunet_sth8(wunet_half* p, wunet_h8 v) { *reinterpret_cast<wunet_h8*>(p) = v; }
__device__ __forceinline__ void wunet_sth4(wunet_half* p, const wunet_half (&h)[4])
{
    typedef unsigned wunet_u2 __attribute__((ext_vector_type(2)));
    *reinterpret_cast<wunet_u2*>(p) = wunet_u2{(unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16)};
}
__device__ __forceinline__ wunet_h8 wunet_selh8(bool ok, wunet_h8 v) { return ok ? v : wunet_h8{0, 0, 0, 0, 0, 0, 0, 0}; }
__device__ __forceinline__ void wunet_put_half(wunet_h8& v, int e, wunet_half h) { v[e] = __builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ unsigned wunet_fbits(float f) { return __float_as_uint(f); }
// v_mfma_f32_16x16x32_f16: lane l holds A[i=l&15][k=(l>>4)*8..+7], B[k=(l>>4)*8..+7][j=l&15]; D as for 16x16x4.
__device__ __forceinline__ wunet_f4 wunet_mfma16h(wunet_h8 a, wunet_h8 b, wunet_f4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
// v_mfma_f32_16x16x32_bf16, same lane layout: the bf16 mode (one pass on bf16 operands, BASELINE configs[4]) - the 16-bit words
// of a wunet_h8 are then bf16 bit patterns
__device__ __forceinline__ wunet_f4 wunet_mfma16b(wunet_h8 a, wunet_h8 b, wunet_f4 c)
{
    typedef __bf16 wunet_b8 __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(wunet_b8, a), __builtin_bit_cast(wunet_b8, b), c, 0, 0, 0);
}
// Transposed LDS read (ds_read_b64_tr_b16), twice: within each 16-lane group the lanes' 8-byte pieces form a
// [4 rows][16 columns] half matrix (row j = lanes 4j..4j+3, 4 columns each) and lane i receives column i.  Measured on
// gfx950: out[lane i][j] = in[lane 4j + (i >> 2)].half[i & 3].  p0 feeds halfs 0-3 of the result, p1 halfs 4-7: an MFMA
// operand whose 8 consecutive K live in 8 different rows of a K-major LDS image.
__device__ __forceinline__ wunet_h8 wunet_ldtr8(const wunet_half* p0, const wunet_half* p1)
{
    typedef short wunet_s4 __attribute__((ext_vector_type(4)));
    typedef short wunet_s8 __attribute__((ext_vector_type(8)));
    const wunet_s4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wunet_s4 __attribute__((address_space(3)))*)p0);
    const wunet_s4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wunet_s4 __attribute__((address_space(3)))*)p1);
    const wunet_s8 r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(wunet_h8, r);
}

// 8 halfs starting O halfs into the 24 halfs p[0] | p[1] | p[2] (O in 0..16, compile time): the tap shift of the
// weight-gradient B operand, done with v_alignbit on aligned 16-byte LDS pieces
typedef unsigned wunet_u4 __attribute__((ext_vector_type(4)));
template <int O>
__device__ __forceinline__ wunet_h8 wunet_funnel(const wunet_h8 (&p)[3])
{
    const wunet_u4 a = __builtin_bit_cast(wunet_u4, p[0]), b = __builtin_bit_cast(wunet_u4, p[1]), c = __builtin_bit_cast(wunet_u4, p[2]);
    const unsigned d[13] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3], c[0], c[1], c[2], c[3], 0u};
    constexpr int w = O / 2;
    wunet_u4 r;
    if (O & 1) {
        r[0] = __builtin_amdgcn_alignbit(d[w + 1], d[w], 16); r[1] = __builtin_amdgcn_alignbit(d[w + 2], d[w + 1], 16);
        r[2] = __builtin_amdgcn_alignbit(d[w + 3], d[w + 2], 16); r[3] = __builtin_amdgcn_alignbit(d[w + 4], d[w + 3], 16);
    } else {
        r[0] = d[w]; r[1] = d[w + 1]; r[2] = d[w + 2]; r[3] = d[w + 3];
    }
    return __builtin_bit_cast(wunet_h8, r);
}
// global -> LDS DMA, 16 bytes per lane (global_load_lds_dwordx4): lane i's 16 bytes land at lds_wave_base + 16*i; inactive
// lanes write nothing; the data is visible after the issuing waves' vmcnt wait + a barrier (__syncthreads does both)
__device__ __forceinline__ void wunet_dma16(const void* g, void* lds_wave_base)
{
    __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
// all of this wave's DMAs have landed (explicit: __syncthreads() also drains vmcnt while a DMA is in flight, but the data
// hazard should not hang on a compiler habit)
__device__ __forceinline__ void wunet_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// wave issue priority 0..3 (s_setprio)
#define wunet_setprio(N_) __builtin_amdgcn_s_setprio(N_)
// The same DMA issued from inline asm (conv_h3d_kernel): hipcc does not see a VMEM / LDS-DMA instruction, so it neither puts its
// conservative "s_waitcnt vmcnt(0)" in front of every later LDS read (it cannot tell the buffers apart) nor counts the piece in
// its own vmcnt arithmetic (extra pieces in flight only make its counted waits wait longer, loads return in order).  The kernel
// places every wait itself: wunet_wait_dma_barrier() before the first read of a DMA-written buffer, wunet_wait_lds_barrier()
// before a buffer is handed back to the DMA engine.  M0 (the LDS base of the instruction) is saved and restored.
typedef unsigned wunet_lds_t;      // LDS byte address
__device__ __forceinline__ wunet_lds_t wunet_lds_addr(const void* p)
{
    return (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) const char*)p;
}
__device__ __forceinline__ int wunet_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }   // v is the same in every lane of the wave
__device__ __forceinline__ void wunet_dma16a(const void* g, wunet_lds_t lds_wave_base)
{
    const unsigned lds = __builtin_amdgcn_readfirstlane(lds_wave_base);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds) : "memory");
}
// The same DMA in the scalar-base form: address = sbase (wave-uniform, an SGPR pair) + voff (32-bit unsigned byte offset per lane).
// A piece then costs NO per-lane 64-bit address arithmetic: the per-lane part is one stage-invariant register per piece, the part
// that moves with the stage / tile is a scalar add (conv_h3d_kernel: ~9 VALU per piece before - 330 VALU beside the 180 MFMAs of a
// stage, 3.0 VALU per MFMA by the SQ counters, the kernels with the most VALU per MFMA the least busy matrix pipes).
__device__ __forceinline__ void wunet_dma16s(const void* sbase, unsigned voff, wunet_lds_t lds_wave_base)
{
    const unsigned lds = __builtin_amdgcn_readfirstlane(lds_wave_base);
    const unsigned long long b = (unsigned long long)(__UINTPTR_TYPE__)sbase;
    const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)b), bhi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    const unsigned long long sb = ((unsigned long long)bhi << 32) | blo;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sb), "s"(lds) : "memory");
}
// ... skipped when the wave-uniform pred is 0, by a branch INSIDE the asm statement: the caller's code stays one basic block, so
// hipcc can schedule the address arithmetic of the pieces between the MFMAs around them
__device__ __forceinline__ void wunet_dma16a_if(int pred, const void* g, wunet_lds_t lds_wave_base)
{
    const unsigned lds = __builtin_amdgcn_readfirstlane(lds_wave_base);
    pred = __builtin_amdgcn_readfirstlane(pred);      // (an "s" operand the compiler holds in a VGPR is NOT moved to an SGPR for us)
    unsigned keep;
    asm volatile("s_cmp_lg_u32 %3, 0\n\ts_cbranch_scc0 .Lwunet_skip%=\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0\n.Lwunet_skip%=:"
                 : "=&s"(keep) : "v"(g), "s"(lds), "s"(pred) : "memory", "scc");
}
// every DMA piece this wave issued has landed, then the workgroup barrier: all pieces of all waves are visible in LDS
__device__ __forceinline__ void wunet_wait_dma_barrier() { asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory"); }
// every LDS read / write this wave issued has completed, then the barrier (no vmcnt wait: DMAs stay in flight across it)
__device__ __forceinline__ void wunet_wait_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void wunet_wait_lds_barrier_if(int pred)       // (pred wave- and block-uniform; no basic-block split)
{
    pred = __builtin_amdgcn_readfirstlane(pred);
    asm volatile("s_cmp_lg_u32 %0, 0\n\ts_cbranch_scc0 .Lwunet_nobar%=\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier\n.Lwunet_nobar%=:" :: "s"(pred) : "memory", "scc");
}
// (tools/conv_bench.py --trace-realtime builds with WUNET_TRACE_REALTIME: the 100 MHz counter all XCDs share, for block START times)
#ifdef WUNET_TRACE_REALTIME
__device__ __forceinline__ unsigned long long wunet_memtime() { return __builtin_amdgcn_s_memrealtime(); }
#else
__device__ __forceinline__ unsigned long long wunet_memtime() { return __builtin_amdgcn_s_memtime(); }
#endif
// nothing is scheduled across this point (hipcc otherwise sinks LDS prefetches to their first use)
#define wunet_sched_fence() __builtin_amdgcn_sched_barrier(0)
// the value becomes opaque to the optimiser (no hoisting of what is derived from it)
__device__ __forceinline__ void wunet_opaque(int& v) { asm volatile("" : "+v"(v)); }
#endif

// hi/lo fp16 split of s*x (s a power of two chosen so that |s*x| stays far below 65504)
__device__ __forceinline__ void wunet_split_h(float x, wunet_half& hi, wunet_half& lo)
{
    hi = wunet_f2h(x);
    lo = wunet_f2h(x - wunet_h2f(hi));
}

// fp32 -> bf16 bits, round to nearest even (NaN stays NaN); written out so that the GPU and the test emulator agree bit for bit
__device__ __forceinline__ wunet_half wunet_f2b(float x)
{
    unsigned u = wunet_fbits(x);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (wunet_half)((u >> 16) | 0x40u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (wunet_half)(u >> 16);
}
// operand conversion of the GEMM paths: fp16 hi/lo split, or (BF) one bf16 word (lo unused)
template <bool BF>
__device__ __forceinline__ void wunet_split(float x, wunet_half& hi, wunet_half& lo)
{
    if (BF) { hi = wunet_f2b(x); lo = 0; }
    else wunet_split_h(x, hi, lo);
}

// Power-of-two scale 2^k with 2^k * bound in [2^T, 2^(T+1)) - multiplying by it is exact, so the split operands keep their 22 bits
// whatever the magnitude of the tensor (the conv -> BatchNorm pair makes the weight scale a free gauge, gamma / beta set the
// activations').  bound = 0 / inf / nan: scale 1.  fp16 tops out at 65504 < 2^16: T = 13 leaves a factor 4 above the bound.
#define WUNET_SCALE_T 13
__device__ __forceinline__ void wunet_pow2_scale(float bound, float& s, float& inv)
{
    s = 1.0f; inv = 1.0f;
    const unsigned u = wunet_fbits(bound) & 0x7fffffffu;
    if (u >= 0x00800000u && u < 0x7f800000u) {
        int k = WUNET_SCALE_T - ((int)(u >> 23) - 127);
        k = k > 100 ? 100 : (k < -100 ? -100 : k);
        s = ldexpf(1.0f, k);
        inv = ldexpf(1.0f, -k);
    }
}

// max |v| folded into *slot (non-negative floats order like their bit patterns; NaN / inf saturate to the largest finite float so a
// scale derived from it stays defined).  Order independent: the result does not depend on which block arrives when.
__device__ __forceinline__ void wunet_atomic_absmax(float* slot, float v)
{
    const unsigned u = wunet_fbits(v) & 0x7fffffffu, w = u < 0x7f800000u ? u : 0x7f7fffffu;
    // (a maximum only grows: a block whose value is not above what the slot already holds has nothing to add - one plain load instead of
    //  one more atomic on the ONE address every block of the launch goes to)
    if (w > *reinterpret_cast<volatile unsigned*>(slot)) atomicMax(reinterpret_cast<unsigned*>(slot), w);
}
#define WUNET_THREADS 256
#define WUNET_WAVES 4
#define WUNET_SLOPE 0.1f

// LeakyReLU(0.1): slope < 1 so max(v, 0.1 v) is exact and branch-free
__device__ __forceinline__ float wunet_lrelu(float v) { return fmaxf(v, WUNET_SLOPE * v); }

// ATen upsample_linear1d(align_corners=True) source coordinate, fp32 arithmetic on purpose
// (reference model/unet_basic.py:93 -> ATen UpSample.h area_pixel_compute_source_index +
// guard_index_and_lambda).  scale = (float)(Lin-1)/(Lout-1) is computed on the host the same way.
// The multiply must round to fp32 before the subtraction: hipcc's default -ffp-contract=fast would fuse
// `scale*j - i0` into one fma (an exact product), which is 4e-4 away from the reference end to end.
__device__ __forceinline__ void wunet_up_coord(int j, int Lin, float scale, int& i0, int& i1, float& l0, float& l1)
{
#ifdef __clang__
#pragma clang fp contract(off)
#endif
    float src = scale * (float)j;
#if !defined(WUNET_EMU)
    asm volatile("" : "+v"(src));     // belt and braces: the rounded product is opaque to later fusion
#endif
    int a = (int)floorf(src);
    a = a > Lin - 1 ? Lin - 1 : a;
    float lam = src - (float)a;
    lam = lam < 0.0f ? 0.0f : (lam > 1.0f ? 1.0f : lam);
    i0 = a;
    i1 = a + (a < Lin - 1 ? 1 : 0);
    l1 = lam;
    l0 = 1.0f - lam;
}
