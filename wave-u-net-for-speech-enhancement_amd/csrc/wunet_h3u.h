// conv_h3u_kernel: the 5-tap conv of a decoder level (unet_basic.py:25-26, 93-95) with the operand pass INSIDE the kernel.
// conv_h3d_kernel reads its input from the split operand arrays an elementwise pass (prep_h3_kernel) wrote: per value 4 bytes
// written + 4 read, on top of the fp32 sources the pass reads.  Here the block builds the x tile itself from what the producers
// left in HBM - the raw fp32 conv outputs z of the previous decoder level (half resolution) and of the skip encoder level -
// BatchNorm scale / shift, LeakyReLU, the x2 linear upsample with ATen's fp32 coordinates, the concat, the power-of-two scale and
// the hi / lo split all happen on the way into LDS.  prep_h3_kernel disappears for the layer and the conv reads 2 (upsampled
// half) or 4 (skip half) bytes per operand value instead of 4 + 4.
//
// Work split inside the block (512 threads, one block per CU): WAVE SPECIALISATION.  Waves 0-3 are the conv_h3d_kernel MFMA
// waves (same LDS images, same fragments, same MFMA order per accumulator); waves 4-7 are loaders - loader wave w owns channel
// group w of the stage's chunk of 32 channels: global loads -> VALU -> ds_write_b128 of the NEXT stage's x tile, and the LDS-DMA
// of the next W sub-tile.  Every SIMD hosts one MFMA wave and one loader wave, so the matrix pipe and the vector ALU / memory
// pipes of a SIMD are fed by different instruction streams and overlap without any software pipelining inside a wave.  x tile
// and W sub-tile are double-buffered: ONE workgroup barrier per stage (the buffer a stage reads was completed before the
// barrier; the buffer the loaders fill was last read in the previous stage).
// No K tail (the forward pack of such a layer is built without one), no K split (levels >= 256 samples only).
// The loader waves are ISSUE bound (a wave issues one instruction per ~4.4 cycles whatever the unit; profiles/r5_h3u_loader_diet.txt): their
// loop is kept lean - stage cursors instead of divisions, scalar-base loads, per-item interpolation weights, the row-edge selects in an
// instantiation only the first tile of a row takes, the operand scale folded into the BatchNorm coefficients, nothing for empty channel
// groups - and what remains of a stage is mostly the 17 memory instructions per wave (~100 cycles each at the CU's shared address path).
#pragma once
#include "wunet_h3.h"
// WUNET_H3U_ABL: ablation builds of tools/h3u_ablation.sh (parts of the kernel compiled out: 1 no global loads, 4 no MFMAs, 8 no W DMA,
// 16 no conversion at all (no LDS writes), 32 no fragment reads and no MFMAs, 64 the loads kept alive but nothing converted, 128 no LDS writes);
// 0 in the product
#ifndef WUNET_H3U_ABL
#define WUNET_H3U_ABL 0
#endif

struct ConvH3uArgs {
    const float* z0; const float* a0; const float* s0;     // upsampled branch: producer's z [B][C0][L/2], its BatchNorm scale / shift
    const float* z1; const float* a1; const float* s1;     // skip branch: [B][C1][L]
    const float* xb0; const float* xb1;                    // activation bounds of the two producers (their slot [4]): the x scale derives from them
    float* xsc;                                            // {scale, 1/scale} of the operand, published by block 0
    float up_scale;                                        // (float)(Lt/2 - 1) / (Lt - 1), computed on the host like ATen does
    int Lt;                                                // samples of a row that exist (<= L, the row stride)
    int C0, C1, C8;                                        // channels of the two branches (multiples of 8), groups of 8 in all
    const wunet_half* wh; unsigned wdelta;                 // forward pack without K tail; the lo pack lies wdelta bytes behind
    const float* bias; const float* sc2;                   // conv bias; {scale, 1/scale} of the packed weights
    float* out; float* stats;                              // [B][Cout][L]; nullptr or [Cout][ntiles][2]
    const float* ev_a; const float* ev_s; float* xrows;    // eval mode: this layer's BatchNorm scale / shift, its xb slot (see ConvH3Args)
    wunet_half* oxh; wunet_half* oxl;                      // nullptr, or the split operand [B][C8][L][8] written out as well (training: the weight gradient reads it)
    int B, Cout, L, logL, NS, ntiles, mblocks;
    unsigned long long* trace;                             // nullptr, or (measurement builds, -DWUNET_H3U_TRACE) clock stamps of the stages, tools/h3u_trace.py
};

// (the helpers of the elementwise operand pass - wunet_x_scale, wunet_split_rt - live in wunet_h3_elem.h, which the host translation
// units include; the kernel only needs this one)
__device__ __forceinline__ void wunet_h3u_x_scale(const float* xb0, const float* xb1, float& s, float& inv)
{
    wunet_pow2_scale(fmaxf(xb0[0], xb1 ? xb1[0] : 0.0f), s, inv);
}

// Loader waves' barrier: every LDS write and every DMA piece of the wave has landed - but NOT the `younger` (0, 9 or 10, wave-uniform)
// global loads the wave issued AFTER its last DMA piece: those are the prefetch of a tile two stages ahead and stay in flight across
// the barrier (memory operations return in order: at most `younger` outstanding means everything older has completed).  Waiting for
// vmcnt(0) here made every stage one full memory latency long (2.5 - 3 us per stage instead of ~1).
// Training: the loaders also copy the operand to HBM (the weight gradient reads it) - WUNET_H3U_NST store instructions per converted tile,
// issued BEHIND the stage's prefetch loads, so they are younger operations too: a barrier / wait that does not allow for them waits for the
// prefetch loads instead (rounds 5: vmcnt(0) whenever the copy was on - the two-stage prefetch collapsed to one memory latency per
// stage, decoder.11 172 us against 104 us without the copy).  The loaders keep count (wave-uniform: did the tile this wave converted last /
// before last write its copy) and pick the wait that allows exactly the loads AND those stores.  WUNET_H3U_NST is a LOWER bound of the
// stores a copying conversion issues (8 x 16 bytes + 2 x 2 bytes = 10 instructions; counting 8 waits for two of the oldest stores more than
// needed, never for fewer operations than are in front of the loads - tools/check_h3u_isa.py counts the stores of every conversion in
// the ISA of each build).
#define WUNET_H3U_NST 8
__device__ __forceinline__ void wunet_loader_barrier(int younger)
{
#ifdef WUNET_EMU
    (void)younger;
    emu::block_barrier();
#else
    if (younger >= 10 + WUNET_H3U_NST) asm volatile("s_waitcnt vmcnt(18) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else if (younger >= 10) asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else if (younger == 9) asm volatile("s_waitcnt vmcnt(9) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    wunet_sched_fence();
#endif
}

// The per-item barrier of the statistics hand-over (the MFMA waves' row sums go through LDS once per work item): same wait, but it does
// NOT end a stage - the operands are written in the other order (`lgkmcnt(0) vmcnt(10)`) so that tools/check_h3u_isa.py, which counts
// STAGE barriers behind a prefetch load, tells the two kinds apart in the ISA.
__device__ __forceinline__ void wunet_loader_stats_barrier(int younger)
{
#ifdef WUNET_EMU
    (void)younger;
    emu::block_barrier();
#else
    if (younger >= 10 + WUNET_H3U_NST) asm volatile("s_waitcnt lgkmcnt(0) vmcnt(18)\n\ts_barrier" ::: "memory");
    else if (younger >= 10) asm volatile("s_waitcnt lgkmcnt(0) vmcnt(10)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(0) vmcnt(0)\n\ts_barrier" ::: "memory");
    wunet_sched_fence();
#endif
}

// ---- loader side.  Loader wave `cw` builds channel group ch*4 + cw of the x tile [hi|lo][4][COLS, de-interleaved][8]: the 260 columns
// 6 .. 265 a 5-tap fragment reads (column c = sample l0 - 8 + c).  Lane k owns the four samples P .. P+3, P = l0 - 2 + 4k, of the
// group's 8 channels: ONE 16-byte load per channel - skip branch: the samples themselves; upsampled branch: the four source samples
// wb .. wb+3, wb = (P-2)/2, which are exactly what ATen's coordinates of the four outputs touch ((i0, i1) = (wb, wb+1), (wb+1, wb+2),
// (wb+1, wb+2), (wb+2, wb+3): the planner checks exactly that on the host for the layer's length - wunet_plan.cpp up_pairs_regular,
// ATen's fp32 coordinates over the whole row - and a length that fails it keeps prep_h3_kernel + conv_h3d_kernel; the row's first
// outputs take the EDGE instantiation's selects) - so a lane activates 4 source values per channel instead of 8 and a tile costs 8 load instructions per
// wave, few enough to keep THREE tiles in flight (WunetH3uRaw: 34 registers).  The last 4 columns (262 .. 265) are one value per lane
// of lanes 0 .. 31.  Two steps: wunet_h3u_issue (the loads) and, stages later, wunet_h3u_convert (prep_h3_kernel's arithmetic:
// BatchNorm scale / shift, LeakyReLU, ATen's upsample weights, scale, split; LDS writes; in training also the operand's copy in HBM).
struct WunetH3uRaw { wunet_f4 q[8]; float ma, mb; };
struct WunetH3uTile { int b, l0, mt0, ch; };
// Where a loader stands in the block's sequence of stages: chunk `ch` of the work item (position tile, row block mb).  Advanced stage by
// stage with adds and compares: deriving it from the stage number costs two integer divisions per tile - with two tiles per stage ~90 of
// the ~330 instructions a loader wave spent between the stage's barrier and its last prefetch load (a wave issues one instruction per
// ~4.4 cycles whatever the unit: the loaders are ISSUE bound, profiles/r5_h3u_stage_timeline.txt).
struct WunetH3uCursor { int ch, tile, mb; };
struct WunetH3uStep { int NS, mblocks, qG, rG; };                 // G = qG * mblocks + rG: the next item of a block is G items further
__device__ __forceinline__ void wunet_h3u_advance(WunetH3uCursor& c, const WunetH3uStep& S)
{
    if (++c.ch == S.NS) {
        c.ch = 0; c.tile += S.qG; c.mb += S.rG;
        if (c.mb >= S.mblocks) { c.mb -= S.mblocks; ++c.tile; }
    }
}
// ATen's interpolation weights of the lane's four outputs P .. P+3 and of its value of the last 4 columns: they depend on the position
// tile only, so they are computed once per work item (NS stages) and kept - not once per stage.
struct WunetH3uCoord { float w0[4], w1[4], mw0, mw1; };
__device__ __forceinline__ void wunet_h3u_coord(const ConvH3uArgs& A, int l0, int lane, WunetH3uCoord& K)
{
    const int Lth = A.Lt >> 1, P = l0 - 2 + 4 * lane;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int i0, i1;
        wunet_up_coord(P + j < 0 ? 0 : P + j, Lth, A.up_scale, i0, i1, K.w0[j], K.w1[j]);
    }
    const int p_m = l0 + 254 + (lane & 3);
    int i0, i1;
    wunet_up_coord((lane < 32 && p_m < A.Lt) ? p_m : 0, Lth, A.up_scale, i0, i1, K.mw0, K.mw1);
}
// ... and the two source samples of that value (the loads need them three stages before the weights are used)
__device__ __forceinline__ void wunet_h3u_mini_src(const ConvH3uArgs& A, int l0, int lane, int& i0, int& i1)
{
    const int p_m = l0 + 254 + (lane & 3);
    float w0, w1;
    wunet_up_coord((lane < 32 && p_m < A.Lt) ? p_m : 0, A.Lt >> 1, A.up_scale, i0, i1, w0, w1);
}

// The prefetch loads are issued from inline asm: hipcc then neither counts them (its own wait-count bookkeeping put the first use of a tile
// loaded two stages ago behind a wait for the loads issued a moment before - and a wait in front of every re-use of a register it believed
// in flight) nor may it touch their destination registers before the loader's own wait (wunet_vm_wait): the values are asm outputs
// the compiler believes defined, so nothing but the hand-placed s_waitcnt orders their use.  (Checked in the ISA of each build by
// tools/check_h3u_isa.py: no instruction touches these registers between the load and the wait.)
// Scalar-base form: address = sbase (wave-uniform SGPR pair: the channel's row) + voff (the lane's byte offset inside the row, the same
// register for the 8 channels of a group) - the per-channel part of the address is a scalar add, no 64-bit vector arithmetic per load.
__device__ __forceinline__ unsigned long long wunet_sgpr64(unsigned long long b)
{
#ifdef WUNET_EMU
    return b;
#else
    const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)b), bhi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    return ((unsigned long long)bhi << 32) | blo;                  // (folded away where the compiler already knows the value uniform)
#endif
}
__device__ __forceinline__ wunet_f4 wunet_ld4s_async(unsigned long long sbase, unsigned voff)
{
#ifdef WUNET_EMU
    wunet_f4 v;
    std::memcpy(&v, reinterpret_cast<const char*>(sbase) + voff, 16);
    return v;
#else
    wunet_f4 v;
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(voff), "s"(sbase) : "memory");
    return v;
#endif
}
__device__ __forceinline__ float wunet_ld1s_async(unsigned long long sbase, unsigned voff)
{
#ifdef WUNET_EMU
    float v;
    std::memcpy(&v, reinterpret_cast<const char*>(sbase) + voff, 4);
    return v;
#else
    float v;
    asm volatile("global_load_dword %0, %1, %2" : "=v"(v) : "v"(voff), "s"(sbase) : "memory");
    return v;
#endif
}
// at most N memory operations of this wave are still outstanding (they return in order: everything older has landed)
template <int N>
__device__ __forceinline__ void wunet_vm_wait()
{
#ifndef WUNET_EMU
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
    wunet_sched_fence();            // (register-only arithmetic on the loaded values must not be scheduled above the wait: the compiler sees no dependency)
#endif
}
// One LDS-DMA piece of a group that shares ONE save / restore of M0 (wunet_m0_save / wunet_m0_restore around the group): M0 = lds_base + IMM
// is written by the add itself.  3 instructions per piece instead of 6 + the address arithmetic of wunet_dma16s.
__device__ __forceinline__ unsigned wunet_m0_save()
{
#ifdef WUNET_EMU
    return 0;
#else
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0" : "=s"(keep) :: "memory");
    return keep;
#endif
}
__device__ __forceinline__ void wunet_m0_restore(unsigned keep)
{
#ifndef WUNET_EMU
    asm volatile("s_mov_b32 m0, %0" :: "s"(keep) : "memory");
#else
    (void)keep;
#endif
}
#ifdef WUNET_EMU
#define wunet_lds_uniform(X_) (X_)
#else
#define wunet_lds_uniform(X_) ((unsigned)__builtin_amdgcn_readfirstlane((int)(X_)))
#endif
template <int IMM>
__device__ __forceinline__ void wunet_dma16m(unsigned long long sbase, unsigned voff, wunet_lds_t lds_base)
{
#ifdef WUNET_EMU
    wunet_dma16(reinterpret_cast<const char*>(sbase) + voff, lds_base + IMM);
#else
    asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_base), "n"(IMM) : "memory", "scc");
#endif
}

// ONE straight sequence of 10 loads whatever the group is (addresses selected, not branches: a value that is "loaded on one path and kept on
// another" becomes a copy of the loaded registers, and a copy waits for the load - the prefetch would be serialised): a group beyond C8
// reads 16 bytes of the first row ten times (its values are never used), the skip branch's second single load repeats the first.  (mi0, mi1): wunet_h3u_mini_src of the tile's item.  Returns the number of load instructions issued (10).
__device__ __forceinline__ int wunet_h3u_issue(const ConvH3uArgs& A, WunetH3uRaw& R, const WunetH3uTile& T, int cw, int lane, int mi0, int mi1)
{
    if (WUNET_H3U_ABL & 1) return 0;
    const int c8 = wunet_uniform(T.ch * 4 + cw);
    const int L = A.L, Lh = L >> 1, Lth = A.Lt >> 1;
    const bool none = c8 >= A.C8, up = !none && c8 * 8 < A.C0;
    const int P = T.l0 - 2 + 4 * lane;
    const int e_m = (lane >> 2) & 7, p_m = T.l0 + 254 + (lane & 3);    // the value of the last 4 columns this lane owns (lanes 0 .. 31)
    const bool in_m = lane < 32 && p_m < A.Lt;
    int wb = (T.l0 >> 1) - 2 + 2 * lane;
    wb = wb < 0 ? 0 : (wb > Lth - 4 ? Lth - 4 : wb);
    const int pc = P < 0 ? 0 : (P > L - 4 ? L - 4 : P);
    const unsigned rs = none ? 0u : up ? (unsigned)Lh : (unsigned)L;     // row stride of the source
    const float* const zrow = up ? A.z0 + ((size_t)T.b * A.C0 + c8 * 8) * Lh : none ? A.z1 : A.z1 + ((size_t)T.b * A.C1 + (c8 * 8 - A.C0)) * L;
    unsigned long long rb = wunet_sgpr64((unsigned long long)(__UINTPTR_TYPE__)zrow);
    const unsigned long long rb0 = rb;
    const unsigned voff = none ? 0u : (unsigned)(up ? wb : pc) * 4u;
    const int m0 = up ? mi0 : (in_m ? p_m : 0), m1 = up ? mi1 : m0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        R.q[e] = wunet_ld4s_async(rb, voff);
        rb = wunet_sgpr64(rb + (unsigned long long)rs * 4u);
    }
    R.ma = wunet_ld1s_async(rb0, none ? 0u : ((unsigned)e_m * rs + (unsigned)m0) * 4u);
    R.mb = wunet_ld1s_async(rb0, none ? 0u : ((unsigned)e_m * rs + (unsigned)m1) * 4u);
    return 10;
}

// EDGE: the tile starts a row (l0 = 0) - lane 0's window was clamped (its loads started at sample 0, not -2) and its first two samples are
// the conv's zero padding.  One tile in L / 256; every other tile runs the instantiation without a single select.
// The BatchNorm coefficients in `coef` are PRE-MULTIPLIED by the operand's power-of-two scale (LeakyReLU and the interpolation commute with
// it exactly).
template <bool EDGE, bool COPY>
__device__ __forceinline__ void wunet_h3u_convert_t(const ConvH3uArgs& A, const WunetH3uRaw& R, const WunetH3uTile& T, wunet_half* xs, int cw,
                                                    int lane, const WunetH3uCoord& K, const float* coef, unsigned long long* tr)
{
#ifdef WUNET_H3U_TRACE
#define WUNET_H3U_SUB(K_) if (tr && cw == 0 && lane == 0) tr[K_] = wunet_memtime();
#else
#define WUNET_H3U_SUB(K_)
    (void)tr;
#endif
    WUNET_H3U_SUB(0)
    constexpr int COLS = 272, Q4 = COLS / 4;
    if (WUNET_H3U_ABL & 16) return;
    if (WUNET_H3U_ABL & 64) {                       // the loads stay alive (their registers are summed), nothing else of the conversion
        float t = R.ma + R.mb;
#pragma unroll
        for (int e = 0; e < 8; ++e) t += R.q[e][0] + R.q[e][1] + R.q[e][2] + R.q[e][3];
        if (t == 123.456f) xs[lane] = 1;
        return;
    }
    const int c8 = wunet_uniform(T.ch * 4 + cw);
    const int L = A.L;
    const bool none = c8 >= A.C8, up = !none && c8 * 8 < A.C0;
    if (none) {                                     // a group beyond the operand's last one (the weight pack's rows of these channels are 0): zeros
        const wunet_h8 z{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = 4 * lane + 6 + j;
            const int pw = (cw * COLS + (col & 3) * Q4 + (col >> 2)) * 8;
            wunet_sth8(xs + pw, z);
            wunet_sth8(xs + 4 * COLS * 8 + pw, z);
        }
        if (lane < 32) {
            const int col = 262 + (lane & 3);
            const int pw = (cw * COLS + (col & 3) * Q4 + (col >> 2)) * 8 + (lane >> 2);
            xs[pw] = 0;
            xs[4 * COLS * 8 + pw] = 0;
        }
        return;
    }
    const bool write_out = COPY && T.mt0 == 0;      // (COPY: the training instantiation, ConvH3uArgs::oxh set)
    // BatchNorm scale / shift of the group's channels from the block's LDS table (a loaded from global memory here would be one more
    // memory round trip per stage - and its wait would drain the prefetched tiles with it)
    const int cc = c8 * 8;
    float av[8], sv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { av[e] = coef[cc + e]; sv[e] = coef[A.C8 * 8 + cc + e]; }
    WUNET_H3U_SUB(1)
    const int P = T.l0 - 2 + 4 * lane;
    const bool edge = EDGE && P < 0;
    float vals[4][8];                               // [sample P + j][channel]
    if (up) {
        // ATen's coordinates of the outputs P .. P+3 (wunet_up_coord: fp32, as the reference computes them).  Their source pairs are
        // (wb, wb+1), (wb+1, wb+2), (wb+1, wb+2), (wb+2, wb+3) with wb = (P-2)/2 - i0(j) = (j-1) >> 1 for every j >= 1 at these lengths;
        // output 0 of a row has i0 = 0 with weight 1 on it, which the clamped source index -1 -> 0 reproduces.
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float u[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) u[k] = wunet_lrelu(av[e] * R.q[e][k] + sv[e]);
            // sources wb .. wb+3 with the row's left clamp: (0, 0, 0, 1) from a window that started at 0
            const float t0 = u[0], t1 = edge ? u[0] : u[1], t2 = edge ? u[0] : u[2], t3 = edge ? u[1] : u[3];
            vals[0][e] = K.w0[0] * t0 + K.w1[0] * t1;
            vals[1][e] = K.w0[1] * t1 + K.w1[1] * t2;
            vals[2][e] = K.w0[2] * t1 + K.w1[2] * t2;
            vals[3][e] = K.w0[3] * t2 + K.w1[3] * t3;
        }
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float u[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) u[k] = wunet_lrelu(av[e] * R.q[e][k] + sv[e]);
            vals[0][e] = u[0]; vals[1][e] = u[1];
            vals[2][e] = edge ? u[0] : u[2];             // (a window that started at sample 0: samples 0, 1 are its first two)
            vals[3][e] = edge ? u[1] : u[3];
        }
    }
    WUNET_H3U_SUB(2)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        wunet_h8 h, l;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            wunet_half x, y;
            wunet_split_h((edge && j < 2) ? 0.0f : vals[j][e], x, y);      // samples -2, -1: the conv's zero padding
            wunet_put_half(h, e, x);
            wunet_put_half(l, e, y);
        }
        const int col = 4 * lane + 6 + j;
        const int pw = (cw * COLS + (col & 3) * Q4 + (col >> 2)) * 8;
        if (!(WUNET_H3U_ABL & 128) || h[0] == (wunet_half)0x1234) {      // (ablation 128: the arithmetic stays, the LDS writes go)
        wunet_sth8(xs + pw, h);
        wunet_sth8(xs + 4 * COLS * 8 + pw, l);
        }
        // (training) the tile's own 256 samples of the operand also go to HBM, once per tile: the weight gradient reads them
        if (write_out && col >= 8 && col < 264) {
            const size_t o = (((size_t)T.b * A.C8 + c8) * L + (size_t)(P + j)) * 8;
            wunet_sth8(A.oxh + o, h);
            wunet_sth8(A.oxl + o, l);
        }
    }
    WUNET_H3U_SUB(3)
    // columns 262 .. 265 (samples l0 + 254 .. 257): channel lane >> 2, sample lane & 3 of lanes 0 .. 31, two bytes per plane each
    if (lane < 32) {
        const int e_m = lane >> 2, p_m = T.l0 + 254 + (lane & 3);
        const bool in_m = p_m < A.Lt;
        const float am = coef[cc + e_m], sm = coef[A.C8 * 8 + cc + e_m];
        const float ua = wunet_lrelu(am * R.ma + sm);
        const float v = up ? K.mw0 * ua + K.mw1 * wunet_lrelu(am * R.mb + sm) : ua;
        wunet_half x, y;
        wunet_split_h(in_m ? v : 0.0f, x, y);
        const int col = 262 + (lane & 3);
        const int pw = (cw * COLS + (col & 3) * Q4 + (col >> 2)) * 8 + e_m;
        xs[pw] = x;
        xs[4 * COLS * 8 + pw] = y;
        if (write_out && col < 264) {
            const size_t o = (((size_t)T.b * A.C8 + c8) * L + (size_t)p_m) * 8 + e_m;
            A.oxh[o] = x;
            A.oxl[o] = y;
        }
    }
#undef WUNET_H3U_SUB
}
template <bool COPY>
__device__ __forceinline__ void wunet_h3u_convert(const ConvH3uArgs& A, const WunetH3uRaw& R, const WunetH3uTile& T, wunet_half* xs, int cw,
                                                  int lane, const WunetH3uCoord& K, const float* coef, unsigned long long* tr = nullptr)
{
    if (wunet_uniform(T.l0) == 0) wunet_h3u_convert_t<true, COPY>(A, R, T, xs, cw, lane, K, coef, tr);
    else wunet_h3u_convert_t<false, COPY>(A, R, T, xs, cw, lane, K, coef, tr);
}

// (-DWUNET_H3U_TRACE, tools/h3u_trace.py: lane 0 of MFMA wave 0 / loader wave 0 stamps its arrival at and its release from every stage
//  barrier - trace[((block * 2 + role) * 128 + stage) * 2 + {0, 1}] - and the loader four points inside its conversion)
#ifdef WUNET_H3U_TRACE
#define WUNET_H3U_SUBPTR (A.trace && t < 128 ? A.trace + 256 * 2 * 128 * 2 + ((size_t)blockIdx.x * 128 + t) * 4 : nullptr)
#define WUNET_H3U_STAMP(ROLE_, STAGE_, WHICH_)                                                                     \
    if (A.trace && cw == 0 && lane == 0 && (STAGE_) < 128)                                                         \
        A.trace[(((size_t)blockIdx.x * 2 + (ROLE_)) * 128 + (STAGE_)) * 2 + (WHICH_)] = wunet_memtime();
#else
#define WUNET_H3U_SUBPTR nullptr
#define WUNET_H3U_STAMP(ROLE_, STAGE_, WHICH_)
#endif

static_assert(10 + WUNET_H3U_NST == 18, "the barrier spellings above");
static_assert(20 + 2 * (2 * 4 + 1) + 2 * WUNET_H3U_NST <= 63, "vmcnt is a 6-bit field");
// COPY: the training instantiation - the loaders also write the operand to A.oxh / A.oxl and count those stores in their waits; the eval
// instantiation holds none of that code
template <int M_REP, bool COPY = false>
__global__ __launch_bounds__(2 * WUNET_THREADS, 1) void conv_h3u_kernel(ConvH3uArgs A)
{
    constexpr int PAD = 2, TG = 5;
    constexpr int COLS = 272, Q4 = COLS / 4;
    constexpr int XP = 2 * 4 * COLS;              // 16-byte pieces of one x tile (hi + lo)
    constexpr int WPM = TG * 64;
    constexpr int WP = 2 * M_REP * WPM;           // pieces of one W sub-tile
    static_assert(WPM == 320, "W run = 256 + 64 pieces");
    WUNET_DYN_SMEM(smem);
    wunet_half* const xs0 = reinterpret_cast<wunet_half*>(smem);          // [2][hi|lo][4][COLS, de-interleaved][8]
    wunet_half* const ws0 = xs0 + 2 * XP * 8;                              // [2][hi|lo][M_REP][TG][4][16][8]
    float* const red = reinterpret_cast<float*>(ws0 + 2 * WP * 8);        // [4 waves][M_REP * 16][2] statistics hand-over
    float* const coef = red + WUNET_WAVES * M_REP * 32 + 4;                // [a | s][C8 * 8]: BatchNorm scale / shift of the operand's channels (x the operand scale)
    const int ER = A.mblocks * M_REP * 16;
    float* const epi = coef + 2 * A.C8 * 8;                          // [bias | eval a | eval s][ER]: the epilogue's per-row constants

    const int tid = threadIdx.x, lane = tid & 63, wave = wunet_uniform(tid >> 6);
    const int cw = wave & 3;                      // MFMA wave / loader wave index
    const int L = A.L;
    const int G = gridDim.x, nitems = A.ntiles * A.mblocks;
    if ((int)blockIdx.x >= nitems) return;
    const int nmy = (nitems - (int)blockIdx.x + G - 1) / G;           // work items (position tile, row block) of this block: blockIdx.x + k G
    const bool want_stats = A.stats != nullptr;

    float xs_ = 1.0f, xinv_ = 1.0f;
    wunet_h3u_x_scale(A.xb0, A.xb1, xs_, xinv_);
    if (blockIdx.x == 0 && tid == 0 && A.xsc) { A.xsc[0] = xs_; A.xsc[1] = xinv_; }

    // [a | s] x C8 groups, times the operand's scale; channels beyond the operand's: 0
    for (int c = tid; c < A.C8 * 8; c += 2 * WUNET_THREADS) {
        const bool u = c < A.C0, k = c < A.C0 + A.C1;
        coef[c] = xs_ * (u ? A.a0[c] : k ? A.a1[c - A.C0] : 0.0f);
        coef[A.C8 * 8 + c] = xs_ * (u ? A.s0[c] : k ? A.s1[c - A.C0] : 0.0f);
    }
    // (from global memory they were one memory round trip per work item with the matrix pipe idle: this kernel has no second block on the
    //  CU to fill it)
    for (int c = tid; c < ER; c += 2 * WUNET_THREADS) {
        const bool in = c < A.Cout;
        epi[c] = (in && A.bias) ? A.bias[c] : 0.0f;
        epi[ER + c] = (in && A.xrows) ? A.ev_a[c] : 0.0f;
        epi[2 * ER + c] = (in && A.xrows) ? A.ev_s[c] : 0.0f;
    }
    __syncthreads();
    // Both roles run the SAME sequence of workgroup barriers: one per stage (the stage's buffers are complete - the loaders waited for
    // their DMA pieces and LDS writes - and every MFMA wave has read the last fragment of the previous stage, whose buffers the
    // loaders may now refill), plus, with statistics, one per item (hand-over of the waves' sums).
    if (wave >= WUNET_WAVES) {
        // ================================================================ loader waves
        const wunet_lds_t ws_a = wunet_lds_addr(ws0);
        const unsigned wo_all = (unsigned)(cw * 64 + lane) * 16u, wo_rest = (256u + (unsigned)lane) * 16u;
        const int T = nmy * A.NS;                 // stages of this block, tile t = (item t / NS, chunk t % NS)
        const WunetH3uStep S{A.NS, A.mblocks, G / A.mblocks, G % A.mblocks};
#define WUNET_H3U_TILE(CUR_, OUT_)                                                                                 \
    WunetH3uTile OUT_;                                                                                             \
    OUT_.ch = (CUR_).ch; OUT_.mt0 = (CUR_).mb * M_REP;                                                             \
    OUT_.b = ((CUR_).tile * 256) >> A.logL; OUT_.l0 = ((CUR_).tile * 256) & (L - 1);
        // W sub-tile of a stage: per (plane, m-tile) a contiguous run of 320 pieces in the pack and in LDS; wave cw moves pieces
        // cw * 64 .. + 63 of every run and the last 64 of the runs with sub % 4 == cw
        const unsigned long long wrun = (unsigned long long)A.NS * (TG * 64 * 16);
#define WUNET_H3U_DMA_W(TL_, P_)                                                                                   \
    if (!(WUNET_H3U_ABL & 8)) {                                                                                    \
        const unsigned long long wb_ = (unsigned long long)(__UINTPTR_TYPE__)A.wh                                  \
                                     + (unsigned long long)((size_t)(TL_).mt0 * A.NS + (TL_).ch) * (TG * 64 * 16); \
        const wunet_lds_t la_ = ws_a + ((P_) * WP + cw * 64) * 16, lr_ = ws_a + ((P_) * WP + 256) * 16;            \
        const unsigned keep_ = wunet_m0_save();                                                                    \
        _Pragma("unroll") for (int sub = 0; sub < 2 * M_REP; ++sub) {                                              \
            const unsigned long long run_ = wunet_sgpr64(wb_ + (unsigned long long)(sub % M_REP) * wrun + (sub >= M_REP ? (unsigned long long)A.wdelta : 0ull)); \
            WUNET_H3U_DMA_PIECE(sub, run_, wo_all, la_)                                                            \
            if (cw == (sub & 3)) { WUNET_H3U_DMA_PIECE(sub, run_, wo_rest, lr_) }                                  \
        }                                                                                                          \
        wunet_m0_restore(keep_);                                                                                   \
    }
        // (the LDS offset of run `sub` is an immediate of the instruction that writes M0: a switch over the unrolled loop's constant)
#define WUNET_H3U_DMA_PIECE(SUB_, RUN_, VOFF_, LB_)                                                                \
    switch (SUB_) {                                                                                                \
    case 0: wunet_dma16m<0 * WPM * 16>(RUN_, VOFF_, wunet_lds_uniform(LB_)); break;                                \
    case 1: wunet_dma16m<1 * WPM * 16>(RUN_, VOFF_, wunet_lds_uniform(LB_)); break;                                \
    case 2: wunet_dma16m<2 * WPM * 16>(RUN_, VOFF_, wunet_lds_uniform(LB_)); break;                                \
    case 3: wunet_dma16m<3 * WPM * 16>(RUN_, VOFF_, wunet_lds_uniform(LB_)); break;                                \
    case 4: wunet_dma16m<4 * WPM * 16>(RUN_, VOFF_, wunet_lds_uniform(LB_)); break;                                \
    case 5: wunet_dma16m<5 * WPM * 16>(RUN_, VOFF_, wunet_lds_uniform(LB_)); break;                                \
    case 6: wunet_dma16m<6 * WPM * 16>(RUN_, VOFF_, wunet_lds_uniform(LB_)); break;                                \
    default: wunet_dma16m<7 * WPM * 16>(RUN_, VOFF_, wunet_lds_uniform(LB_)); break;                               \
    }
        static_assert(M_REP <= 4, "eight runs per W sub-tile");
        // Tile t's loads are issued three stages before the MFMA waves need it: in stage t the loaders convert tile t + 1 (loaded two
        // stages ago) into the other buffer while the loads of tiles t + 2 and t + 3 are in flight - memory latency under load (2 - 3 us)
        // is several stage times.  R0 / R1 / R2 rotate by unrolling, not by moves.
        // (Every stage of the steady loop issues its 10 prefetch loads UNCONDITIONALLY - past the block's last tile it re-reads that tile - and
        // the loop body has no join between an "issued" and a "not issued" path: at such a join hipcc's wait-count bookkeeping has to assume
        // the older loads are the youngest ones outstanding, and the first use of a tile loaded two stages ago then waits for the loads
        // issued a moment before it - the prefetch collapses to one memory latency per stage.)
        WunetH3uRaw R0, R1, R2;
        // did this wave's conversion of a tile issue the operand's copy (training; the first row block's loaders write it; a channel group
        // beyond the operand writes nothing): of the tile converted last / before last (see WUNET_H3U_NST)
        int wo_p = 0, wo_pp = 0;
#define WUNET_H3U_COPIES(T_) ((COPY && (T_).mt0 == 0 && (T_).ch * 4 + cw < A.C8) ? 1 : 0)
        WunetH3uCursor c1, c3;                    // the tiles t + 1 (converted in stage t) and min(t + 3, T - 1) (loaded in stage t)
        int i3;                                   // c3's stage number
        WunetH3uCoord K;                          // of c1's item
        int mi0, mi1;                             // of c3's item
        {
            const int tile0 = (int)blockIdx.x / A.mblocks;
            c1 = WunetH3uCursor{0, tile0, (int)blockIdx.x - tile0 * A.mblocks};
            WUNET_H3U_TILE(c1, t0)
            c3 = c1; i3 = 0;
            if (T > 1) { wunet_h3u_advance(c3, S); ++i3; }
            WUNET_H3U_TILE(c3, t1)
            wunet_h3u_coord(A, t0.l0, lane, K);
            wunet_h3u_mini_src(A, t0.l0, lane, mi0, mi1);
            wunet_h3u_issue(A, R0, t0, cw, lane, mi0, mi1);
            if (t1.ch == 0) wunet_h3u_mini_src(A, t1.l0, lane, mi0, mi1);
            wunet_h3u_issue(A, R1, t1, cw, lane, mi0, mi1);
            WUNET_H3U_DMA_W(t0, 0)
            if (T > 2) { wunet_h3u_advance(c3, S); ++i3; }
            WUNET_H3U_TILE(c3, t2)
            if (t2.ch == 0) wunet_h3u_mini_src(A, t2.l0, lane, mi0, mi1);
            wunet_h3u_issue(A, R2, t2, cw, lane, mi0, mi1);
            wunet_vm_wait<0>();
            wunet_h3u_convert<COPY>(A, R0, t0, xs0, cw, lane, K, coef);
            wo_p = WUNET_H3U_COPIES(t0);
        }
        // stage t (< T - 1): CUR_ holds the loads of tile t + 1, FREE_ (tile t's, converted a stage ago) takes those of tile t + 3; 10 loads
        // (tile t + 3's) are younger than the stage's last DMA piece at the next barrier
#define WUNET_H3U_LOADER_STAGE(CUR_, FREE_)                                                                        \
    {                                                                                                              \
        WUNET_H3U_STAMP(1, t, 0)                                                                                   \
        if (COPY) wunet_loader_barrier((WUNET_H3U_ABL & 1) ? 0 : 10 + wo_p * WUNET_H3U_NST);                       \
        else wunet_loader_barrier((WUNET_H3U_ABL & 1) ? 0 : 10);         /* (eval: none of the bookkeeping) */     \
        WUNET_H3U_STAMP(1, t, 1)                                                                                   \
        wunet_h3u_advance(c1, S);                                                                                  \
        WUNET_H3U_TILE(c1, tn)                                                                                     \
        WUNET_H3U_DMA_W(tn, (t + 1) & 1)                                                                           \
        if (i3 < T - 1) { wunet_h3u_advance(c3, S); ++i3; }                                                        \
        WUNET_H3U_TILE(c3, tnn)                                                                                    \
        if (tnn.ch == 0) wunet_h3u_mini_src(A, tnn.l0, lane, mi0, mi1);                                            \
        wunet_h3u_issue(A, FREE_, tnn, cw, lane, mi0, mi1);                                                        \
        if (tn.ch == 0) wunet_h3u_coord(A, tn.l0, lane, K);                                                        \
        /* tile t + 1's loads have landed: younger than them are the loads of tiles t + 2 and t + 3 (20), the DMA pieces of two     \
           stages (>= 2 M_REP + 1 per wave and stage) and the copies (training) of the two tiles converted since, WUNET_H3U_NST each */ \
        if (COPY) {                                                                                                \
            const int nwo_ = wunet_uniform(wo_p + wo_pp);                                                          \
            if (nwo_ == 0) wunet_vm_wait<20 + 2 * (2 * M_REP + 1)>();                                              \
            else if (nwo_ == 1) wunet_vm_wait<20 + 2 * (2 * M_REP + 1) + WUNET_H3U_NST>();                         \
            else wunet_vm_wait<20 + 2 * (2 * M_REP + 1) + 2 * WUNET_H3U_NST>();                                    \
        } else wunet_vm_wait<20 + 2 * (2 * M_REP + 1)>();                                                          \
        wunet_h3u_convert<COPY>(A, CUR_, tn, xs0 + ((t + 1) & 1) * XP * 8, cw, lane, K, coef, WUNET_H3U_SUBPTR);    \
        if (COPY) { wo_pp = wo_p; wo_p = WUNET_H3U_COPIES(tn); }                                                   \
        if (want_stats && tn.ch == 0) wunet_loader_stats_barrier((WUNET_H3U_ABL & 1) ? 0 : 10 + wo_p * WUNET_H3U_NST); \
        ++t;                                                                                                       \
    }
        int t = 0;
        while (t < T - 1) {
            WUNET_H3U_LOADER_STAGE(R1, R0)
            if (t >= T - 1) break;
            WUNET_H3U_LOADER_STAGE(R2, R1)
            if (t >= T - 1) break;
            WUNET_H3U_LOADER_STAGE(R0, R2)
        }
        // the last stage: nothing left to prepare
        wunet_loader_barrier(0);
        if (want_stats) wunet_loader_stats_barrier(0);
#undef WUNET_H3U_LOADER_STAGE
#undef WUNET_H3U_COPIES
#undef WUNET_H3U_DMA_W
#undef WUNET_H3U_DMA_PIECE
#undef WUNET_H3U_TILE
        return;
    }

    // ================================================================ MFMA waves (conv_h3d_kernel's fragments and MFMA order)
    const int q = lane >> 4, i16 = lane & 15;
    const int ll0 = cw * 64 + i16 * 4;            // this lane's 4 positions cw*64 + 4*i16 .. +3
    const int boff = (q * COLS + (ll0 >> 2)) * 8;
    const int aoff = (q * 16 + i16) * 8;
    int par = 0;                                  // buffer the coming stage reads
    float amax_run = 0.0f;
    const float inv12 = xinv_ * (A.sc2 ? A.sc2[1] : 1.0f);       // un-scale of operand and weights (two powers of two: the product is exact)
    for (int k = 0; k < nmy; ++k) {
        const int v = (int)blockIdx.x + k * G, tile = v / A.mblocks, mblk = v - tile * A.mblocks;
        const int b = (tile * 256) >> A.logL, l0 = (tile * 256) & (L - 1), mt0 = mblk * M_REP;
        wunet_f4 acc[M_REP][4];
#pragma unroll
        for (int mt = 0; mt < M_REP; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = wunet_f4{0.f, 0.f, 0.f, 0.f};
        for (int st = 0; st < A.NS; ++st) {
            WUNET_H3U_STAMP(0, k * A.NS + st, 0)
            wunet_wait_lds_barrier();
            WUNET_H3U_STAMP(0, k * A.NS + st, 1)
            if (WUNET_H3U_ABL & 32) { par ^= 1; continue; }
            const wunet_half* const xs = xs0 + par * XP * 8;
            const wunet_half* const ws = ws0 + par * WP * 8;
            wunet_h8 fh[TG + 3], fl[TG + 3];
#pragma unroll
            for (int e = 0; e < TG + 3; ++e) {
                const int ec = e + 8 - PAD;
                const int po = ((ec & 3) * Q4 + (ec >> 2)) * 8;
                fh[e] = wunet_ldh8(xs + boff + po);
                fl[e] = wunet_ldh8(xs + 4 * COLS * 8 + boff + po);
            }
            wunet_h8 ah[2][M_REP], al[2][M_REP];
#define WUNET_H3U_LOAD_A(BUF_, TL_)                                                                               \
    _Pragma("unroll") for (int mt = 0; mt < M_REP; ++mt) {                                                        \
        ah[BUF_][mt] = wunet_ldh8(ws + ((mt * TG + (TL_)) * 64) * 8 + aoff);                                      \
        al[BUF_][mt] = wunet_ldh8(ws + ((M_REP + mt) * TG + (TL_)) * 64 * 8 + aoff);                              \
    }
#define WUNET_H3U_PASS(WHICH_, BUF_, TL_)                                                                         \
    _Pragma("unroll") for (int mt = 0; mt < M_REP; ++mt)                                                          \
        _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) {                                                        \
            if (WUNET_H3U_ABL & 4) { if ((WHICH_) == 0 && (TL_) == 0) acc[mt][nt][0] += (float)al[BUF_][mt][0] + (float)fh[nt][0] + (float)ah[BUF_][mt][1] + (float)fl[nt][1]; } \
            else if ((WHICH_) == 0) acc[mt][nt] = wunet_mfma16h(al[BUF_][mt], fh[(TL_) + nt], acc[mt][nt]);       \
            else if ((WHICH_) == 1) acc[mt][nt] = wunet_mfma16h(ah[BUF_][mt], fl[(TL_) + nt], acc[mt][nt]);       \
            else acc[mt][nt] = wunet_mfma16h(ah[BUF_][mt], fh[(TL_) + nt], acc[mt][nt]);                          \
        }
#define WUNET_H3U_STEP(TL_)                                                                                       \
    {                                                                                                             \
        constexpr int tl = (TL_);                                                                                 \
        if (tl + 1 < TG) {                                                                                        \
            if (tl & 1) { WUNET_H3U_LOAD_A(0, tl + 1) } else { WUNET_H3U_LOAD_A(1, tl + 1) }                      \
        }                                                                                                         \
        wunet_sched_fence();                                                                                      \
        if (tl & 1) { WUNET_H3U_PASS(0, 1, tl) WUNET_H3U_PASS(1, 1, tl) WUNET_H3U_PASS(2, 1, tl) }               \
        else { WUNET_H3U_PASS(0, 0, tl) WUNET_H3U_PASS(1, 0, tl) WUNET_H3U_PASS(2, 0, tl) }                      \
    }
            WUNET_H3U_LOAD_A(0, 0)
            WUNET_H3U_STEP(0) WUNET_H3U_STEP(1) WUNET_H3U_STEP(2) WUNET_H3U_STEP(3) WUNET_H3U_STEP(4)
#undef WUNET_H3U_STEP
#undef WUNET_H3U_PASS
#undef WUNET_H3U_LOAD_A
            par ^= 1;
        }

        // ---- epilogue (conv_h3d_kernel's): un-scale, bias, store, BatchNorm statistics of the bias-free conv / eval activation bound.
        // The loaders are filling the next item's first buffers meanwhile.
        {
            float amax = 0.0f;
            float bvs[M_REP][4];
#pragma unroll
            for (int mt = 0; mt < M_REP; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) bvs[mt][r] = epi[(mt0 + mt) * 16 + q * 4 + r];
            float* const prow = A.out + ((size_t)b * A.Cout + mt0 * 16 + q * 4) * L + (l0 + ll0);
            const bool full = (mt0 + M_REP) * 16 <= A.Cout;
#define WUNET_H3U_ROWS(STATS_, GUARD_, EVAL_)                                                                     \
    float eas[M_REP][4], ess[M_REP][4];                                                                           \
    if (EVAL_) {                                                                                                  \
        _Pragma("unroll") for (int mt = 0; mt < M_REP; ++mt)                                                      \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                       \
                const int co = (mt0 + mt) * 16 + q * 4 + r;                                                       \
                eas[mt][r] = epi[ER + co];                                                                        \
                ess[mt][r] = epi[2 * ER + co];                                                                    \
            }                                                                                                     \
    }                                                                                                             \
    _Pragma("unroll") for (int mt = 0; mt < M_REP; ++mt) {                                                        \
        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};                                         \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                           \
            const int co = (mt0 + mt) * 16 + q * 4 + r;                                                           \
            const float bv = bvs[mt][r];                                                                          \
            wunet_f4 o;                                                                                           \
            _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) {                                                    \
                const float vv = acc[mt][nt][r] * inv12;                                                          \
                if (STATS_) { s1[r] += vv; s2[r] = fmaf(vv, vv, s2[r]); }                                         \
                o[nt] = vv + bv;                                                                                  \
            }                                                                                                     \
            if (!(GUARD_) || co < A.Cout) {                                                                       \
                wunet_st4(prow + (size_t)(mt * 16 + r) * L, o);                                                   \
                if (EVAL_) {                                                                                      \
                    const float ea = eas[mt][r], es = ess[mt][r];                                                 \
                    _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) amax = fmaxf(amax, fabsf(ea * o[nt] + es));  \
                }                                                                                                 \
            }                                                                                                     \
        }                                                                                                         \
        if (STATS_) {                                                                                             \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                       \
                s1[r] = wunet_row16_sum(s1[r]);                                                                   \
                s2[r] = wunet_row16_sum(s2[r]);                                                                   \
                if (i16 == 0) {                                                                                   \
                    float* rp = red + ((cw * M_REP + mt) * 16 + q * 4 + r) * 2;                                   \
                    rp[0] = s1[r];                                                                                \
                    rp[1] = s2[r];                                                                                \
                }                                                                                                 \
            }                                                                                                     \
        }                                                                                                         \
    }
            if (A.xrows) { WUNET_H3U_ROWS(false, true, true) }
            else if (want_stats) { if (full) { WUNET_H3U_ROWS(true, false, false) } else { WUNET_H3U_ROWS(true, true, false) } }
            else { if (full) { WUNET_H3U_ROWS(false, false, false) } else { WUNET_H3U_ROWS(false, true, false) } }
#undef WUNET_H3U_ROWS
            if (A.xrows) {                        // eval: the wave's running maximum of the activation bound (no hand-over: one atomic per wave at the end)
#pragma unroll
                for (int m = 1; m < 64; m <<= 1) amax = fmaxf(amax, wunet_shfl_xor(amax, m));
                amax_run = fmaxf(amax_run, amax);
            }
        }
        // one statistics row per tile (256 positions): the four MFMA waves' sums are added in wave order
        if (want_stats) {
            wunet_wait_lds_barrier();
            if (tid < M_REP * 16) {
                const float* rp = red + tid * 2;
                float t1 = 0.0f, t2 = 0.0f;
#pragma unroll
                for (int w = 0; w < WUNET_WAVES; ++w) {
                    t1 += rp[w * M_REP * 32];
                    t2 += rp[w * M_REP * 32 + 1];
                }
                const int co = mt0 * 16 + tid;
                if (co < A.Cout) {
                    float* stp = A.stats + ((size_t)co * A.ntiles + tile) * 2;
                    stp[0] = t1;
                    stp[1] = t2;
                }
            }
        }
    }
    if (A.xrows && lane == 0) wunet_atomic_absmax(A.xrows, amax_run);      // one atomic per MFMA wave into the layer's xb slot
}
