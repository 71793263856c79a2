// Elementwise side of the fp16-split ("h3") path (see wunet_h3.h): gradient scale, fp32 -> hi/lo split into the
// channel-group-major layout, weight pack.  Included by the host translation unit only.
#pragma once
#include "wunet_elementwise.h"
#include "wunet_h3.h"

// ---------------------------------------------------------------------------- power-of-two scales of the split operands
// Per-layer scale slots in the forward segment of the workspace, 8 floats per conv layer:
//   [0],[1]  scale, 1/scale of the layer's conv INPUT x (written by the operand pass that produces x)
//   [2],[3]  scale, 1/scale of the layer's weights (written by pack_h3_kernel from the partial maxima below)
//   [4]      xb: an upper bound of |LeakyReLU(BN(z))| of the layer's OUTPUT - what the consumers' x scales derive from.
//            Training mode: max_c |gamma_c| sqrt(N) + |beta_c|, N = B*L positions (batch statistics: (z - mean)^2 <= N var, so
//            |zhat| <= sqrt(N) - rigorous and data independent, known before the forward starts; measured values sit a factor
//            sqrt(N)/6 below it, which costs nothing: the error floor of a split value is 2^-25 in scaled units against
//            actual maxima of 2^5 or more).  Eval mode (running statistics, activations not normalised by the batch): the
//            measured maximum of |a z + s|, act_max_kernel.
#define WUNET_SLOT_FLOATS 8
#define WUNET_WMAX_PARTS 64
struct ScaleDesc {
    const float* w; unsigned wn;                  // conv weight (nullptr: no split pack of this layer)
    const float* gamma; const float* beta; int C; // BatchNorm affine parameters of the layer
    float sqrtn;                                  // sqrt(B * L)
    float* zp0; float* zp1;                       // nullptr or the 16-byte zero pads behind the layer's split conv input / split g_z: cleared here
};
struct ScaleTable { ScaleDesc d[WUNET_MAX_CONV_LAYERS]; float* wmax; float* slots; int training; };

// grid (WUNET_WMAX_PARTS, layers): partial max |W| per layer; block (0, layer) also writes the layer's xb (training) or clears
// it (eval: act_max_kernel accumulates into it with atomicMax)
static __global__ __launch_bounds__(WUNET_THREADS) void h3_scales_kernel(ScaleTable T)
{
    __shared__ float red[WUNET_THREADS];
    const ScaleDesc& d = T.d[blockIdx.y];
    const int tid = threadIdx.x;
    if (d.w) {
        float m = 0.0f;
        for (unsigned i = blockIdx.x * WUNET_THREADS + tid; i < d.wn; i += gridDim.x * WUNET_THREADS) m = fmaxf(m, fabsf(d.w[i]));
        red[tid] = m;
        __syncthreads();
        for (int s = WUNET_THREADS / 2; s > 0; s >>= 1) {
            if (tid < s) red[tid] = fmaxf(red[tid], red[tid + s]);
            __syncthreads();
        }
        if (tid == 0) T.wmax[blockIdx.y * WUNET_WMAX_PARTS + blockIdx.x] = red[0];
        __syncthreads();
    }
    if (blockIdx.x == 0) {
        float m = 0.0f;
        if (T.training)
            for (int c = tid; c < d.C; c += WUNET_THREADS) m = fmaxf(m, fabsf(d.gamma[c]) * d.sqrtn + fabsf(d.beta[c]));
        red[tid] = m;
        __syncthreads();
        for (int s = WUNET_THREADS / 2; s > 0; s >>= 1) {
            if (tid < s) red[tid] = fmaxf(red[tid], red[tid + s]);
            __syncthreads();
        }
        if (tid == 0) T.slots[blockIdx.y * WUNET_SLOT_FLOATS + 4] = red[0];
        // the zero pads the DMA pieces outside a tensor fetch (conv_h3d_kernel: halo beyond an item, channel groups beyond C8)
        if (tid < 4) {
            if (d.zp0) d.zp0[tid] = 0.0f;
            if (d.zp1) d.zp1[tid] = 0.0f;
        }
    }
}

// eval mode: absolute row sums ||W_c||_1 of the conv weights of the layers whose epilogue scales the next operand by a rigorous bound
// (ConvH3Args::op_wl1; built with the weight packs, reused with them).  grid (8, layers): block (x, layer) the rows x, x + 8, ...
struct RowL1Desc { const float* w; float* dst; int rows, rowlen; };
struct RowL1Table { RowL1Desc d[WUNET_MAX_CONV_LAYERS]; };
static __global__ __launch_bounds__(WUNET_THREADS) void w_rowl1_kernel(RowL1Table T)
{
    __shared__ float red[WUNET_WAVES];
    const RowL1Desc& d = T.d[blockIdx.y];
    for (int r = blockIdx.x; r < d.rows; r += gridDim.x) {
        const float* wr = d.w + (size_t)r * d.rowlen;
        float m = 0.0f;
        for (int i = threadIdx.x; i < d.rowlen; i += WUNET_THREADS) m += fabsf(wr[i]);
#pragma unroll
        for (int k = 1; k < 64; k <<= 1) m += wunet_shfl_xor(m, k);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
        __syncthreads();
        // (an upper bound is what is needed: rounding of the sum is covered by the factor 4 of head room above every operand bound)
        if (threadIdx.x == 0) d.dst[r] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}

// eval mode with reused weight packs (WUNET_FWD_PACKS_VALID): what h3_scales_kernel does beside the weight maxima - the activation
// bounds cleared (the conv epilogues / act_max_kernel fold this call's maxima into them), the DMA zero pads rewritten.  One block.
static __global__ __launch_bounds__(WUNET_THREADS) void h3_slots_clear_kernel(ScaleTable T, int nl)
{
    for (int i = threadIdx.x; i < nl * 5; i += WUNET_THREADS) {
        const int layer = i / 5, k = i - layer * 5;
        const ScaleDesc& d = T.d[layer];
        if (k == 4) T.slots[layer * WUNET_SLOT_FLOATS + 4] = 0.0f;
        else if (d.zp0) d.zp0[k] = 0.0f;
    }
}

// eval mode: xb = max |a_c z + s_c| over the layer (>= |LeakyReLU(.)|), block maxima combined with atomicMax on the bit
// pattern of the non-negative float (exact and order independent: deterministic)
static __global__ __launch_bounds__(WUNET_THREADS) void act_max_kernel(const float* z, const float* a, const float* s, int C, int logL,
                                                                 size_t n4, float* xb)
{
    __shared__ float red[WUNET_THREADS];
    float m = 0.0f;
    for (size_t i = (size_t)blockIdx.x * WUNET_THREADS + threadIdx.x; i < n4; i += (size_t)gridDim.x * WUNET_THREADS) {
        const int c = (int)((i >> (logL - 2)) % (size_t)C);
        const float av = a[c], sv = s[c];
        const wunet_f4 v = wunet_ld4(z + 4 * i);
#pragma unroll
        for (int j = 0; j < 4; ++j) m = fmaxf(m, fabsf(av * v[j] + sv));
    }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int st = WUNET_THREADS / 2; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + st]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        // NaN / inf maxima saturate to the largest finite float: the scale stays defined (the data is garbage either way)
        const unsigned u = wunet_fbits(red[0]) & 0x7fffffffu;
        atomicMax(reinterpret_cast<unsigned*>(xb), u < 0x7f800000u ? u : 0x7f7fffffu);
    }
}

// runtime form of wunet_split<BF> for the elementwise producers (bf: the bf16 mode, one word per value, lo array unused)
__device__ __forceinline__ void wunet_split_rt(int bf, float x, wunet_half& hi, wunet_half& lo)
{
    if (bf) { hi = wunet_f2b(x); lo = 0; }
    else wunet_split_h(x, hi, lo);
}

// x scale of a conv input whose sources' activation bounds are xb0 (and xb1): every thread derives the same value
__device__ __forceinline__ void wunet_x_scale(const float* xb0, const float* xb1, float& s, float& inv)
{
    wunet_pow2_scale(fmaxf(xb0[0], xb1 ? xb1[0] : 0.0f), s, inv);
}

// fp32 [B][C][L]  ->  hi / lo [B][C8][L][8] halfs of sc[0]*x (sc == nullptr: unscaled).  One thread per
// (channel group, 4 samples): 8 float4 loads, 4+4 16-byte stores.
static __global__ __launch_bounds__(WUNET_THREADS) void split_act_kernel(const float* x, wunet_half* hi, wunet_half* lo, const float* sc,
                                                                   const float* xb0, const float* xb1, float* xsc,
                                                                   int B, int C, int C8, int L, int logL, int bf)
{
    const int l4n = L >> 2;
    float s = sc ? sc[0] : 1.0f;
    if (xb0) {                                  // scale from the sources' activation bounds; block 0 publishes it for the GEMMs
        float inv;
        wunet_x_scale(xb0, xb1, s, inv);
        if (blockIdx.x == 0 && threadIdx.x == 0) { xsc[0] = s; xsc[1] = inv; }
    }
    const size_t total = (size_t)B * C8 * l4n;
    for (size_t i = (size_t)blockIdx.x * WUNET_THREADS + threadIdx.x; i < total; i += (size_t)gridDim.x * WUNET_THREADS) {
        const int l4 = (int)(i & (size_t)(l4n - 1));
        const size_t row = i >> (logL - 2);
        const int b = (int)(row / (size_t)C8), c8 = (int)(row - (size_t)b * C8);
        wunet_f4 v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = c8 * 8 + e;
            const bool ok = c < C;
            v[e] = wunet_sel4(ok, wunet_ld4(x + ((size_t)b * C + (ok ? c : 0)) * L + 4 * l4));
        }
        wunet_half* ph = hi + (row * L + 4 * (size_t)l4) * 8;
        wunet_half* pl = lo + (row * L + 4 * (size_t)l4) * 8;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            wunet_h8 h, l;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                wunet_half a, d;
                wunet_split_rt(bf, s * v[e][j], a, d);
                wunet_put_half(h, e, a);
                wunet_put_half(l, e, d);
            }
            wunet_sth8(ph + 8 * j, h);
            if (!bf) wunet_sth8(pl + 8 * j, l);
        }
    }
}

// ---------------------------------------------------------------------------- split-K reduce of wgrad_h3_kernel
// Sums the tile-major partials of wgrad_h3_kernel over the splits (fixed order: 16 split lanes per group of outputs,
// fp32 in groups of 8, fp64 across - like wgrad_reduce_kernel) and scatters the result into dW [Cout][Cin][TAPS].
// One float4 of the tile-major layout = rows q*4 .. q*4+3 of one (m-tile, ci, tap).
struct WgradH3ReduceArgs {
    const float* part; size_t part_stride; int splits;
    float* dw;
    int Cout, Cin, taps, mrep, tw, nblocks, mblocks, cib;
};

// dW [Cout][Cin][TAPS] <- the four rows of tile-major float4 i = ((((bm*nblocks + bn)*4 + wave)*mrep + mt)*tw + t)*64 + lane
__device__ __forceinline__ void wgrad_h3_scatter(const WgradH3ReduceArgs& A, size_t i, const double* u)
{
    size_t j = i;
    const int lane = (int)(j & 63); j >>= 6;
    const int t = (int)(j % A.tw); j /= A.tw;
    const int mt = (int)(j % A.mrep); j /= A.mrep;
    const int wave = (int)(j & 3); j >>= 2;
    const int bn = (int)(j % A.nblocks), bm = (int)(j / A.nblocks);
    const int grp = A.taps == 15 ? wave >> 1 : wave, t0 = A.taps == 15 ? (wave & 1) * 8 : 0;
    const int ci = bn * A.cib + grp * 16 + (lane & 15), tap = t0 + t;
    const int co = (bm * A.mrep + mt) * 16 + (lane >> 4) * 4;
    if (ci < A.Cin && tap < A.taps) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (co + r < A.Cout) A.dw[((size_t)(co + r) * A.Cin + ci) * A.taps + tap] = (float)u[r];
    }
}

// Few splits and many outputs (the deep levels: 9-64 splits of a 0.1-3 MB dW): one thread per float4 walks the splits itself -
// coalesced 16-byte loads, eight in flight, no LDS, no barrier.  (The 16-split-lane form below spends these layers waiting
// on two barriers per 16 outputs: 29 us for 33 MB at the 32-sample level.)  fp32 in groups of 8 consecutive splits, fp64
// across, fixed order.
static __global__ __launch_bounds__(WUNET_THREADS) void wgrad_h3_reduce_serial_kernel(WgradH3ReduceArgs A)
{
    const size_t n4 = (size_t)A.mblocks * A.nblocks * WUNET_WAVES * A.mrep * A.tw * 64;
    for (size_t i = (size_t)blockIdx.x * WUNET_THREADS + threadIdx.x; i < n4; i += (size_t)gridDim.x * WUNET_THREADS) {
        double u[4] = {0.0, 0.0, 0.0, 0.0};
        int r = 0;
        for (; r + 8 <= A.splits; r += 8) {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const wunet_f4 v = wunet_ld4(A.part + (size_t)(r + k) * A.part_stride + 4 * i);
                a0 += v[0]; a1 += v[1]; a2 += v[2]; a3 += v[3];
            }
            u[0] += (double)a0; u[1] += (double)a1; u[2] += (double)a2; u[3] += (double)a3;
        }
        for (; r < A.splits; ++r) {
            const wunet_f4 v = wunet_ld4(A.part + (size_t)r * A.part_stride + 4 * i);
            u[0] += (double)v[0]; u[1] += (double)v[1]; u[2] += (double)v[2]; u[3] += (double)v[3];
        }
        wgrad_h3_scatter(A, i, u);
    }
}

static __global__ __launch_bounds__(WUNET_THREADS) void wgrad_h3_reduce_kernel(WgradH3ReduceArgs A)
{
    __shared__ double red[16][16][4];
    const int og = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const size_t n4 = (size_t)A.mblocks * A.nblocks * WUNET_WAVES * A.mrep * A.tw * 64;
    for (size_t base = (size_t)blockIdx.x * 16; base < n4; base += (size_t)gridDim.x * 16) {
        const size_t i = base + og;
        double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
        if (i < n4) {
            int r = sl;
            for (; r + 7 * 16 < A.splits; r += 8 * 16) {
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const wunet_f4 v = wunet_ld4(A.part + (size_t)(r + k * 16) * A.part_stride + 4 * i);
                    a0 += v[0]; a1 += v[1]; a2 += v[2]; a3 += v[3];
                }
                t0 += (double)a0; t1 += (double)a1; t2 += (double)a2; t3 += (double)a3;
            }
            for (; r < A.splits; r += 16) {
                const wunet_f4 v = wunet_ld4(A.part + (size_t)r * A.part_stride + 4 * i);
                t0 += (double)v[0]; t1 += (double)v[1]; t2 += (double)v[2]; t3 += (double)v[3];
            }
        }
        __syncthreads();
        red[sl][og][0] = t0; red[sl][og][1] = t1; red[sl][og][2] = t2; red[sl][og][3] = t3;
        __syncthreads();
        if (sl == 0 && i < n4) {
            double u[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int k = 0; k < 16; ++k) { u[0] += red[k][og][0]; u[1] += red[k][og][1]; u[2] += red[k][og][2]; u[3] += red[k][og][3]; }
            wgrad_h3_scatter(A, i, u);
        }
    }
}

// ---------------------------------------------------------------------------- g_z, produced split
// g_z = k1*g + k2*z + k3 (gz_materialize_kernel) written straight into the scaled hi / lo [B][C8][L][8] layout.  The
// power-of-two scale comes from an upper bound of max |g_z| (bn_finalize_bwd_kernel's bound[c], max over channels) so
// no pass over g_z is needed before writing it: 2^k * bound in [512, 1024).  Every block derives the same scale;
// block 0 publishes {scale, 1/scale} for the GEMMs.  One thread per (channel group, 4 samples).
// FIN (levels with at most WUNET_GZ_FIN_LOADS channels x partial rows of pass A, C <= WUNET_GZ_FIN_C): the BatchNorm-backward
// finalize (bn_finalize_bwd_kernel: sums of the partial rows -> d gamma, d beta, k1..k3, the bound) runs in EVERY block's prologue - a
// few hundred loads that hit the L2 - instead of in a launch of its own in front of this one: on those levels that launch is 4 us of
// latency + a kernel boundary on the backward's critical path.  Block 0 publishes the results.  The sums run over the rows in order,
// in double: the same numbers as the tree of the separate kernel to the last bit of the float results.
// HEAD (the last decoder layer, whose only consumer is the 1x1 head): its g = wh[c] * gh * LeakyReLU'(a z + s) is a function of the
// one-channel head gradient gh [B][L] (pass_a_kernel<A_HEAD>'s arithmetic, bit for bit) - recomputed here from gh (1 / C of the
// bytes) instead of written by pass A and read back: the largest layer's g (100 MB at batch 64) never exists.
// ENC (an encoder layer whose data gradients are whole tensors): its g = (dXdec[b, coff+c, l] + (l even ? dXenc[b, c, l/2] : 0)) * LeakyReLU'
// (pass_a_kernel<A_ENC>'s arithmetic, bit for bit) is recomputed here from the two data gradients (6 bytes per value read instead of
// the 4 of g) so that pass A does not WRITE g (4 bytes per value; a written byte costs 1.4 x a read one on this part).
// UP (a layer that feeds an upsample and whose BatchNorm-backward sums came out of the consumer's data-gradient epilogue, conv_h3d_kernel<.., BSUM>:
// no pass A ran, no g exists): g = upsample^T(dX of the next decoder layer) * LeakyReLU' is formed here from dX [B][Cg0][2L] (8 bytes per value
// read instead of the 4 of a stored g; wunet_upT_row - pass_a_kernel<A_UP>'s arithmetic).
struct GzHeadArgs {
    const float* gh; const float* wh; const float* a; const float* s;      // HEAD (gh != nullptr); a, s also ENC and UP
    const float* gd; const float* ge; int Cg0, coff;                        // ENC (gd != nullptr): dXdec [B][Cg0][L], dXenc [B][C][L/2]
    const float* gu; float up_scale;                                        // UP (gu != nullptr): dX [B][Cg0][2L]; (float)(Lt-1)/(2Lt-1)
    const float* gq;                                                        // UPH (gq != nullptr): dXh [B][C][L], completed by pass_a_kernel<A_UPH>
};
#define WUNET_GZ_FIN_LOADS 1536   // (sweep 1152 / 2048 / 3100 / all: 5.41 / 5.42 / 5.46 / 5.50 ms per step - beyond the 512-sample level the prologue costs more than the launch)
#define WUNET_GZ_FIN_C 512
// GM (compile time): where g comes from.  Rounds 3 - 6 selected the mode by run-time tests of GzHeadArgs' pointers inside the unrolled loop over
// the thread's eight channels: hipcc then cannot move a channel's loads across the branches - the ISA held `load z, branch, load g, s_waitcnt
// vmcnt(0)` per channel, eight serialised memory round trips per thread at 197 VGPRs (two waves per SIMD) - and the pass ran at HALF the rate of
// a kernel that moves the same bytes (tools/microbench/elem_passes.hip: 92.6 us against 46 us on decoder.10's geometry).  With the mode a
// template parameter every load of the iteration is issued before the first value is used, as in prep_h3_kernel.
enum { GZ_G = 0, GZ_HEAD = 1, GZ_ENC = 2, GZ_UPH = 3, GZ_UP = 4 };
template <bool FIN, int GM, bool BFM>
__global__ __launch_bounds__(WUNET_THREADS) void gz_split_h3_kernel(const float* g, const float* z, const float* k1, const float* k2,
                                                                     const float* k3, const float* bound, float* sc, wunet_half* hi,
                                                                     wunet_half* lo, int B, int C, int C8, int L, int logL, int Lt,
                                                                     BnBwdArgs F, GzHeadArgs H)
{
    __shared__ float red[WUNET_WAVES];
    __shared__ float ks[FIN ? 3 * WUNET_GZ_FIN_C : 3];
    float m = 0.0f;
    if (FIN) {
        for (int c = threadIdx.x; c < C; c += WUNET_THREADS) {
            double s1 = 0.0, s2 = 0.0;
            float mg = 0.0f, mz = 0.0f;
            const float gam_ = F.gamma[c], rs_ = F.rstd[c], mu_ = F.mean[c];     // (with the rows' loads, not behind their sums: one round trip less in front of the data)
            for (int r0 = 0; r0 < F.rows; r0 += 8) {
                float p1[8], p2[8], q1[8], q2[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) {                         // eight rows' loads first (clamped rows, selected below)
                    const size_t o = ((size_t)(r0 + r < F.rows ? r0 + r : 0) * F.C + c) * 2;
                    p1[r] = F.part[o]; p2[r] = F.part[o + 1];
                    q1[r] = F.pmax[o]; q2[r] = F.pmax[o + 1];
                }
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    if (r0 + r < F.rows) {
                        s1 += (double)p1[r]; s2 += (double)p2[r];
                        mg = fmaxf(mg, q1[r]); mz = fmaxf(mz, q2[r]);
                    }
                }
            }
            const double m1 = s1 / F.count, m2 = s2 / F.count;
            const double a = (double)gam_ * (double)rs_;
            const float k1c = (float)a, k2c = (float)(-a * m2 * (double)rs_);
            const float k3c = (float)(a * m2 * (double)rs_ * (double)mu_ - a * m1);
            const float bc = (float)(fabs(a) * (double)mg + fabs(a * m2 * (double)rs_) * (double)mz + fabs(a * m1));
            ks[c] = k1c; ks[WUNET_GZ_FIN_C + c] = k2c; ks[2 * WUNET_GZ_FIN_C + c] = k3c;
            m = fmaxf(m, bc);
            if (blockIdx.x == 0) {
                F.dgamma[c] = (float)s2; F.dbeta[c] = (float)s1; F.dbias[c] = 0.0f;
                F.k1[c] = k1c; F.k2[c] = k2c; F.k3[c] = k3c; F.bound[c] = bc;
            }
        }
#pragma unroll
        for (int x = 32; x > 0; x >>= 1) m = fmaxf(m, wunet_shfl_xor(m, x));
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
        __syncthreads();
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    } else {
        // every WAVE takes the maximum over all channels itself (a maximum does not depend on the order: every wave of every block derives the
        // same scale) - no LDS tree, no barrier in front of the data loads
        for (int c = (int)(threadIdx.x & 63); c < C; c += 64) m = fmaxf(m, bound[c]);
#pragma unroll
        for (int x = 32; x > 0; x >>= 1) m = fmaxf(m, wunet_shfl_xor(m, x));
    }
    float s = 1.0f, inv = 1.0f;
    {
        const unsigned u = wunet_fbits(m);
        if (u != 0 && u < 0x7f800000u) {
            int k = 9 - ((int)((u >> 23) & 0xffu) - 127);
            k = k > 100 ? 100 : (k < -100 ? -100 : k);
            s = ldexpf(1.0f, k);
            inv = ldexpf(1.0f, -k);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { sc[0] = s; sc[1] = inv; }
    // One thread computes 8 channels x 4 consecutive samples (16-byte loads of g and z).  Its four 16-byte pieces per array
    // would leave every store instruction of a wave 16-byte pieces at a 64-byte stride (measured on the operand passes and
    // here: partial-line scattered stores cost 1.3 % of the step in this kernel alone), so the wave's 4 KiB of hi and of lo
    // are turned in LDS: lane l then stores pieces l, 64+l, 128+l, 192+l - 1 KiB contiguous per instruction.  The output
    // address of thread i is linear in i (piece 4*i + j), whatever the row boundaries.  (The turn is wave-private - a wave reads
    // only what its own lanes wrote; with the two block barriers replaced by a wave-level fence the kernel measured the same:
    // 49.4 us, step 5.06 / 5.06 ms - the barriers stay.)
    __shared__ wunet_h8 tb[BFM ? 1 : 2][WUNET_THREADS * 4];
    const int l4n = L >> 2;
    const size_t total = (size_t)B * C8 * l4n;
    const int lane = (int)(threadIdx.x & 63), wbase = (int)(threadIdx.x & ~63u);
    for (size_t base = (size_t)blockIdx.x * WUNET_THREADS; base < total; base += (size_t)gridDim.x * WUNET_THREADS) {
        const size_t i = base + threadIdx.x;
        if (i < total) {
            const int l4 = (int)(i & (size_t)(l4n - 1));
            const size_t row = i >> (logL - 2);
            const int b = (int)(row / (size_t)C8), c8 = (int)(row - (size_t)b * C8);
            const int c0_ = c8 * 8;
            // ---- phase 1: every load of the iteration (clamped channels instead of branches around loads; values of channels >= C are zeroed below)
            wunet_f4 zv[8], gv[8];
            float2 e2[GM == GZ_ENC ? 8 : 1];
            float whv[GM == GZ_HEAD ? 8 : 1];
            wunet_f4 gh4 = wunet_f4{0.f, 0.f, 0.f, 0.f};
            if (GM == GZ_HEAD) gh4 = wunet_ld4(H.gh + (size_t)b * L + 4 * l4);
            WunetUpT U;
            if (GM == GZ_UP) wunet_upT_coords(4 * l4, Lt, H.up_scale, U);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int cc = c0_ + e < C ? c0_ + e : C - 1;
                const size_t o = ((size_t)b * C + cc) * L + 4 * l4;
                zv[e] = wunet_ld4(z + o);
                if (GM == GZ_G) gv[e] = wunet_ld4(g + o);
                else if (GM == GZ_UPH) gv[e] = wunet_ld4(H.gq + o);        // (tile-edge values completed in place by pass_a_kernel<A_UPH>)
                else if (GM == GZ_ENC) {
                    gv[e] = wunet_ld4(H.gd + ((size_t)b * H.Cg0 + H.coff + cc) * L + 4 * l4);
                    e2[e] = *reinterpret_cast<const float2*>(H.ge + ((size_t)b * C + cc) * (L >> 1) + 2 * l4);
                } else if (GM == GZ_HEAD) whv[e] = H.wh[cc];
            }
            // the eight channels' constants as 16-byte loads (k1 .. k3 and the BatchNorm scale / shift of the recompute modes: five arrays whose
            // rows are padded to 64 floats - the loads past C stay inside them and their values are selected away)
            float kA[8], kB[8], kD[8], hA[8], hS[8];
            if (FIN) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { kA[e] = ks[c0_ + e]; kB[e] = ks[WUNET_GZ_FIN_C + c0_ + e]; kD[e] = ks[2 * WUNET_GZ_FIN_C + c0_ + e]; }
            } else {
                const wunet_f4 a0 = wunet_ld4(k1 + c0_), a1 = wunet_ld4(k1 + c0_ + 4), b0 = wunet_ld4(k2 + c0_), b1 = wunet_ld4(k2 + c0_ + 4);
                const wunet_f4 d0 = wunet_ld4(k3 + c0_), d1 = wunet_ld4(k3 + c0_ + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { kA[e] = a0[e]; kA[4 + e] = a1[e]; kB[e] = b0[e]; kB[4 + e] = b1[e]; kD[e] = d0[e]; kD[4 + e] = d1[e]; }
            }
            if (GM != GZ_G) {
                const wunet_f4 a0 = wunet_ld4(H.a + c0_), a1 = wunet_ld4(H.a + c0_ + 4), s0 = wunet_ld4(H.s + c0_), s1 = wunet_ld4(H.s + c0_ + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { hA[e] = a0[e]; hA[4 + e] = a1[e]; hS[e] = s0[e]; hS[4 + e] = s1[e]; }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) hA[e] = hS[e] = 0.0f;
            }
            // ---- phase 2: g (the recompute modes: pass_a_kernel's arithmetic in its order), g_z, the split
            wunet_f4 v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bool ok = c0_ + e < C;
                const float ha = hA[e], hs = hS[e];
                wunet_f4 gg = gv[e];
                if (GM == GZ_HEAD) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float t = whv[e] * gh4[j];
                        if (!(ha * zv[e][j] + hs > 0.0f)) t *= WUNET_SLOPE;
                        gg[j] = t;
                    }
                } else if (GM == GZ_ENC) {
                    gg[0] = gv[e][0] + e2[e].x; gg[1] = gv[e][1]; gg[2] = gv[e][2] + e2[e].y; gg[3] = gv[e][3];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (!(ha * zv[e][j] + hs > 0.0f)) gg[j] *= WUNET_SLOPE;
                } else if (GM == GZ_UPH) {
                    // (a layer behind an upsample whose consumer's data gradient arrives at this resolution: pass_a_kernel<A_UPH>'s arithmetic)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (!(ha * zv[e][j] + hs > 0.0f)) gg[j] *= WUNET_SLOPE;
                } else if (GM == GZ_UP) {
                    float gu[4];
                    const int cc = ok ? c0_ + e : C - 1;
                    wunet_upT_row(H.gu + ((size_t)b * H.Cg0 + cc) * (size_t)(2 * L), 4 * l4, 2 * L, 2 * Lt, Lt, H.up_scale, U, gu);
#pragma unroll
                    for (int j = 0; j < 4; ++j) gg[j] = (ha * zv[e][j] + hs > 0.0f) ? gu[j] : gu[j] * WUNET_SLOPE;
                }
                const float a = kA[e], bb = kB[e], d = kD[e];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[e][j] = (ok && 4 * l4 + j < Lt) ? s * (a * gg[j] + bb * zv[e][j] + d) : 0.0f;     // (row padding: no gradient)
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                wunet_h8 h, l;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    wunet_half x, y;
                    wunet_split_rt(BFM ? 1 : 0, v[e][j], x, y);
                    wunet_put_half(h, e, x);
                    wunet_put_half(l, e, y);
                }
                tb[0][threadIdx.x * 4 + j] = h;
                if (!BFM) tb[1][threadIdx.x * 4 + j] = l;
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int piece = 64 * j + lane;                               // of this wave's 256
            const size_t src = base + wbase + (piece >> 2);                // the thread that produced it
            if (src < total) {
                const size_t o = (4 * (base + wbase) + piece) * 8;
                wunet_sth8(hi + o, tb[0][wbase * 4 + piece]);
                if (!BFM) wunet_sth8(lo + o, tb[1][wbase * 4 + piece]);
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------- conv input, produced split
// prep_decim_kernel / prep_upcat_kernel (wunet_elementwise.h) writing the activated conv input straight into the
// hi / lo [B][C8][L][8] layout: the fp32 tensor and its split pass do not exist for these layers.  One thread per
// (channel group, 4 samples).  kind 0: decimate (C1 = 0), 1: upsample x2 + skip concat.
struct PrepH3Args {
    const float* z0; const float* a0; const float* s0;
    const float* z1; const float* a1; const float* s1;
    wunet_half* xh; wunet_half* xl;
    int B, C0, C1, C8, L, logL, kind;
    float up_scale;
    // kind 0 only: the same pass over the producer's z also writes its full-resolution activation - the skip half of the
    // decoder input that concatenates it (unet_basic.py:95) - into that layer's split arrays at channel group sc8off
    wunet_half* sh; wunet_half* sl; int SC8, sc8off;
    int up_only;         // kind 1: the skip half was written by the encoder-side pass: only the C0/8 upsampled groups
    // power-of-two scale of the split values: derived from the activation bounds of the sources (slot [4] of their layers),
    // published as {scale, 1/scale} in xsc by block 0; ssb0 / ssb1: the sources of the decoder layer whose skip half this pass
    // writes (its own pass derives and publishes the same scale)
    const float* xb0; const float* xb1; float* xsc;
    const float* ssb0; const float* ssb1;
    int bf;              // bf16 mode: one bf16 word per value into xh / sh, the lo arrays are not written
    int Lt;              // samples of a row that exist (<= L, the power-of-two row stride; PrepArgs::Lt); the skip destination has 2 Lt
};

// One thread = 8 channels of ONE sample, consecutive lanes = consecutive samples: every store instruction of a wave
// writes 64 x 16 contiguous bytes.  (The round-1 form - 4 consecutive samples per thread with 16-byte loads - left
// each store instruction 16-byte pieces at a 64-byte stride; measured with the stores permuted to lane-contiguous and
// nothing else changed: 37.4 -> 30.3 us average over the upsampling launches, 6.43 -> 6.32 ms per step; this form:
// 689 -> 523 us of operand passes per step, 6.36 -> 6.19 ms.  The same change to gz_split_h3_kernel, which reads twice
// what it writes, measured 0.4 % slower and was dropped.)
// MODE (compile time): 0 decimate, 1 decimate + skip destination, 2 upsample (+) skip concat, 3 upsample only.
// Every load of an iteration is issued before the first value is used (clamped indices instead of branches around loads, the
// conversion in a second phase): written with `if (c < C)` around each element's load, hipcc put one s_waitcnt vmcnt(0) per
// element - 8 to 24 serialised memory round trips per thread and iteration.
template <int MODE>
__global__ __launch_bounds__(WUNET_THREADS) void prep_h3_kernel(PrepH3Args A)
{
    constexpr int KIND = MODE >= 2 ? 1 : 0;
    constexpr bool SKIP_DST = MODE == 1, UP_ONLY = MODE == 3;
    const int Lh = A.L >> 1, C = A.C0 + A.C1;
    const int ngrp = UP_ONLY ? A.C0 / 8 : A.C8;              // channel groups this launch produces
    const size_t total = (size_t)A.B * ngrp * A.L;
    float xs_ = 1.0f, xinv_ = 1.0f, ss_ = 1.0f;
    wunet_x_scale(A.xb0, A.xb1, xs_, xinv_);
    if (blockIdx.x == 0 && threadIdx.x == 0) { A.xsc[0] = xs_; A.xsc[1] = xinv_; }
    if (SKIP_DST) { float si_; wunet_x_scale(A.ssb0, A.ssb1, ss_, si_); }
    for (size_t i = (size_t)blockIdx.x * WUNET_THREADS + threadIdx.x; i < total; i += (size_t)gridDim.x * WUNET_THREADS) {
        const int p = (int)(i & (size_t)(A.L - 1));
        const unsigned grow = (unsigned)(i >> A.logL);
        const int b = (int)(grow / (unsigned)ngrp), c8 = (int)(grow - (unsigned)b * ngrp);
        const size_t row = (size_t)b * A.C8 + c8;
        int i0 = 0, i1 = 0;
        float l0 = 0.0f, l1 = 0.0f;
        if (KIND != 0) wunet_up_coord(p, A.Lt >> 1, A.up_scale, i0, i1, l0, l1);     // once for the 8 channels
        // skip half of the decoder input at the producer's resolution (SKIP_DST): the wave's 64 samples p0 .. p0+63 come from the
        // 128 source samples 2*p0 .. 2*p0+127; this lane activates and writes source samples 2*p0+lane and 2*p0+64+lane
        // (lane-contiguous again).  Rows shorter than a wave: the thread's own pair 2p, 2p+1.
        const int lane = (int)(threadIdx.x & 63);
        const int q0 = A.L >= 64 ? 2 * (p - lane) + lane : 2 * p;
        const int qstep = A.L >= 64 ? 64 : 1;
        // ---- phase 1: every load of the iteration
        float za[8], zb[8], av[8], sv[8], zq[2][8];
        bool from_up[8];
        // the group's BatchNorm scale / shift as four 16-byte loads where its eight channels lie in ONE source (always for the decimating
        // kinds; for the concat when the upsampled half ends on a group boundary) - the arrays' rows are padded to 64 floats, values past
        // the last channel are zeroed below - instead of sixteen one-float loads beside the 8 - 16 that carry data
        const bool vec_c = KIND == 0 || UP_ONLY || (A.C0 & 7) == 0;
        if (vec_c) {
            const bool upg = KIND == 0 || UP_ONLY || c8 * 8 < A.C0;
            const float* const ap = upg ? A.a0 + c8 * 8 : A.a1 + (c8 * 8 - A.C0);
            const float* const sp = upg ? A.s0 + c8 * 8 : A.s1 + (c8 * 8 - A.C0);
            const wunet_f4 a0 = wunet_ld4(ap), a1 = wunet_ld4(ap + 4), s0 = wunet_ld4(sp), s1 = wunet_ld4(sp + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { av[e] = a0[e]; av[4 + e] = a1[e]; sv[e] = s0[e]; sv[4 + e] = s1[e]; }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = c8 * 8 + e;
            const int cc = c < C ? c : C - 1;                // (values of channels beyond C are zeroed below)
            from_up[e] = true;
            if (KIND == 0) {
                const float* zr = A.z0 + ((size_t)b * A.C0 + cc) * (size_t)(2 * A.L);
                za[e] = zr[2 * p];
                zb[e] = 0.0f;
                if (SKIP_DST) { zq[0][e] = zr[q0]; zq[1][e] = zr[q0 + qstep]; }
            } else {
                const bool up = UP_ONLY || cc < A.C0;
                from_up[e] = up;
                const int cu = up ? cc : 0, cs = up ? 0 : cc - A.C0;
                const float* zr = up ? A.z0 + ((size_t)b * A.C0 + cu) * Lh : A.z1 + ((size_t)b * A.C1 + cs) * A.L;
                za[e] = zr[up ? i0 : p];
                zb[e] = zr[up ? i1 : p];
                if (!vec_c) {
                    av[e] = up ? A.a0[cu] : A.a1[cs];
                    sv[e] = up ? A.s0[cu] : A.s1[cs];
                }
            }
        }
        // ---- phase 2: activate, scale, split, store
        {
            wunet_h8 h, l;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int c = c8 * 8 + e;
                float v;
                if (KIND == 0) v = wunet_lrelu(av[e] * za[e] + sv[e]);
                else {
                    const float u0 = wunet_lrelu(av[e] * za[e] + sv[e]), u1 = wunet_lrelu(av[e] * zb[e] + sv[e]);
                    v = from_up[e] ? l0 * u0 + l1 * u1 : u0;
                }
                if (c >= C || p >= A.Lt) v = 0.0f;                // (channels beyond C, row padding: zeros)
                wunet_half a, d;
                wunet_split_rt(A.bf, xs_ * v, a, d);
                wunet_put_half(h, e, a);
                wunet_put_half(l, e, d);
            }
            wunet_sth8(A.xh + (row * A.L + (size_t)p) * 8, h);
            if (!A.bf) wunet_sth8(A.xl + (row * A.L + (size_t)p) * 8, l);
        }
        if (SKIP_DST) {
            const size_t srow = (size_t)b * A.SC8 + A.sc8off + c8;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int q = q0 + k * qstep;
                wunet_h8 h, l;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int c = c8 * 8 + e;
                    const float v = (c < C && q < 2 * A.Lt) ? wunet_lrelu(av[e] * zq[k][e] + sv[e]) : 0.0f;
                    wunet_half a, d;
                    wunet_split_rt(A.bf, ss_ * v, a, d);
                    wunet_put_half(h, e, a);
                    wunet_put_half(l, e, d);
                }
                wunet_sth8(A.sh + (srow * (size_t)(2 * A.L) + (size_t)q) * 8, h);
                if (!A.bf) wunet_sth8(A.sl + (srow * (size_t)(2 * A.L) + (size_t)q) * 8, l);
            }
        }
    }
}

// ---------------------------------------------------------------------------- weight pack
// dst[mt][stage][slot][q][i][e], 5 slots per stage.  Full stage (chunk ch of 32 K channels, tap group tg; stage = ch * taps/5 + tg):
// W(row = mt*16+i, k-channel = ch*32+q*8+e, tap = tg*5+slot) - i.e. [mt][chunk][tap][q][i][e].  Tail stage g (ntt > 0; after the
// nfull full chunks: one per left-over group of 8 channels, conv_h3d_kernel's K tail): W(row, k-channel = nfull*32+g*8+e,
// tap = q*ntt+slot) for slot < ntt, zero for slot >= ntt or tap >= taps.  Forward: W = w[row][kch][tap];
// data gradient (transposed): W = w[kch][row][TAPS-1-tap].  hi and lo arrays.
struct PackH3Desc {
    const float* w;
    wunet_half* hi;
    wunet_half* lo;
    int Cout, Cin, taps;   // shape of w
    int rows, kch;         // GEMM rows / K channels (Cout,Cin forward; Cin,Cout transposed)
    int mtiles, nch;       // padded m-tiles, chunks of 32 K channels
    int transposed;
    const float* wmax;     // WUNET_WMAX_PARTS partial maxima of |w| (h3_scales_kernel)
    float* wsc;            // {scale, 1/scale} of the packed weights, published by block 0 (nullptr: another pack of this layer did)
    int bf;                // bf16 mode: one bf16 word per weight into hi
    int ntt;               // 0: every chunk full stages (nch chunks); else the steps of a tail stage, and
    int nfull, ns;         // ... the full chunks / the stages in all
};
struct PackH3Table { PackH3Desc d[WUNET_MAX_CONV_LAYERS]; };

static __global__ __launch_bounds__(WUNET_THREADS) void pack_h3_kernel(PackH3Table tab)
{
    const PackH3Desc& d = tab.d[blockIdx.y];
    const int ntg = d.taps / 5, ns = d.ntt ? d.ns : d.nch * ntg, nfs = d.ntt ? d.nfull * ntg : ns;
    const int total = d.mtiles * ns * 5 * 512;
    // the layer's weight scale: max |w| -> [2^13, 2^14); every block derives the same value from the partial maxima
    __shared__ float wsc_sh[2];
    if (threadIdx.x < 64) {
        float m = threadIdx.x < WUNET_WMAX_PARTS ? d.wmax[threadIdx.x] : 0.0f;
#pragma unroll
        for (int k = 1; k < 64; k <<= 1) m = fmaxf(m, wunet_shfl_xor(m, k));
        if (threadIdx.x == 0) {
            float s_, i_;
            wunet_pow2_scale(m, s_, i_);
            wsc_sh[0] = s_; wsc_sh[1] = i_;
            if (blockIdx.x == 0 && d.wsc) { d.wsc[0] = s_; d.wsc[1] = i_; }
        }
    }
    __syncthreads();
    const float wscale = wsc_sh[0];
    for (int idx = blockIdx.x * WUNET_THREADS + threadIdx.x; idx < total; idx += gridDim.x * WUNET_THREADS) {
        const int e = idx & 7, i = (idx >> 3) & 15, q = (idx >> 7) & 3;
        int r = idx >> 9;
        const int slot = r % 5; r /= 5;
        const int stg = r % ns, mt = r / ns;
        const int row = mt * 16 + i;
        int k, t;
        if (stg < nfs) { k = (stg / ntg) * 32 + q * 8 + e; t = (stg % ntg) * 5 + slot; }
        else { k = d.nfull * 32 + (stg - nfs) * 8 + e; t = slot < d.ntt ? q * d.ntt + slot : d.taps; }
        float v = 0.0f;
        if (row < d.rows && k < d.kch && t < d.taps)
            v = d.transposed ? d.w[((size_t)k * d.Cin + row) * d.taps + (d.taps - 1 - t)] : d.w[((size_t)row * d.Cin + k) * d.taps + t];
        wunet_half a, b;
        wunet_split_rt(d.bf, wscale * v, a, b);
        d.hi[idx] = a;
        if (!d.bf) d.lo[idx] = b;
    }
}

