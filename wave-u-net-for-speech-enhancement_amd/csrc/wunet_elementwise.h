// Memory-bound kernels of the Wave-U-Net hot path: weight packing, BatchNorm statistics
// finalisation (forward and backward), the output head, the gradient-assembly pass
// ("pass A": upsample^T / zero-stuffed decimation^T / skip add + LeakyReLU' + BN-backward sums),
// split-K reduction of weight gradients, and the losses.
#pragma once
#include "wunet_dev.h"

// ---------------------------------------------------------------------------- weight packing
// One descriptor per (layer, direction).  dst[mt][c][t][i] = W[o = mt*16+i][c][t] (forward) or the
// flipped/transposed tensor for the data gradient: dst[mt][c][t][i] = W[co = c][ci = mt*16+i][TAPS-1-t].
struct PackDesc {
    const float* w;   // reference layout [Cout][Cin][TAPS]
    float* dst;       // [Mtiles][CP][TAPS][16]
    int Cout, Cin, taps;
    int M;            // rows of the packed GEMM (Cout forward, Cin for dgrad)
    int CP;           // padded K-channel count
    int mtiles;       // padded m-tile count
    int transposed;   // 0 forward, 1 dgrad
};

#define WUNET_MAX_CONV_LAYERS 34
struct PackTable { PackDesc d[WUNET_MAX_CONV_LAYERS]; };

static __global__ __launch_bounds__(WUNET_THREADS) void pack_weights_kernel(PackTable tab)
{
    const PackDesc& d = tab.d[blockIdx.y];
    const int total = d.mtiles * d.CP * d.taps * 16;
    const int kc = d.transposed ? d.Cout : d.Cin;   // valid K-channels
    for (int idx = blockIdx.x * WUNET_THREADS + threadIdx.x; idx < total; idx += gridDim.x * WUNET_THREADS) {
        const int i = idx & 15;
        int r = idx >> 4;
        const int t = r % d.taps; r /= d.taps;
        const int c = r % d.CP;
        const int mt = r / d.CP;
        const int m = mt * 16 + i;
        float v = 0.0f;
        if (m < d.M && c < kc) {
            v = d.transposed ? d.w[((size_t)c * d.Cin + m) * d.taps + (d.taps - 1 - t)]
                             : d.w[((size_t)m * d.Cin + c) * d.taps + t];
        }
        d.dst[idx] = v;
    }
}

// ---------------------------------------------------------------------------- block reduce helper
// sums two doubles over the 256-thread block; result valid in thread 0. red = 2*WUNET_THREADS doubles of LDS (8 used).
// Butterfly inside each wave (no barrier), the four wave sums through LDS: two barriers instead of the nine of a 256-wide LDS tree -
// these reductions end kernels that are a few microseconds of pure latency (BatchNorm finalizes, split-K sums).  Fixed order.
__device__ __forceinline__ void block_sum2(double& a, double& b, double* red)
{
    const int tid = threadIdx.x;
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) {
        a += wunet_shfl_xor_d(a, m);
        b += wunet_shfl_xor_d(b, m);
    }
    __syncthreads();                           // (the previous user of red is done with it)
    if ((tid & 63) == 0) {
        red[tid >> 6] = a;
        red[WUNET_WAVES + (tid >> 6)] = b;
    }
    __syncthreads();
    a = (red[0] + red[1]) + (red[2] + red[3]);
    b = (red[WUNET_WAVES] + red[WUNET_WAVES + 1]) + (red[WUNET_WAVES + 2] + red[WUNET_WAVES + 3]);
}

// max of two non-negative floats over the block (same LDS buffer, after a block_sum2); result valid in thread 0
__device__ __forceinline__ void block_max2(float& a, float& b, double* red)
{
    float* fr = reinterpret_cast<float*>(red);
    const int tid = threadIdx.x;
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) {
        a = fmaxf(a, wunet_shfl_xor(a, m));
        b = fmaxf(b, wunet_shfl_xor(b, m));
    }
    __syncthreads();
    if ((tid & 63) == 0) {
        fr[tid >> 6] = a;
        fr[WUNET_WAVES + (tid >> 6)] = b;
    }
    __syncthreads();
    a = fmaxf(fmaxf(fr[0], fr[1]), fmaxf(fr[2], fr[3]));
    b = fmaxf(fmaxf(fr[WUNET_WAVES], fr[WUNET_WAVES + 1]), fmaxf(fr[WUNET_WAVES + 2], fr[WUNET_WAVES + 3]));
}

// ---------------------------------------------------------------------------- BN forward finalize
// grid = C.  Training: reduce the conv epilogue partials (bias-free sums) -> batch mean / biased var,
// scale/shift for the consumers, saved mean/rstd for backward, running-stat update (unbiased var,
// momentum 0.1) and num_batches_tracked += 1 (nn.BatchNorm1d defaults, reference unet_basic.py:12,25,55).
// Eval: scale/shift from the running statistics.
struct BnFwdArgs {
    const float* stats;   // [C][rows][2] or nullptr (eval)
    int rows;
    const float* bias;    // conv bias [C]
    const float* gamma;
    const float* beta;
    float* running_mean;
    float* running_var;
    long long* nbt;
    float* a;             // out: scale  gamma*rstd
    float* s;             // out: shift  beta - mean*a
    float* mean;          // out (training): batch mean of z
    float* rstd;          // out (training)
    float* cst;           // out (training) or nullptr: the same four as ONE 16-byte row per channel {a, s, mean, rstd} (conv_h3d_kernel<.., BSUM> loads a row's constants with one instruction)
    int C;
    double count;         // B*L
    int training;
};

// What bn_finalize_core reads of channel c besides the sums.  Loaded at the TOP of the kernels that end in it (every thread loads the same
// words: one broadcast line): as loads at their uses in thread 0's tail - between stores the compiler must assume they alias - they were five
// dependent memory round trips behind the reduction, most of a 4.7 us kernel that sits on the layer-to-layer chain 18 times per step.
struct BnFwdPre { float bias, gamma, beta, rmean, rvar; long long nbt; };
__device__ __forceinline__ BnFwdPre bn_finalize_preload(const BnFwdArgs& A, int c)
{
    BnFwdPre P;
    P.bias = A.bias[c]; P.gamma = A.gamma[c]; P.beta = A.beta[c]; P.rmean = A.running_mean[c]; P.rvar = A.running_var[c];
    P.nbt = c == 0 ? *A.nbt : 0;
    return P;
}
// thread 0 of the block owning channel c: turn (sum, sum of squares) of the bias-free conv into everything BN needs
__device__ __forceinline__ void bn_finalize_core(const BnFwdArgs& A, int c, double s1, double s2, const BnFwdPre& P)
{
    const double m0 = s1 / A.count;                   // mean of the bias-free conv
    double var = s2 / A.count - m0 * m0;              // biased variance (bias does not change it)
    var = var < 0.0 ? 0.0 : var;
    const double mean = m0 + (double)P.bias;
    const float rstd = (float)(1.0 / sqrt(var + 1e-5));
    const float a = P.gamma * rstd;
    const float sh = P.beta - (float)mean * a;
    A.a[c] = a;
    A.s[c] = sh;
    A.mean[c] = (float)mean;
    A.rstd[c] = rstd;
    if (A.cst) { A.cst[4 * c] = a; A.cst[4 * c + 1] = sh; A.cst[4 * c + 2] = (float)mean; A.cst[4 * c + 3] = rstd; }
    const double unbiased = A.count > 1.0 ? var * A.count / (A.count - 1.0) : var;
    A.running_mean[c] = (float)(0.9 * (double)P.rmean + 0.1 * mean);
    A.running_var[c] = (float)(0.9 * (double)P.rvar + 0.1 * unbiased);
    if (c == 0) *A.nbt = P.nbt + 1;
}

__device__ __forceinline__ void bn_eval_core(const BnFwdArgs& A, int c)
{
    const float rstd = (float)(1.0 / sqrt((double)A.running_var[c] + 1e-5));
    const float a = A.gamma[c] * rstd;
    A.a[c] = a;
    A.s[c] = A.beta[c] - A.running_mean[c] * a;
}

// Eval mode: the BatchNorm scale / shift of EVERY layer in one launch - they only depend on the running statistics, so nothing
// of it belongs on the layer-to-layer chain (25 launches of ~4 us less per eval forward).  grid = (ceil(maxC / 256), layers).
struct BnEvalDesc { const float* gamma; const float* beta; const float* running_mean; const float* running_var; float* a; float* s; int C; };
struct BnEvalTable { BnEvalDesc d[WUNET_MAX_CONV_LAYERS]; };
static __global__ __launch_bounds__(WUNET_THREADS) void bn_eval_all_kernel(BnEvalTable T)
{
    const BnEvalDesc& d = T.d[blockIdx.y];
    const int c = blockIdx.x * WUNET_THREADS + threadIdx.x;
    if (c >= d.C) return;
    BnFwdArgs A{};
    A.gamma = d.gamma; A.beta = d.beta; A.running_mean = const_cast<float*>(d.running_mean); A.running_var = const_cast<float*>(d.running_var);
    A.a = d.a; A.s = d.s;
    bn_eval_core(A, c);
}

static __global__ __launch_bounds__(WUNET_THREADS) void bn_finalize_fwd_kernel(BnFwdArgs A)
{
    __shared__ double red[2 * WUNET_THREADS];
    const int c = blockIdx.x, tid = threadIdx.x;
    if (!A.training) {
        if (tid == 0) bn_eval_core(A, c);
        return;
    }
    const BnFwdPre P = bn_finalize_preload(A, c);
    double s1 = 0.0, s2 = 0.0;
    const float2* st = reinterpret_cast<const float2*>(A.stats) + (size_t)c * A.rows;     // [C][rows][2]
    // (four loads in flight: a plain loop over the run-time row count waits for every load - up to 16 serialised round trips on
    // the long levels, whose tiles write 4096 rows per channel; same order of additions)
    int r = tid;
    for (; r + 3 * WUNET_THREADS < A.rows; r += 4 * WUNET_THREADS) {
        const float2 v0 = st[r], v1 = st[r + WUNET_THREADS], v2 = st[r + 2 * WUNET_THREADS], v3 = st[r + 3 * WUNET_THREADS];
        s1 += (double)v0.x; s2 += (double)v0.y;
        s1 += (double)v1.x; s2 += (double)v1.y;
        s1 += (double)v2.x; s2 += (double)v2.y;
        s1 += (double)v3.x; s2 += (double)v3.y;
    }
    for (; r < A.rows; r += WUNET_THREADS) {
        const float2 v = st[r];
        s1 += (double)v.x;
        s2 += (double)v.y;
    }
    block_sum2(s1, s2, red);
    if (tid == 0) bn_finalize_core(A, c, s1, s2, P);
}

// Split-K forward conv: sum the z-slices' partial outputs, add the bias, write z, and reduce the BatchNorm
// statistics.  grid = (C, rsplit): with rsplit == 1 (short levels) BatchNorm is finished in the same launch,
// otherwise each block writes one partial row [blockIdx.y][C][2] for bn_finalize_fwd_kernel.
// Sum of ksplit split-K partials (16 bytes each, `stride` floats apart) in split order, four loads in flight: written as a plain
// loop over a run-time ksplit every load was followed by s_waitcnt vmcnt(0) - ksplit serialised memory round trips per thread.
__device__ __forceinline__ wunet_f4 wunet_sum_splits4(const float* p, int ksplit, size_t stride)
{
    wunet_f4 v = wunet_f4{0.f, 0.f, 0.f, 0.f};
    int k = 0;
    for (; k + 4 <= ksplit; k += 4) {
        const wunet_f4 t0 = wunet_ld4(p + (size_t)k * stride), t1 = wunet_ld4(p + (size_t)(k + 1) * stride);
        const wunet_f4 t2 = wunet_ld4(p + (size_t)(k + 2) * stride), t3 = wunet_ld4(p + (size_t)(k + 3) * stride);
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] += t0[j]; v[j] += t1[j]; v[j] += t2[j]; v[j] += t3[j]; }
    }
    for (; k < ksplit; ++k) {
        const wunet_f4 t = wunet_ld4(p + (size_t)k * stride);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += t[j];
    }
    return v;
}

// xb (eval mode, a layer that feeds split operands; nullptr otherwise): the layer's activation bound max |a_c z + s_c| is folded into
// *xb by ONE atomic max per block (a / s are bn_eval_all_kernel's) - act_max_kernel's pass over z in the launch that writes z.
static __global__ __launch_bounds__(WUNET_THREADS) void conv_reduce_bn_kernel(BnFwdArgs A, const float* part, int ksplit,
                                                                        size_t split_stride, float* z, int B, int L, int logL,
                                                                        float* stats_rows, int Lt, float* xb)
{
    __shared__ double red[2 * WUNET_THREADS];
    const int c = blockIdx.x, tid = threadIdx.x;
    const float bias = A.bias[c];
    double s1 = 0.0, s2 = 0.0;
    const int total = B * L;
    if (!A.training) {
        // eval: z = sum of the splits + bias; no statistics (the coefficients come from the running ones, already finalised)
        const float ea = xb ? A.a[c] : 0.0f, es = xb ? A.s[c] : 0.0f;
        float m = 0.0f, m2 = 0.0f;
        if ((L & 3) == 0) {
            const int total4 = total >> 2;
            const int per = (total4 + gridDim.y - 1) / gridDim.y;
            const int beg = blockIdx.y * per, end = beg + per < total4 ? beg + per : total4;
            for (int p4 = beg + tid; p4 < end; p4 += WUNET_THREADS) {
                const int p = p4 << 2;
                const int b = p >> logL, l = p & (L - 1);
                const size_t off = ((size_t)b * A.C + c) * L + l;
                const wunet_f4 v = wunet_sum_splits4(part + off, ksplit, split_stride);
                wunet_f4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    o[j] = v[j] + bias;
                    m = fmaxf(m, fabsf(ea * o[j] + es));
                }
                wunet_st4(z + off, o);
            }
        } else {
            const int per = (total + gridDim.y - 1) / gridDim.y;
            const int beg = blockIdx.y * per, end = beg + per < total ? beg + per : total;
            for (int p = beg + tid; p < end; p += WUNET_THREADS) {
                const int b = p >> logL, l = p & (L - 1);
                const size_t off = ((size_t)b * A.C + c) * L + l;
                float v = 0.0f;
                for (int k = 0; k < ksplit; ++k) v += part[(size_t)k * split_stride + off];
                z[off] = v + bias;
                m = fmaxf(m, fabsf(ea * (v + bias) + es));
            }
        }
        if (xb) {
            block_max2(m, m2, red);
            if (tid == 0) wunet_atomic_absmax(xb, m);
        }
        return;
    }
    BnFwdPre P{};
    if (gridDim.y == 1) P = bn_finalize_preload(A, c);           // (this launch finishes BatchNorm itself)
    if ((L & 3) == 0) {
        // four samples per thread (16-byte loads and stores); every level of >= 4 samples
        const int total4 = total >> 2;
        const int per = (total4 + gridDim.y - 1) / gridDim.y;
        const int beg = blockIdx.y * per, end = beg + per < total4 ? beg + per : total4;
        for (int p4 = beg + tid; p4 < end; p4 += WUNET_THREADS) {
            const int p = p4 << 2;
            const int b = p >> logL, l = p & (L - 1);
            const size_t off = ((size_t)b * A.C + c) * L + l;
            const wunet_f4 v = wunet_sum_splits4(part + off, ksplit, split_stride);
            wunet_f4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[j] = v[j] + bias;
                if (l + j < Lt) {                                 // (row padding does not count)
                    s1 += (double)v[j];
                    s2 += (double)v[j] * (double)v[j];
                }
            }
            wunet_st4(z + off, o);
        }
    } else {
        const int per = (total + gridDim.y - 1) / gridDim.y;
        const int beg = blockIdx.y * per, end = beg + per < total ? beg + per : total;
        for (int p = beg + tid; p < end; p += WUNET_THREADS) {
            const int b = p >> logL, l = p & (L - 1);
            const size_t off = ((size_t)b * A.C + c) * L + l;
            float v = 0.0f;
            for (int k = 0; k < ksplit; ++k) v += part[(size_t)k * split_stride + off];
            z[off] = v + bias;
            if (l < Lt) {
                s1 += (double)v;
                s2 += (double)v * (double)v;
            }
        }
    }
    block_sum2(s1, s2, red);
    if (tid == 0) {
        if (gridDim.y > 1) {
            float* st = stats_rows + ((size_t)c * gridDim.y + blockIdx.y) * 2;
            st[0] = (float)s1;
            st[1] = (float)s2;
        } else bn_finalize_core(A, c, s1, s2, P);
    }
}

// sum of split-K partial tensors (data gradient of the short levels)
static __global__ __launch_bounds__(WUNET_THREADS) void split_sum_kernel(const float* part, int ksplit, size_t n, float* out,
                                                                   const float* bias, int C, int logL)
{
    if ((n & 3) == 0 && (bias == nullptr || logL >= 2)) {      // 16-byte loads and stores (4 samples of one row)
        const size_t n4 = n >> 2;
        for (size_t i4 = (size_t)blockIdx.x * WUNET_THREADS + threadIdx.x; i4 < n4; i4 += (size_t)gridDim.x * WUNET_THREADS) {
            const size_t i = i4 << 2;
            wunet_f4 v = wunet_sum_splits4(part + i, ksplit, n);
            const float bv = bias ? bias[(i >> logL) % (size_t)C] : 0.0f;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += bv;
            wunet_st4(out + i, v);
        }
        return;
    }
    for (size_t i = (size_t)blockIdx.x * WUNET_THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * WUNET_THREADS) {
        float v = 0.0f;
        for (int k = 0; k < ksplit; ++k) v += part[(size_t)k * n + i];
        out[i] = v + (bias ? bias[(i >> logL) % (size_t)C] : 0.0f);
    }
}

struct HeadFwdArgs {
    const float* z;    // [B][C][T] raw conv output of the last decoder layer
    const float* a;
    const float* s;
    const float* in;   // [B][T]
    const float* wh;   // [C+1]
    const float* bh;   // [1]
    float* out;        // [B][T]
    int B, C, T, logT;
};

// One thread = 4 consecutive samples; the channels are walked 24 (then eight) at a time with all the 16-byte loads of a group issued before the first is used
// (rounds 1 - 5: one sample per thread and a run-time loop over the channels - 24 dependent 4-byte round trips per thread, 2.3 - 2.7 TB/s;
// same order of additions, bit-identical results).
static __global__ __launch_bounds__(WUNET_THREADS) void head_fwd_kernel(HeadFwdArgs A)
{
    const size_t total4 = ((size_t)A.B * A.T) >> 2;            // T is a power of two >= 4
    const float bh = A.bh[0], win = A.wh[A.C];
    for (size_t q4 = (size_t)blockIdx.x * WUNET_THREADS + threadIdx.x; q4 < total4; q4 += (size_t)gridDim.x * WUNET_THREADS) {
        const size_t p = q4 << 2;
        const size_t b = p >> A.logT, t = p & (size_t)(A.T - 1);
        const float* zr = A.z + b * A.C * A.T + t;
        const wunet_f4 xin = wunet_ld4(A.in + p);
        wunet_f4 acc = wunet_f4{bh, bh, bh, bh};
        int c = 0;
        for (; c + 24 <= A.C; c += 24) {           // (the reference's 24 channels: every load of the thread in flight at once)
            wunet_f4 v[24];
#pragma unroll
            for (int e = 0; e < 24; ++e) v[e] = wunet_ld4(zr + (size_t)(c + e) * A.T);
#pragma unroll
            for (int e = 0; e < 24; ++e) {
                const float av = A.a[c + e], sv = A.s[c + e], wv = A.wh[c + e];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] += wv * wunet_lrelu(av * v[e][j] + sv);
            }
        }
        for (; c + 8 <= A.C; c += 8) {
            wunet_f4 v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = wunet_ld4(zr + (size_t)(c + e) * A.T);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float av = A.a[c + e], sv = A.s[c + e], wv = A.wh[c + e];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] += wv * wunet_lrelu(av * v[e][j] + sv);
            }
        }
        for (; c < A.C; ++c) {
            const wunet_f4 v = wunet_ld4(zr + (size_t)c * A.T);
            const float av = A.a[c], sv = A.s[c], wv = A.wh[c];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] += wv * wunet_lrelu(av * v[j] + sv);
        }
        wunet_f4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = tanhf(acc[j] + win * xin[j]);
        wunet_st4(A.out + p, o);
    }
}

// ---------------------------------------------------------------------------- output head backward
// gh = gout * (1 - out^2); partial sums for d wh[c] = sum gh*y_c (c<C), d wh[C] = sum gh*in, d bh = sum gh.
// grid = (nblk); partial layout part[blk][C+2].
struct HeadBwdArgs {
    const float* z; const float* a; const float* s; const float* in;
    const float* out; const float* gout;
    float* gh;        // [B][T]
    float* part;      // [gridDim.x][2]: sum gh*input, sum gh (the per-channel sums come from pass_a_kernel<A_HEAD>)
    int B, C, T, logT;
};

static __global__ __launch_bounds__(WUNET_THREADS) void head_bwd_kernel(HeadBwdArgs A)
{
    __shared__ double red[2 * WUNET_THREADS];
    const size_t total = (size_t)A.B * A.T;
    const size_t per = (total + gridDim.x - 1) / gridDim.x;
    const size_t beg = (size_t)blockIdx.x * per, end = beg + per < total ? beg + per : total;
    // pass 1: gh and the two channel-free sums
    double sb = 0.0, sin_ = 0.0;
    // four trips' loads in flight (the store of a trip kept the next trip's loads behind it: eight serialised round trips per thread, 9.3 us for
    // 16 MB); the trips are added in the one-trip loop's order - the same bits
    size_t p = beg + threadIdx.x;
    for (; p + 3 * WUNET_THREADS < end; p += 4 * WUNET_THREADS) {
        float o[4], go[4], xi[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { o[u] = A.out[p + u * WUNET_THREADS]; go[u] = A.gout[p + u * WUNET_THREADS]; xi[u] = A.in[p + u * WUNET_THREADS]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float g = go[u] * (1.0f - o[u] * o[u]);
            A.gh[p + u * WUNET_THREADS] = g;
            sb += (double)g;
            sin_ += (double)g * (double)xi[u];
        }
    }
    for (; p < end; p += WUNET_THREADS) {
        const float o = A.out[p];
        const float g = A.gout[p] * (1.0f - o * o);
        A.gh[p] = g;
        sb += (double)g;
        sin_ += (double)g * (double)A.in[p];
    }
    block_sum2(sb, sin_, red);
    if (threadIdx.x == 0) {
        A.part[(size_t)blockIdx.x * 2] = (float)sin_;
        A.part[(size_t)blockIdx.x * 2 + 1] = (float)sb;
    }
}

// sums partial rows: out[j] = sum_r part[r][j]   (grid = n, one block per column)
static __global__ __launch_bounds__(WUNET_THREADS) void rows_sum_kernel(const float* part, int rows, int n, float* out0, int n0, float* out1)
{
    __shared__ double red[2 * WUNET_THREADS];
    const int j = blockIdx.x;
    double s = 0.0, dummy = 0.0;
    for (int r = threadIdx.x; r < rows; r += WUNET_THREADS) s += (double)part[(size_t)r * n + j];
    block_sum2(s, dummy, red);
    if (threadIdx.x == 0) {
        if (j < n0) out0[j] = (float)s; else out1[j - n0] = (float)s;
    }
}

// ---------------------------------------------------------------------------- pass A (gradient assembly)
// Builds g_pre = dL/d(BN output) of one layer from its consumers' data gradients, applies the
// LeakyReLU derivative, writes g_pre and the partial sums  sum g_pre, sum g_pre*xhat  per channel.
//   A_HEAD : g_y[b,c,l] = wh[c] * gh[b,l]                                  (last decoder layer)
//   A_UP   : g_y[b,c,i] = sum_j w(j->i) dX[b,c,j], dX of the next decoder layer at 2L, upsample^T
//   A_ENC  : g_y[b,c,l] = dXdec[b, coff+c, l] + (l even ? dXenc[b,c,l/2] : 0)   (skip + decimation^T)
// grid = (C, nsplit); part[split][C][2].
//   A_UPH  : g_y[b,c,i] = dXh[b,c,i] (+ the tile-edge terms): the next decoder layer's data gradient arrives already pulled back through
//            the upsample (conv_h3d_kernel<.., 3>, ConvH3Args::uh_*) - a plain elementwise pass
enum { A_HEAD = 0, A_UP = 1, A_ENC = 2, A_UPH = 3 };

// The tile-edge terms of a half-resolution data gradient (ConvH3Args::uh_spill [2][C][ntiles]; a tile = 128 inputs of one row): input l, the
// first of a tile, is owed sp[0][c][tile - 1]; the last of a tile sp[1][c][tile + 1].  g: the four inputs l .. l + 3 (l a multiple of 4) of
// channel c, item b; tpr = tiles per row.  Row ends are owed nothing (the terms there are exact zeros, but the neighbour tile may not exist).
// pass_a_kernel<A_UPH> - one channel per thread, no loads to batch - adds them and writes the two completed values back into the array, so the
// second reader (gz_split_h3_kernel's UPH mode, eight channels per thread) reads finished values with nothing conditional among its loads.
__device__ __forceinline__ void wunet_uph_edges(const float* sp, int C, int ntiles, int tpr, int b, int c, int l, int L, float* gq_at_l, float& g_first, float& g_last)
{
    const int tile = b * tpr + (l >> 7);
    if ((l & 127) == 0 && l > 0) { g_first += sp[(size_t)c * ntiles + tile - 1]; gq_at_l[0] = g_first; }
    if ((l & 127) == 124 && l + 4 < L) { g_last += sp[((size_t)C + c) * ntiles + tile + 1]; gq_at_l[3] = g_last; }
}

// Transposed x2 upsample of ONE row, four inputs at a time, for callers that walk several channels at the same positions (gz_split_h3_kernel's
// UP mode: 8 channels per thread): the coordinates of outputs 2l - 1 .. 2l + 8 once (wunet_upT_coords), then per row the ten data-gradient
// values and sixteen multiply-adds (wunet_upT_row) - pass_a_kernel<A_UP>'s arithmetic in pass_a_kernel's order: input i receives, in ascending
// j, l1(2i-1) d[2i-1] + l1(2i) d[2i] + l0(2i+1) d[2i+1] + l0(2i+2) d[2i+2].  `fast`: interior positions whose outputs read the regular
// source pairs ((j-1) >> 1, +1) (checked against ATen's fp32 coordinates); the row ends take the general form.
struct WunetUpT { float c0[10], c1[10]; bool fast; };
__device__ __forceinline__ void wunet_upT_coords(int l, int Lt, float up_scale, WunetUpT& U)
{
    const int j0 = 2 * l - 4;
    U.fast = l >= 4 && l + 8 <= Lt;
    if (U.fast) {
#pragma unroll
        for (int k = 3; k <= 12; ++k) {
            int i0, i1;
            wunet_up_coord(j0 + k, Lt, up_scale, i0, i1, U.c0[k - 3], U.c1[k - 3]);
            U.fast = U.fast && i0 == ((j0 + k - 1) >> 1) && i1 == i0 + 1;
        }
    }
}
// row: the data gradient of one (item, channel) at 2L samples (row stride Lo = 2L, Lot = 2 Lt of them exist); g: inputs l .. l + 3
__device__ __forceinline__ void wunet_upT_row(const float* row, int l, int Lo, int Lot, int Lt, float up_scale, const WunetUpT& U, float (&g)[4])
{
    const int j0 = 2 * l - 4;
    g[0] = g[1] = g[2] = g[3] = 0.0f;
    if (U.fast) {
        const float dm = row[j0 + 3];
        const wunet_f4 da = wunet_ld4(row + j0 + 4), db = wunet_ld4(row + j0 + 8);
        const float dl = row[j0 + 12];
        const float d[10] = {dm, da[0], da[1], da[2], da[3], db[0], db[1], db[2], db[3], dl};      // outputs 2l - 1 .. 2l + 8
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            g[m] += U.c1[2 * m] * d[2 * m];
            g[m] += U.c1[2 * m + 1] * d[2 * m + 1];
            g[m] += U.c0[2 * m + 2] * d[2 * m + 2];
            g[m] += U.c0[2 * m + 3] * d[2 * m + 3];
        }
        return;
    }
#pragma unroll
    for (int k = 2; k <= 12; ++k) {                   // j = 2l-2 .. 2l+8, ascending like ATen's backward loop
        const int j = j0 + k;
        if (j >= 0 && j < Lot) {
            const float dv = row[j];
            int i0, i1; float l0, l1;
            wunet_up_coord(j, Lt, up_scale, i0, i1, l0, l1);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const float w = (i0 == l + m ? l0 : 0.0f) + (i1 == l + m ? l1 : 0.0f);
                g[m] += w * dv;
            }
        }
    }
    (void)Lo;
}

struct PassAArgs {
    const float* z;      // [B][C][L]
    const float* a;      // scale/shift of this layer (sign of the pre-activation)
    const float* s;
    const float* mean;
    const float* rstd;
    float* gpre;         // out [B][C][L]
    float* part;         // [nsplit][C][2]
    float* pmax;         // nullptr or [nsplit][C][2]: max |g_pre|, max |z - mean| (bound of |g_z| for the fp16-split scale)
    float* hpart;        // HEAD: [nsplit][C] partial head-weight gradients  sum gh * act(z_c)  (same pass over z and gh)
    const float* g0;     // HEAD: gh [B][L];  UP: dX [B][Cg0][2L];  ENC: dXdec [B][Cg0][L];  UPH: dXh [B][C][L] (its tile-edge values are completed IN PLACE)
    const float* g1;     // HEAD: wh;         ENC: dXenc [B][C][L/2]
    int Cg0;             // channel count of the g0 tensor
    int g1_splits;       // ENC: > 1 - g1 points at the split-K partials of the data gradient (g1_stride floats apart): this pass is their
    size_t g1_stride;    //      only reader and adds them itself, in split order, instead of a split_sum_kernel launch in front of it
    int coff;            // ENC: channel offset of the skip part inside dXdec
    int B, C, L, logL;
    float up_scale;      // UP: (float)(Lt-1)/(2Lt-1)
    int swap;            // grid = (splits, C) instead of (C, splits)
    int Lt;              // samples of a row that exist (<= L, the power-of-two row stride; PrepArgs::Lt): the padding gets no gradient
    const float* sp;     // UPH: the tile-edge terms [2][C][ntiles] (g0 = dXh [B][C][L])
    int ntiles, tpr;     // UPH: tiles of the consumer's data gradient, tiles per row (L / 128)
    // E0 (the first layer, Cin = 1, 15 taps): its weight gradient dW[c][k] = sum g_z[b,c,l] x[b,l+k-7] with g_z = k1 g + k2 z + k3 is three sums per
    // tap that need no BatchNorm constant - sum g x, sum z x, sum x over the positions that exist - taken by THIS pass, which reads z and forms g
    // anyway; bn_finalize_bwd_kernel combines them.  x0: the input waveform [B][L] (zero beyond the samples that exist); e0rows [pieces][C][48].
    const float* x0; float* e0rows;
    // FUSE (a whole channel in one pass of one block: B*L <= 1024): BatchNorm-backward finalize and g_z in the same launch -
    // gpre receives g_z = k1*g + k2*z + k3 directly, part is not written (bn_finalize_bwd_kernel + gz_materialize_kernel)
    const float* gamma; float* dgamma; float* dbeta; float* dbias; float* k1; float* k2; float* k3; double count;
};

// The pass's closing reduction: two sums (three in HEAD mode) and two maxima over the block behind ONE barrier - butterflies inside each wave, the four
// waves' values through LDS, added in block_sum2's order (the same bits).  Rounds 1 - 6 ran block_sum2, block_max2 (and a second block_sum2
// in HEAD mode) one after the other: four to seven barriers at the end of blocks that are only 4 - 8 loop trips long.  Sums valid in every thread,
// maxima too.  red: 5 * WUNET_WAVES doubles.
__device__ __forceinline__ void block_reduce_pass_a(double& s1, double& s2, double& s3, bool want3, float& mg, float& mz, double* red)
{
    const int tid = threadIdx.x;
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) {
        s1 += wunet_shfl_xor_d(s1, m);
        s2 += wunet_shfl_xor_d(s2, m);
        if (want3) s3 += wunet_shfl_xor_d(s3, m);
        mg = fmaxf(mg, wunet_shfl_xor(mg, m));
        mz = fmaxf(mz, wunet_shfl_xor(mz, m));
    }
    if ((tid & 63) == 0) {
        const int w = tid >> 6;
        red[w] = s1; red[WUNET_WAVES + w] = s2; red[2 * WUNET_WAVES + w] = s3;
        red[3 * WUNET_WAVES + w] = (double)mg; red[4 * WUNET_WAVES + w] = (double)mz;        // (a float survives the round trip through a double)
    }
    __syncthreads();
    s1 = (red[0] + red[1]) + (red[2] + red[3]);
    s2 = (red[WUNET_WAVES] + red[WUNET_WAVES + 1]) + (red[WUNET_WAVES + 2] + red[WUNET_WAVES + 3]);
    s3 = (red[2 * WUNET_WAVES] + red[2 * WUNET_WAVES + 1]) + (red[2 * WUNET_WAVES + 2] + red[2 * WUNET_WAVES + 3]);
    mg = fmaxf(fmaxf((float)red[3 * WUNET_WAVES], (float)red[3 * WUNET_WAVES + 1]), fmaxf((float)red[3 * WUNET_WAVES + 2], (float)red[3 * WUNET_WAVES + 3]));
    mz = fmaxf(fmaxf((float)red[4 * WUNET_WAVES], (float)red[4 * WUNET_WAVES + 1]), fmaxf((float)red[4 * WUNET_WAVES + 2], (float)red[4 * WUNET_WAVES + 3]));
}

// What one loop trip of pass A loads for its four samples (everything that comes from memory, nothing derived): two trips' loads are issued
// before the first trip's values are used.
template <int MODE, bool E0 = false>
struct PassALoads {
    wunet_f4 z, g;                       // z; HEAD: gh, UPH: dXh, ENC: dXdec
    float e0, e1;                        // ENC: dXenc at l/2, l/2 + 1 (the split-K partials already added, in split order)
    float d[MODE == A_UP ? 13 : 1];      // UP: dX at 2l - 4 .. 2l + 8
    wunet_f4 xw[E0 ? 5 : 1];             // E0: the input waveform at l - 8 .. l + 11
    size_t zi; int b, l;
};

// G1S (ENC only, compile time): g1 points at split-K partials (A.g1_splits > 1) - as a run-time test inside the trip it kept the two trips' loads apart
template <int MODE, bool FUSE = false, bool G1S = false, bool E0 = false>
__global__ __launch_bounds__(WUNET_THREADS) void pass_a_kernel(PassAArgs A)
{
    static_assert(!E0 || (MODE == A_ENC && !FUSE && !G1S), "E0: the first layer's gradient assembly");
    __shared__ double red[5 * WUNET_WAVES];
    __shared__ float e0x[E0 ? WUNET_WAVES * 48 : 1];
    float w1[E0 ? 15 : 1], w2[E0 ? 15 : 1], w3[E0 ? 15 : 1];          // E0: sum g x, sum z x, sum x per tap (this thread's positions)
    if (E0) {
#pragma unroll
        for (int k = 0; k < 15; ++k) w1[E0 ? k : 0] = w2[E0 ? k : 0] = w3[E0 ? k : 0] = 0.0f;
    }
    // grid (C, splits), or (splits, C) with A.swap: consecutive blocks then walk consecutive pieces of ONE channel row
    const int c = A.swap ? blockIdx.y : blockIdx.x;
    const unsigned by = A.swap ? blockIdx.x : blockIdx.y, ny = A.swap ? gridDim.x : gridDim.y;
    // each thread produces four consecutive samples (aligned float4 traffic); L is a power of two >= 4
    const size_t total4 = ((size_t)A.B * A.L) >> 2;
    const size_t per = (total4 + ny - 1) / ny;
    const size_t beg = (size_t)by * per, end = beg + per < total4 ? beg + per : total4;
    const float a = A.a[c], s = A.s[c], mu = A.mean[c], rstd = A.rstd[c];
    const float wh = MODE == A_HEAD ? A.g1[c] : 0.0f;
    const float gam = FUSE ? A.gamma[c] : 0.0f;          // (FUSE: needed behind the reduction - loaded here, not there)
    double s1 = 0.0, s2 = 0.0, s3 = 0.0;
    float mg = 0.0f, mz = 0.0f;
    wunet_f4 keep_g = wunet_f4{0.f, 0.f, 0.f, 0.f}, keep_z = keep_g;
    size_t keep_i = 0;
    int keep_l = 0;
    bool have = false;

    // ---- the loads of one trip
    auto load = [&](size_t q4, PassALoads<MODE, E0>& t) {
        const size_t p = q4 << 2;
        t.b = (int)(p >> A.logL); t.l = (int)(p & (size_t)(A.L - 1));
        const int b = t.b, l = t.l;
        t.zi = ((size_t)b * A.C + c) * A.L + l;
        t.z = wunet_ld4(A.z + t.zi);
        t.e0 = t.e1 = 0.0f;
        if (MODE == A_HEAD) t.g = wunet_ld4(A.g0 + (size_t)b * A.L + l);
        else if (MODE == A_UPH) {
            // the tile-edge terms (wunet_uph_edges) with the trip's other loads: clamped indices, the values of the lanes that are owed nothing
            // are not used (as a conditional load inside the trip, every wave - each spans two tiles - waited for all its loads there)
            t.g = wunet_ld4(A.g0 + t.zi);
            const int tile = b * A.tpr + (l >> 7);
            t.e0 = A.sp[(size_t)c * A.ntiles + (tile > 0 ? tile - 1 : 0)];
            t.e1 = A.sp[((size_t)A.C + c) * A.ntiles + (tile + 1 < A.ntiles ? tile + 1 : tile)];
        }
        else if (MODE == A_ENC) {
            t.g = wunet_ld4(A.g0 + ((size_t)b * A.Cg0 + A.coff + c) * A.L + l);
            const float* ge = A.g1 + ((size_t)b * A.C + c) * (A.L >> 1) + (l >> 1);
            float e0, e1;
            if (G1S) {                                        // (four loads in flight, the order of split_sum_kernel's additions)
                e0 = e1 = 0.0f;
                int k = 0;
                for (; k + 4 <= A.g1_splits; k += 4) {
                    const float2 t0 = *reinterpret_cast<const float2*>(ge + (size_t)k * A.g1_stride);
                    const float2 t1 = *reinterpret_cast<const float2*>(ge + (size_t)(k + 1) * A.g1_stride);
                    const float2 t2 = *reinterpret_cast<const float2*>(ge + (size_t)(k + 2) * A.g1_stride);
                    const float2 t3 = *reinterpret_cast<const float2*>(ge + (size_t)(k + 3) * A.g1_stride);
                    e0 += t0.x; e0 += t1.x; e0 += t2.x; e0 += t3.x;
                    e1 += t0.y; e1 += t1.y; e1 += t2.y; e1 += t3.y;
                }
                if (k < A.g1_splits) {
                    // the one to three splits left over: loaded together (clamped indices), added in split order under selects (one at a time
                    // they were up to three more dependent round trips per trip of a launch that is latency from end to end)
                    const int last = A.g1_splits - 1;
                    const float2 t0 = *reinterpret_cast<const float2*>(ge + (size_t)k * A.g1_stride);
                    const float2 t1 = *reinterpret_cast<const float2*>(ge + (size_t)(k + 1 < last ? k + 1 : last) * A.g1_stride);
                    const float2 t2 = *reinterpret_cast<const float2*>(ge + (size_t)(k + 2 < last ? k + 2 : last) * A.g1_stride);
                    e0 += t0.x; e1 += t0.y;           // (selects, not branches: hipcc sinks a load whose only use is conditional into the branch)
                    e0 += k + 1 <= last ? t1.x : 0.0f; e1 += k + 1 <= last ? t1.y : 0.0f;
                    e0 += k + 2 <= last ? t2.x : 0.0f; e1 += k + 2 <= last ? t2.y : 0.0f;
                }
            } else { const float2 tt = *reinterpret_cast<const float2*>(ge); e0 = tt.x; e1 = tt.y; }
            t.e0 = e0; t.e1 = e1;
            if (E0) {
                // l and L are multiples of 4: an aligned 16-byte piece of the row lies wholly inside [0, L) or wholly outside (zero padding of the conv)
                const float* xr = A.x0 + (size_t)b * A.L;
#pragma unroll
                for (int v = 0; v < 5; ++v) {
                    const int lv = l - 8 + 4 * v;
                    const bool ok = lv >= 0 && lv < A.L;
                    t.xw[E0 ? v : 0] = wunet_sel4(ok, wunet_ld4(xr + (ok ? lv : 0)));
                }
            }
        } else {
            const int Lo = 2 * A.L;                               // row stride of the upsampled tensor
            const float* row = A.g0 + ((size_t)b * A.Cg0 + c) * Lo;
            const int j0 = 2 * l - 4;                         // 16-byte aligned
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                const int jv = j0 + 4 * v;
                const bool ok = jv >= 0 && jv < Lo;
                const wunet_f4 tt = wunet_sel4(ok, wunet_ld4(row + (ok ? jv : 0)));
                t.d[(MODE == A_UP ? 4 * v : 0)] = tt[0]; t.d[(MODE == A_UP ? 4 * v + 1 : 0)] = tt[1];
                t.d[(MODE == A_UP ? 4 * v + 2 : 0)] = tt[2]; t.d[(MODE == A_UP ? 4 * v + 3 : 0)] = tt[3];
            }
            t.d[MODE == A_UP ? 12 : 0] = (j0 + 12 < Lo) ? row[j0 + 12] : 0.0f;
            t.g = wunet_f4{0.f, 0.f, 0.f, 0.f};
        }
    };

    // ---- g of the trip's four samples, LeakyReLU', the sums
    auto compute = [&](const PassALoads<MODE, E0>& t) {
        const int b = t.b, l = t.l;
        const wunet_f4 z = t.z;
        float g[4];
        if (MODE == A_HEAD) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                g[j] = wh * t.g[j];
                s3 += (double)(t.g[j] * wunet_lrelu(a * z[j] + s));      // d(head weight of channel c)
            }
        } else if (MODE == A_UPH) {
            float* const gq = const_cast<float*>(A.g0) + t.zi;
            g[0] = t.g[0]; g[1] = t.g[1]; g[2] = t.g[2]; g[3] = t.g[3];
            // wunet_uph_edges on the values loaded above: the first input of a tile is owed sp[0][c][tile - 1], the last sp[1][c][tile + 1]; row ends nothing
            // (selects, not branches around the additions: hipcc sinks a load whose only use is conditional back into the branch)
            const bool first = (l & 127) == 0 && l > 0, last = (l & 127) == 124 && l + 4 < A.L;
            g[0] += first ? t.e0 : 0.0f;
            g[3] += last ? t.e1 : 0.0f;
            if (first) gq[0] = g[0];
            if (last) gq[3] = g[3];
        } else if (MODE == A_ENC) {
            g[0] = t.g[0] + t.e0; g[1] = t.g[1]; g[2] = t.g[2] + t.e1; g[3] = t.g[3];
        } else {
            // transpose of ATen's upsample_linear1d: output j contributes l0 to input i0(j) and l1 to i1(j), with the
            // fp32-computed coordinates; inputs l..l+3 can only be hit by outputs j in [2l-2, 2l+8], walked in
            // ascending j like ATen's backward loop
            const int Lot = 2 * A.Lt;                             // samples that exist of the upsampled tensor
            const float* d = t.d;
            const float dlast = t.d[MODE == A_UP ? 12 : 0];
            const int j0 = 2 * l - 4;
            g[0] = g[1] = g[2] = g[3] = 0.0f;
            // interior threads: ATen's source pair of output j is ((j-1)>>1, +1) (checked against the exact coordinates), so
            // input i receives, in ascending j, l1(2i-1) d[2i-1] + l1(2i) d[2i] + l0(2i+1) d[2i+1] + l0(2i+2) d[2i+2]: four
            // multiply-adds instead of eleven rounds of compare-and-select (the pass is VALU-heavy: ~300 -> ~130 instructions)
            bool fast = l >= 4 && l + 8 <= A.Lt;
            float c0[10], c1[10];
            if (fast) {
#pragma unroll
                for (int k = 3; k <= 12; ++k) {
                    int i0, i1;
                    wunet_up_coord(j0 + k, A.Lt, A.up_scale, i0, i1, c0[k - 3], c1[k - 3]);
                    fast = fast && i0 == ((j0 + k - 1) >> 1) && i1 == i0 + 1;
                }
            }
            if (fast) {
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    g[m] += c1[2 * m] * d[MODE == A_UP ? 2 * m + 3 : 0];
                    g[m] += c1[2 * m + 1] * d[MODE == A_UP ? 2 * m + 4 : 0];
                    g[m] += c0[2 * m + 2] * d[MODE == A_UP ? 2 * m + 5 : 0];
                    g[m] += c0[2 * m + 3] * (m == 3 ? dlast : d[MODE == A_UP ? (m == 3 ? 0 : 2 * m + 6) : 0]);
                }
            } else
#pragma unroll
            for (int k = 2; k <= 12; ++k) {                   // j = 2l-2 .. 2l+8
                const int j = j0 + k;
                const float dv = k < 12 ? d[MODE == A_UP ? (k < 12 ? k : 0) : 0] : dlast;
                if (j >= 0 && j < Lot) {
                    int i0, i1; float l0, l1;
                    wunet_up_coord(j, A.Lt, A.up_scale, i0, i1, l0, l1);
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        const float w = (i0 == l + m ? l0 : 0.0f) + (i1 == l + m ? l1 : 0.0f);
                        g[m] += w * dv;
                    }
                }
            }
        }
        wunet_f4 go;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float gv = g[j];
            if (!(a * z[j] + s > 0.0f)) gv *= WUNET_SLOPE;
            const bool pad = l + j >= A.Lt;                       // row padding: no gradient, no part in the sums or bounds
            if (pad) gv = 0.0f;
            const float zc = pad ? 0.0f : z[j] - mu;
            go[j] = gv;
            s1 += (double)gv;
            s2 += (double)(gv * (zc * rstd));
            mg = fmaxf(mg, fabsf(gv));
            mz = fmaxf(mz, fabsf(zc));
        }
        if (E0) {
            float xs[20];
#pragma unroll
            for (int v = 0; v < 5; ++v) { xs[4 * v] = t.xw[E0 ? v : 0][0]; xs[4 * v + 1] = t.xw[E0 ? v : 0][1]; xs[4 * v + 2] = t.xw[E0 ? v : 0][2]; xs[4 * v + 3] = t.xw[E0 ? v : 0][3]; }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool pad = l + j >= A.Lt;
                const float zz = pad ? 0.0f : z[j], one = pad ? 0.0f : 1.0f;
#pragma unroll
                for (int k = 0; k < 15; ++k) {                    // x[l + j + k - 7] = xs[j + k + 1]
                    w1[E0 ? k : 0] = fmaf(go[j], xs[j + k + 1], w1[E0 ? k : 0]);
                    w2[E0 ? k : 0] = fmaf(zz, xs[j + k + 1], w2[E0 ? k : 0]);
                }
                if (c == 0) {                                     // (sum x does not depend on the channel: channel 0's blocks take it for all)
#pragma unroll
                    for (int k = 0; k < 15; ++k) w3[E0 ? k : 0] = fmaf(one, xs[j + k + 1], w3[E0 ? k : 0]);
                }
            }
        }
        if (FUSE) { keep_g = go; keep_z = z; keep_i = t.zi; keep_l = l; have = true; }      // the thread's only trip
        else if (A.gpre) wunet_st4(A.gpre + t.zi, go);          // (nullptr: the consumer recomputes g - gz_split_h3_kernel's recompute modes)
    };

    // two trips' loads in flight per thread (rounds 1 - 6: one trip's two 16-byte loads, waited for, used, then the next trip's); the trips are
    // computed in the order of the one-trip loop (HEAD / UPH: every gradient bit-identical to the one-trip form on the GPU, tools/grad_dump.py;
    // UP: hipcc contracts the transposed-upsample sums differently in this shape - 1e-7 relative).  Measured: pass A alone 44.6 -> 38.4 us on
    // decoder.10's geometry back to back (tools/microbench/elem_passes.hip), 36 us per step of serial kernel time, nothing on the two-stream step -
    // in the step the pass runs at what the memory system gives it behind a data gradient's 300 MB of write-backs
    for (size_t q4 = beg + threadIdx.x; q4 < end; q4 += 2 * WUNET_THREADS) {
        PassALoads<MODE, E0> t0, t1;
        const bool two = !FUSE && q4 + WUNET_THREADS < end;
        load(q4, t0);
        if (!FUSE) load(two ? q4 + WUNET_THREADS : q4, t1);
        compute(t0);
        if (two) compute(t1);
    }
    const bool want_max = A.pmax != nullptr;
    if (E0) {
        // the block's 45 sums: 16-lane rows by DPP, the four rows of a wave by two shuffle steps, the four waves through LDS in wave order (fixed order)
#pragma unroll
        for (int q = 0; q < 45; ++q) {
            float v = q < 15 ? w1[E0 ? q : 0] : q < 30 ? w2[E0 ? q - 15 : 0] : w3[E0 ? q - 30 : 0];
            v = wunet_row16_sum(v);
            v += wunet_shfl_xor(v, 16);
            v += wunet_shfl_xor(v, 32);
            if ((threadIdx.x & 63) == 0) e0x[E0 ? (threadIdx.x >> 6) * 48 + q : 0] = v;
        }
        __syncthreads();
        if (threadIdx.x < 45) {
            const int q = threadIdx.x;
            A.e0rows[((size_t)by * A.C + c) * 48 + q] = (e0x[E0 ? q : 0] + e0x[E0 ? 48 + q : 0]) + (e0x[E0 ? 96 + q : 0] + e0x[E0 ? 144 + q : 0]);
        }
    }
    block_reduce_pass_a(s1, s2, s3, MODE == A_HEAD, mg, mz, red);
    if (FUSE) {
        // bn_finalize_bwd_kernel's arithmetic on the (float-rounded, as if through part[]) sums, then gz_materialize_kernel's
        const double t1 = (double)(float)s1, t2 = (double)(float)s2;
        const double m1 = t1 / A.count, m2 = t2 / A.count;
        const double ar = (double)gam * (double)rstd;
        const float k1 = (float)ar, k2 = (float)(-ar * m2 * (double)rstd), k3 = (float)(ar * m2 * (double)rstd * (double)mu - ar * m1);
        if (threadIdx.x == 0) {
            A.dgamma[c] = (float)t2;
            A.dbeta[c] = (float)t1;
            A.dbias[c] = 0.0f;
            A.k1[c] = k1; A.k2[c] = k2; A.k3[c] = k3;
        }
        if (have) {
            wunet_f4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = keep_l + j < A.Lt ? k1 * keep_g[j] + k2 * keep_z[j] + k3 : 0.0f;
            wunet_st4(A.gpre + keep_i, o);
        }
        return;
    }
    if (threadIdx.x == 0) {
        float* pr = A.part + ((size_t)by * A.C + c) * 2;
        pr[0] = (float)s1;
        pr[1] = (float)s2;
        if (want_max) {
            float* pm = A.pmax + ((size_t)by * A.C + c) * 2;
            pm[0] = mg;
            pm[1] = mz;
        }
        if (MODE == A_HEAD && A.hpart) A.hpart[(size_t)by * A.C + c] = (float)s3;
    }
}

// ---------------------------------------------------------------------------- BN backward finalize
// grid = C.  dgamma = sum g*xhat, dbeta = sum g, and the folded coefficients of
// g_z = gamma*rstd*(g - mean(g) - xhat*mean(g*xhat)) = k1*g + k2*z + k3.
struct BnBwdArgs {
    const float* part; int rows;
    const float* gamma; const float* mean; const float* rstd;
    float* dgamma; float* dbeta;
    float* dbias;        // conv bias feeding training-mode BN: exact zero gradient
    float* k1; float* k2; float* k3;
    int C; double count;
    const float* pmax;   // nullptr or pass A's [rows][C][2] maxima
    float* bound;        // [C]: |k1| max|g| + |a m2 rstd| max|z - mean| + |a m1|  >=  max |g_z| of the channel
    const float* e0rows; // nullptr, or the first layer's [rows][C][48] partial sums (pass_a_kernel<.., E0>): its weight gradient is finished here
    float* dw0;          // [C][15]: dW[c][k] = k1 sum g x + k2 sum z x + k3 sum x
};

static __global__ __launch_bounds__(WUNET_THREADS) void bn_finalize_bwd_kernel(BnBwdArgs A)
{
    __shared__ double red[5 * WUNET_WAVES];
    const int c = blockIdx.x, tid = threadIdx.x;
    // the channel's constants first, the sums' and the maxima's rows in ONE trip, one closing barrier (rounds 1 - 6: sums, two barriers, thread 0's
    // loads of gamma / rstd / mean between its stores, then the maxima, two more barriers, the same loads again - a chain of dependent round
    // trips in a kernel that is nothing else, ten times per step on the backward's chain).  Same sums in the same order: the same bits.
    const float gam = A.gamma[c], rs = A.rstd[c], mu = A.mean[c];
    double s1 = 0.0, s2 = 0.0, s3 = 0.0;
    float mg = 0.0f, mz = 0.0f;
    for (int r = tid; r < A.rows; r += WUNET_THREADS) {
        const float2 pr = *reinterpret_cast<const float2*>(A.part + ((size_t)r * A.C + c) * 2);
        float2 pm = float2{0.0f, 0.0f};
        if (A.pmax) pm = *reinterpret_cast<const float2*>(A.pmax + ((size_t)r * A.C + c) * 2);
        s1 += (double)pr.x;
        s2 += (double)pr.y;
        mg = fmaxf(mg, pm.x);
        mz = fmaxf(mz, pm.y);
    }
    block_reduce_pass_a(s1, s2, s3, false, mg, mz, red);
    if (tid == 0) {
        const double m1 = s1 / A.count, m2 = s2 / A.count;
        const double a = (double)gam * (double)rs;
        A.dgamma[c] = (float)s2;
        A.dbeta[c] = (float)s1;
        A.dbias[c] = 0.0f;
        A.k1[c] = (float)a;
        A.k2[c] = (float)(-a * m2 * (double)rs);
        A.k3[c] = (float)(a * m2 * (double)rs * (double)mu - a * m1);
        if (A.pmax) A.bound[c] = (float)(fabs(a) * (double)mg + fabs(a * m2 * (double)rs) * (double)mz + fabs(a * m1));
    }
    if (A.e0rows) {
        // the first layer's weight gradient from the three sums per tap: the rows in order, in double; the coefficients as g_z would have had them
        // (the float k1, k2, k3)
        __shared__ double e0s[48];
        if (tid < 45) {
            double t = 0.0;
            for (int r0 = 0; r0 < A.rows; r0 += 32) {          // (32 rows' loads in flight: 128 rows are four round trips on the backward's last stretch)
                float v[32];
#pragma unroll
                for (int r = 0; r < 32; ++r) v[r] = A.e0rows[((size_t)(r0 + r < A.rows ? r0 + r : 0) * A.C + (tid < 30 ? c : 0)) * 48 + tid];      // (sum x: taken by channel 0's blocks)
#pragma unroll
                for (int r = 0; r < 32; ++r) if (r0 + r < A.rows) t += (double)v[r];
            }
            e0s[tid] = t;
        }
        __syncthreads();
        if (tid < 15) {
            const double m1 = s1 / A.count, m2 = s2 / A.count;
            const double a = (double)gam * (double)rs;
            const float k1 = (float)a, k2 = (float)(-a * m2 * (double)rs), k3 = (float)(a * m2 * (double)rs * (double)mu - a * m1);
            A.dw0[c * 15 + tid] = (float)((double)k1 * e0s[tid] + (double)k2 * e0s[15 + tid] + (double)k3 * e0s[30 + tid]);      // (e0s[30 ..]: channel 0's rows)
        }
    }
}

// The same finalize for a layer whose sums were taken by its consumers' data-gradient epilogues (conv_h3d_kernel<.., BSUM>): up to two sets
// of per-tile rows [channel][tiles][4] = {sum g, sum g xhat, bound of max |g|, max |z - mean|} - the skip consumer's and the decimating
// consumer's for an encoder layer, one set for a layer that feeds an upsample.  Sums: tiles in order, set 0 then set 1, in double;
// max |g| <= the sum of the two sets' bounds (g is the sum of the two consumers' gradients).
struct BnBwdTilesArgs {
    const float* part0; int tiles0;      // rows of THIS layer's channel c start at part0 + c * tiles0 * 4 (the caller applied the row offset)
    const float* part1; int tiles1;      // nullptr: one set
    const float* gamma; const float* mean; const float* rstd;
    float* dgamma; float* dbeta; float* dbias; float* k1; float* k2; float* k3; float* bound;
    int C; double count;
};
static __global__ __launch_bounds__(WUNET_THREADS) void bn_finalize_bwd_tiles_kernel(BnBwdTilesArgs A)
{
    __shared__ double red[2 * WUNET_THREADS];
    const int c = blockIdx.x, tid = threadIdx.x;
    double s1 = 0.0, s2 = 0.0;
    float mg0 = 0.0f, mg1 = 0.0f, mz = 0.0f;
    {
        const float* p = A.part0 + (size_t)c * A.tiles0 * 4;
        int r = tid;
        for (; r + 3 * WUNET_THREADS < A.tiles0; r += 4 * WUNET_THREADS) {      // (four loads in flight, same order of additions)
            const wunet_f4 v0 = wunet_ld4(p + 4 * (size_t)r), v1 = wunet_ld4(p + 4 * (size_t)(r + WUNET_THREADS));
            const wunet_f4 v2 = wunet_ld4(p + 4 * (size_t)(r + 2 * WUNET_THREADS)), v3 = wunet_ld4(p + 4 * (size_t)(r + 3 * WUNET_THREADS));
            s1 += (double)v0[0]; s2 += (double)v0[1]; s1 += (double)v1[0]; s2 += (double)v1[1];
            s1 += (double)v2[0]; s2 += (double)v2[1]; s1 += (double)v3[0]; s2 += (double)v3[1];
            mg0 = fmaxf(fmaxf(mg0, v0[2]), fmaxf(fmaxf(v1[2], v2[2]), v3[2]));
            mz = fmaxf(fmaxf(mz, v0[3]), fmaxf(fmaxf(v1[3], v2[3]), v3[3]));
        }
        for (; r < A.tiles0; r += WUNET_THREADS) {
            const wunet_f4 v = wunet_ld4(p + 4 * (size_t)r);
            s1 += (double)v[0]; s2 += (double)v[1]; mg0 = fmaxf(mg0, v[2]); mz = fmaxf(mz, v[3]);
        }
    }
    if (A.part1) {
        const float* p = A.part1 + (size_t)c * A.tiles1 * 4;
        for (int r = tid; r < A.tiles1; r += WUNET_THREADS) {
            const wunet_f4 v = wunet_ld4(p + 4 * (size_t)r);
            s1 += (double)v[0]; s2 += (double)v[1]; mg1 = fmaxf(mg1, v[2]); mz = fmaxf(mz, v[3]);
        }
    }
    block_sum2(s1, s2, red);
    __syncthreads();
    block_max2(mg0, mg1, red);
    __syncthreads();
    float dummy = 0.0f;
    block_max2(mz, dummy, red);
    if (tid == 0) {
        A.dgamma[c] = (float)s2;
        A.dbeta[c] = (float)s1;
        A.dbias[c] = 0.0f;
        const double m1 = s1 / A.count, m2 = s2 / A.count;
        const double a = (double)A.gamma[c] * (double)A.rstd[c];
        A.k1[c] = (float)a;
        A.k2[c] = (float)(-a * m2 * (double)A.rstd[c]);
        A.k3[c] = (float)(a * m2 * (double)A.rstd[c] * (double)A.mean[c] - a * m1);
        A.bound[c] = (float)(fabs(a) * ((double)mg0 + (double)mg1) + fabs(a * m2 * (double)A.rstd[c]) * (double)mz + fabs(a * m1));
    }
}

// ---------------------------------------------------------------------------- split-K reduce of dW
// dw[i] = sum over `splits` partial tensors, in a fixed order (deterministic).  dW is small (10^4..10^6 values) and
// the splits are many (up to one per CU), so the work is spread over outputs AND splits: a block owns 16 float4 groups
// of outputs; its 16 "split lanes" each sum the splits s = lane, lane+16, ... in fp32 groups of 8 and fp64 across
// groups, and the lanes are combined in lane order through LDS in fp64.
static __global__ __launch_bounds__(WUNET_THREADS) void wgrad_reduce_kernel(const float* part, int splits, size_t n, float* dw)
{
    __shared__ double red[16][16][4];
    const int og = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const size_t n4 = n >> 2;
    for (size_t base = (size_t)blockIdx.x * 16; base < n4; base += (size_t)gridDim.x * 16) {
        const size_t i = base + og;
        double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
        if (i < n4) {
            int r = sl;
            for (; r + 7 * 16 < splits; r += 8 * 16) {
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const wunet_f4 v = wunet_ld4(part + (size_t)(r + k * 16) * n + 4 * i);
                    a0 += v[0]; a1 += v[1]; a2 += v[2]; a3 += v[3];
                }
                t0 += (double)a0; t1 += (double)a1; t2 += (double)a2; t3 += (double)a3;
            }
            for (; r < splits; r += 16) {
                const wunet_f4 v = wunet_ld4(part + (size_t)r * n + 4 * i);
                t0 += (double)v[0]; t1 += (double)v[1]; t2 += (double)v[2]; t3 += (double)v[3];
            }
        }
        __syncthreads();
        red[sl][og][0] = t0; red[sl][og][1] = t1; red[sl][og][2] = t2; red[sl][og][3] = t3;
        __syncthreads();
        if (sl == 0 && i < n4) {
            double u0 = 0.0, u1 = 0.0, u2 = 0.0, u3 = 0.0;
#pragma unroll
            for (int k = 0; k < 16; ++k) { u0 += red[k][og][0]; u1 += red[k][og][1]; u2 += red[k][og][2]; u3 += red[k][og][3]; }
            wunet_f4 o;
            o[0] = (float)u0; o[1] = (float)u1; o[2] = (float)u2; o[3] = (float)u3;
            wunet_st4(dw + 4 * i, o);
        }
    }
    for (size_t i = (n4 << 2) + (size_t)blockIdx.x * WUNET_THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * WUNET_THREADS) {
        double tot = 0.0;
        for (int r = 0; r < splits; ++r) tot += (double)part[(size_t)r * n + i];
        dw[i] = (float)tot;
    }
}

// ---------------------------------------------------------------------------- first encoder conv (Cin = 1)
// encoder[0] is 2*K*Cout flop per output sample against 4*Cout bytes written: HBM-bound by a wide margin, and a
// single input channel wastes 31/32 of a matrix-core K block.  Plain fp32 FMAs: one thread = 4 consecutive samples
// x all Cout channels (the K+3 inputs sit in registers, the weights are wave-uniform scalar loads), one 16-byte
// store per channel; BN statistics of the bias-free conv per (wave, channel) like the MFMA kernels
// (stats [Cout][gridDim.x*4][2], one row per wave = 256 samples).  L >= 256: a wave lies inside one batch item.
// Eval mode: ev_a / ev_s (this layer's BatchNorm scale / shift, known before the conv) and xrows (ONE float, the layer's xb slot, cleared by
// h3_scales_kernel: every block folds its maximum into it with one atomic max - order independent, so deterministic): the block's
// max |a z + s|, the activation bound the consumer's split-operand scale derives from.
template <int K>
__global__ __launch_bounds__(WUNET_THREADS) void conv_first_kernel(const float* x, const float* w, const float* bias, float* out,
                                                                    float* stats, int B, int Cout, int L, int logL,
                                                                    const float* ev_a, const float* ev_s, float* xrows, int Lt)
{
    constexpr int PAD = K / 2, NX = ((8 + 4 + PAD + 3) / 4) * 4;          // aligned window [l0 - 8, l0 + NX - 8)
    static_assert(PAD <= 8 && NX >= 8 + 4 + PAD, "window");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t p = ((size_t)blockIdx.x * WUNET_THREADS + threadIdx.x) * 4;
    const int b = (int)(p >> logL), l0 = (int)(p & (size_t)(L - 1));
    const bool live = b < B;
    const bool counted = live && l0 < Lt;       // row padding (Lt = the samples that exist, a multiple of 4 at this level: all four or none
                                                // of the thread's) is computed and stored - finite values - but not counted
    float xv[NX];
#pragma unroll
    for (int v = 0; v < NX / 4; ++v) {
        const int l = l0 - 8 + 4 * v;
        const bool ok = live && l >= 0 && l < L;
        const wunet_f4 t = wunet_sel4(ok, wunet_ld4(x + (ok ? (size_t)b * L + l : 0)));
        xv[4 * v] = t[0]; xv[4 * v + 1] = t[1]; xv[4 * v + 2] = t[2]; xv[4 * v + 3] = t[3];
    }
    const size_t rows = (size_t)gridDim.x * WUNET_WAVES, row = (size_t)blockIdx.x * WUNET_WAVES + wave;
    float amax = 0.0f;
#pragma unroll 2
    for (int co = 0; co < Cout; ++co) {
        const float* wr = w + (size_t)co * K;
        wunet_f4 acc = wunet_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < K; ++t) {
            const float wv = wr[t];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = fmaf(wv, xv[8 - PAD + t + j], acc[j]);
        }
        float s1 = counted ? (acc[0] + acc[1]) + (acc[2] + acc[3]) : 0.0f;
        float s2 = counted ? (acc[0] * acc[0] + acc[1] * acc[1]) + (acc[2] * acc[2] + acc[3] * acc[3]) : 0.0f;
        if (live) {
            const float bv = bias ? bias[co] : 0.0f;
            wunet_f4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = acc[j] + bv;
            wunet_st4(out + ((size_t)b * Cout + co) * L + l0, o);
            if (xrows) {
                const float ea = ev_a[co], es = ev_s[co];
#pragma unroll
                for (int j = 0; j < 4; ++j) amax = fmaxf(amax, fabsf(ea * o[j] + es));
            }
        }
        if (stats) {
            // the wave's sums: the 16-lane steps by DPP moves (wunet_row16_sum: the butterfly's first four steps, bit for bit, without the
            // LDS crossbar), then the two cross-row steps - 4 instead of 12 ds_bpermute per channel
            s1 = wunet_row16_sum(s1);
            s2 = wunet_row16_sum(s2);
#pragma unroll
            for (int m = 16; m < 64; m <<= 1) {
                s1 += wunet_shfl_xor(s1, m);
                s2 += wunet_shfl_xor(s2, m);
            }
            if (lane == 0) {
                float* stp = stats + ((size_t)co * rows + row) * 2;
                stp[0] = s1;
                stp[1] = s2;
            }
        }
    }
    if (xrows) {
        __shared__ float xm[WUNET_WAVES];
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) amax = fmaxf(amax, wunet_shfl_xor(amax, m));
        if (lane == 0) xm[wave] = amax;
        __syncthreads();
        if (threadIdx.x == 0) wunet_atomic_absmax(xrows, fmaxf(fmaxf(xm[0], xm[1]), fmaxf(xm[2], xm[3])));
    }
}

// ---------------------------------------------------------------------------- conv input materialisation
// One fused pass per layer builds the tensor its conv consumes (and its weight gradient re-reads):
//   encoder i+1 / middle : x[b,c,l] = lrelu(a[c]*z[b,c,2l] + s[c])                (unet_basic.py:13,86)
//   decoder j            : c <  C0: l0*act(zp[b,c,i0]) + l1*act(zp[b,c,i1])        (unet_basic.py:93, ATen fp32 coordinates)
//                          c >= C0: lrelu(a1*zs[b,c-C0,l] + s1)                    (unet_basic.py:95 cat([up, skip]))
// Each thread produces one aligned float4 of x.
struct PrepArgs {
    const float* z0; const float* a0; const float* s0;   // producer (previous level) raw conv output + BN scale/shift
    const float* z1; const float* a1; const float* s1;   // skip producer (decoder only)
    float* x;                                            // [B][C0+C1][L]
    int B, C0, C1, L, logL;
    float up_scale;                                      // (float)(Lt/2-1)/(Lt-1)
    int Lt;                                              // samples of a row that exist (<= L, the power-of-two row stride): lengths m*2^n
                                                         // are carried in rows padded to the next power of two, the padding holds zeros
};

static __global__ __launch_bounds__(WUNET_THREADS) void prep_decim_kernel(PrepArgs A)
{
    const int l4n = A.L >> 2;
    const size_t total = (size_t)A.B * A.C0 * l4n;
    for (size_t i = (size_t)blockIdx.x * WUNET_THREADS + threadIdx.x; i < total; i += (size_t)gridDim.x * WUNET_THREADS) {
        const size_t row = i >> (A.logL - 2);
        const int l4 = (int)(i & (size_t)(l4n - 1));
        const int c = (int)(row % (size_t)A.C0);
        const float a = A.a0[c], s = A.s0[c];
        const float4* src = reinterpret_cast<const float4*>(A.z0 + row * (size_t)(2 * A.L) + 8 * l4);
        const float4 u = src[0], v = src[1];
        float4 o;
        o.x = wunet_lrelu(a * u.x + s); o.y = wunet_lrelu(a * u.z + s);
        o.z = wunet_lrelu(a * v.x + s); o.w = wunet_lrelu(a * v.z + s);
        if (4 * l4 + 3 >= A.Lt) {                                // row padding: zeros (the conv's zero padding at the true end of the row)
            if (4 * l4 >= A.Lt) o.x = 0.0f;
            if (4 * l4 + 1 >= A.Lt) o.y = 0.0f;
            if (4 * l4 + 2 >= A.Lt) o.z = 0.0f;
            o.w = 0.0f;
        }
        reinterpret_cast<float4*>(A.x)[i] = o;
    }
}

// channels [cbeg, cend) of the decoder input: the skip half only depends on an encoder level, so the host
// runs it on a side stream during the encoder phase and only the upsampled half sits on the critical path
static __global__ __launch_bounds__(WUNET_THREADS) void prep_upcat_kernel(PrepArgs A, int cbeg, int cend)
{
    const int l4n = A.L >> 2, C = A.C0 + A.C1, Lh = A.L >> 1, nc = cend - cbeg;
    const size_t total = (size_t)A.B * nc * l4n;
    for (size_t i = (size_t)blockIdx.x * WUNET_THREADS + threadIdx.x; i < total; i += (size_t)gridDim.x * WUNET_THREADS) {
        const size_t row = i >> (A.logL - 2);
        const int l4 = (int)(i & (size_t)(l4n - 1));
        const int b = (int)(row / (size_t)nc), c = cbeg + (int)(row - (size_t)b * nc);
        float4 o;
        if (c < A.C0) {
            const float a = A.a0[c], s = A.s0[c];
            const float* zr = A.z0 + ((size_t)b * A.C0 + c) * Lh;
            float r[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int i0, i1; float l0, l1;
                wunet_up_coord(4 * l4 + j, A.Lt >> 1, A.up_scale, i0, i1, l0, l1);
                r[j] = l0 * wunet_lrelu(a * zr[i0] + s) + l1 * wunet_lrelu(a * zr[i1] + s);
            }
            o.x = r[0]; o.y = r[1]; o.z = r[2]; o.w = r[3];
        } else {
            const int cs = c - A.C0;
            const float a = A.a1[cs], s = A.s1[cs];
            const float4 v = reinterpret_cast<const float4*>(A.z1 + ((size_t)b * A.C1 + cs) * A.L)[l4];
            o.x = wunet_lrelu(a * v.x + s); o.y = wunet_lrelu(a * v.y + s);
            o.z = wunet_lrelu(a * v.z + s); o.w = wunet_lrelu(a * v.w + s);
        }
        if (4 * l4 + 3 >= A.Lt) {                                // row padding: zeros
            if (4 * l4 >= A.Lt) o.x = 0.0f;
            if (4 * l4 + 1 >= A.Lt) o.y = 0.0f;
            if (4 * l4 + 2 >= A.Lt) o.z = 0.0f;
            o.w = 0.0f;
        }
        reinterpret_cast<float4*>(A.x + ((size_t)b * C + c) * A.L)[l4] = o;
    }
}

// g_z = k1*g + k2*z + k3 (BatchNorm backward folded to three per-channel coefficients), materialised once
// per layer for its data-gradient and weight-gradient GEMMs
static __global__ __launch_bounds__(WUNET_THREADS) void gz_materialize_kernel(const float* g, const float* z, const float* k1, const float* k2,
                                                                        const float* k3, int C, int logL, size_t n4, float* gz, int Lt)
{
    for (size_t i = (size_t)blockIdx.x * WUNET_THREADS + threadIdx.x; i < n4; i += (size_t)gridDim.x * WUNET_THREADS) {
        const int c = (int)((i >> (logL - 2)) % (size_t)C);
        const float a = k1[c], b = k2[c], d = k3[c];
        const float4 gv = reinterpret_cast<const float4*>(g)[i], zv = reinterpret_cast<const float4*>(z)[i];
        float4 o;
        o.x = a * gv.x + b * zv.x + d; o.y = a * gv.y + b * zv.y + d;
        o.z = a * gv.z + b * zv.z + d; o.w = a * gv.w + b * zv.w + d;
        const int l = (int)((i << 2) & (((size_t)1 << logL) - 1));
        if (l + 3 >= Lt) {                                       // row padding: no gradient
            if (l >= Lt) o.x = 0.0f;
            if (l + 1 >= Lt) o.y = 0.0f;
            if (l + 2 >= Lt) o.z = 0.0f;
            o.w = 0.0f;
        }
        reinterpret_cast<float4*>(gz)[i] = o;
    }
}

// ---------------------------------------------------------------------------- data input: aligned window crops (f4)
// out[b][j] = flat[starts[b] + j] for the mixture and the clean array alike (the reference's sample_fixed_length_data_aligned,
// util/utils.py:101-113, for a whole batch).  A window start is any sample, so the loads are dwords (64 consecutive per wave
// instruction: coalesced, never 16-byte aligned in general); a thread gathers 4 consecutive samples and stores them as one
// aligned float4 per array when the row length allows it.  HBM-bound: 16 B read + 16 B written per thread.
static __global__ __launch_bounds__(WUNET_THREADS) void crop_windows_kernel(const float* mix_flat, const float* clean_flat, const long long* starts,
                                                                            long long total, int length, float* mix, float* clean)
{
    const int b = blockIdx.y;
    long long s0 = starts[b];
    if (s0 < 0) s0 = 0;                                   // (a corrupt start must not read outside the shard)
    if (s0 > total - length) s0 = total - length;
    const int j0 = (blockIdx.x * WUNET_THREADS + threadIdx.x) * 4;
    if (j0 >= length) return;
    const float* ms = mix_flat + s0 + j0;
    const float* cs = clean_flat + s0 + j0;
    float* md = mix + (size_t)b * length + j0;
    float* cd = clean + (size_t)b * length + j0;
    if (j0 + 4 <= length && (length & 3) == 0) {
        const wunet_f4 m = {ms[0], ms[1], ms[2], ms[3]};
        const wunet_f4 c = {cs[0], cs[1], cs[2], cs[3]};
        wunet_st4(md, m);
        wunet_st4(cd, c);
    } else {
        for (int k = 0; k < 4 && j0 + k < length; ++k) { md[k] = ms[k]; cd[k] = cs[k]; }
    }
}

static __global__ __launch_bounds__(WUNET_THREADS) void fill_kernel(float* p, size_t n, float v)
{
    for (size_t i = (size_t)blockIdx.x * WUNET_THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * WUNET_THREADS) p[i] = v;
}

// ---------------------------------------------------------------------------- losses
// loss(clean, enhanced) of trainer/trainer.py:36; kind 0 = MSELoss, 1 = L1Loss (model/loss.py:3-7),
// 2 = SmoothL1Loss(beta=1).  mean reduction.
__device__ __forceinline__ float loss_term(int kind, float d)
{
    if (kind == 0) return d * d;
    const float ad = fabsf(d);
    if (kind == 1) return ad;
    return ad < 1.0f ? 0.5f * d * d : ad - 0.5f;
}
__device__ __forceinline__ float loss_dterm(int kind, float d)   // derivative w.r.t. d = enhanced - clean
{
    if (kind == 0) return 2.0f * d;
    const float sg = (d > 0.0f ? 1.0f : 0.0f) - (d < 0.0f ? 1.0f : 0.0f);
    if (kind == 1) return sg;
    return fabsf(d) < 1.0f ? d : sg;
}

static __global__ __launch_bounds__(WUNET_THREADS) void loss_partial_kernel(int kind, const float* clean, const float* enh, size_t n, double* part)
{
    __shared__ double red[2 * WUNET_THREADS];
    double s = 0.0, dummy = 0.0;
    // four trips' loads in flight (a thread's 16 trips at batch 64 x 16384 were 16 dependent memory round trips: 10.4 us for 8 MB); the terms are
    // added in the one-trip loop's order - the same bits
    const size_t stride = (size_t)gridDim.x * WUNET_THREADS;
    size_t i = (size_t)blockIdx.x * WUNET_THREADS + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        float e[4], c[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { e[u] = enh[i + u * stride]; c[u] = clean[i + u * stride]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) s += (double)loss_term(kind, e[u] - c[u]);
    }
    for (; i < n; i += stride) s += (double)loss_term(kind, enh[i] - clean[i]);
    block_sum2(s, dummy, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

static __global__ __launch_bounds__(WUNET_THREADS) void loss_final_kernel(const double* part, int rows, size_t n, float* loss)
{
    __shared__ double red[2 * WUNET_THREADS];
    double s = 0.0, dummy = 0.0;
    for (int r = threadIdx.x; r < rows; r += WUNET_THREADS) s += part[r];
    block_sum2(s, dummy, red);
    if (threadIdx.x == 0) *loss = (float)(s / (double)n);
}

// grad_enh[i] = gscale[0] * dterm(enh - clean) / n
static __global__ __launch_bounds__(WUNET_THREADS) void loss_bwd_kernel(int kind, const float* clean, const float* enh, const float* gscale, size_t n, float* genh)
{
    const float sc = gscale[0] / (float)n;
    const size_t stride = (size_t)gridDim.x * WUNET_THREADS;
    size_t i = (size_t)blockIdx.x * WUNET_THREADS + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {             // (four trips' loads ahead of the first store)
        float e[4], c[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { e[u] = enh[i + u * stride]; c[u] = clean[i + u * stride]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) genh[i + u * stride] = sc * loss_dterm(kind, e[u] - c[u]);
    }
    for (; i < n; i += stride) genh[i] = sc * loss_dterm(kind, enh[i] - clean[i]);
}

// ---------------------------------------------------------------------------- fused Adam
// torch.optim.Adam(lr, betas, eps=1e-8, weight_decay=0, amsgrad=False) as the reference builds it
// (train.py:31-35), for up to WUNET_ADAM_MAX tensors per launch: blockIdx.y = tensor, blockIdx.x strides inside it.
// Same operation order as torch's single-tensor path: m.lerp_(g, 1-b1); v = v*b2 + (1-b2)*g*g;
// denom = sqrt(v)/sqrt(1-b2^t) + eps; p -= (lr/(1-b1^t)) * m/denom.
#define WUNET_ADAM_MAX 64
struct AdamTable {
    float* p[WUNET_ADAM_MAX];
    const float* g[WUNET_ADAM_MAX];
    float* m[WUNET_ADAM_MAX];
    float* v[WUNET_ADAM_MAX];
    unsigned n[WUNET_ADAM_MAX];
};

// gscale: the gradient is multiplied by it first (1/world_size of the data-parallel average folded into the step: the same
// rounding as a separate g.mul_(1/world) pass).  hyper != nullptr: {step_size, bc2_sqrt} come from device memory
// (adam_hyper_kernel: the step counter lives on the device, so a captured graph replays the right bias corrections).
static __global__ __launch_bounds__(WUNET_THREADS) void adam_kernel(AdamTable T, float one_minus_b1, float b2, float one_minus_b2,
                                                              float bc2_sqrt, float eps, float step_size, float gscale,
                                                              const float* hyper)
{
    if (hyper) { step_size = hyper[0]; bc2_sqrt = hyper[1]; }
    const int t = blockIdx.y;
    float* p = T.p[t];
    const float* g = T.g[t];
    float* m = T.m[t];
    float* v = T.v[t];
    const unsigned n = T.n[t];
    for (unsigned i = blockIdx.x * WUNET_THREADS + threadIdx.x; i < n; i += gridDim.x * WUNET_THREADS) {
        const float gi = gscale == 1.0f ? g[i] : g[i] * gscale;
        float mi = m[i], vi = v[i];
        mi = mi + one_minus_b1 * (gi - mi);
        vi = vi * b2 + one_minus_b2 * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = p[i] - step_size * (mi / denom);
    }
}

// step counter on the device: *step += 1, then the bias-corrected step size lr / (1 - b1^t) and sqrt(1 - b2^t) in double like
// torch's host arithmetic (one thread)
static __global__ void adam_hyper_kernel(long long* step, double lr, double beta1, double beta2, float* hyper)
{
    const long long t = *step + 1;
    *step = t;
    const double bc1 = 1.0 - pow(beta1, (double)t), bc2 = 1.0 - pow(beta2, (double)t);
    hyper[0] = (float)(lr / bc1);
    hyper[1] = (float)sqrt(bc2);
}
