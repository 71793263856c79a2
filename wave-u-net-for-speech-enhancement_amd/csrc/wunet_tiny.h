// Scalar kernels for the levels shorter than 4 samples (L = 1 or 2: the bottom of a net whose depth approaches
// log2(T), e.g. 16 levels at 65536 samples - reference model/unet_basic.py accepts any n_layers <= log2(T)).
// The vectorised kernels assume 16-byte rows; these levels hold a few hundred KB and a few MFLOP, so they run as
// plain one-thread-per-output VALU kernels with the same arithmetic.
#pragma once
#include "wunet_elementwise.h"

// conv input (see prep_decim_kernel / prep_upcat_kernel), one thread per element
static __global__ __launch_bounds__(WUNET_THREADS) void prep_scalar_kernel(PrepArgs A, int upcat)
{
    const int C = A.C0 + A.C1;
    const size_t total = (size_t)A.B * C * A.L;
    for (size_t i = (size_t)blockIdx.x * WUNET_THREADS + threadIdx.x; i < total; i += (size_t)gridDim.x * WUNET_THREADS) {
        const int l = (int)(i & (size_t)(A.L - 1));
        const size_t row = i >> A.logL;
        const int b = (int)(row / (size_t)C), c = (int)(row - (size_t)b * C);
        float v;
        if (!upcat) {
            v = wunet_lrelu(A.a0[c] * A.z0[((size_t)b * A.C0 + c) * (2 * A.L) + 2 * l] + A.s0[c]);
        } else if (c < A.C0) {
            const int Lh = A.L >> 1;
            const float* zr = A.z0 + ((size_t)b * A.C0 + c) * Lh;
            int i0, i1; float l0, l1;
            wunet_up_coord(l, Lh, A.up_scale, i0, i1, l0, l1);
            v = l0 * wunet_lrelu(A.a0[c] * zr[i0] + A.s0[c]) + l1 * wunet_lrelu(A.a0[c] * zr[i1] + A.s0[c]);
        } else {
            const int cs = c - A.C0;
            v = wunet_lrelu(A.a1[cs] * A.z1[((size_t)b * A.C1 + cs) * A.L + l] + A.s1[cs]);
        }
        A.x[i] = v;
    }
}

// gradient assembly (see pass_a_kernel), one sample per thread
template <int MODE>
__global__ __launch_bounds__(WUNET_THREADS) void pass_a_scalar_kernel(PassAArgs A)
{
    __shared__ double red[2 * WUNET_THREADS];
    const int c = blockIdx.x;
    const size_t total = (size_t)A.B * A.L;
    const size_t per = (total + gridDim.y - 1) / gridDim.y;
    const size_t beg = (size_t)blockIdx.y * per, end = beg + per < total ? beg + per : total;
    const float a = A.a[c], s = A.s[c], mu = A.mean[c], rstd = A.rstd[c];
    double s1 = 0.0, s2 = 0.0;
    for (size_t p = beg + threadIdx.x; p < end; p += WUNET_THREADS) {
        const int b = (int)(p >> A.logL), l = (int)(p & (size_t)(A.L - 1));
        const size_t zi = ((size_t)b * A.C + c) * A.L + l;
        const float z = A.z[zi];
        float g;
        if (MODE == A_HEAD) {
            g = A.g1[c] * A.g0[(size_t)b * A.L + l];
        } else if (MODE == A_ENC) {
            g = A.g0[((size_t)b * A.Cg0 + A.coff + c) * A.L + l];
            if ((l & 1) == 0) g += A.g1[((size_t)b * A.C + c) * (A.L >> 1) + (l >> 1)];
        } else {
            const int Lo = 2 * A.L;
            const float* row = A.g0 + ((size_t)b * A.Cg0 + c) * Lo;
            g = 0.0f;
            for (int j = 2 * l - 2; j <= 2 * l + 2; ++j) {
                if (j >= 0 && j < Lo) {
                    int i0, i1; float l0, l1;
                    wunet_up_coord(j, A.L, A.up_scale, i0, i1, l0, l1);
                    g += ((i0 == l ? l0 : 0.0f) + (i1 == l ? l1 : 0.0f)) * row[j];
                }
            }
        }
        if (!(a * z + s > 0.0f)) g *= WUNET_SLOPE;
        A.gpre[zi] = g;
        s1 += (double)g;
        s2 += (double)(g * ((z - mu) * rstd));
    }
    block_sum2(s1, s2, red);
    if (threadIdx.x == 0) {
        float* pr = A.part + ((size_t)blockIdx.y * A.C + c) * 2;
        pr[0] = (float)s1;
        pr[1] = (float)s2;
    }
}

static __global__ __launch_bounds__(WUNET_THREADS) void gz_scalar_kernel(const float* g, const float* z, const float* k1, const float* k2,
                                                                   const float* k3, int C, int logL, size_t n, float* gz)
{
    for (size_t i = (size_t)blockIdx.x * WUNET_THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * WUNET_THREADS) {
        const int c = (int)((i >> logL) % (size_t)C);
        gz[i] = k1[c] * g[i] + k2[c] * z[i] + k3[c];
    }
}

// out[b,r,l] = sum_{c,k} W(r,c,k) * x[b,c,l+k-pad];  forward: W = w[r][c][k] (w = [R][C][K]);
// data gradient (transposed): W = w[c][r][K-1-k] (w = [C][R][K]).  No bias.
static __global__ __launch_bounds__(WUNET_THREADS) void tiny_conv_kernel(const float* x, const float* w, float* out, int B, int C, int R,
                                                                   int L, int logL, int K, int transposed)
{
    const int pad = K / 2;
    const size_t total = (size_t)B * R * L;
    for (size_t i = (size_t)blockIdx.x * WUNET_THREADS + threadIdx.x; i < total; i += (size_t)gridDim.x * WUNET_THREADS) {
        const int l = (int)(i & (size_t)(L - 1));
        const size_t row = i >> logL;
        const int b = (int)(row / (size_t)R), r = (int)(row - (size_t)b * R);
        float acc = 0.0f;
        for (int c = 0; c < C; ++c) {
            const float* xr = x + ((size_t)b * C + c) * L;
            for (int k = 0; k < K; ++k) {
                const int p = l + k - pad;
                if (p >= 0 && p < L) {
                    const float wv = transposed ? w[((size_t)c * R + r) * K + (K - 1 - k)] : w[((size_t)r * C + c) * K + k];
                    acc = fmaf(wv, xr[p], acc);
                }
            }
        }
        out[i] = acc;
    }
}

// dw[co][ci][k] = sum_{b,l} g[b,co,l] * x[b,ci,l+k-pad]
static __global__ __launch_bounds__(WUNET_THREADS) void tiny_wgrad_kernel(const float* g, const float* x, float* dw, int B, int Cin, int Cout,
                                                                    int L, int K)
{
    const int pad = K / 2;
    const size_t total = (size_t)Cout * Cin * K;
    for (size_t i = (size_t)blockIdx.x * WUNET_THREADS + threadIdx.x; i < total; i += (size_t)gridDim.x * WUNET_THREADS) {
        const int k = (int)(i % (size_t)K);
        const size_t r = i / (size_t)K;
        const int ci = (int)(r % (size_t)Cin), co = (int)(r / (size_t)Cin);
        float acc = 0.0f;
        for (int b = 0; b < B; ++b) {
            const float* gr = g + ((size_t)b * Cout + co) * L;
            const float* xr = x + ((size_t)b * Cin + ci) * L;
            for (int l = 0; l < L; ++l) {
                const int p = l + k - pad;
                if (p >= 0 && p < L) acc = fmaf(gr[l], xr[p], acc);
            }
        }
        dw[i] = acc;
    }
}
