// MFMA kernels of the Wave-U-Net hot path for gfx950 (CDNA4).
//
//   conv_mfma_kernel  : z = W (*) x  as an implicit GEMM  M=Cout, N=B*L positions, K=Cin*taps
//                       (forward conv of every layer, and - with flipped/transposed packed weights
//                       and the GZ loader - the data gradient of every layer).
//   wgrad_mfma_kernel : dW = g_z (*) x  as a GEMM  M=Cout, N=(ci,tap), K=B*L positions.
//
// Both read a *virtual* input: BatchNorm scale/shift + LeakyReLU + decimation ([:, :, ::2]) or
// linear x2 upsample + skip concat are applied while staging the tile into LDS, so none of those
// tensors is ever materialised (reference model/unet_basic.py:82-96 materialises all of them).
// The MFMA is v_mfma_f32_16x16x4_f32: exact fp32 (an fmaf chain), 157 TF peak = fp32 vector peak.
//
// Layouts (all fp32, reference layout (batch, channel, sample), sample contiguous):
//   packed weights  [m-tile][ci (padded to KC)][tap][16 co]   -> a K-chunk of one m-tile is one
//                   contiguous run, A fragments are bank-conflict free (tap stride 16 dwords with
//                   TAPS odd => the two k-quarters of a 32-lane group land on disjoint bank halves).
//   LDS x tile      [ci][rowp], rowp == 16 (mod 32)           -> B fragments conflict free.
#pragma once
#include "wunet_dev.h"

enum { SRC_RAW = 0, SRC_DECIM = 1, SRC_UPCAT = 2, SRC_GZ = 3 };

// Virtual input x[b, c, l], c < C, l < L.
//  RAW   : p0[b, c, l]                                             (network input, encoder[0])
//  DECIM : lrelu(a0[c] * p0[b, c, 2l] + s0[c])                     (unet_basic.py:86 fused, p0 = raw conv out of previous level)
//  UPCAT : c <  C0: l0*act0(p0[b,c,i0]) + l1*act0(p0[b,c,i1])      (unet_basic.py:93 F.interpolate x2, align_corners)
//          c >= C0: lrelu(a1[c-C0] * p1[b, c-C0, l] + s1[c-C0])    (unet_basic.py:95 cat([up, skip]))
//  GZ    : a0[c]*p0[b,c,l] + s0[c]*p1[b,c,l] + a1[c]               (BatchNorm backward folded to k1*g + k2*z + k3)
struct SrcDesc {
    const float* p0;
    const float* a0;
    const float* s0;
    const float* p1;
    const float* a1;
    const float* s1;
    int C0;        // UPCAT: channels coming from the upsampled branch; otherwise == C
    int C;         // virtual channel count
    int L;         // virtual length (power of two)
    int Lsrc0;     // row length of p0 (DECIM: 2L, UPCAT: L/2, else L)
    int logL;      // log2(L)
    float up_scale;  // UPCAT: (float)(Lsrc0-1)/(L-1)
};

// Per-thread description of one staged LDS column: where in global memory it comes from.
struct ColRef {
    unsigned off0, off1;   // offsets (floats) into p0 / p1 rows for channel 0 of batch item
    float l0, l1;          // UPCAT interpolation weights
    unsigned offu1;        // UPCAT: second tap offset
    bool valid;
};

template <int MODE>
__device__ __forceinline__ void col_prepare(const SrcDesc& d, int b, int l, bool inb, ColRef& r)
{
    r.valid = inb && l >= 0 && l < d.L;
    const int lc = r.valid ? l : 0;
    const unsigned bb = r.valid ? (unsigned)b : 0u;
    r.off0 = r.off1 = r.offu1 = 0;
    r.l0 = r.l1 = 0.0f;
    if (MODE == SRC_RAW || MODE == SRC_GZ) {
        r.off0 = bb * (unsigned)d.C * (unsigned)d.L + (unsigned)lc;
        r.off1 = r.off0;
    } else if (MODE == SRC_DECIM) {
        r.off0 = bb * (unsigned)d.C * (unsigned)d.Lsrc0 + 2u * (unsigned)lc;
    } else {
        int i0, i1;
        wunet_up_coord(lc, d.Lsrc0, d.up_scale, i0, i1, r.l0, r.l1);
        r.off0 = bb * (unsigned)d.C0 * (unsigned)d.Lsrc0 + (unsigned)i0;
        r.offu1 = bb * (unsigned)d.C0 * (unsigned)d.Lsrc0 + (unsigned)i1;
        r.off1 = bb * (unsigned)(d.C - d.C0) * (unsigned)d.L + (unsigned)lc;
    }
}

// value of virtual channel c at a prepared column (c is wave-uniform)
template <int MODE>
__device__ __forceinline__ float col_load(const SrcDesc& d, const ColRef& r, int c)
{
    if (!r.valid || c >= d.C) return 0.0f;
    if (MODE == SRC_RAW) {
        return d.p0[r.off0 + (unsigned)c * (unsigned)d.L];
    } else if (MODE == SRC_DECIM) {
        const float v = d.p0[r.off0 + (unsigned)c * (unsigned)d.Lsrc0];
        return wunet_lrelu(d.a0[c] * v + d.s0[c]);
    } else if (MODE == SRC_GZ) {
        const unsigned o = r.off0 + (unsigned)c * (unsigned)d.L;
        return d.a0[c] * d.p0[o] + d.s0[c] * d.p1[o] + d.a1[c];
    } else {
        if (c < d.C0) {
            const float a = d.a0[c], s = d.s0[c];
            const unsigned base = (unsigned)c * (unsigned)d.Lsrc0;
            const float v0 = wunet_lrelu(a * d.p0[r.off0 + base] + s);
            const float v1 = wunet_lrelu(a * d.p0[r.offu1 + base] + s);
            return r.l0 * v0 + r.l1 * v1;
        } else {
            const int cs = c - d.C0;
            const float v = d.p1[r.off1 + (unsigned)cs * (unsigned)d.L];
            return wunet_lrelu(d.a1[cs] * v + d.s1[cs]);
        }
    }
}

// Geometry of a position tile: TN flattened (b,l) positions = nseg segments of seg positions,
// each segment lies inside one batch item and is staged with a halo of PAD on both sides.
struct TileGeom {
    int seg, seg_shift;   // seg = min(L, TN)
    int nseg;             // TN / seg
    int segw;             // seg + 2*PAD
    int rowlen;           // staged columns per channel row
    int rowp;             // LDS row stride (floats)
    unsigned segw_magic;  // ceil(2^20 / segw)
};

// column index -> (batch item, sample index incl. halo) for a tile starting at flattened position n0
__device__ __forceinline__ void col_to_bl(const TileGeom& g, int logL, int L, int n0, int col, int pad, int& b, int& l)
{
    const int sg = (int)(((unsigned)col * g.segw_magic) >> 20);
    const int w = col - sg * g.segw - pad;
    const int gpos = n0 + (sg << g.seg_shift);
    b = gpos >> logL;
    l = (gpos & (L - 1)) + w;
}

struct ConvArgs {
    SrcDesc src;
    TileGeom geo;
    const float* wpk;   // packed weights [Mtiles_padded][CinP][TAPS][16]
    const float* bias;  // [Cout] or nullptr
    float* out;         // [B][Cout][L]
    float* stats;       // nullptr or [gridDim.x*4][Cout][2]  (sum, sum of squares of the bias-free conv)
    int B, Cout, CinP;
};

template <int TAPS, int MODE, int M_REP, int N_REP>
__global__ __launch_bounds__(WUNET_THREADS) void conv_mfma_kernel(ConvArgs A)
{
    constexpr int PAD = TAPS / 2;
    constexpr int KC = (TAPS == 15) ? 4 : 12;
    constexpr int TN = 64 * N_REP;
    constexpr int WCHUNK = KC * TAPS * 16;   // floats of one m-tile's K-chunk
    WUNET_DYN_SMEM(smem);
    float* xs = smem;
    float* ws = smem + KC * A.geo.rowp;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, i16 = lane & 15;
    const int n0 = blockIdx.x * TN;
    const int mt0 = blockIdx.y * M_REP;
    const SrcDesc& S = A.src;
    const int L = S.L;

    // columns this thread stages (same for every K-chunk)
    ColRef cr[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int col = tid + it * WUNET_THREADS;
        int b, l;
        col_to_bl(A.geo, S.logL, L, n0, col, PAD, b, l);
        col_prepare<MODE>(S, b, l, col < A.geo.rowlen && b < A.B, cr[it]);
    }

    int xoff[N_REP];
#pragma unroll
    for (int nt = 0; nt < N_REP; ++nt) {
        const int tp = wave * 16 * N_REP + nt * 16 + i16;
        const int sg = tp >> A.geo.seg_shift;
        xoff[nt] = q * A.geo.rowp + sg * A.geo.segw + (tp & (A.geo.seg - 1));
    }
    const int aoff = q * TAPS * 16 + i16;

    wunet_f4 acc[M_REP][N_REP];
#pragma unroll
    for (int mt = 0; mt < M_REP; ++mt)
#pragma unroll
        for (int nt = 0; nt < N_REP; ++nt) acc[mt][nt] = wunet_f4{0.f, 0.f, 0.f, 0.f};

    for (int c0 = 0; c0 < A.CinP; c0 += KC) {
        __syncthreads();
        // ---- stage x tile [KC][rowlen]
#pragma unroll
        for (int cl = 0; cl < KC; ++cl) {
            const int c = c0 + cl;
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int col = tid + it * WUNET_THREADS;
                if (col < A.geo.rowlen) xs[cl * A.geo.rowp + col] = col_load<MODE>(S, cr[it], c);
            }
        }
        // ---- stage packed weights: M_REP contiguous runs of WCHUNK floats
#pragma unroll
        for (int mt = 0; mt < M_REP; ++mt) {
            const float4* src = reinterpret_cast<const float4*>(A.wpk + ((size_t)(mt0 + mt) * A.CinP + c0) * (TAPS * 16));
            float4* dst = reinterpret_cast<float4*>(ws + mt * WCHUNK);
            for (int i = tid; i < WCHUNK / 4; i += WUNET_THREADS) dst[i] = src[i];
        }
        __syncthreads();
        // ---- MFMA over the chunk
#pragma unroll
        for (int s = 0; s < KC / 4; ++s) {
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                float af[M_REP], bf[N_REP];
#pragma unroll
                for (int mt = 0; mt < M_REP; ++mt) af[mt] = ws[(mt * KC + 4 * s) * TAPS * 16 + tap * 16 + aoff];
#pragma unroll
                for (int nt = 0; nt < N_REP; ++nt) bf[nt] = xs[4 * s * A.geo.rowp + tap + xoff[nt]];
#pragma unroll
                for (int mt = 0; mt < M_REP; ++mt)
#pragma unroll
                    for (int nt = 0; nt < N_REP; ++nt) acc[mt][nt] = wunet_mfma16(af[mt], bf[nt], acc[mt][nt]);
            }
        }
    }

    // ---- epilogue: bias, store, per-channel partial statistics of the bias-free conv
#pragma unroll
    for (int mt = 0; mt < M_REP; ++mt) {
        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nt = 0; nt < N_REP; ++nt) {
            const int n = n0 + wave * 16 * N_REP + nt * 16 + i16;
            const int b = n >> S.logL, l = n & (L - 1);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = (mt0 + mt) * 16 + q * 4 + r;
                const float v = acc[mt][nt][r];
                s1[r] += v;
                s2[r] += v * v;
                if (co < A.Cout && b < A.B)
                    A.out[((size_t)b * A.Cout + co) * L + l] = v + (A.bias ? A.bias[co] : 0.0f);
            }
        }
        if (A.stats) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int m = 1; m < 16; m <<= 1) {
                    s1[r] += wunet_shfl_xor(s1[r], m);
                    s2[r] += wunet_shfl_xor(s2[r], m);
                }
                const int co = (mt0 + mt) * 16 + q * 4 + r;
                if (i16 == 0 && co < A.Cout) {
                    float* st = A.stats + ((size_t)(blockIdx.x * WUNET_WAVES + wave) * A.Cout + co) * 2;
                    st[0] = s1[r];
                    st[1] = s2[r];
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Weight gradient.  grid = (ksplit, n-blocks, m-blocks).  Block tile: M_REP*16 output channels x
// (4 waves * NW) n-tiles; an n-tile is 16 (ci,tap) columns: TAPS=15 -> one ci (tap 15 is a dummy
// column), TAPS=5 -> three ci x 5 taps (column 15 is a dummy).  K runs over positions in chunks of
// TP=64; partial results go to part[ksplit][Cout][Cin][TAPS] and are summed by wgrad_reduce_kernel
// (fixed order => deterministic).
struct WgradArgs {
    SrcDesc x;       // virtual layer input  (RAW / DECIM / UPCAT), C = Cin
    SrcDesc g;       // GZ descriptor of the layer's output gradient, C = Cout
    TileGeom geo;    // geometry of a TP-position chunk of x (halo PAD)
    float* part;     // [ksplit][Cout][Cin][TAPS]
    int B, Cout, Cin;
    int chunks_per_split;
};

template <int TAPS, int MODE, int M_REP, int NW>
__global__ __launch_bounds__(WUNET_THREADS) void wgrad_mfma_kernel(WgradArgs A)
{
    constexpr int PAD = TAPS / 2;
    constexpr int TP = 64;
    constexpr int GROW = TP + 2;                                // == 2 (mod 32): A fragments conflict free
    constexpr int CI_PER_NT = (TAPS == 15) ? 1 : 3;
    constexpr int CIB = WUNET_WAVES * NW * CI_PER_NT;           // input channels per block
    WUNET_DYN_SMEM(smem);
    float* gs = smem;                          // [M_REP*16][GROW]
    float* xs = smem + M_REP * 16 * GROW;      // [CIB][rowp]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, i16 = lane & 15;
    const int split = blockIdx.x;
    const int ci0 = blockIdx.y * CIB;
    const int co0 = blockIdx.z * M_REP * 16;
    const SrcDesc& X = A.x;
    const SrcDesc& G = A.g;
    const int L = X.L;
    const int rowp = A.geo.rowp;

    // B-fragment lane offsets
    int boff[NW];
    {
        const int j = (TAPS == 15) ? i16 : (i16 < 15 ? i16 : 14);
        const int lane_off = (TAPS == 15) ? (q + j) : (q + (j % 5) + (j / 5) * rowp);
#pragma unroll
        for (int k = 0; k < NW; ++k) boff[k] = lane_off + (wave * NW + k) * CI_PER_NT * rowp;
    }
    const int aoff = i16 * GROW + q;

    wunet_f4 acc[M_REP][NW];
#pragma unroll
    for (int mt = 0; mt < M_REP; ++mt)
#pragma unroll
        for (int k = 0; k < NW; ++k) acc[mt][k] = wunet_f4{0.f, 0.f, 0.f, 0.f};

    for (int ch = 0; ch < A.chunks_per_split; ++ch) {
        const int p0 = (split * A.chunks_per_split + ch) * TP;
        __syncthreads();
        // ---- stage g_z tile: thread -> (position tid&63, channel rows tid>>6 + 4*it)
        {
            const int p = p0 + (tid & 63);
            const int b = p >> G.logL, l = p & (L - 1);
            ColRef gr;
            col_prepare<SRC_GZ>(G, b, l, b < A.B, gr);
#pragma unroll
            for (int it = 0; it < M_REP * 4; ++it) {
                const int cl = (tid >> 6) + 4 * it;
                gs[cl * GROW + (tid & 63)] = col_load<SRC_GZ>(G, gr, co0 + cl);
            }
        }
        // ---- stage x tile [CIB][rowlen]
        {
            ColRef cr[2];
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int col = tid + it * WUNET_THREADS;
                int b, l;
                col_to_bl(A.geo, X.logL, L, p0, col, PAD, b, l);
                col_prepare<MODE>(X, b, l, col < A.geo.rowlen && b < A.B, cr[it]);
            }
            for (int cl = 0; cl < CIB; ++cl) {
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int col = tid + it * WUNET_THREADS;
                    if (col < A.geo.rowlen) xs[cl * rowp + col] = col_load<MODE>(X, cr[it], ci0 + cl);
                }
            }
        }
        __syncthreads();
#pragma unroll 4
        for (int s = 0; s < TP / 4; ++s) {
            const int t4 = 4 * s;
            const int col = (t4 >> A.geo.seg_shift) * A.geo.segw + (t4 & (A.geo.seg - 1));
            float af[M_REP], bf[NW];
#pragma unroll
            for (int mt = 0; mt < M_REP; ++mt) af[mt] = gs[mt * 16 * GROW + aoff + t4];
#pragma unroll
            for (int k = 0; k < NW; ++k) bf[k] = xs[boff[k] + col];
#pragma unroll
            for (int mt = 0; mt < M_REP; ++mt)
#pragma unroll
                for (int k = 0; k < NW; ++k) acc[mt][k] = wunet_mfma16(af[mt], bf[k], acc[mt][k]);
        }
    }

    // ---- write this split's partial dW in the reference layout [Cout][Cin][TAPS]
    float* part = A.part + (size_t)split * A.Cout * A.Cin * TAPS;
#pragma unroll
    for (int mt = 0; mt < M_REP; ++mt)
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            const int nt = wave * NW + k;
            int ci, tap;
            if (TAPS == 15) { ci = ci0 + nt; tap = i16; }
            else { ci = ci0 + nt * 3 + i16 / 5; tap = i16 % 5; }
            if (i16 < 15 && ci < A.Cin) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = co0 + mt * 16 + q * 4 + r;
                    if (co < A.Cout) part[((size_t)co * A.Cin + ci) * TAPS + tap] = acc[mt][k][r];
                }
            }
        }
}
