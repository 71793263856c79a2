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
    unsigned out_off;      // offset of (b, channel 0, l) in a materialised [B][C][L] copy of the virtual input
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
    r.out_off = bb * (unsigned)d.C * (unsigned)d.L + (unsigned)lc;
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

// ---- loaders.  All global loads are UNCONDITIONAL (addresses are clamped to something valid and the
// result is selected afterwards): a per-element branch around a load makes hipcc serialise the loads
// (one basic block and one vmcnt(0) wait per element - cdna_hip_programming.md "three .s-level traps" (c)).
// Two-phase form for software pipelining: raw global loads now (col_fetch), arithmetic later (col_finish).
struct RawX { float v0, v1; };
struct ChanK { float a, s, t; };   // per-channel constants (wave-uniform): scale, shift, (GZ: k3)

template <int MODE>
__device__ __forceinline__ ChanK chan_consts(const SrcDesc& d, int c)
{
    ChanK k{0.f, 0.f, 0.f};
    const int cc = c < d.C ? c : d.C - 1;
    if (MODE == SRC_DECIM) { k.a = d.a0[cc]; k.s = d.s0[cc]; }
    else if (MODE == SRC_GZ) { k.a = d.a0[cc]; k.s = d.s0[cc]; k.t = d.a1[cc]; }
    else if (MODE == SRC_UPCAT) {
        const bool up = cc < d.C0;
        const float* pa = up ? d.a0 : d.a1;
        const float* ps = up ? d.s0 : d.s1;
        const int ci = up ? cc : cc - d.C0;
        k.a = pa[ci]; k.s = ps[ci];
    }
    return k;
}

template <int MODE>
__device__ __forceinline__ RawX col_fetch(const SrcDesc& d, const ColRef& r, int c)
{
    RawX x{0.f, 0.f};
    const unsigned cc = (unsigned)(c < d.C ? c : d.C - 1);
    if (MODE == SRC_RAW) {
        x.v0 = d.p0[r.off0 + cc * (unsigned)d.L];
    } else if (MODE == SRC_DECIM) {
        x.v0 = d.p0[r.off0 + cc * (unsigned)d.Lsrc0];
    } else if (MODE == SRC_GZ) {
        const unsigned o = r.off0 + cc * (unsigned)d.L;
        x.v0 = d.p0[o];
        x.v1 = d.p1[o];
    } else {
        const bool up = cc < (unsigned)d.C0;                       // wave-uniform
        const float* p = up ? d.p0 : d.p1;
        const unsigned cb = up ? cc * (unsigned)d.Lsrc0 : (cc - (unsigned)d.C0) * (unsigned)d.L;
        x.v0 = p[(up ? r.off0 : r.off1) + cb];
        x.v1 = p[(up ? r.offu1 : r.off1) + cb];
    }
    return x;
}

template <int MODE>
__device__ __forceinline__ float col_finish(const SrcDesc& d, const ColRef& r, int c, const RawX& x, const ChanK& k)
{
    float v;
    if (MODE == SRC_RAW) v = x.v0;
    else if (MODE == SRC_DECIM) v = wunet_lrelu(k.a * x.v0 + k.s);
    else if (MODE == SRC_GZ) v = k.a * x.v0 + k.s * x.v1 + k.t;
    else {
        const bool up = c < d.C0;                                  // wave-uniform
        const float w0 = up ? r.l0 : 1.0f, w1 = up ? r.l1 : 0.0f;
        v = w0 * wunet_lrelu(k.a * x.v0 + k.s) + w1 * wunet_lrelu(k.a * x.v1 + k.s);
    }
    return (r.valid && c < d.C) ? v : 0.0f;
}

// value of virtual channel c at a prepared column (c is wave-uniform)
template <int MODE>
__device__ __forceinline__ float col_load(const SrcDesc& d, const ColRef& r, int c)
{
    const ChanK k = chan_consts<MODE>(d, c);
    const RawX x = col_fetch<MODE>(d, r, c);
    return col_finish<MODE>(d, r, c, x, k);
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
    float* xout;           // nullptr or [B][C][L]: the staged (activated / g_z) input tile is also written here by
                           // the blockIdx.y == 0 blocks, so the weight-gradient GEMM reads it instead of re-deriving it
    int kc_per_split;      // split-K over gridDim.z: padded input channels per z-slice (== CinP when unsplit)
    size_t split_stride;   // floats between the partial outputs of consecutive z-slices
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

    // columns that are tile interior (not halo) are also materialised to xout
    bool xw[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int col = tid + it * WUNET_THREADS;
        const int sg = (int)(((unsigned)col * A.geo.segw_magic) >> 20);
        const int w = col - sg * A.geo.segw - PAD;
        xw[it] = A.xout != nullptr && blockIdx.y == 0 && cr[it].valid && w >= 0 && w < A.geo.seg && col < A.geo.rowlen;
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

    const int cbeg = blockIdx.z * A.kc_per_split;
    const int cend = cbeg + A.kc_per_split < A.CinP ? cbeg + A.kc_per_split : A.CinP;
    float* const outp = A.out + (size_t)blockIdx.z * A.split_stride;

    // software pipeline: the global loads of chunk k+1 are in flight while chunk k runs on the matrix cores
    RawX rx[KC][2];
    ChanK ck[KC];
    float4 wreg[M_REP];
    const int wtid = tid < WCHUNK / 4 ? tid : 0;      // clamped: threads >= WCHUNK/4 load a dummy, never store it
#define WUNET_PREFETCH(C0_)                                                                                     \
    {                                                                                                           \
        _Pragma("unroll") for (int cl = 0; cl < KC; ++cl) {                                                     \
            ck[cl] = chan_consts<MODE>(S, (C0_) + cl);                                                          \
            _Pragma("unroll") for (int it = 0; it < 2; ++it) rx[cl][it] = col_fetch<MODE>(S, cr[it], (C0_) + cl); \
        }                                                                                                       \
        _Pragma("unroll") for (int mt = 0; mt < M_REP; ++mt)                                                    \
            wreg[mt] = reinterpret_cast<const float4*>(A.wpk + ((size_t)(mt0 + mt) * A.CinP + (C0_)) * (TAPS * 16))[wtid]; \
    }
    WUNET_PREFETCH(cbeg)

    for (int c0 = cbeg; c0 < cend; c0 += KC) {
        __syncthreads();      // every wave is done reading the previous chunk from LDS
        // ---- registers -> LDS (BN scale/shift, LeakyReLU, interpolation applied here)
#pragma unroll
        for (int cl = 0; cl < KC; ++cl) {
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int col = tid + it * WUNET_THREADS;
                const float xv = col_finish<MODE>(S, cr[it], c0 + cl, rx[cl][it], ck[cl]);
                if (col < A.geo.rowlen) xs[cl * A.geo.rowp + col] = xv;
                if (xw[it] && c0 + cl < S.C) A.xout[cr[it].out_off + (unsigned)(c0 + cl) * (unsigned)L] = xv;
            }
        }
        if (tid < WCHUNK / 4) {
#pragma unroll
            for (int mt = 0; mt < M_REP; ++mt) reinterpret_cast<float4*>(ws + mt * WCHUNK)[tid] = wreg[mt];
        }
        __syncthreads();
        if (c0 + KC < cend) WUNET_PREFETCH(c0 + KC)
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

#undef WUNET_PREFETCH
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
                    outp[((size_t)b * A.Cout + co) * L + l] = v + (A.bias ? A.bias[co] : 0.0f);
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
// Weight gradient  dW[co][ci][tap] = sum_{b,l} g_z[b,co,l] * x[b,ci,l+tap-PAD]  as a GEMM M=Cout,
// N=(ci,tap), K=B*L positions, on MATERIALISED operands: x is the activated layer input written by the
// forward conv's loader (ConvArgs::xout), g_z is written by the data-gradient's loader.  Both are plain
// [B][C][L] tensors, so staging is aligned float4 copies, software-pipelined one chunk ahead.
//   grid = (ksplit, n-blocks, m-blocks); block tile = M_REP*16 output channels x (4 waves * NW) n-tiles.
//   n-tile = 16 (ci,tap) columns: TAPS=15 -> one ci (+1 dummy tap), TAPS=5 -> 3 ci x 5 taps (+1 dummy).
//   K runs over chunks of 64 positions; a chunk is nseg segments of seg=min(L,64) positions, each staged
//   with 8 floats of halo on both sides (16-byte aligned rows; out-of-range float4s are zero).
//   WSPLIT (Cin < 4 n-tiles, i.e. encoder[0]): all four waves work on n-tile 0 and split the K-steps;
//   each wave writes its own partial row.
//   Partials part[row][Cout][Cin][TAPS] are summed by wgrad_reduce_kernel in a fixed order (deterministic).
// component-wise select (a struct-level ?: on float4 goes through scratch memory with hipcc)
__device__ __forceinline__ float4 wunet_sel4(bool ok, const float4& v)
{
    float4 r;
    r.x = ok ? v.x : 0.0f; r.y = ok ? v.y : 0.0f; r.z = ok ? v.z : 0.0f; r.w = ok ? v.w : 0.0f;
    return r;
}

struct WgradArgs {
    const float* x;      // [B][Cin][L]
    const float* g;      // [B][Cout][L]
    float* part;
    int B, Cin, Cout, L, logL;
    int chunks_per_split;
    int seg, seg_shift, segw;     // segw = seg + 16
    int rowp;                     // LDS x row stride (floats, multiple of 4)
    int r4;                       // float4 per staged x row = nseg*segw/4
    int sw4;                      // float4 per segment = segw/4
    unsigned r4_magic, sw4_magic; // ceil(2^20 / r4), ceil(2^20 / sw4)
};

template <int TAPS, int M_REP, int NW, int XIT, bool WSPLIT>
__global__ __launch_bounds__(WUNET_THREADS) void wgrad_mfma_kernel(WgradArgs A)
{
    constexpr int PAD = TAPS / 2;
    constexpr int TP = 64;
    constexpr int GROW = TP + 2;                                // == 2 (mod 32): A fragments conflict free
    constexpr int CI_PER_NT = (TAPS == 15) ? 1 : 3;
    constexpr int CIB = WSPLIT ? 1 : WUNET_WAVES * NW * CI_PER_NT;   // input channels per block
    WUNET_DYN_SMEM(smem);
    float* gs = smem;                          // [M_REP*16][GROW]
    float* xs = smem + M_REP * 16 * GROW;      // [CIB][rowp]  (16-byte aligned: M_REP*16*66*4 is a multiple of 16)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, i16 = lane & 15;
    const int split = blockIdx.x;
    const int ci0 = blockIdx.y * CIB;
    const int co0 = blockIdx.z * M_REP * 16;
    const int L = A.L;
    const int rowp = A.rowp;

    // ---- per-thread staging slots (independent of the chunk)
    // g: float4 slot it -> channel row (tid>>4)+16*it, positions 4*(tid&15)..+3 of the chunk
    const int gq = tid & 15;
    // x: float4 slot it -> flattened index f = tid + 256*it over [CIB][r4]
    int xrow[XIT], xsg[XIT], xrel[XIT], xlds[XIT];
#pragma unroll
    for (int it = 0; it < XIT; ++it) {
        const int f = tid + it * WUNET_THREADS;
        const int row = (int)(((unsigned)f * A.r4_magic) >> 20);
        const int c4 = f - row * A.r4;
        const int sg = (int)(((unsigned)c4 * A.sw4_magic) >> 20);
        const int w4 = c4 - sg * A.sw4;
        xrow[it] = row < CIB ? row : -1;
        xsg[it] = sg;
        xrel[it] = 4 * w4 - 8;
        xlds[it] = row * rowp + sg * A.segw + 4 * w4;
    }

    // B-fragment lane offsets
    int boff[NW];
    {
        const int j = (TAPS == 15) ? i16 : (i16 < 15 ? i16 : 14);
        const int lane_off = (8 - PAD) + ((TAPS == 15) ? (q + j) : (q + (j % 5) + (j / 5) * rowp));
#pragma unroll
        for (int k = 0; k < NW; ++k) boff[k] = lane_off + (WSPLIT ? 0 : (wave * NW + k) * CI_PER_NT * rowp);
    }
    const int aoff = i16 * GROW + q;

    wunet_f4 acc[M_REP][NW];
#pragma unroll
    for (int mt = 0; mt < M_REP; ++mt)
#pragma unroll
        for (int k = 0; k < NW; ++k) acc[mt][k] = wunet_f4{0.f, 0.f, 0.f, 0.f};

    float4 greg[M_REP], xreg[XIT];
#define WUNET_WG_PREFETCH(P0_)                                                                                   \
    {                                                                                                            \
        const int p_ = (P0_) + 4 * gq;                                                                           \
        const int b_ = p_ >> A.logL, l_ = p_ & (L - 1);                                                          \
        _Pragma("unroll") for (int it = 0; it < M_REP; ++it) {                                                   \
            const int co_ = co0 + (tid >> 4) + 16 * it;                                                          \
            const bool ok_ = b_ < A.B && co_ < A.Cout;                                                           \
            const size_t o_ = ok_ ? ((size_t)b_ * A.Cout + co_) * L + l_ : 0;                                    \
            const float4 v_ = *reinterpret_cast<const float4*>(A.g + o_);                                        \
            greg[it] = wunet_sel4(ok_, v_);                                                                      \
        }                                                                                                        \
        _Pragma("unroll") for (int it = 0; it < XIT; ++it) {                                                     \
            const int gp_ = (P0_) + (xsg[it] << A.seg_shift);                                                    \
            const int xb_ = gp_ >> A.logL, xl_ = (gp_ & (L - 1)) + xrel[it];                                     \
            const int ci_ = ci0 + xrow[it];                                                                      \
            const bool ok_ = xrow[it] >= 0 && ci_ < A.Cin && xb_ < A.B && xl_ >= 0 && xl_ < L;                   \
            const size_t o_ = ok_ ? ((size_t)xb_ * A.Cin + ci_) * L + xl_ : 0;                                   \
            const float4 v_ = *reinterpret_cast<const float4*>(A.x + o_);                                        \
            xreg[it] = wunet_sel4(ok_, v_);                                                                      \
        }                                                                                                        \
    }
    const int pbeg = split * A.chunks_per_split * TP;
    WUNET_WG_PREFETCH(pbeg)

    for (int ch = 0; ch < A.chunks_per_split; ++ch) {
        __syncthreads();
        // ---- registers -> LDS
#pragma unroll
        for (int it = 0; it < M_REP; ++it) {
            float* dst = gs + ((tid >> 4) + 16 * it) * GROW + 4 * gq;      // 8-byte aligned
            reinterpret_cast<float2*>(dst)[0] = float2{greg[it].x, greg[it].y};
            reinterpret_cast<float2*>(dst)[1] = float2{greg[it].z, greg[it].w};
        }
#pragma unroll
        for (int it = 0; it < XIT; ++it)
            if (xrow[it] >= 0) *reinterpret_cast<float4*>(xs + xlds[it]) = xreg[it];
        __syncthreads();
        if (ch + 1 < A.chunks_per_split) WUNET_WG_PREFETCH(pbeg + (ch + 1) * TP)
#pragma unroll 4
        for (int s = 0; s < TP / 4; ++s) {
            if (WSPLIT && (s & 3) != wave) continue;
            const int t4 = 4 * s;
            const int col = (t4 >> A.seg_shift) * A.segw + (t4 & (A.seg - 1));
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
#undef WUNET_WG_PREFETCH

    // ---- write this block's partial dW in the reference layout [Cout][Cin][TAPS]
    const int prow = WSPLIT ? split * WUNET_WAVES + wave : split;
    float* part = A.part + (size_t)prow * A.Cout * A.Cin * TAPS;
#pragma unroll
    for (int mt = 0; mt < M_REP; ++mt)
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            const int nt = WSPLIT ? 0 : wave * NW + k;
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
