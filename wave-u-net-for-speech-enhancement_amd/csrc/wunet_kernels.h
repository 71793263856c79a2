// MFMA kernels of the Wave-U-Net hot path for gfx950 (CDNA4).
//
//   conv_mfma_kernel  : z = W (*) x  as an implicit GEMM  M=Cout, N=B*L positions, K=Cin*taps
//                       (forward conv of every layer and - with the flipped/transposed weight pack
//                       and x = g_z - the data gradient of every layer).
//   wgrad_mfma_kernel : dW = g_z (*) x  as a GEMM  M=Cout, N=(ci,tap), K=B*L positions.
//
// All three GEMMs read MATERIALISED operands: one fused elementwise kernel per layer (prep_*_kernel)
// writes the conv's activated input x = decimate / upsample+concat of LeakyReLU(BN(z_prev)) once and the
// forward conv and the weight gradient both consume it; gz_materialize_kernel writes the BatchNorm-backward
// gradient g_z once for the data gradient and the weight gradient.  Measured on MI355X this beats loaders
// that re-derive those values inside the GEMM (the path is MFMA-bound at 3 % of HBM bandwidth, so bytes
// are cheap and issue slots beside the matrix pipe are not): PMC showed 20-35 % of wave cycles issuing
// loader VALU/VMEM and 25-45 % parked for the fused form against 12-16 % / 11-17 % for the pure GEMM.
// The MFMA is v_mfma_f32_16x16x4_f32: exact fp32 (an fmaf chain), 157 TF peak = fp32 vector peak.
//
// Layouts (all fp32, reference layout (batch, channel, sample), sample contiguous):
//   packed weights  [m-tile][ci (padded to KC)][tap][16 co]   -> a K-chunk of one m-tile is one
//                   contiguous run, A fragments are bank-conflict free (tap stride 16 dwords with
//                   TAPS odd => the two k-quarters of a 32-lane group land on disjoint bank halves).
//   LDS x tile      [ci][rowp], rowp == 16 (mod 32), each segment staged with 8 floats of halo on both
//                   sides so every global load / LDS store is an aligned float4.
#pragma once
#include "wunet_dev.h"

// A position tile of TN flattened (b,l) positions = nseg segments of seg = min(L, TN) positions; every
// segment lies inside one batch item and is staged as seg+16 floats (halo of 8 left and right).
struct ConvArgs {
    const float* x;     // [B][Cin][L] materialised input (zero padding is implied by the bounds)
    const float* wpk;   // packed weights [Mtiles_padded][CinP][TAPS][16]
    const float* bias;  // [Cout] or nullptr
    float* out;         // [B][Cout][L]  (+ z-slice * split_stride when split-K)
    float* stats;       // nullptr or [Cout][gridDim.x*4][2]  (sum, sum of squares of the bias-free conv)
    int B, Cin, Cout, CinP, L, logL;
    int seg, seg_shift, segw;      // segw = seg + 16
    int rowp;                      // LDS row stride
    int r4, sw4;                   // float4 per staged row / per segment
    unsigned r4_magic, sw4_magic;  // ceil(2^20 / r4), ceil(2^20 / sw4)
    int kc_per_split;      // split-K over gridDim.z: padded input channels per z-slice (== CinP when unsplit)
    size_t split_stride;   // floats between the partial outputs of consecutive z-slices
};

template <int TAPS, int M_REP, int N_REP>
__global__ __launch_bounds__(WUNET_THREADS) void conv_mfma_kernel(ConvArgs A)
{
    constexpr int PAD = TAPS / 2;
    constexpr int KC = (TAPS == 15) ? 4 : 12;
    constexpr int XIT = (TAPS == 15) ? 2 : 4;       // float4 x slots per thread: KC * r4 <= 256 * XIT
    constexpr int TN = 64 * N_REP;
    constexpr int WCHUNK = KC * TAPS * 16;          // floats of one m-tile's K-chunk (240 float4)
    WUNET_DYN_SMEM(smem);
    float* xs = smem;                               // [KC][rowp]
    float* ws = smem + KC * A.rowp;                 // [M_REP][KC][TAPS][16]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, i16 = lane & 15;
    const int n0 = blockIdx.x * TN;
    const int mt0 = blockIdx.y * M_REP;
    const int L = A.L;

    // ---- per-thread x staging slots (the tile is fixed for the block, only the channel chunk moves)
    int xrow[XIT], xlds[XIT];
    unsigned xg[XIT];
#pragma unroll
    for (int it = 0; it < XIT; ++it) {
        const int f = tid + it * WUNET_THREADS;
        const int row = (int)(((unsigned)f * A.r4_magic) >> 20);
        const int c4 = f - row * A.r4;
        const int sg = (int)(((unsigned)c4 * A.sw4_magic) >> 20);
        const int w4 = c4 - sg * A.sw4;
        const int gpos = n0 + (sg << A.seg_shift);
        const int b = gpos >> A.logL, l = (gpos & (L - 1)) + 4 * w4 - 8;
        const bool ok = row < KC && b < A.B && l >= 0 && l < L;
        xrow[it] = ok ? row : -1 - (row < KC ? row : KC);     // < 0: zero fill (still a valid LDS slot when row < KC)
        xlds[it] = row * A.rowp + sg * A.segw + 4 * w4;
        xg[it] = ok ? (unsigned)b * (unsigned)A.Cin * (unsigned)L + (unsigned)l : 0u;
    }

    int xoff[N_REP];
#pragma unroll
    for (int nt = 0; nt < N_REP; ++nt) {
        const int tp = wave * 16 * N_REP + nt * 16 + i16;
        const int sg = tp >> A.seg_shift;
        xoff[nt] = q * A.rowp + sg * A.segw + (tp & (A.seg - 1)) + (8 - PAD);
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
    wunet_f4 xreg[XIT], wreg[M_REP];
    const int wtid = tid < WCHUNK / 4 ? tid : 0;      // clamped: threads >= WCHUNK/4 load a dummy, never store it
#define WUNET_PREFETCH(C0_)                                                                                     \
    {                                                                                                           \
        _Pragma("unroll") for (int it = 0; it < XIT; ++it) {                                                    \
            const int c_ = (C0_) + (xrow[it] >= 0 ? xrow[it] : 0);                                              \
            const bool ok_ = xrow[it] >= 0 && c_ < A.Cin;                                                       \
            xreg[it] = wunet_ld4(A.x + (ok_ ? xg[it] + (unsigned)c_ * (unsigned)L : 0u));   /* raw: select at store time */ \
        }                                                                                                       \
        _Pragma("unroll") for (int mt = 0; mt < M_REP; ++mt)                                                    \
            wreg[mt] = wunet_ld4(A.wpk + ((size_t)(mt0 + mt) * A.CinP + (C0_)) * (TAPS * 16) + 4 * wtid);      \
    }
    WUNET_PREFETCH(cbeg)

    for (int c0 = cbeg; c0 < cend; c0 += KC) {
        __syncthreads();      // every wave is done reading the previous chunk from LDS
        // the bounds select happens HERE, not at load time: a select next to the load makes hipcc wait for the
        // data before the MFMA phase (vmcnt + v_cndmask straight after the global_load), un-hiding its latency
#pragma unroll
        for (int it = 0; it < XIT; ++it)
            if (xrow[it] > -1 - KC) wunet_st4(xs + xlds[it], wunet_sel4(xrow[it] >= 0 && c0 + xrow[it] < A.Cin, xreg[it]));
        if (tid < WCHUNK / 4) {
#pragma unroll
            for (int mt = 0; mt < M_REP; ++mt) wunet_st4(ws + mt * WCHUNK + 4 * tid, wreg[mt]);
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
                for (int nt = 0; nt < N_REP; ++nt) bf[nt] = xs[4 * s * A.rowp + tap + xoff[nt]];
#pragma unroll
                for (int mt = 0; mt < M_REP; ++mt)
#pragma unroll
                    for (int nt = 0; nt < N_REP; ++nt) acc[mt][nt] = wunet_mfma16(af[mt], bf[nt], acc[mt][nt]);
            }
        }
    }
#undef WUNET_PREFETCH

    // ---- epilogue: bias, store, per-channel partial statistics of the bias-free conv
    // (the biases in ONE batch of loads: loaded at their use, each was waited for with vmcnt(0) - together with the stores before it)
    float bvs[M_REP][4];
#pragma unroll
    for (int mt = 0; mt < M_REP; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = (mt0 + mt) * 16 + q * 4 + r;
            bvs[mt][r] = (A.bias && co < A.Cout) ? A.bias[co] : 0.0f;
        }
#pragma unroll
    for (int mt = 0; mt < M_REP; ++mt) {
        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nt = 0; nt < N_REP; ++nt) {
            const int n = n0 + wave * 16 * N_REP + nt * 16 + i16;
            const int b = n >> A.logL, l = n & (L - 1);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = (mt0 + mt) * 16 + q * 4 + r;
                const float v = acc[mt][nt][r];
                s1[r] += v;
                s2[r] = fmaf(v, v, s2[r]);
                if (co < A.Cout && b < A.B)
                    outp[((size_t)b * A.Cout + co) * L + l] = v + bvs[mt][r];
            }
        }
        if (A.stats) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s1[r] = wunet_row16_sum(s1[r]);
                s2[r] = wunet_row16_sum(s2[r]);
                const int co = (mt0 + mt) * 16 + q * 4 + r;
                if (i16 == 0 && co < A.Cout) {
                    // [channel][row][2]: the finalize kernel then reads each channel's rows contiguously
                    float* st = A.stats + ((size_t)co * (gridDim.x * WUNET_WAVES) + (blockIdx.x * WUNET_WAVES + wave)) * 2;
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
    // WSPLIT (the first layer, Cin = 1: nothing else reads its g_z) with z != nullptr: g holds the gradient at the BatchNorm input's
    // activation and the kernel forms g_z = k1[c] g + k2[c] z + k3[c] (zero in the row padding l >= Lt) while it stages the chunk -
    // gz_materialize_kernel's pass (100 MB written and read back at the end of the backward's critical path) is not run
    const float* z; const float* k1; const float* k2; const float* k3; int Lt;
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

    wunet_f4 greg[M_REP], xreg[XIT];
    const bool fuse_gz = WSPLIT && A.z != nullptr;
    wunet_f4 zreg[WSPLIT ? M_REP : 1];
    float k1v[WSPLIT ? M_REP : 1], k2v[WSPLIT ? M_REP : 1], k3v[WSPLIT ? M_REP : 1];
    if (WSPLIT) {
#pragma unroll
        for (int it = 0; it < M_REP; ++it) {
            const int co_ = co0 + (tid >> 4) + 16 * it;
            const bool ok_ = fuse_gz && co_ < A.Cout;
            k1v[it] = ok_ ? A.k1[co_] : 1.0f;
            k2v[it] = ok_ ? A.k2[co_] : 0.0f;
            k3v[it] = ok_ ? A.k3[co_] : 0.0f;
            zreg[it] = wunet_f4{0.f, 0.f, 0.f, 0.f};
        }
    }
#define WUNET_WG_PREFETCH(P0_)                                                                                   \
    {                                                                                                            \
        const int p_ = (P0_) + 4 * gq;                                                                           \
        const int b_ = p_ >> A.logL, l_ = p_ & (L - 1);                                                          \
        _Pragma("unroll") for (int it = 0; it < M_REP; ++it) {                                                   \
            const int co_ = co0 + (tid >> 4) + 16 * it;                                                          \
            const bool ok_ = b_ < A.B && co_ < A.Cout;                                                           \
            const size_t o_ = ok_ ? ((size_t)b_ * A.Cout + co_) * L + l_ : 0;                                    \
            greg[it] = wunet_ld4(A.g + o_);                                                                      \
            if (WSPLIT) { if (fuse_gz) zreg[it] = wunet_ld4(A.z + o_); }                                         \
        }                                                                                                        \
        _Pragma("unroll") for (int it = 0; it < XIT; ++it) {                                                     \
            const int gp_ = (P0_) + (xsg[it] << A.seg_shift);                                                    \
            const int xb_ = gp_ >> A.logL, xl_ = (gp_ & (L - 1)) + xrel[it];                                     \
            const int ci_ = ci0 + xrow[it];                                                                      \
            const bool ok_ = xrow[it] >= 0 && ci_ < A.Cin && xb_ < A.B && xl_ >= 0 && xl_ < L;                   \
            const size_t o_ = ok_ ? ((size_t)xb_ * A.Cin + ci_) * L + xl_ : 0;                                   \
            xreg[it] = wunet_ld4(A.x + o_);                                                                      \
        }                                                                                                        \
    }
    const int pbeg = split * A.chunks_per_split * TP;
    WUNET_WG_PREFETCH(pbeg)

    for (int ch = 0; ch < A.chunks_per_split; ++ch) {
        __syncthreads();
        // ---- registers -> LDS; out-of-range loads are zeroed here (a select at load time would make hipcc wait
        //      for the data before the MFMA phase)
        {
            const int pc = pbeg + ch * TP;
            const int p_ = pc + 4 * gq;
            const int b_ = p_ >> A.logL;
#pragma unroll
            for (int it = 0; it < M_REP; ++it) {
                wunet_f4 gv = wunet_sel4(b_ < A.B && co0 + (tid >> 4) + 16 * it < A.Cout, greg[it]);
                if (WSPLIT) {
                    if (fuse_gz) {
                        const int l_ = p_ & (L - 1);
                        const wunet_f4 zv = wunet_sel4(b_ < A.B && co0 + (tid >> 4) + 16 * it < A.Cout, zreg[it]);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float v = k1v[it] * gv[j] + k2v[it] * zv[j] + k3v[it];
                            gv[j] = l_ + j < A.Lt ? v : 0.0f;
                        }
                    }
                }
                float* dst = gs + ((tid >> 4) + 16 * it) * GROW + 4 * gq;      // 8-byte aligned
                reinterpret_cast<float2*>(dst)[0] = float2{gv[0], gv[1]};
                reinterpret_cast<float2*>(dst)[1] = float2{gv[2], gv[3]};
            }
#pragma unroll
            for (int it = 0; it < XIT; ++it) {
                const int gp_ = pc + (xsg[it] << A.seg_shift);
                const int xb_ = gp_ >> A.logL, xl_ = (gp_ & (L - 1)) + xrel[it];
                const bool ok_ = xrow[it] >= 0 && ci0 + xrow[it] < A.Cin && xb_ < A.B && xl_ >= 0 && xl_ < L;
                if (xrow[it] >= 0) wunet_st4(xs + xlds[it], wunet_sel4(ok_, xreg[it]));
            }
        }
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
