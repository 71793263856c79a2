// fp16-split ("h3") GEMM path for the large levels: every fp32 operand x is carried as hi + lo with hi, lo fp16
// (22 significant bits together, gradients pre-scaled by a power of two into fp16's range) and a product is
// hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_f16 with fp32 accumulation.  Measured against the reference this is
// indistinguishable from fp32 (end-to-end 4e-6 on the output, <=1.1e-5 on gradients - the fp32 noise floor; a 3xbf16
// split was measured at 4.5e-5 / 1.1e-4 and rejected), while the matrix pipe runs 16x faster per instruction:
// 3 passes => ~5x the fp32 MFMA rate.
//
// Layouts:
//   activations  hi / lo arrays  [B][C8][L][8] halfs  (C8 = ceil(C/8); 16 bytes per (channel group, sample)), so a
//                position tile of one channel group is a contiguous run of 16-byte pieces (linear LDS image) and a
//                B fragment (8 consecutive K = 8 channels at one tap) is one ds_read_b128.
//   weights      hi / lo arrays  [m-tile][chunk of 32 ch][tap][4 quarters][16 rows][8 ch] halfs.
#pragma once
#include "wunet_dev.h"

// ---------------------------------------------------------------------------- conv / data gradient
// Implicit GEMM like conv_mfma_kernel, 256 positions x M_REP*16 rows per block (L >= 256: a tile lies inside one batch
// item), K walked as chunks of 32 channels x groups of TG=5 taps.  Per stage the block stages the W sub-tile (and, for
// the first tap group of a chunk, the x tile: 4 channel groups x 272 columns, hi and lo) and each wave issues
// 5 taps x M_REP x 4 tiles x 3 MFMAs.
struct ConvH3Args {
    const wunet_half* xh; const wunet_half* xl;   // [B][C8][L][8]
    const wunet_half* wh; const wunet_half* wl;   // packed
    const float* bias;                            // [Cout] or nullptr
    const float* sc;                              // nullptr or {scale, 1/scale} of the input: the result is multiplied by sc[1]
    float* out;                                   // [B][Cout][L] fp32
    float* stats;                                 // nullptr or [Cout][gridDim.x*4][2]
    int B, Cout, C8, NCH, L, logL;
};

template <int TAPS, int M_REP>
__global__ __launch_bounds__(WUNET_THREADS) void conv_h3_kernel(ConvH3Args A)
{
    constexpr int PAD = TAPS / 2;
    constexpr int TG = 5;                         // taps per stage
    constexpr int NTG = TAPS / TG;
    constexpr int COLS = 272;                     // 256 + 8 + 8
    constexpr int XP = 2 * 4 * COLS;              // 16-byte pieces of the x tile (hi + lo)
    constexpr int XIT = (XP + WUNET_THREADS - 1) / WUNET_THREADS;     // 9
    constexpr int WPM = TG * 64;                  // pieces per (m-tile, hi|lo) sub-tile
    constexpr int WP = 2 * M_REP * WPM;
    constexpr int WIT = (WP + WUNET_THREADS - 1) / WUNET_THREADS;
    WUNET_DYN_SMEM(smem);
    wunet_half* xs = reinterpret_cast<wunet_half*>(smem);             // [hi|lo][4][COLS][8]
    wunet_half* ws = xs + XP * 8;                                      // [hi|lo][M_REP][TG][4][16][8]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, i16 = lane & 15;
    const int n0 = blockIdx.x * 256;
    const int b = n0 >> A.logL, l0 = n0 & (A.L - 1);
    const int mt0 = blockIdx.y * M_REP;
    const int L = A.L;

    // x slots: piece f -> (which, c8 local, column)
    int xc8[XIT], xcol[XIT];
#pragma unroll
    for (int it = 0; it < XIT; ++it) {
        const int f = tid + it * WUNET_THREADS;
        const int r = f % (4 * COLS);
        xc8[it] = f < XP ? ((r / COLS) | (f >= 4 * COLS ? 8 : 0)) : -1;       // bits 0-2: channel group in the chunk, bit 3: lo array
        xcol[it] = r % COLS;
    }
    // B-fragment base (halfs): plane q, column of this lane
    int boff[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) boff[nt] = (q * COLS + wave * 64 + nt * 16 + i16 + (8 - PAD)) * 8;
    const int aoff = (q * 16 + i16) * 8;

    wunet_f4 acc[M_REP][4];
#pragma unroll
    for (int mt = 0; mt < M_REP; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = wunet_f4{0.f, 0.f, 0.f, 0.f};

    wunet_h8 xreg[XIT], wreg[WIT];
    const int nstage = A.NCH * NTG;
#define WUNET_H3_PREFETCH(ST_)                                                                                    \
    {                                                                                                             \
        const int ch_ = (ST_) / NTG, tg_ = (ST_) - ch_ * NTG;                                                    \
        if (tg_ == 0) {                                                                                           \
            _Pragma("unroll") for (int it = 0; it < XIT; ++it) {                                                  \
                const int c8g_ = ch_ * 4 + (xc8[it] & 7);                                                         \
                const int l_ = l0 - 8 + xcol[it];                                                                 \
                const bool ok_ = xc8[it] >= 0 && c8g_ < A.C8 && b < A.B && l_ >= 0 && l_ < L;                     \
                const wunet_half* src_ = (xc8[it] & 8) ? A.xl : A.xh;                                             \
                xreg[it] = wunet_ldh8(src_ + (ok_ ? (((size_t)b * A.C8 + c8g_) * L + l_) * 8 : 0));               \
            }                                                                                                     \
        }                                                                                                         \
        _Pragma("unroll") for (int it = 0; it < WIT; ++it) {                                                      \
            const int f_ = tid + it * WUNET_THREADS;                                                              \
            const int g_ = f_ < WP ? f_ : 0;                                                                      \
            const int which_ = g_ / (M_REP * WPM), r_ = g_ % (M_REP * WPM), mt_ = r_ / WPM, p_ = r_ % WPM;        \
            const wunet_half* src_ = which_ ? A.wl : A.wh;                                                        \
            wreg[it] = wunet_ldh8(src_ + ((((size_t)(mt0 + mt_) * A.NCH + ch_) * TAPS + tg_ * TG) * 64 + p_) * 8); \
        }                                                                                                         \
    }
    WUNET_H3_PREFETCH(0)

    for (int st = 0; st < nstage; ++st) {
        const int ch = st / NTG, tg = st - ch * NTG;
        __syncthreads();
        if (tg == 0) {
#pragma unroll
            for (int it = 0; it < XIT; ++it) {
                const int f = tid + it * WUNET_THREADS;
                const int c8g = ch * 4 + (xc8[it] & 7);
                const int l = l0 - 8 + xcol[it];
                const bool ok = xc8[it] >= 0 && c8g < A.C8 && b < A.B && l >= 0 && l < L;
                if (f < XP) wunet_sth8(xs + (size_t)f * 8, wunet_selh8(ok, xreg[it]));
            }
        }
#pragma unroll
        for (int it = 0; it < WIT; ++it) {
            const int f = tid + it * WUNET_THREADS;
            if (f < WP) wunet_sth8(ws + (size_t)f * 8, wreg[it]);
        }
        __syncthreads();
        if (st + 1 < nstage) WUNET_H3_PREFETCH(st + 1)
#pragma unroll
        for (int tl = 0; tl < TG; ++tl) {
            const int tap = tg * TG + tl;
            wunet_h8 ah[M_REP], al[M_REP], bh[4], bl[4];
#pragma unroll
            for (int mt = 0; mt < M_REP; ++mt) {
                ah[mt] = wunet_ldh8(ws + ((mt * TG + tl) * 64) * 8 + aoff);
                al[mt] = wunet_ldh8(ws + ((M_REP + mt) * TG + tl) * 64 * 8 + aoff);
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                bh[nt] = wunet_ldh8(xs + boff[nt] + tap * 8);
                bl[nt] = wunet_ldh8(xs + 4 * COLS * 8 + boff[nt] + tap * 8);
            }
#pragma unroll
            for (int mt = 0; mt < M_REP; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    acc[mt][nt] = wunet_mfma16h(al[mt], bh[nt], acc[mt][nt]);
                    acc[mt][nt] = wunet_mfma16h(ah[mt], bl[nt], acc[mt][nt]);
                    acc[mt][nt] = wunet_mfma16h(ah[mt], bh[nt], acc[mt][nt]);
                }
        }
    }
#undef WUNET_H3_PREFETCH

    // ---- epilogue (as conv_mfma_kernel): un-scale, bias, store, BN statistics of the bias-free conv
    const float inv = A.sc ? A.sc[1] : 1.0f;
#pragma unroll
    for (int mt = 0; mt < M_REP; ++mt) {
        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int l = l0 + wave * 64 + nt * 16 + i16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = (mt0 + mt) * 16 + q * 4 + r;
                const float v = acc[mt][nt][r] * inv;
                s1[r] += v;
                s2[r] += v * v;
                if (co < A.Cout && b < A.B) A.out[((size_t)b * A.Cout + co) * L + l] = v + (A.bias ? A.bias[co] : 0.0f);
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
                    float* stp = A.stats + ((size_t)co * (gridDim.x * WUNET_WAVES) + (blockIdx.x * WUNET_WAVES + wave)) * 2;
                    stp[0] = s1[r];
                    stp[1] = s2[r];
                }
            }
        }
    }
}
