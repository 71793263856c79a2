// fp16-split ("h3") GEMM path for the levels >= 32 samples: every fp32 operand x is carried as hi + lo with hi, lo fp16
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

// 16-byte zero page the LDS-DMA pieces that lie outside a tensor are fetched from (halo beyond an item, channel groups beyond C8):
// one copy per translation unit, in the code object - nothing to allocate or clear per step
#ifdef WUNET_EMU
static const unsigned wunet_zero16[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#else
static __device__ __attribute__((aligned(16))) const unsigned wunet_zero16[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif

// ---------------------------------------------------------------------------- conv / data gradient
// Implicit GEMM like conv_mfma_kernel, 256 positions x M_REP*16 rows per block, K walked as chunks of 32 channels x
// groups of TG=5 taps.  NSEG = 1: L >= 256, the tile lies inside one batch item; NSEG = 2 .. 16: L = 128 .. 16, the tile is
// NSEG whole items, each with its own zero halo in the LDS image (a lane's 4 positions never straddle items).
// Optional split-K over gridDim.y (short levels: too few position tiles to fill the chip): bias-free partial results
// [split][B][Cout][L], summed by conv_reduce_bn_kernel / split_sum_kernel like the fp32 path.  Per stage the block stages the W sub-tile (and, for
// the first tap group of a chunk, the x tile: 4 channel groups x 272 columns, hi and lo) and each wave issues
// 5 taps x M_REP x 4 tiles x 3 MFMAs.  The kernel is conv_h3d_kernel (wunet_h3d.h); its register-staged predecessor and the
// paired-tile variant of round 1 / 2 are gone (measurements: HISTORY.md sections 7, 8).
struct ConvH3Args {
    const wunet_half* xh; const wunet_half* xl;   // [B][C8][L][8]
    const wunet_half* wh; const wunet_half* wl;   // packed
    // conv_h3d_kernel addresses its DMA pieces as SGPR base + unsigned 32-bit offset: the lo plane / lo pack must lie xdelta / wdelta
    // bytes BEHIND the hi one (< 4 GiB), and zpad - 16 zero bytes that pieces outside the tensor fetch - behind both planes
    unsigned xdelta, wdelta;
    const void* zpad;
    const float* bias;                            // [Cout] or nullptr
    const float* sc;                              // nullptr or {scale, 1/scale} of the input: the result is multiplied by sc[1]
    const float* sc2;                             // nullptr or {scale, 1/scale} of the packed weights: ... and by sc2[1]
    float* out;                                   // [B][Cout][L] fp32
    float* stats;                                 // nullptr or [Cout][ntiles][2]: sum, sum of squares per 256-position tile
    // eval mode (BatchNorm coefficients known before the conv): ev_a / ev_s = scale / shift of this layer's BatchNorm, xrows
    // (one float, pre-cleared: the layer's xb slot) receives max |a (v + bias) + s| over all blocks by atomic max - the activation
    // bound the consumers' operand scale derives from
    const float* ev_a; const float* ev_s; float* xrows;
    unsigned long long* trace;                    // nullptr, or [gridDim.x][64] shader-clock stamps of the block's phases (conv_h3d_kernel; tools/conv_trace.py)
    int B, Cout, C8, NCH, L, logL;
    int NS, NFS;                                  // conv_h3d_kernel: K stages of the pack, the first NFS of them full (TG taps of a chunk); the rest: K tail
    int ntiles, mblocks;                          // grid.x = ntiles * mblocks blocks
    int stages_per_split;                         // grid.y splits of the K stages (1 split: all of them)
    int epi_eval;                                 // the block's LDS table of per-row constants also holds ev_a / ev_s (eval mode, where it fits two blocks per CU)
    size_t split_stride;                          // floats between the partial results of two splits
    // EVOP (eval mode, an encoder level whose consumer is the next encoder level; conv_h3d_kernel<.., EVOP = true>): the epilogue also
    // writes the NEXT layer's split operand - LeakyReLU(a z + s) at the even samples, scaled, hi / lo - into op_h / op_l
    // [B][op_C8][L/2][8]: prep_h3_kernel<0> disappears for that layer.  Its power-of-two scale cannot wait for the layer's measured
    // maximum (that exists when the launch ends), so it derives from a RIGOROUS bound every block computes for itself before its first
    // tile: max_c |a_c| (||W_c||_1 xmax + |bias_c|) + |s_c| with xmax = the measured maximum of THIS layer's input (op_xmax) and
    // ||W_c||_1 the absolute row sums of the weights (op_wl1) - loose by 2^4 .. 2^6 against the measured maximum, which costs nothing: a
    // split value's error floor is 2^-25 in scaled units against maxima of 2^7 or more.  Block 0 publishes {scale, 1/scale} (op_xsc).
    wunet_half* op_h; wunet_half* op_l;
    const float* op_wl1; const float* op_xmax; float* op_xsc;
    int op_C8;
    // BSUM (training backward, a data gradient on un-split whole-row tiles; conv_h3d_kernel<.., BSUM = 1 | 2>): the rows this launch
    // writes are dL/d(conv input) = the gradients w.r.t. the ACTIVATIONS of the layers that produced that input.  BatchNorm backward
    // of a producer needs sum g and sum g * xhat over its positions with g = dL/d(activation) * LeakyReLU'(a z + s) - linear in
    // the data gradient, so the epilogue takes them from the accumulators while they are still in registers (it reads the producer's raw z
    // tile, 4 bytes per value) and pass_a_kernel - a whole pass over z and the data gradients - disappears for that producer:
    //   BSUM = 1, a decoder layer (model/unet_basic.py:93-95 backwards): rows < bs_c0 came through the x2 upsample of producer A
    //            (z at L/2): sum_q mask_q sum_p U[p,q] dx[p] = sum_p dx[p] (U mask)[p], i.e. the sums are taken against the UPSAMPLED
    //            mask and mask * xhat (ATen's fp32 coordinates, pairs ((p-1) >> 1, +1): wunet_plan.cpp up_pairs_regular) - the
    //            transposed upsample itself is left to gz_split_h3_kernel; rows >= bs_c0 are the skip of producer B (z at L, same
    //            position);
    //   BSUM = 2, an encoder-side layer (:86 backwards): every row is the decimated activation of producer A (z at 2L, even samples).
    // Per tile and row {sum g, sum g xhat, bound of max |g|, max |z - mean|} go to bs_part [rows][ntiles][4] (fixed order: the four
    // waves' parts are added in wave order, bn_finalize_bwd_tiles_kernel adds the tiles in order).  A producer whose pointers are null
    // is skipped (it keeps pass_a_kernel).
    const float* bs_z[2];                         // raw conv outputs of producers A, B
    const float* bs_cst[2];                       // their BatchNorm constants [C][4] = {a, s, mean, rstd} (bn_finalize_core writes the table)
    int bs_C[2];                                  // channel counts of the producers (row strides of their z)
    int bs_c0;                                    // first row of producer B
    float bs_up_scale;                            // BSUM = 1, 3: (float)(L/2 - 1) / (float)(L - 1)
    float* bs_part;
    // BSUM = 3 ("UPT", training backward of a decoder layer, un-split whole-row tiles): the rows < bs_c0 - the gradient w.r.t. the UPSAMPLED
    // half of the conv input (model/unet_basic.py:93 backwards) - leave the kernel already pulled back through the x2 upsample: the transpose
    // of ATen's upsample_linear1d (fp32 coordinates, pass_a_kernel<A_UP>'s arithmetic in its order) is applied to the accumulators, a lane's
    // four outputs p0 .. p0 + 3 making the two inputs p0 / 2, p0 / 2 + 1 with one value from each neighbour lane (a DPP move; across the
    // four waves through LDS), and uh_out [B][bs_c0][L/2] is written instead of those rows of `out` - half the bytes written here, half the
    // bytes read by the gradient assembly, which becomes a plain elementwise pass.  What a tile owes its neighbour TILES (the first / last
    // input of a tile gets one term from the tile before / behind it) goes to uh_spill [2][bs_c0][ntiles]: [0] = the term for the next
    // tile's first input, [1] = the term for the previous tile's last input; the reader adds them (2 of 128 positions).
    float* uh_out; float* uh_spill;
};

// ---------------------------------------------------------------------------- weight gradient
// dW[co][ci][t] = sum_{b,p} g_z[b][co][p] * x[b][ci][p + t - PAD] as a GEMM over positions (K) on the fp16 split:
// A = g_z (rows co), B = x with one n-tile = 16 input channels at ONE tap, so the tap shift is uniform per MFMA.
// Both operands come from the split layout [B][C8][L][8] (the one the conv kernels use) by straight 16-byte copies
// into LDS images [hi|lo][channel group][position][8]; an MFMA operand (8 consecutive positions of one channel) is two
// transposed reads (ds_read_b64_tr_b16) of that position-major image.  Plane strides are 4 mod 16 pieces: the 32 lanes
// of a transposed read then cover all 64 banks.  The tap shift is just a row offset of the transposed read.
//   TAPS = 15: block = M_REP*16 co x 32 ci; wave = (ci group, tap half of 8)  -> M_REP*8 accumulator tiles
//   TAPS =  5: block = M_REP*16 co x 64 ci; wave = ci group, all 5 taps        -> M_REP*5 accumulator tiles
// Split-K over gridDim.x like wgrad_mfma_kernel; the partial dW of a split is stored tile-major (see the epilogue) and
// reduced + scattered into [Cout][Cin][TAPS] by wgrad_h3_reduce_kernel.
struct WgradH3Args {
    const wunet_half* xh; const wunet_half* xl;   // [B][XC8][L][8]
    const wunet_half* gh; const wunet_half* gl;   // [B][GC8][L][8]  scaled g_z
    const float* sc;     // {scale, 1/scale} of g_z
    const float* sc2;    // {scale, 1/scale} of x
    float* part;
    int B, Cin, Cout, XC8, GC8, L, logL, chunks_per_split;
    size_t part_stride;                           // floats between two splits' partial results
};

template <int TAPS, int M_REP, int NSEG, int TP, bool BF = false>
__global__ __launch_bounds__(WUNET_THREADS, (M_REP <= 2 ? 2 : 1)) void wgrad_h3_kernel(WgradH3Args A)
{
    constexpr int GP = TP + 4;                      // TP positions per chunk; plane stride (pieces) of the g_z image, 4 mod 16
    constexpr int LSEG = TP / NSEG, SWX = LSEG + 16;        // NSEG > 1: L < TP, a chunk is NSEG items, each with its own halo rows
    constexpr int XROWS = NSEG == 1 ? TP + 20 : NSEG * SWX; // x rows staged per plane (8 halo + samples + halo)
    constexpr int XPOS = ((XROWS + 11) / 16) * 16 + 4;      // plane stride of the x image (4 mod 16)
    constexpr int WG = TAPS == 15 ? 2 : 4;          // ci groups of 16 per block
    constexpr int TW = TAPS == 15 ? 8 : 5;          // taps per wave
    constexpr int OB = TAPS == 15 ? 1 : 8 - TAPS / 2;    // funnel offset of the wave's first tap
    constexpr int CIB = WG * 16, XG = CIB / 8, GG = M_REP * 2;
    constexpr int NPL = BF ? 1 : 2;                 // operand planes: hi (+ lo); BF: one bf16 word per value
    constexpr int GPC = NPL * GG * TP;              // pieces staged per chunk: g_z, x
    constexpr int XPC = NPL * XG * XROWS;
    constexpr int GIT = (GPC + WUNET_THREADS - 1) / WUNET_THREADS;
    constexpr int XIT = (XPC + WUNET_THREADS - 1) / WUNET_THREADS;
    WUNET_DYN_SMEM(smem);
    wunet_half* gs = reinterpret_cast<wunet_half*>(smem);                      // [hi|lo][GG][GP][8]
    wunet_half* xs = gs + NPL * GG * GP * 8;                                     // [hi|lo][XG][XPOS][8] (+ slack behind it)
    static_assert(XROWS <= XPOS, "x image");

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, i16 = lane & 15;
    const int grp = TAPS == 15 ? wave >> 1 : wave;
    const int t0 = TAPS == 15 ? (wave & 1) * 8 : 0;
    const int co0 = blockIdx.z * M_REP * 16, ci0 = blockIdx.y * CIB;
    const int L = A.L;
    const long long nchunks = ((long long)A.B * L + TP - 1) / TP;      // L = 64 and an odd batch: the last chunk is half empty
    const long long kbeg = (long long)blockIdx.x * A.chunks_per_split;
    long long kend = kbeg + A.chunks_per_split;
    if (kend > nchunks) kend = nchunks;

    // transposed-read bases (halfs): lane (j = i16>>2, cq = i16&3) feeds row j, channels 4cq..4cq+3 of its 16-lane group
    const int tr_row = (i16 >> 2) + q * 8, tr_pl = (i16 & 3) >> 1, tr_h = (i16 & 1) * 4;
    const int gbase = (tr_pl * GP + tr_row) * 8 + tr_h;
    // x rows: item (p / LSEG) starts at row item * SWX (its own 8 + 8 halo rows); a lane quarter's 8 positions stay in one item
    const int qrow = ((q * 8) / LSEG) * SWX + (q * 8) % LSEG;
    const int xbase = ((grp * 2 + tr_pl) * XPOS + (i16 >> 2) + qrow + t0) * 8 + tr_h;

    wunet_f4 acc[M_REP][TW];
#pragma unroll
    for (int mt = 0; mt < M_REP; ++mt)
#pragma unroll
        for (int tw = 0; tw < TW; ++tw) acc[mt][tw] = wunet_f4{0.f, 0.f, 0.f, 0.f};

    wunet_h8 greg[GIT], xreg[XIT];
#define WUNET_WH3_PREFETCH(K_)                                                                                    \
    {                                                                                                             \
        const long long n0_ = (K_) * TP;                                                                          \
        const int b_ = (int)(n0_ >> A.logL), l0_ = (int)(n0_ & (L - 1));                                          \
        _Pragma("unroll") for (int it = 0; it < GIT; ++it) {                                                      \
            const int f_ = tid + it * WUNET_THREADS;                                                              \
            const int pl_ = f_ / TP, pos_ = f_ - pl_ * TP;                 /* plane = which * GG + group */       \
            const int c8_ = (co0 >> 3) + (pl_ % GG);                                                              \
            const int p_ = l0_ + pos_;                                     /* L = 64: the second item of the chunk */ \
            const bool ok_ = f_ < GPC && c8_ < A.GC8 && b_ + (p_ >> A.logL) < A.B;                               \
            const wunet_half* src_ = pl_ >= GG ? A.gl : A.gh;                                                     \
            greg[it] = wunet_ldh8(src_ + (ok_ ? (((size_t)(b_ + (p_ >> A.logL)) * A.GC8 + c8_) * L + (p_ & (L - 1))) * 8 : 0)); \
        }                                                                                                         \
        _Pragma("unroll") for (int it = 0; it < XIT; ++it) {                                                      \
            const int f_ = tid + it * WUNET_THREADS;                                                              \
            const int pl_ = f_ / XROWS, pos_ = f_ - pl_ * XROWS;                                                  \
            const int c8_ = (ci0 >> 3) + (pl_ % XG);                                                              \
            const int sg_ = NSEG == 1 ? 0 : pos_ / SWX;                                                           \
            const int l_ = (NSEG == 1 ? l0_ : 0) - 8 + pos_ - sg_ * SWX;                                          \
            const bool ok_ = f_ < XPC && c8_ < A.XC8 && l_ >= 0 && l_ < L && b_ + sg_ < A.B;                      \
            const wunet_half* src_ = pl_ >= XG ? A.xl : A.xh;                                                     \
            xreg[it] = wunet_ldh8(src_ + (ok_ ? (((size_t)(b_ + sg_) * A.XC8 + c8_) * L + l_) * 8 : 0));          \
        }                                                                                                         \
    }
    if (kbeg < kend) WUNET_WH3_PREFETCH(kbeg)

    for (long long k = kbeg; k < kend; ++k) {
        const int l0 = (int)((k * TP) & (L - 1));
        __syncthreads();
#pragma unroll
        for (int it = 0; it < GIT; ++it) {
            const int f = tid + it * WUNET_THREADS;
            const int pl = f / TP, pos = f - pl * TP;
            const bool ok = (co0 >> 3) + (pl % GG) < A.GC8 && (int)((k * TP + pos) >> A.logL) < A.B;
            if (f < GPC) wunet_sth8(gs + ((size_t)pl * GP + pos) * 8, wunet_selh8(ok, greg[it]));
        }
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            const int f = tid + it * WUNET_THREADS;
            const int pl = f / XROWS, pos = f - pl * XROWS;
            const int sg = NSEG == 1 ? 0 : pos / SWX;
            const int l = (NSEG == 1 ? l0 : 0) - 8 + pos - sg * SWX;
            const int bb = (int)((k * TP) >> A.logL) + sg;
            const bool ok = (ci0 >> 3) + (pl % XG) < A.XC8 && l >= 0 && l < L && bb < A.B;
            if (f < XPC) wunet_sth8(xs + ((size_t)pl * XPOS + pos) * 8, wunet_selh8(ok, xreg[it]));
        }
        __syncthreads();
        if (k + 1 < kend) WUNET_WH3_PREFETCH(k + 1)
#pragma unroll
        for (int ks = 0; ks < TP / 32; ++ks) {
            wunet_h8 ah[M_REP], al[M_REP];
#pragma unroll
            for (int mt = 0; mt < M_REP; ++mt) {
                const wunet_half* p = gs + gbase + ((mt * 2) * GP + ks * 32) * 8;
                ah[mt] = wunet_ldtr8(p, p + 32);
                if (!BF) al[mt] = wunet_ldtr8(p + GG * GP * 8, p + GG * GP * 8 + 32);
            }
            // the tap shift is a row offset of the transposed read (rows are 16-byte pieces: any alignment)
#pragma unroll
            for (int tw = 0; tw < TW; ++tw) {
                {                   // (k15: the second tap half computes a 16th, unused tap rather than branch around MFMAs)
                    const int krow = ((ks * 32) / LSEG) * SWX + (ks * 32) % LSEG;                   // first x row of this K step
                    const wunet_half* p = xs + xbase + (krow + OB + tw) * 8;
                    const wunet_h8 bh = wunet_ldtr8(p, p + 32);
                    wunet_h8 bl = bh;
                    if (!BF) bl = wunet_ldtr8(p + XG * XPOS * 8, p + XG * XPOS * 8 + 32);
#pragma unroll
                    for (int mt = 0; mt < M_REP; ++mt) {
                        if (BF) acc[mt][tw] = wunet_mfma16b(ah[mt], bh, acc[mt][tw]);
                        else {
                            acc[mt][tw] = wunet_mfma16h(al[mt], bh, acc[mt][tw]);
                            acc[mt][tw] = wunet_mfma16h(ah[mt], bl, acc[mt][tw]);
                            acc[mt][tw] = wunet_mfma16h(ah[mt], bh, acc[mt][tw]);
                        }
                    }
                }
            }
        }
    }
#undef WUNET_WH3_PREFETCH

    // partial result, tile-major: [split][co block][ci block][wave][mt][tw][lane][4 rows] - every store is a contiguous
    // KiB per wave (the dW layout [co][ci][tap] would be 64 scattered dwords per store); wgrad_h3_reduce_kernel sums the
    // splits in this layout and scatters only the final dW
    const float inv = A.sc[1], inv2 = A.sc2[1];
    float* part = A.part + (size_t)blockIdx.x * A.part_stride
                + ((((size_t)blockIdx.z * gridDim.y + blockIdx.y) * WUNET_WAVES + wave) * (M_REP * TW)) * 256 + lane * 4;
#pragma unroll
    for (int mt = 0; mt < M_REP; ++mt)
#pragma unroll
        for (int tw = 0; tw < TW; ++tw) {
            wunet_f4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = acc[mt][tw][r] * inv * inv2;
            wunet_st4(part + (mt * TW + tw) * 256, o);
        }
}

// ---------------------------------------------------------------------------- weight gradient, DMA staging
// wgrad_h3_kernel for L >= 128 with its chunk staging done by the LDS-DMA engine (global_load_lds_dwordx4): the LDS
// images are straight copies of the split layout, so no staging registers, no VALU and no ds_write are needed, and
// with the freed registers and LDS the tiles are DOUBLE buffered: the DMA of chunk k+1 runs while the MFMAs of chunk k
// do, one barrier per chunk.  (This kernel runs one block per CU: nothing else hides its staging.)  Lanes whose piece
// lies outside the tensor (halo beyond the item, channel groups beyond C8) fetch from a 16-byte zero page.
struct WgradH3dArgs {
    const wunet_half* xh; const wunet_half* xl;
    const wunet_half* gh; const wunet_half* gl;
    const float* sc;          // {scale, 1/scale} of g_z
    const float* sc2;         // {scale, 1/scale} of x
    float* part;
    int B, Cin, Cout, XC8, GC8, L, logL, chunks_per_split;
    size_t part_stride;
    int cin_active;           // waves whose input channels start at or beyond it skip their MFMAs (= Cin; A/B switch: INT_MAX)
    // XCD-aware walk (xcd_walk != 0; the grid is then 1-D): the nblocks * mblocks blocks that stream the SAME position range (one K split) -
    // they re-read its g_z chunks once per input-channel block and its x chunks once per row block - get block ids 8 apart, i.e. the same
    // XCD (block b runs on XCD b % 8: observed placement, for speed only) next to each other in dispatch order, and share those chunks in
    // that XCD's L2; 8 consecutive splits interleave over the 8 XCDs.  Off: grid (ksplit, nblocks, mblocks), a split's blocks ksplit ids apart.
    int xcd_walk, ksplit, nblocks, mblocks;
};

// TP = 64 with DB: chunks of 64 positions, double buffered within the LDS budget of ONE 128-position buffer - two blocks per CU AND
// the DMA of chunk k+1 under the MFMAs of chunk k (the halo rows of the x image weigh 31 % instead of 16 %).
template <int TAPS, int M_REP, bool DB, bool BF = false, int TP = 128>
__global__ __launch_bounds__(WUNET_THREADS, ((DB && TP == 128) ? 1 : 2)) void wgrad_h3d_kernel(WgradH3dArgs A)
{
    constexpr int GP = TP + 4, XPOS = TP + 20;                    // plane strides (pieces), both 4 mod 16
    static_assert(TP == 128 || TP == 64, "chunk");
    constexpr int WG = TAPS == 15 ? 2 : 4;
    constexpr int TW = TAPS == 15 ? 8 : 5;
    constexpr int OB = TAPS == 15 ? 1 : 8 - TAPS / 2;
    constexpr int CIB = WG * 16, XG = CIB / 8, GG = M_REP * 2;
    constexpr int NPL = BF ? 1 : 2;                 // operand planes: hi (+ lo); BF: one bf16 word per value
    constexpr int GPCS = NPL * GG * GP;             // pieces of the g_z image (rows 128..131 of a plane are never read or written)
    constexpr int XPCS = NPL * XG * XPOS;           // pieces of the x image
    constexpr int GIT = (GPCS + WUNET_THREADS - 1) / WUNET_THREADS;
    constexpr int XIT = (XPCS + WUNET_THREADS - 1) / WUNET_THREADS;
    constexpr int BUF = (GPCS + XPCS + 8) * 8;      // halfs per buffer (+ slack behind the x image)
    WUNET_DYN_SMEM(smem);
    wunet_half* lds = reinterpret_cast<wunet_half*>(smem);            // [2][ g_z image | x image | slack ]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, i16 = lane & 15;
    const int grp = TAPS == 15 ? wave >> 1 : wave;
    const int t0 = TAPS == 15 ? (wave & 1) * 8 : 0;
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (A.xcd_walk) {
        const int nyz = A.nblocks * A.mblocks;
        const int grp8 = (int)blockIdx.x / (8 * nyz), rem = (int)blockIdx.x - grp8 * 8 * nyz;
        const int yz = rem >> 3;
        bx = grp8 * 8 + (rem & 7); by = yz % A.nblocks; bz = yz / A.nblocks;
        if (bx >= A.ksplit) return;                  // (the last group of 8 splits may be short)
    }
    const int co0 = bz * M_REP * 16, ci0 = by * CIB;
    const int L = A.L;
    const long long nchunks = ((long long)A.B * L) / TP;
    const long long kbeg = (long long)bx * A.chunks_per_split;
    long long kend = kbeg + A.chunks_per_split;
    if (kend > nchunks) kend = nchunks;

    const int tr_row = (i16 >> 2) + q * 8, tr_pl = (i16 & 3) >> 1, tr_h = (i16 & 1) * 4;
    const int gbase = (tr_pl * GP + tr_row) * 8 + tr_h;
    const int xbase = GPCS * 8 + ((grp * 2 + tr_pl) * XPOS + tr_row + t0) * 8 + tr_h;

    // per-thread source descriptors of its pieces (chunk independent): element offset of the (plane, row) inside one
    // batch item, or -1 for a piece that is never fetched; x rows also keep their sample offset for the halo test
    long long gsrc[GIT]; int gsel[GIT];
    long long xsrc[XIT]; int xrow[XIT], xsel[XIT];
#pragma unroll
    for (int it = 0; it < GIT; ++it) {
        const int f = tid + it * WUNET_THREADS;
        const int pl = f / GP, r = f - pl * GP;
        const int c8 = (co0 >> 3) + (pl % GG);
        gsel[it] = (f < GPCS && r < TP) ? ((c8 < A.GC8 ? 1 : 0) | (pl >= GG ? 2 : 0)) : -1;       // bit 0: real data, bit 1: lo array
        gsrc[it] = ((long long)c8 * L + r) * 8;
    }
#pragma unroll
    for (int it = 0; it < XIT; ++it) {
        const int f = tid + it * WUNET_THREADS;
        const int pl = f / XPOS, r = f - pl * XPOS;
        const int c8 = (ci0 >> 3) + (pl % XG);
        xsel[it] = f < XPCS ? ((c8 < A.XC8 ? 1 : 0) | (pl >= XG ? 2 : 0)) : -1;
        xrow[it] = r - 8;
        xsrc[it] = ((long long)c8 * L + (r - 8)) * 8;
    }
    const wunet_half* const zero_ = reinterpret_cast<const wunet_half*>(wunet_zero16);
    // issue the DMA of chunk K_ into buffer BUF_
#define WUNET_WH3D_DMA(K_, BUF_)                                                                                  \
    {                                                                                                             \
        const long long n0_ = (K_) * TP;                                                                          \
        const int b_ = (int)(n0_ >> A.logL), l0_ = (int)(n0_ & (L - 1));                                          \
        wunet_half* dst_ = lds + (size_t)(BUF_) * BUF + (size_t)wave * 64 * 8;                                    \
        _Pragma("unroll") for (int it = 0; it < GIT; ++it) {                                                      \
            if (gsel[it] >= 0) {                                                                                  \
                const wunet_half* src_ = (gsel[it] & 2) ? A.gl : A.gh;                                            \
                const wunet_half* p_ = (gsel[it] & 1) ? src_ + (size_t)b_ * A.GC8 * L * 8 + gsrc[it] + (size_t)l0_ * 8 : zero_; \
                wunet_dma16(p_, dst_ + (size_t)it * WUNET_THREADS * 8);                                           \
            }                                                                                                     \
        }                                                                                                         \
        _Pragma("unroll") for (int it = 0; it < XIT; ++it) {                                                      \
            if (xsel[it] >= 0) {                                                                                  \
                const int l_ = l0_ + xrow[it];                                                                    \
                const wunet_half* src_ = (xsel[it] & 2) ? A.xl : A.xh;                                            \
                const wunet_half* p_ = ((xsel[it] & 1) && l_ >= 0 && l_ < L)                                      \
                                           ? src_ + (size_t)b_ * A.XC8 * L * 8 + xsrc[it] + (size_t)l0_ * 8 : zero_; \
                wunet_dma16(p_, dst_ + (size_t)GPCS * 8 + (size_t)it * WUNET_THREADS * 8);                        \
            }                                                                                                     \
        }                                                                                                         \
    }

    wunet_f4 acc[M_REP][TW];
#pragma unroll
    for (int mt = 0; mt < M_REP; ++mt)
#pragma unroll
        for (int tw = 0; tw < TW; ++tw) acc[mt][tw] = wunet_f4{0.f, 0.f, 0.f, 0.f};

    // a wave whose 16 input channels all lie beyond Cin (the last channel block of a Cin that is not a multiple of 16 WG) leaves the
    // matrix pipe of its SIMD to the other resident block; its zero accumulators are stored like any other (the reduce kernel
    // drops those columns)
    const bool active = wunet_uniform(ci0 + grp * 16 < A.cin_active ? 1 : 0) != 0;
    if (DB && kbeg < kend) WUNET_WH3D_DMA(kbeg, 0)
    for (long long k = kbeg; k < kend; ++k) {
        const int cur = DB ? (int)((k - kbeg) & 1) : 0;
        if (!DB) {                                 // single buffer, two blocks per CU: the partner block computes while this DMA lands
            __syncthreads();
            WUNET_WH3D_DMA(k, 0)
        }
        wunet_dma_wait();
        __syncthreads();                           // chunk k has landed (vmcnt wait + barrier); nobody reads the other buffer any more
        if (DB && k + 1 < kend) WUNET_WH3D_DMA(k + 1, cur ^ 1)
        const wunet_half* gs = lds + (size_t)cur * BUF;
        // fragments are read ONE (K step, tap) ahead of the MFMAs that use them (hipcc sinks a fragment read to its first use and
        // waits for it with lgkmcnt(0): one LDS round trip per tap with nothing from this wave in the matrix pipe); the MFMAs of a
        // tap are issued pass-major (per accumulator the order lo*hi, hi*lo, hi*hi is unchanged: bit-identical results)
        constexpr int KS = TP / 32;
        if (!active) continue;                     // (it only took part in the staging)
        wunet_h8 ah[2][M_REP], al[2][M_REP], bh[2], bl[2];
#define WUNET_WH3D_LOAD_A(BUF_, KS_)                                                                              \
    _Pragma("unroll") for (int mt = 0; mt < M_REP; ++mt) {                                                        \
        const wunet_half* p = gs + gbase + ((mt * 2) * GP + (KS_) * 32) * 8;                                      \
        ah[BUF_][mt] = wunet_ldtr8(p, p + 32);                                                                    \
        if (!BF) al[BUF_][mt] = wunet_ldtr8(p + GG * GP * 8, p + GG * GP * 8 + 32);                               \
    }
#define WUNET_WH3D_LOAD_B(BUF_, KS_, TW_)                                                                         \
    {                                                                                                             \
        const wunet_half* p = gs + xbase + ((KS_) * 32 + OB + (TW_)) * 8;                                         \
        bh[BUF_] = wunet_ldtr8(p, p + 32);                                                                        \
        if (!BF) bl[BUF_] = wunet_ldtr8(p + XG * XPOS * 8, p + XG * XPOS * 8 + 32);                               \
    }
        WUNET_WH3D_LOAD_A(0, 0)
        WUNET_WH3D_LOAD_B(0, 0, 0)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int tw = 0; tw < TW; ++tw) {
                const int sb = (ks * TW + tw) & 1, ab = ks & 1;
                if (tw + 1 < TW) {
                    if (sb) { WUNET_WH3D_LOAD_B(0, ks, tw + 1) } else { WUNET_WH3D_LOAD_B(1, ks, tw + 1) }
                } else if (ks + 1 < KS) {
                    if (ab) { WUNET_WH3D_LOAD_A(0, ks + 1) } else { WUNET_WH3D_LOAD_A(1, ks + 1) }
                    if (sb) { WUNET_WH3D_LOAD_B(0, ks + 1, 0) } else { WUNET_WH3D_LOAD_B(1, ks + 1, 0) }
                }
                wunet_sched_fence();
                if (BF) {
#pragma unroll
                    for (int mt = 0; mt < M_REP; ++mt) acc[mt][tw] = wunet_mfma16b(ah[ab][mt], bh[sb], acc[mt][tw]);
                } else {
#pragma unroll
                    for (int mt = 0; mt < M_REP; ++mt) acc[mt][tw] = wunet_mfma16h(al[ab][mt], bh[sb], acc[mt][tw]);
#pragma unroll
                    for (int mt = 0; mt < M_REP; ++mt) acc[mt][tw] = wunet_mfma16h(ah[ab][mt], bl[sb], acc[mt][tw]);
#pragma unroll
                    for (int mt = 0; mt < M_REP; ++mt) acc[mt][tw] = wunet_mfma16h(ah[ab][mt], bh[sb], acc[mt][tw]);
                }
            }
        }
#undef WUNET_WH3D_LOAD_A
#undef WUNET_WH3D_LOAD_B
    }
#undef WUNET_WH3D_DMA

    const float inv = A.sc[1], inv2 = A.sc2[1];
    float* part = A.part + (size_t)bx * A.part_stride
                + ((((size_t)bz * (A.xcd_walk ? (unsigned)A.nblocks : gridDim.y) + by) * WUNET_WAVES + wave) * (M_REP * TW)) * 256 + lane * 4;
#pragma unroll
    for (int mt = 0; mt < M_REP; ++mt)
#pragma unroll
        for (int tw = 0; tw < TW; ++tw) {
            wunet_f4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = acc[mt][tw][r] * inv * inv2;
            wunet_st4(part + (mt * TW + tw) * 256, o);
        }
}
