// conv_h3d_kernel: the GEMM of conv_h3_kernel (same LDS images, same fragments, same order of the MFMAs of every accumulator:
// bit-identical results) with the staging rebuilt around the LDS-DMA engine so that a block fetches while it computes:
//   * the x tile is staged by global_load_lds_dwordx4 as well - the de-interleave of the columns is done by the per-lane SOURCE
//     address (lane i's 16 bytes always land at wave base + 16 i); halo pieces outside the item and channel groups beyond C8 fetch
//     a 16-byte zero page.  No staging registers, no ds_write pass, no selects.
//   * every LDS buffer is handed back to the DMA engine as soon as its last fragment has been read, not at the end of the
//     stage: the 8 B-fragment pairs of a stage are read up front, so the x tile is free right after them and the NEXT x tile
//     (next chunk, or the first chunk of the block's next work item) lands under this stage's MFMAs; the A fragments are read one
//     tap ahead, so the W sub-tile is free before the last tap and the next one lands under the rest of the MFMAs.
//     The waits are hand-placed (s_waitcnt lgkmcnt(0) + s_barrier to release a buffer, vmcnt(0) + s_barrier at the top of a stage);
//     the DMA instructions are issued from inline asm so that hipcc's conservative "vmcnt(0) before any LDS read that follows an
//     LDS-DMA" cannot serialise the pipeline again.
//   * blocks are persistent: a block walks work items (position tile, row block) blockIdx.x, blockIdx.x + gridDim.x, ...; the first
//     stage of the next item is fetched under the last stage and the epilogue of the current one.
//   * K tail (A.NFS < A.NS): the channel groups left over after the last full chunk of 32 channels are not padded to a chunk (15 steps
//     for 1-3 groups of 8 channels with 3/4 .. 1/4 of every MFMA multiplying zeros) but run one TAIL stage each: the four K quarters
//     of an MFMA carry four TAP ranges of the one group (quarter q: taps q*NTT .. q*NTT + NTT - 1, NTT = ceil(TAPS / 4); the weight
//     pack holds zeros for taps >= TAPS), i.e. 4 steps per group for 15 taps, 2 for 5.  Same LDS images: the tail chunk's x tile is
//     staged like any other, a quarter just reads its columns NTT*q further right.
// The MFMAs of a tap are issued pass-major (all accumulators' lo*hi, then hi*lo, then hi*hi): 4*M_REP independent MFMAs between two
// MFMAs on the same accumulator instead of hipcc's back-to-back chains.
#pragma once
#include "wunet_h3.h"
// WUNET_ABL: ablation builds of tools/conv_ablation.sh (parts of the kernel compiled out, bits listed there); 0 in the product
#ifndef WUNET_ABL
#define WUNET_ABL 0
#endif


// (no tail stages in the two shapes at the register limit: they would spill)
#define WUNET_H3D_HAS_TAIL(M_REP_, NSEG_) ((M_REP_) < 4 && (NSEG_) < 16)
template <int TAPS, int M_REP, int NSEG, bool BF = false, bool EVOP = false, int BSUM = 0>
__global__ __launch_bounds__(WUNET_THREADS, 2) void conv_h3d_kernel(ConvH3Args A)
{
    static_assert(BSUM == 0 || (NSEG == 1 && !BF && !EVOP), "BSUM / UPT: a training data gradient on whole-row tiles");
    static_assert(BSUM != 3 || TAPS == 5, "UPT: a decoder layer");
    constexpr int PAD = TAPS / 2;
    constexpr int TG = 5;                         // taps per stage
    constexpr int NTG = TAPS / TG;
    constexpr int LSEG = 256 / NSEG, SW = LSEG + 16;
    constexpr int COLS = NSEG * SW, Q4 = COLS / 4;
    constexpr int NPL = BF ? 1 : 2;
    constexpr int XP = NPL * 4 * COLS;            // 16-byte pieces of the x tile
    constexpr int WPM = TG * 64;
    constexpr int WP = NPL * M_REP * WPM;         // pieces of the W sub-tile of a stage
    constexpr int WIT = (WP + WUNET_THREADS - 1) / WUNET_THREADS;
    static_assert(XP % 64 == 0 && WP % 64 == 0, "whole DMA instructions per wave");
    WUNET_DYN_SMEM(smem);
    wunet_half* xs = reinterpret_cast<wunet_half*>(smem);             // [hi|lo][4][COLS, de-interleaved][8]
    wunet_half* ws = xs + XP * 8;                                      // [hi|lo][M_REP][TG][4][16][8]
    float* red = reinterpret_cast<float*>(ws + WP * 8);                // [4 waves][M_REP * 16][2] statistics hand-over (BSUM: [4] per row), + 4 maxima
    // un-segmented tiles (the levels of 256 samples and more): [bias | eval a | eval s][ER], the epilogue's per-row constants, once per block
    // instead of one global round trip per work item (the eval parts only where A.epi_eval says they fit).  (BSUM: a data gradient has
    // no bias - no table; its LDS goes into the wider hand-over rows.)
    constexpr bool EPI_LDS = NSEG == 1 && BSUM == 0;
    constexpr int RW = BSUM ? 64 : 32;                                 // hand-over floats per 16 rows
    float* const epi = red + WUNET_WAVES * M_REP * RW + 4;
    const int ER = A.mblocks * M_REP * 16;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, i16 = lane & 15;
    const int L = A.L;

    // ---- per-thread source descriptors of its DMA pieces (independent of the work item).  A DMA instruction of a wave writes one
    // RUN of 64 consecutive pieces; the runs of the x image ([plane][4 groups][COLS, de-interleaved]: RP runs per plane) are dealt
    // so that the plane of an instruction is a compile-time constant: instruction it < NPL*XF covers runs 4 (it % XF) + wave of
    // plane it / XF, the XR left-over runs of every plane share one last instruction (wave -> plane wave / XR, run 4 XF + wave % XR).
    // Piece p of a plane is (channel group c8 = p / COLS, column col = 4 (w % Q4) + w / Q4 with w = p % COLS - the de-interleave).
    // Kept per piece: xoffb = bytes from 8 samples in front of the tile's first sample in the chunk's first channel group of the hi
    // plane (never negative: the DMA takes an SGPR base + an unsigned 32-bit offset per lane; the lo plane lies A.xdelta bytes behind
    // the hi plane and is part of the offset); per thread, one bit per
    // piece: m_live (the wave has a run in the shared instruction), m_lo / m_hi (the piece is left / right halo: outside the item
    // for the first / last tile of an item, always for NSEG > 1), m_pl (plane of the shared instruction); c8pk / segpk = the
    // piece's channel group (2 bits) / item of the tile (4 bits).
    constexpr int RP = 4 * COLS / 64, XF = RP / 4, XR = RP % 4;
    constexpr int XIT = NPL * XF + (XR ? 1 : 0);
    static_assert(XR * NPL <= 4, "left-over runs fit one instruction");
    static_assert(XIT <= 16, "descriptor bit fields");
    int xoffb[XIT];
    unsigned m_live = 0, m_lo = 0, m_hi = 0, m_pl = 0, c8pk = 0;
    unsigned long long segpk = 0;
    int xrun = 0;                                   // run (within its plane) of this wave in the last, shared instruction
#pragma unroll
    for (int it = 0; it < XIT; ++it) {
        int pl = it / XF, run = 4 * (it % XF) + wave;
        bool live = true;
        if (it == NPL * XF) { pl = wave / (XR ? XR : 1); run = 4 * XF + wave % (XR ? XR : 1); live = wave < XR * NPL; xrun = run; }
        const int p = run * 64 + lane, c8 = p / COLS, w = p % COLS;
        const int col = (WUNET_ABL & 1024) ? w : 4 * (w % Q4) + w / Q4, seg = col / SW, lrel = col - seg * SW - 8;      // (ablation 1024: no de-interleave - every DMA instruction 1 KiB contiguous, wrong columns)
        xoffb[it] = ((seg * A.C8 + c8) * L + lrel + 8) * 16 + (NPL > 1 && (pl & 1) ? (int)A.xdelta : 0);
        m_live |= (unsigned)live << it;
        m_lo |= (unsigned)(lrel < 0) << it;
        m_hi |= (unsigned)(lrel >= LSEG) << it;
        m_pl |= (unsigned)(pl & 1) << it;
        c8pk |= (unsigned)c8 << (2 * it);
        segpk |= (unsigned long long)seg << (4 * it);
    }
    // W sub-tile [hi|lo][M_REP][TG][64 pieces]: per (plane, m-tile) a contiguous run of TG * 64 = 320 pieces in the pack and in the LDS
    // image alike - one DMA instruction of all four waves (pieces 0 .. 255, lane offset 16 tid) plus one of a single wave (pieces
    // 256 .. 319); the run's start is a scalar.  Two per-lane offsets serve every W piece of the kernel.
    static_assert(WPM == 320, "W run = 256 + 64 pieces");
    const unsigned wo_all = (unsigned)tid * 16u, wo_rest = (256u + (unsigned)lane) * 16u;
    const wunet_lds_t xs_a = wunet_lds_addr(xs), ws_a = wunet_lds_addr(ws);
    const int wave_u = wunet_uniform(wave);
    // pieces of the tile that lie inside the tensor: bit per piece.  Halo pieces of an item's first / last tile (every halo piece
    // for NSEG > 1: each item of the tile carries its own zero padding), items beyond the batch, channel groups beyond C8 (only the
    // last chunk can have them: the test is skipped elsewhere)
#define WUNET_H3D_VALID(B_, L0_, CH_, OUT_)                                                                       \
    unsigned OUT_ = m_live;                                                                                       \
    {                                                                                                             \
        if (NSEG > 1 || (L0_) == 0) OUT_ &= ~m_lo;                                                                \
        if (NSEG > 1 || (L0_) + 256 >= L) OUT_ &= ~m_hi;                                                          \
        if (NSEG > 1 && (B_) + NSEG > A.B) {                                                                      \
            _Pragma("unroll") for (int it = 0; it < XIT; ++it)                                                    \
                if ((B_) + (int)((segpk >> (4 * it)) & 15) >= A.B) OUT_ &= ~(1u << it);                           \
        }                                                                                                         \
        if ((CH_) * 4 + 4 > A.C8) {                                                                               \
            _Pragma("unroll") for (int it = 0; it < XIT; ++it)                                                    \
                if ((CH_) * 4 + (int)((c8pk >> (2 * it)) & 3) >= A.C8) OUT_ &= ~(1u << it);                       \
        }                                                                                                         \
    }
    // A piece: SGPR base of the tile (+ stage) + the piece's stage-invariant 32-bit offset; a piece outside the tensor takes the offset of
    // the operand's 16-byte zero pad (A.zpad, behind both planes) instead.  No 64-bit address per lane, nothing for hipcc to hoist.
#define WUNET_H3D_X_PIECE(IT_, VALID_, BASE_, ZSEL_)                                                              \
    {                                                                                                             \
        const unsigned o_ = (((VALID_) >> (IT_)) & 1) && !(WUNET_ABL & 128) ? (unsigned)xoffb[IT_] : (ZSEL_);                             \
        const int run_ = (IT_) < NPL * XF ? 4 * ((IT_) % XF) + wave_u : wunet_uniform(xrun);                      \
        const int pl_ = (IT_) < NPL * XF ? (IT_) / XF : wunet_uniform((m_pl >> (IT_)) & 1);                       \
        wunet_dma16s(BASE_, o_, xs_a + (pl_ * 4 * COLS + run_ * 64) * 16);                                        \
    }
#define WUNET_H3D_ISSUE_X(B_, L0_, CH_)                                                                           \
    {                                                                                                             \
        const char* const base_ = reinterpret_cast<const char*>(A.xh) + (long long)((((size_t)(B_) * A.C8 + (CH_) * 4) * L + (L0_)) * 16) - 128; \
        const unsigned zsel_ = (unsigned)(reinterpret_cast<const char*>(A.zpad) - base_);                         \
        WUNET_H3D_VALID(B_, L0_, CH_, valid_)                                                                     \
        if (!(WUNET_ABL & (2 | 8))) _Pragma("unroll") for (int it = 0; it < XIT; ++it) {                                                      \
            if (it < NPL * XF) WUNET_H3D_X_PIECE(it, valid_, base_, zsel_)                                        \
            else if ((m_live >> it) & 1) WUNET_H3D_X_PIECE(it, valid_, base_, zsel_)                              \
        }                                                                                                         \
    }
#define WUNET_H3D_ISSUE_W(MT0_, ST_)                                                                              \
    {                                                                                                             \
        const char* const base_ = reinterpret_cast<const char*>(A.wh) + (long long)((((size_t)(MT0_) * A.NS + (ST_)) * TG * 64) * 16); \
        if (!(WUNET_ABL & (2 | 16))) _Pragma("unroll") for (int sub = 0; sub < NPL * M_REP; ++sub) {                                           \
            const char* const run_ = (WUNET_ABL & 256) ? reinterpret_cast<const char*>(A.wh) : base_ + (long long)(sub % M_REP) * A.NS * (TG * 64 * 16) + (sub >= M_REP ? (long long)A.wdelta : 0LL); \
            wunet_dma16s(run_, wo_all, ws_a + (sub * WPM + wave_u * 64) * 16);                                    \
            if (wave_u == (sub & 3)) wunet_dma16s(run_, wo_rest, ws_a + (sub * WPM + 256) * 16);                  \
        }                                                                                                         \
    }
    // work item v -> (position tile, row block): the row blocks of one tile on ONE XCD (they share its x tile in that L2);
    // gridDim.x is a multiple of 8 whenever a block walks more than one item, so the XCD of an item is the block's
#define WUNET_H3D_ITEM(V_, TILE_, MBLK_)                                                                          \
    if ((A.ntiles & 7) == 0) {                                                                                    \
        const int xcd_ = (V_) & 7, k_ = (V_) >> 3;                                                                \
        MBLK_ = k_ % A.mblocks;                                                                                   \
        TILE_ = (k_ / A.mblocks) * 8 + xcd_;                                                                      \
    } else {                                                                                                      \
        MBLK_ = (V_) % A.mblocks;                                                                                 \
        TILE_ = (V_) / A.mblocks;                                                                                 \
    }
#define WUNET_H3D_STAMP(K_)                                                                                       \
    if (A.trace) {                                                                                                       \
        const unsigned long long t_ = wunet_memtime();                                                            \
        if (tid == 0 && (K_) < 64) A.trace[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 64 + (K_)] = t_;                                  \
    }

    const int G = gridDim.x, nitems = A.ntiles * A.mblocks;
    int v = blockIdx.x;
    if (v >= nitems) return;
    int tile, mblk;
    WUNET_H3D_ITEM(v, tile, mblk)
    int b = (tile * 256) >> A.logL, l0 = NSEG == 1 ? ((tile * 256) & (L - 1)) : 0, mt0 = mblk * M_REP;
    const bool split = gridDim.y > 1;
    const int st_beg = blockIdx.y * A.stages_per_split;
    const int nstage = (st_beg + A.stages_per_split < A.NS) ? st_beg + A.stages_per_split : A.NS;
    constexpr bool KT = WUNET_H3D_HAS_TAIL(M_REP, NSEG);
    const int nfs = KT ? A.NFS : 0x7fffffff, tch = nfs / NTG;        // full stages (TG taps of a chunk of 4 channel groups); the tail stages' chunk
    constexpr int NTT = (TAPS + 3) / 4;            // steps of a tail stage
    // the epilogue's per-row constants once per block, not one global round trip per work item; the un-scale of operand and weights
    // (two powers of two: their product is exact) likewise
    for (int c = threadIdx.x; EPI_LDS && c < ER; c += WUNET_THREADS) {
        const bool in = c < A.Cout;
        epi[c] = (in && A.bias && gridDim.y == 1) ? A.bias[c] : 0.0f;
        if (A.epi_eval) { epi[ER + c] = in ? A.ev_a[c] : 0.0f; epi[2 * ER + c] = in ? A.ev_s[c] : 0.0f; }
    }
    const float inv12 = (A.sc ? A.sc[1] : 1.0f) * (A.sc2 ? A.sc2[1] : 1.0f);
    if (EPI_LDS) wunet_wait_lds_barrier();
    int stamp = 0;
    float amax_run = 0.0f;                          // eval mode: the block's running maximum of the activation bound over its work items
    float op_scale = 1.0f;                          // EVOP: scale of the consumer's operand (ConvH3Args: from the rigorous bound of this layer's activation)
    if (EVOP) {
        const float xm = A.op_xmax[0];
        float m = 0.0f;
        for (int c = tid; c < A.Cout; c += WUNET_THREADS)
            m = fmaxf(m, fabsf(A.ev_a[c]) * (A.op_wl1[c] * xm + (A.bias ? fabsf(A.bias[c]) : 0.0f)) + fabsf(A.ev_s[c]));
#pragma unroll
        for (int k = 1; k < 64; k <<= 1) m = fmaxf(m, wunet_shfl_xor(m, k));
        if (lane == 0) red[wave] = m;
        wunet_wait_lds_barrier();
        float inv_;
        wunet_pow2_scale(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])), op_scale, inv_);
        if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) { A.op_xsc[0] = op_scale; A.op_xsc[1] = inv_; }
        wunet_wait_lds_barrier();                     // (red is the epilogue's hand-over area)
    }
    WUNET_H3D_STAMP(stamp) ++stamp;
    if (st_beg < nstage) {
        WUNET_H3D_ISSUE_X(b, l0, (!KT || st_beg < nfs ? st_beg / NTG : tch))
        WUNET_H3D_ISSUE_W(mt0, st_beg)
    }

    // this lane's 4 positions wave*64 + 4*i16 .. +3 lie in ONE batch item of the tile: item lseg, first sample ll0
    const int lpos = wave * 64 + i16 * 4;
    const int lseg = lpos / LSEG, ll0 = lpos - lseg * LSEG;
    const int boff = (q * COLS + ((lseg * SW + ll0) >> 2)) * 8;
    const int boff_t = ((lseg * SW + ll0) >> 2) * 8;        // tail stages: + the group's plane
    const int aoff = (q * 16 + i16) * 8;

    for (;;) {
        const bool more = v + G < nitems;
        int ntile = 0, nmblk = 0;
        if (more) { WUNET_H3D_ITEM(v + G, ntile, nmblk) }
        const int nb = (ntile * 256) >> A.logL, nl0 = NSEG == 1 ? ((ntile * 256) & (L - 1)) : 0, nmt0 = nmblk * M_REP;

        wunet_f4 acc[M_REP][4];
#pragma unroll
        for (int mt = 0; mt < M_REP; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = wunet_f4{0.f, 0.f, 0.f, 0.f};
        for (int st = st_beg; st < nstage; ++st) {
            const bool tail = KT && st >= nfs;
            const int ch = tail ? tch : st / NTG, tg = tail ? st - nfs : st - ch * NTG;     // (tail: tg = the group of its chunk)
            const bool last = st + 1 == nstage, has_next = !last || more;
            const int nst = last ? st_beg : st + 1, nch = (!KT || nst < nfs) ? nst / NTG : tch;
            const bool x_next = has_next && (last || nch != ch);
            wunet_setprio(0);
            if (WUNET_ABL & 32) __syncthreads(); else
            wunet_wait_dma_barrier();             // this stage's x tile and W sub-tile have landed (every wave waited for its own pieces)
            WUNET_H3D_STAMP(stamp) ++stamp;
            // the two blocks of a CU take turns at the higher issue priority, stage by stage: at equal priority the older block of the pair
            // wins every tie, finishes its items at 0.78 of the kernel and leaves its partner alone on the CU for the rest (block end times
            // on the real-time counter, profiles/r4_conv_prio_alternate_ab.txt: bimodal 50 / 64 us -> one mode 57 - 65 us; -1.3 % kernel time)
            if ((st + (int)(blockIdx.x >= (unsigned)(G >> 1))) & 1) wunet_setprio(3); else wunet_setprio(1);
            // B fragments slide: with the interleaved column mapping fragment (n-tile nt, tap) is column 4*lane + nt + tap = F[nt + tap]
            // (tail stage: every quarter reads the plane of the stage's group and its taps start at q * NTT; NTT + 3 fragments, the rest
            // repeat the first).  ONE stage body for both kinds - a second copy of the MFMA block behind a branch cost 30 registers
#define WUNET_H3D_LOAD_F(TAILS_)                                                                                  \
    wunet_h8 fh[TG + 3], fl[TG + 3];                                                                              \
    {                                                                                                             \
        const int fb_ = ((TAILS_) && tail) ? boff_t + tg * COLS * 8 : boff;                                       \
        const int e0_ = ((TAILS_) && tail) ? q * NTT : tg * TG;                                                   \
        _Pragma("unroll") for (int e = 0; e < TG + 3; ++e) {                                                      \
            const int ec = e0_ + ((TAILS_) && e >= NTT + 3 && tail ? 0 : e) + 8 - PAD;                            \
            const int po = ((ec & 3) * Q4 + (ec >> 2)) * 8;                                                       \
            fh[e] = wunet_ldh8(xs + fb_ + po);                                                                    \
            if (!BF) fl[e] = wunet_ldh8(xs + 4 * COLS * 8 + fb_ + po);                                            \
        }                                                                                                         \
    }
            wunet_h8 ah[2][M_REP], al[2][M_REP];
#define WUNET_H3D_LOAD_A(BUF_, TL_)                                                                               \
    _Pragma("unroll") for (int mt = 0; mt < M_REP; ++mt) {                                                        \
        ah[BUF_][mt] = wunet_ldh8(ws + ((mt * TG + (TL_)) * 64) * 8 + aoff);                                      \
        if (!BF) al[BUF_][mt] = wunet_ldh8(ws + ((M_REP + mt) * TG + (TL_)) * 64 * 8 + aoff);                     \
    }
#define WUNET_H3D_PASS(WHICH_, BUF_, TL_)                                                                         \
    if (!(WUNET_ABL & 1) && !((WUNET_ABL & 512) && (WHICH_) == 0)) _Pragma("unroll") for (int mt = 0; mt < M_REP; ++mt)                                                          \
        _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) {                                                        \
            if (BF) { if ((WHICH_) == 2) acc[mt][nt] = wunet_mfma16b(ah[BUF_][mt], fh[(TL_) + nt], acc[mt][nt]); } \
            else if ((WHICH_) == 0) acc[mt][nt] = wunet_mfma16h(al[BUF_][mt], fh[(TL_) + nt], acc[mt][nt]);       \
            else if ((WHICH_) == 1) acc[mt][nt] = wunet_mfma16h(ah[BUF_][mt], fl[(TL_) + nt], acc[mt][nt]);       \
            else acc[mt][nt] = wunet_mfma16h(ah[BUF_][mt], fh[(TL_) + nt], acc[mt][nt]);                          \
        }
            // After the B fragments: the x tile is free once every wave holds them; the prefetch of the next step's A fragments is
            // ISSUED at the fence, not sunk to its first use; with the last A fragments of the stage in flight the W sub-tile is free
            // once every wave has them.  A tail stage runs the first NTT steps only.
            WUNET_H3D_LOAD_F(true)
            WUNET_H3D_LOAD_A(0, 0)
            if (x_next) {
                wunet_wait_lds_barrier();
                if (last) { WUNET_H3D_ISSUE_X(nb, nl0, nch) } else { WUNET_H3D_ISSUE_X(b, l0, nch) }
            }
#define WUNET_H3D_RELEASE_W()                                                                                     \
    {                                                                                                             \
        WUNET_H3D_STAMP(stamp)                                                                                    \
        wunet_wait_lds_barrier();                                                                                 \
        if (has_next) {                                                                                           \
            if (last) { WUNET_H3D_ISSUE_W(nmt0, nst) } else { WUNET_H3D_ISSUE_W(mt0, nst) }                       \
        }                                                                                                         \
    }
#define WUNET_H3D_STEP(TL_)                                                                                       \
    {                                                                                                             \
        constexpr int tl = (TL_);                                                                                 \
        if (tl + 1 < TG) {                                                                                        \
            if (tl & 1) { WUNET_H3D_LOAD_A(0, tl + 1) } else { WUNET_H3D_LOAD_A(1, tl + 1) }                      \
        }                                                                                                         \
        wunet_sched_fence();                                                                                      \
        if (tl & 1) { WUNET_H3D_PASS(0, 1, tl) } else { WUNET_H3D_PASS(0, 0, tl) }                                \
        if (tl == TG - 2) WUNET_H3D_RELEASE_W()                                                                   \
        else if (TG - 2 >= NTT && tl == NTT - 2) { if (tail) WUNET_H3D_RELEASE_W() }                              \
        if (tl & 1) { WUNET_H3D_PASS(1, 1, tl) WUNET_H3D_PASS(2, 1, tl) } else { WUNET_H3D_PASS(1, 0, tl) WUNET_H3D_PASS(2, 0, tl) } \
    }
            // (the steps a tail stage skips behind ONE uniform branch; TG - 2 >= NTT, 5 taps: the W sub-tile of a full stage is
            // released inside them, that of a tail stage at its step NTT - 2)
            static_assert(TG == 5 && (NTT == 4 || NTT == 2), "step list below");
            WUNET_H3D_STEP(0) WUNET_H3D_STEP(1)
            if (NTT == 4) {
                WUNET_H3D_STEP(2) WUNET_H3D_STEP(3)
                if (!tail) { WUNET_H3D_STEP(4) }
            } else if (!tail) {
                WUNET_H3D_STEP(2) WUNET_H3D_STEP(3) WUNET_H3D_STEP(4)
            }
#undef WUNET_H3D_STEP
#undef WUNET_H3D_RELEASE_W
            ++stamp;
            WUNET_H3D_STAMP(stamp) ++stamp;
        }
#undef WUNET_H3D_LOAD_A
#undef WUNET_H3D_LOAD_F
#undef WUNET_H3D_PASS

        // ---- epilogue (conv_h3_kernel's): un-scale, bias, store, BN statistics of the bias-free conv; a K split stores its
        // bias-free partial sum (statistics then come from the reduce kernel).  The DMAs of the next item's first stage are in flight.
        wunet_setprio(0);
        float* outp = A.out + (size_t)blockIdx.y * A.split_stride;
        const int bo = b + lseg;
        float amax = 0.0f;
        // the per-row constants in ONE batch of loads: loaded row by row where they are used, each load was waited for with
        // vmcnt(0) - i.e. together with the previous row's store - and the 4*M_REP serialised round trips were a third of a shallow
        // layer's block time (phase stamps, tools/conv_bench.py --trace)
        float bvs[M_REP][4];
#pragma unroll
        for (int mt = 0; mt < M_REP; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = (mt0 + mt) * 16 + q * 4 + r;
                bvs[mt][r] = EPI_LDS ? epi[co] : (A.bias && !split && co < A.Cout) ? A.bias[co] : 0.0f;
            }
        // row (mt, r) of this lane: prow + (mt * 16 + r) * L  (uniform strides: no 64-bit multiply per row)
        float* const prow = outp + ((size_t)bo * A.Cout + mt0 * 16 + q * 4) * L + (l0 + ll0);
        const bool want_stats = A.stats && !split;
        const bool full = (mt0 + M_REP) * 16 <= A.Cout && (NSEG == 1 || b + NSEG <= A.B);     // no row / item of the tile outside the tensor
        // STATS_: Sigma, Sigma^2 of the bias-free conv per row; GUARD_: rows beyond Cout / items beyond B exist; EVAL_: the activation
        // bound of eval mode.  Copies of the loop behind uniform branches: the un-guarded, statistics-free one (data gradients, K
        // splits) is half the instructions
#define WUNET_H3D_ROWS(STATS_, GUARD_, EVAL_)                                                                     \
    float eas[M_REP][4], ess[M_REP][4];                                                                           \
    if (EVAL_) {                                                  /* one batch of loads, like the biases */      \
        _Pragma("unroll") for (int mt = 0; mt < M_REP; ++mt)                                                      \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                       \
                const int co = (mt0 + mt) * 16 + q * 4 + r;                                                       \
                eas[mt][r] = (EPI_LDS && A.epi_eval) ? epi[ER + co] : co < A.Cout ? A.ev_a[co] : 0.0f;            \
                ess[mt][r] = (EPI_LDS && A.epi_eval) ? epi[2 * ER + co] : co < A.Cout ? A.ev_s[co] : 0.0f;        \
            }                                                                                                     \
    }                                                                                                             \
    _Pragma("unroll") for (int mt = 0; mt < M_REP; ++mt) {                                                        \
        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};                                         \
        float ye[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};   /* EVOP: the activation at the lane's two even samples, rows r */ \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                           \
            const int co = (mt0 + mt) * 16 + q * 4 + r;                                                           \
            const float bv = bvs[mt][r];                                                                          \
            wunet_f4 o;                                                                                           \
            _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) {                                                    \
                const float vv = acc[mt][nt][r] * inv12;                                                     \
                if (STATS_) { s1[r] += vv; s2[r] = fmaf(vv, vv, s2[r]); }                                                    \
                o[nt] = vv + bv;                                                                                  \
            }                                                                                                     \
            if ((!(GUARD_) || (co < A.Cout && bo < A.B)) && (!(WUNET_ABL & 4) || o[0] == 123.456f)) {                                                         \
                wunet_st4(prow + (size_t)(mt * 16 + r) * L, o);                                                   \
                if (EVAL_) {                                                                                      \
                    const float ea = eas[mt][r], es = ess[mt][r];                                                 \
                    _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) amax = fmaxf(amax, fabsf(ea * o[nt] + es));  \
                    if (EVOP) { ye[0][r] = wunet_lrelu(ea * o[0] + es); ye[1][r] = wunet_lrelu(ea * o[2] + es); } \
                }                                                                                                 \
            }                                                                                                     \
        }                                                                                                         \
        if (EVOP && (EVAL_)) {         /* rows q*4 .. q*4+3 = half of the 16-byte piece of channel group 2 (mt0 + mt) + (q >> 1) */ \
            const int c8o = (mt0 + mt) * 2 + (q >> 1);                                                            \
            if (c8o < A.op_C8 && bo < A.B) {                                                                      \
                _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                   \
                    wunet_half hh[4], ll[4];                                                                      \
                    _Pragma("unroll") for (int r = 0; r < 4; ++r) wunet_split_h(op_scale * ye[j][r], hh[r], ll[r]); \
                    const size_t oo = ((((size_t)bo * A.op_C8 + c8o) << (A.logL - 1)) + (size_t)(((l0 + ll0) >> 1) + j)) * 8 + (q & 1) * 4; \
                    wunet_sth4(A.op_h + oo, hh);                                                                  \
                    wunet_sth4(A.op_l + oo, ll);                                                                  \
                }                                                                                                 \
            }                                                                                                     \
        }                                                                                                         \
        if (STATS_) {                                                                                             \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                       \
                s1[r] = wunet_row16_sum(s1[r]);                                                                   \
                s2[r] = wunet_row16_sum(s2[r]);                                                                   \
                if (i16 == 0) {                               /* per-wave sums of row mt*16 + q*4 + r -> LDS */   \
                    float* rp = red + ((wave * M_REP + mt) * 16 + q * 4 + r) * 2;                                 \
                    rp[0] = s1[r];                                                                                \
                    rp[1] = s2[r];                                                                                \
                }                                                                                                 \
            }                                                                                                     \
        }                                                                                                         \
    }
        if (BSUM == 3) {
            // ---- UPT (ConvH3Args::uh_out): the rows through the x2 upsample are stored pulled back to the producer's resolution
            // (per-row addresses below = one per-lane base + offsets that are the same in every lane, and the bases derive from values made
            //  opaque here: otherwise everything that does not depend on the work item is hoisted out of the loop over the items and lives
            //  through the K loop, whose registers are spoken for)
            int qe = q, pe = ll0;
            wunet_opaque(qe);
            wunet_opaque(pe);
            const int p0 = l0 + pe, Lh = L >> 1;
            float uw0[4], uw1[4], wp1, wn0;         // ATen's weights of the lane's outputs p0 .. p0 + 3, l1 of output p0 - 1, l0 of output p0 + 4
#pragma unroll
            for (int j = 0; j < 4; ++j) { int i0_, i1_; wunet_up_coord(p0 + j, Lh, A.bs_up_scale, i0_, i1_, uw0[j], uw1[j]); }
            { int i0_, i1_; float t_; wunet_up_coord(p0 > 0 ? p0 - 1 : 0, Lh, A.bs_up_scale, i0_, i1_, t_, wp1); }
            { int i0_, i1_; float t_; wunet_up_coord(p0 + 4 < L ? p0 + 4 : L - 1, Lh, A.bs_up_scale, i0_, i1_, wn0, t_); }
            // (output 0 of a row reads source 0 with weight 1, not the pair (-1, 0): its term goes to the lane's own first input)
            const float ca0 = p0 == 0 ? uw0[0] : uw1[0];
            constexpr int NRW = M_REP * 16;
            // [wave][row of the block][2]: the wave's first and last data-gradient value of the row.  Per-lane bases: the lane's own slot,
            // the previous wave's last value, the next wave's first
            float* const ex_own = red + (wave * NRW + qe * 4) * 2;
            const float* const ex_prev = red + ((wave > 0 ? wave - 1 : 0) * NRW + qe * 4) * 2 + 1;
            const float* const ex_next = red + ((wave < WUNET_WAVES - 1 ? wave + 1 : 0) * NRW + qe * 4) * 2;
#pragma unroll
            for (int mt = 0; mt < M_REP; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (i16 == 0) ex_own[(mt * 16 + r) * 2] = acc[mt][0][r] * inv12;
                    if (i16 == 15) ex_own[(mt * 16 + r) * 2 + 1] = acc[mt][3][r] * inv12;
                }
            wunet_wait_lds_barrier();
            const int rb0 = mt0 * 16 + qe * 4;      // the lane's first row of the block's first m-tile
            float* const hp0 = A.uh_out + ((size_t)bo * A.bs_c0 + rb0) * Lh + (p0 >> 1);
            float* const prow_e = outp + ((size_t)bo * A.Cout + rb0) * L + p0;
            float* const spa = A.uh_spill + (size_t)rb0 * A.ntiles + tile;                 // [0]: owed to the next tile's first input
            float* const spb = spa + (size_t)A.bs_c0 * A.ntiles;                            // [1]: owed to the previous tile's last input
            const bool edge_lo = wave == 0 && i16 == 0, edge_hi = wave == WUNET_WAVES - 1 && i16 == 15;
#pragma unroll
            for (int mt = 0; mt < M_REP; ++mt) {
                const int rb = rb0 + mt * 16;
                const bool live = rb < A.Cout && bo < A.B, up = rb < A.bs_c0;
                // two rows at a time: a lane makes two inputs per row, the lanes of a pair (i16, i16 ^ 1) four consecutive ones - the even lane
                // stores 16 bytes of the first row, the odd lane 16 bytes of the second (8-byte stores measured 11 % slower than the plain epilogue)
#pragma unroll
                for (int rp = 0; rp < 4; rp += 2) {
                    float ga[2], gb[2];
                    wunet_f4 ov[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int ro = mt * 16 + rp + h;      // row offset from the lane's base: the same in every lane
                        wunet_f4 o;
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt) o[nt] = acc[mt][nt][rp + h] * inv12;
                        ov[h] = o;
                        // the neighbours' values: output p0 - 1 (the previous lane's last) and p0 + 4 (the next lane's first); across waves from
                        // LDS, across tiles nothing - that term travels through uh_spill
                        float dp = wunet_row16_shr1(o[3]), dn = wunet_row16_shl1(o[0]);
                        if (i16 == 0) dp = wave > 0 ? ex_prev[ro * 2] : 0.0f;
                        if (i16 == 15) dn = wave < WUNET_WAVES - 1 ? ex_next[ro * 2] : 0.0f;
                        // input i receives, in ascending output j: l1(2i-1) d[2i-1] + l1(2i) d[2i] + l0(2i+1) d[2i+1] + l0(2i+2) d[2i+2]
                        float a_ = wp1 * dp;
                        a_ = fmaf(ca0, o[0], a_); a_ = fmaf(uw0[1], o[1], a_); a_ = fmaf(uw0[2], o[2], a_);
                        float b_ = uw1[1] * o[1];
                        b_ = fmaf(uw1[2], o[2], b_); b_ = fmaf(uw0[3], o[3], b_); b_ = fmaf(wn0, dn, b_);
                        ga[h] = a_; gb[h] = b_;
                    }
                    const bool odd = (i16 & 1) != 0;
                    const float ra = wunet_lane_swap1(odd ? ga[0] : ga[1]), rb_ = wunet_lane_swap1(odd ? gb[0] : gb[1]);
                    const int ro = mt * 16 + rp;
                    if (live && up) {
                        const wunet_f4 w = odd ? wunet_f4{ra, rb_, ga[1], gb[1]} : wunet_f4{ga[0], gb[0], ra, rb_};
                        wunet_st4(odd ? hp0 + (size_t)(ro + 1) * Lh - 2 : hp0 + (size_t)ro * Lh, w);
                        if (edge_lo) { spb[(size_t)ro * A.ntiles] = p0 == 0 ? 0.0f : uw0[0] * ov[0][0]; spb[(size_t)(ro + 1) * A.ntiles] = p0 == 0 ? 0.0f : uw0[0] * ov[1][0]; }
                        if (edge_hi) { spa[(size_t)ro * A.ntiles] = uw1[3] * ov[0][3]; spa[(size_t)(ro + 1) * A.ntiles] = uw1[3] * ov[1][3]; }
                    } else if (live && (!(WUNET_ABL & 4) || ov[0][0] == 123.456f)) {
                        wunet_st4(prow_e + (size_t)ro * L, ov[0]);
                        wunet_st4(prow_e + (size_t)(ro + 1) * L, ov[1]);
                    }
                    wunet_sched_fence();              // (one pair of rows at a time)
                }
            }
        } else
        if (BSUM) {
            // ---- data gradient + the producers' BatchNorm-backward sums (ConvH3Args::bs_*).  Two phases: (A) the sums, from the accumulators
            // and the producers' z rows - loads only, a ring of D rows in flight (hipcc counts loads and stores in ONE counter and waits
            // for vmcnt(0) as soon as both kinds are pending: a load issued behind a store would wait for that store to reach the L2);
            // (B) the stores of the data gradient itself, as in the plain epilogue, drained under the next item's K loop.
            // (everything below derives from values made opaque HERE: what does not depend on the work item would otherwise be hoisted out of the
            //  loop over the items and live through the K loop, whose registers are spoken for)
            int qe = q, pe = ll0;
            wunet_opaque(qe);
            wunet_opaque(pe);
            const int p0 = l0 + pe;                 // first of the lane's four positions in its row
            float uw0[4] = {1.f, 1.f, 1.f, 1.f}, uw1[4] = {0.f, 0.f, 0.f, 0.f};
            if (BSUM == 1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { int i0_, i1_; wunet_up_coord(p0 + j, L >> 1, A.bs_up_scale, i0_, i1_, uw0[j], uw1[j]); }
            }
            constexpr int NR = 4 * M_REP;                                         // rows of this lane: (m-tile, r)
#ifndef WUNET_BS_D
#define WUNET_BS_D 4
#endif
            constexpr int D = WUNET_BS_D < NR ? WUNET_BS_D : NR;                  // rows in flight (8 / 12 registers each)
            wunet_f4 zq[D], zr[D], cq[D];
            bool bs_on[D];
            int bs_pk[D];
#define WUNET_H3D_BS_LOAD(RR_)                                                                                    \
    {                                                                                                             \
        const int sl_ = (RR_) % D, mt_ = (RR_) >> 2, r_ = (RR_) & 3;                                              \
        const int rb_ = (mt0 + mt_) * 16 + qe * 4;                                                                \
        const int pk_ = (BSUM == 1 && rb_ >= A.bs_c0) ? 1 : 0;                                                    \
        const float* const zb_ = pk_ ? A.bs_z[1] : A.bs_z[0];                                                     \
        bs_pk[sl_] = pk_;                                                                                         \
        bs_on[sl_] = rb_ < A.Cout && bo < A.B && zb_ != nullptr;                                                  \
        if (bs_on[sl_]) {                                                                                         \
            const int ch_ = rb_ - (pk_ ? A.bs_c0 : 0) + r_;                                                       \
            const int ls_ = BSUM == 2 ? 2 * L : (pk_ ? L : (L >> 1));          /* the producer's row length */    \
            const float* const zp_ = zb_ + ((size_t)bo * (pk_ ? A.bs_C[1] : A.bs_C[0]) + ch_) * ls_               \
                                     + (BSUM == 2 ? 2 * p0 : pk_ ? p0 : (p0 >> 1) - 1);                           \
            cq[sl_] = wunet_ld4((pk_ ? A.bs_cst[1] : A.bs_cst[0]) + 4 * ch_);                                     \
            if (BSUM == 2) { zq[sl_] = wunet_ld4(zp_); zr[sl_] = wunet_ld4(zp_ + 4); }                            \
            else if (pk_) zq[sl_] = wunet_ld4(zp_);                                                               \
            else zq[sl_] = wunet_ld4u(zp_);                           /* sources (p0 >> 1) - 1 .. + 2 */          \
        }                                                                                                         \
    }
#pragma unroll
            for (int rr = 0; rr < D; ++rr) WUNET_H3D_BS_LOAD(rr)
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) {
                const int sl = rr % D, mt = rr >> 2, r = rr & 3;
                float s1 = 0.0f, s2 = 0.0f, mg = 0.0f, mz = 0.0f;
                if (bs_on[sl]) {
                    wunet_f4 o;
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) o[nt] = acc[mt][nt][r] * inv12;
                    const wunet_f4 k = cq[sl];            // a, s, mean, rstd of the row's producer channel
                    float zs[4];
                    if (BSUM == 2) { zs[0] = zq[sl][0]; zs[1] = zq[sl][2]; zs[2] = zr[sl][0]; zs[3] = zr[sl][2]; }
                    else { zs[0] = zq[sl][0]; zs[1] = zq[sl][1]; zs[2] = zq[sl][2]; zs[3] = zq[sl][3]; }
                    if (BSUM == 1 && !bs_pk[sl]) {
                        // through the x2 upsample: the lane's outputs p0 .. p0 + 3 read the source pairs (0, 1), (1, 2), (1, 2), (2, 3)
                        // of zs = sources (p0 >> 1) - 1 .. + 2; the row's first output reads source 0 with weight 1 (the clamped -1),
                        // the last one has weight 0 on the sample behind the row (up_pairs_regular): selects, never arithmetic
                        if (p0 == 0) zs[0] = zs[1];
                        if (p0 + 4 >= L) zs[3] = zs[2];
                        float m[4], e[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float d = zs[i] - k[2];
                            m[i] = (k[0] * zs[i] + k[1] > 0.0f) ? 1.0f : WUNET_SLOPE;
                            e[i] = m[i] * (d * k[3]);
                            mz = fmaxf(mz, fabsf(d));
                        }
                        const float mm[4] = {uw0[0] * m[0] + uw1[0] * m[1], uw0[1] * m[1] + uw1[1] * m[2],
                                             uw0[2] * m[1] + uw1[2] * m[2], uw0[3] * m[2] + uw1[3] * m[3]};
                        const float me[4] = {uw0[0] * e[0] + uw1[0] * e[1], uw0[1] * e[1] + uw1[1] * e[2],
                                             uw0[2] * e[1] + uw1[2] * e[2], uw0[3] * e[2] + uw1[3] * e[3]};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            s1 = fmaf(o[j], mm[j], s1);
                            s2 = fmaf(o[j], me[j], s2);
                            mg = fmaxf(mg, 2.0f * fabsf(o[j]));      // |g[i]| <= sum_p U[p, i] |dx[p]| <= 2 max |dx|
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float d = zs[j] - k[2];
                            const float g = (k[0] * zs[j] + k[1] > 0.0f) ? o[j] : WUNET_SLOPE * o[j];
                            s1 += g;
                            s2 = fmaf(g, d * k[3], s2);
                            mg = fmaxf(mg, fabsf(g));
                            mz = fmaxf(mz, fabsf(d));
                        }
                    }
                }
                if (rr + D < NR) WUNET_H3D_BS_LOAD(rr + D)
                s1 = wunet_row16_sum(s1);
                s2 = wunet_row16_sum(s2);
                mg = wunet_row16_max(mg);
                mz = wunet_row16_max(mz);
                if (i16 == 0) wunet_st4(red + ((wave * M_REP + mt) * 16 + qe * 4 + r) * 4, wunet_f4{s1, s2, mg, mz});
                wunet_sched_fence();
            }
#undef WUNET_H3D_BS_LOAD
            wunet_sched_fence();
            // (B) the data gradient itself
            if (full) { WUNET_H3D_ROWS(false, false, false) } else { WUNET_H3D_ROWS(false, true, false) }
            // one row of sums per tile and GEMM row: the four waves' parts in wave order
            wunet_wait_lds_barrier();
            if (tid < M_REP * 16) {
                const float* rp = red + tid * 4;
                wunet_f4 t = wunet_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int w = 0; w < WUNET_WAVES; ++w) {
                    const wunet_f4 v = wunet_ld4(rp + w * M_REP * 64);
                    t[0] += v[0]; t[1] += v[1]; t[2] = fmaxf(t[2], v[2]); t[3] = fmaxf(t[3], v[3]);
                }
                const int row = mt0 * 16 + tid;
                const float* const zb = (BSUM == 1 && row >= A.bs_c0) ? A.bs_z[1] : A.bs_z[0];
                if (row < A.Cout && zb != nullptr) wunet_st4(A.bs_part + ((size_t)row * A.ntiles + tile) * 4, t);
            }
        } else
        if (A.xrows) { WUNET_H3D_ROWS(false, true, true) }          // eval mode: the maximum of the activation bound instead of statistics
        else if (want_stats) { if (full) { WUNET_H3D_ROWS(true, false, false) } else { WUNET_H3D_ROWS(true, true, false) } }
        else { if (full) { WUNET_H3D_ROWS(false, false, false) } else { WUNET_H3D_ROWS(false, true, false) } }
#undef WUNET_H3D_ROWS
        if (A.xrows) {                            // eval: block maximum of the activation bound (max is order independent)
#pragma unroll
            for (int m = 1; m < 64; m <<= 1) amax = fmaxf(amax, wunet_shfl_xor(amax, m));
            float* rp = red + WUNET_WAVES * M_REP * 32;
            if (lane == 0) rp[wave] = amax;
            wunet_wait_lds_barrier();
            if (tid == 0) amax_run = fmaxf(amax_run, fmaxf(fmaxf(rp[0], rp[1]), fmaxf(rp[2], rp[3])));
        }
        // one statistics row per tile (256 positions): the four waves' sums are added in wave order
        if (A.stats && !split) {
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
        WUNET_H3D_STAMP(stamp) ++stamp;
        if (!more) break;
        v += G; tile = ntile; b = nb; l0 = nl0; mt0 = nmt0;
    }
    if (A.xrows && tid == 0) wunet_atomic_absmax(A.xrows, amax_run);      // ONE atomic per block into the layer's xb slot
#undef WUNET_H3D_ISSUE_X
#undef WUNET_H3D_ISSUE_W
#undef WUNET_H3D_ITEM
#undef WUNET_H3D_STAMP
}
