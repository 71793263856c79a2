// Argument blocks and launches of the GEMM kernels, the weight packs and the side stream (see wunet_host.h).
#include "wunet_host.h"
#include "wunet_elementwise.h"
#include "wunet_h3_elem.h"

namespace wunet_host {

ConvArgs make_conv_args(const float* x, int kch, const float* wpk, const float* bias, float* out, float* stats,
                        int B, int rows, int L, int taps, const ConvCfg& c, size_t split_stride)
{
    ConvArgs a{};
    a.x = x; a.wpk = wpk; a.bias = bias; a.out = out; a.stats = stats;
    a.B = B; a.Cin = kch; a.Cout = rows; a.CinP = c.cp; a.L = L; a.logL = ilog2(L);
    const int tn = 64 * c.nrep;
    a.seg = L < tn ? L : tn;
    a.seg_shift = ilog2(a.seg);
    a.segw = a.seg + 16;
    const int rowlen = (tn / a.seg) * a.segw;
    int rp = rowlen;
    while (rp % 32 != 16) rp += 4;
    a.rowp = rp;
    a.r4 = rowlen / 4;
    a.sw4 = a.segw / 4;
    a.r4_magic = (unsigned)(((1u << 20) + a.r4 - 1) / a.r4);
    a.sw4_magic = (unsigned)(((1u << 20) + a.sw4 - 1) / a.sw4);
    a.kc_per_split = c.kcps;
    a.split_stride = split_stride;
    return a;
}

int launch_conv(int taps, const ConvArgs& a, const ConvCfg& c, hipStream_t st)
{
    const int kc = kc_of(taps);
    char pname[96];
    snprintf(pname, sizeof pname, "conv_mfma_kernel<%d, %d, %d>", taps, c.mrep, c.nrep);
    // algorithmic work: 2*B*L*rows*kch*taps flops; input read once + output written once
    const double posn = (double)a.B * a.L;
    prof_begin(st, pname, 2.0 * posn * a.Cout * a.Cin * taps, 4.0 * posn * (a.Cout + a.Cin));
    const size_t smem = ((size_t)kc * a.rowp + (size_t)c.mrep * kc * taps * 16) * sizeof(float);
    const dim3 grid(c.grid_x, c.mblocks, c.ksplit);
    const int rc = taps == 15 ? wunet_launch_conv_15(a, c.mrep, c.nrep, grid, smem, st)
                              : wunet_launch_conv_5(a, c.mrep, c.nrep, grid, smem, st);
    prof_end(st);
    if (rc != 0) return fail(WUNET_E_ARG, "no conv kernel for taps=%d mrep=%d nrep=%d", taps, c.mrep, c.nrep);
    return 0;
}
WgradArgs make_wgrad_args(const float* x, const float* g, float* part, int B, int Cin, int Cout, int L, int taps, int cps)
{
    WgradArgs a{};
    a.x = x; a.g = g; a.part = part; a.B = B; a.Cin = Cin; a.Cout = Cout; a.L = L; a.logL = ilog2(L);
    a.chunks_per_split = cps;
    a.seg = L < 64 ? L : 64;
    a.seg_shift = ilog2(a.seg);
    a.segw = a.seg + 16;
    const int nseg = 64 / a.seg;
    a.r4 = nseg * a.segw / 4;
    a.sw4 = a.segw / 4;
    int rp = a.r4 * 4;
    if (taps == 5) while (rp % 32 != 8) rp += 4;     // 3 ci rows per n-tile land on disjoint banks
    a.rowp = rp;
    a.r4_magic = (unsigned)(((1u << 20) + a.r4 - 1) / a.r4);
    a.sw4_magic = (unsigned)(((1u << 20) + a.sw4 - 1) / a.sw4);
    return a;
}

int launch_wgrad_any(int taps, const WgradArgs& a, const WgradCfg& w, hipStream_t st)
{
    const int cib = w.wsplit ? 1 : 4 * w.nw * (taps == 15 ? 1 : 3);
    const size_t smem = ((size_t)w.mrep * 16 * 66 + (size_t)cib * a.rowp) * sizeof(float);
    char pname[96];
    snprintf(pname, sizeof pname, "wgrad_mfma_kernel<%d, %d, %d, %d, %s>", taps, w.mrep, w.nw, w.xit, w.wsplit ? "true" : "false");
    const double posn = (double)a.B * a.L;
    prof_begin(st, pname, 2.0 * posn * a.Cout * a.Cin * taps, 4.0 * posn * (a.Cout + a.Cin));
    const dim3 grid(w.ksplit, w.nblocks, w.mblocks);
    int rc = taps == 15 ? wunet_launch_wgrad_15(a, w.mrep, w.nw, w.xit, w.wsplit, grid, smem, st)
                        : wunet_launch_wgrad_5(a, w.mrep, w.nw, w.xit, w.wsplit, grid, smem, st);
    prof_end(st);
    if (rc != 0) return fail(WUNET_E_ARG, "no wgrad kernel for taps=%d mrep=%d nw=%d xit=%d wsplit=%d", taps, w.mrep, w.nw, w.xit, w.wsplit);
    return 0;
}
// ---- fp16-split helpers
int launch_split(const float* x, wunet_half* hi, wunet_half* lo, const float* sc, const float* xb0, const float* xb1, float* xsc,
                 int B, int C, int L, hipStream_t st, int bf)
{
    const int c8 = (C + 7) / 8;
    const size_t n = (size_t)B * c8 * (L / 4);
    size_t blocks = (n + WUNET_THREADS - 1) / WUNET_THREADS;
    if (blocks > 16384) blocks = 16384;
    WUNET_LAUNCH(split_act_kernel, dim3((unsigned)blocks), dim3(WUNET_THREADS), 0, st, x, hi, lo, sc, xb0, xb1, xsc, B, C, c8, L, ilog2(L), bf);
    return 0;
}
// resident conv_h3d blocks the grid is sized for: two per CU (its launch bounds).  Emulator build only: WUNET_H3_GRID overrides
// (tests: a few blocks walk many work items) - the product reads no environment on a launch path
int h3_grid_cap()
{
#ifdef WUNET_EMU
    if (const char* e = getenv("WUNET_H3_GRID")) { const int v = atoi(e); if (v > 0) return v; }
    return 1 << 30;
#else
    static int cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!cus[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus[dev] = n;
    }
    return 2 * cus[dev];
#endif
}
int launch_conv_h3(int taps, int mrep, int mtiles_p, int sps, const wunet_half* xh, const wunet_half* xl, const wunet_half* wh,
                   const wunet_half* wl, const float* bias, const float* sc, const float* sc2, float* out, float* stats, int B, int rows,
                   int kch, int nch, int L, hipStream_t st, const void* zpad, const float* ev_a, const float* ev_s, float* xrows, int bf, int ntt,
                   const ConvH3OpOut* op, const ConvH3Bsum* bs)
{
    // conv_h3d_kernel addresses its DMA pieces as SGPR base + unsigned 32-bit offset: lo plane / lo pack behind the hi one, the zero pad
    // behind both, everything within 4 GiB of the hi arrays
    const long long xd = bf ? 0 : (const char*)xl - (const char*)xh, wd = bf ? 0 : (const char*)wl - (const char*)wh;
    const long long zd = (const char*)zpad - (const char*)xh, plane = (long long)B * ((kch + 7) / 8) * L * 16;
    if (!zpad || xd < 0 || wd < 0 || xd >= (1LL << 31) || wd >= (1LL << 31) || zd < plane + xd || zd + 256 >= (1LL << 32))
        return fail(WUNET_E_ARG, "conv_h3d operand layout: lo plane / lo pack / zero pad must lie behind the hi array within 4 GiB");
    char pname[96];
    const double posn = (double)B * L;
    ConvH3Args a{};
    a.xh = xh; a.xl = xl; a.wh = wh; a.wl = wl; a.bias = bias; a.sc = sc; a.sc2 = sc2; a.out = out; a.stats = stats;
    a.B = B; a.Cout = rows; a.C8 = (kch + 7) / 8; a.NCH = nch; a.L = L; a.logL = ilog2(L);
    a.ev_a = ev_a; a.ev_s = ev_s; a.xrows = xrows;
    if (op) { a.op_h = op->h; a.op_l = op->l; a.op_wl1 = op->wl1; a.op_xmax = op->xmax; a.op_xsc = op->xsc; a.op_C8 = op->C8; }
    a.xdelta = (unsigned)xd; a.wdelta = (unsigned)wd; a.zpad = zpad;
    const int nseg = L >= 256 ? 1 : 256 / L, nstage = h3_stage_count(kch, taps, ntt);
    a.NS = nstage; a.NFS = ntt ? (a.C8 / 4) * (taps / 5) : nstage;
    const int ksplit = (nstage + sps - 1) / sps;
    a.stages_per_split = sps; a.split_stride = (size_t)B * rows * L;

    a.ntiles = (int)((posn + 255) / 256); a.mblocks = mtiles_p / mrep;
    a.trace = g_h3_trace;
    if (op && (ksplit != 1 || nseg != 1 || !ev_a || !xrows)) return fail(WUNET_E_ARG, "conv_h3d EVOP needs an un-split whole-row eval launch");
    if (bs) {
        const bool upt = bs->kind == 3;
        if (ksplit != 1 || nseg != 1 || bf || op || bias || stats || xrows || (rows & 3) || (bs->c0 & 3) || bs->kind < 1 || bs->kind > 3 ||
            (bs->kind != 2) != (taps == 5) || (upt ? (!bs->uh_out || !bs->uh_spill || bs->c0 > rows) : !bs->part))
            return fail(WUNET_E_ARG, "conv_h3d BSUM / UPT needs an un-split whole-row data gradient with rows in groups of four");
        for (int k = 0; k < 2; ++k) { a.bs_z[k] = bs->z[k]; a.bs_cst[k] = bs->cst[k]; a.bs_C[k] = bs->C[k]; }
        a.bs_c0 = bs->kind == 2 ? rows : bs->c0; a.bs_up_scale = bs->up_scale; a.bs_part = bs->part;
        a.uh_out = bs->uh_out; a.uh_spill = bs->uh_spill;
    }
    // conv_h3d_kernel: x tile and W sub-tile by LDS-DMA, buffers re-filled under the MFMAs, persistent blocks (two per CU)
    snprintf(pname, sizeof pname, bf ? "conv_h3d_kernel<%d, %d, %d, bf16>" : op ? "conv_h3d_kernel<%d, %d, %d, evop>" :
             (bs && bs->kind == 3) ? "conv_h3d_kernel<%d, %d, %d, upt>" : bs ? "conv_h3d_kernel<%d, %d, %d, bsum>" : "conv_h3d_kernel<%d, %d, %d>", taps, mrep, nseg);
    // (BSUM: + the producers' z tiles the epilogue reads - 2 bytes per value through the upsample, 4 for a skip or decimated row)
    double bs_bytes = 0.0;
    if (bs && bs->kind == 1) bs_bytes = posn * ((bs->z[0] ? 2.0 * bs->c0 : 0.0) + (bs->z[1] ? 4.0 * (rows - bs->c0) : 0.0));
    if (bs && bs->kind == 2 && bs->z[0]) bs_bytes = posn * 4.0 * rows;
    if (bs && bs->kind == 3) bs_bytes = -2.0 * posn * bs->c0;          // (UPT: the upsampled rows leave at half the length)
    prof_begin(st, pname, 2.0 * posn * rows * kch * taps, (bf ? 2.0 * posn * kch + 4.0 * posn * rows : 4.0 * posn * (rows + kch)) + (op ? 2.0 * posn * rows : 0.0) + bs_bytes);
    // (eval: BatchNorm scale / shift beside the bias in the block's table of per-row constants unless that costs the second block of a CU)
    a.epi_eval = (xrows && nseg == 1 && 2 * h3d_smem(nseg, mrep, bf, mtiles_p, 1) <= 160u * 1024u) ? 1 : 0;
    const size_t smem = h3d_smem(nseg, mrep, bf, mtiles_p, a.epi_eval, bs ? 1 : 0);
    const int nitems = a.ntiles * a.mblocks;
    // resident blocks: two per CU, one for the 16-segment tile (96 KB); a K-split layer whose items x splits fit them runs one item per block
    int gx = h3_grid_cap() / 2 * h3d_blocks_per_cu(nseg, mrep, bf) / ksplit;
    if (gx < nitems) gx &= ~7;
    if (gx < 8) gx = 8;
    if (gx > nitems) gx = nitems;
    const dim3 grid((unsigned)gx, (unsigned)ksplit);
    const int rc = wunet_launch_conv_h3d(a, taps, mrep, nseg, grid, smem, st, bf != 0, op != nullptr, bs ? bs->kind : 0);
    prof_end(st);
    if (rc != 0) return fail(WUNET_E_ARG, "no conv_h3 kernel for taps=%d mrep=%d nseg=%d (rc %d)", taps, mrep, nseg, rc);
    return 0;
}

// conv_h3u_kernel: one block of 8 waves per CU walks the (tile, row block) items; the caller filled everything of `a` but the tiling
int launch_conv_h3u(const ConvH3uArgs& a0, int mrep, int mtiles_p, int kch, hipStream_t st)
{
    ConvH3uArgs a = a0;
    const double posn = (double)a.B * a.L;
    a.logL = ilog2(a.L);
    a.C8 = (kch + 7) / 8;
    a.NS = (a.C8 + 3) / 4;
    a.ntiles = (int)((posn + 255) / 256); a.mblocks = mtiles_p / mrep;
    a.trace = g_h3_trace ? g_h3_trace + (1 << 20) : nullptr;      // (behind conv_h3d_kernel's region of the caller's trace buffer; written by -DWUNET_H3U_TRACE builds only)
    const size_t smem = (size_t)(2 * 2 * 4 * 272 + 2 * 2 * mrep * 5 * 64) * 16 + (size_t)(WUNET_WAVES * mrep * 32 + 4 + 2 * a.C8 * 8 + 3 * mtiles_p * 16) * sizeof(float);   // (red, the coefficient table, the epilogue constants)
    const int nitems = a.ntiles * a.mblocks;
    int gx = h3_grid_cap() / 2;                  // one block per CU
    if (gx > nitems) gx = nitems;
    char pname[64];
    snprintf(pname, sizeof pname, "conv_h3u_kernel<%d>", mrep);
    // algorithmic bytes: the fp32 sources read once (the upsampled branch at half resolution), the result written once
    prof_begin(st, pname, 2.0 * posn * a.Cout * kch * 5.0, 4.0 * posn * (a.Cout + a.C0 * 0.5 + a.C1));
    const int rc = wunet_launch_conv_h3u(a, mrep, dim3((unsigned)gx), smem, st);
    prof_end(st);
    if (rc != 0) return fail(WUNET_E_ARG, "no conv_h3u kernel for mrep=%d (rc %d)", mrep, rc);
    return 0;
}

// wgrad_h3d_kernel (DMA-staged) runs this layer's weight gradient: whole chunks inside one item (L >= 128), both buffers within
// the 160 KB
size_t h3w_dma_smem(const LayerPlan& l, int bf)
{
    const int xg = l.taps == 15 ? 4 : 8, tp = l.h3w_tp, npl = bf ? 1 : 2;
    return (size_t)2 * (npl * (l.h3w_mrep * 2) * (tp + 4) + npl * xg * (tp + 20) + 8) * 16;
}
bool h3w_is_dma(const LayerPlan& l, int bf)
{
    const int tp = l.h3w_tp, nseg = l.L >= tp ? 1 : tp / l.L;
    return nseg == 1 && tp == 128 && h3w_dma_smem(l, bf) <= 160 * 1024 && l.h3w_mrep <= (l.taps == 15 ? 6 : 5);
}
int launch_wgrad_h3(const LayerPlan& l, const wunet_half* xh, const wunet_half* xl, const wunet_half* gh, const wunet_half* gl,
                    const float* sc, const float* sc2, float* part, int B, hipStream_t st, int bf)
{
    char pname[96];
    const double posn = (double)B * l.L;
    const int xg = l.taps == 15 ? 4 : 8, tp = l.h3w_tp, nseg = l.L >= tp ? 1 : tp / l.L;
    const dim3 grid(l.h3w_ksplit, l.h3w_nblocks, l.h3w_mblocks);
    const int npl = bf ? 1 : 2;
    const size_t smem_d = h3w_dma_smem(l, bf);
    int rc;
    if (h3w_is_dma(l, bf)) {
        WgradH3dArgs a{};
        a.xh = xh; a.xl = xl; a.gh = gh; a.gl = gl; a.sc = sc; a.sc2 = sc2; a.part = part; a.B = B; a.Cin = l.cin; a.Cout = l.cout;
        a.XC8 = (l.cin + 7) / 8; a.GC8 = (l.cout + 7) / 8; a.L = l.L; a.logL = l.logL;
        a.chunks_per_split = l.h3w_cps; a.part_stride = h3w_part_stride(l);
        a.cin_active = l.cin;
        a.ksplit = l.h3w_ksplit; a.nblocks = l.h3w_nblocks; a.mblocks = l.h3w_mblocks;
        // the blocks of one K split on one XCD where the planner found room (plan_h3_wgrad; WgradH3dArgs::xcd_walk)
        a.xcd_walk = l.h3w_xcd;
        snprintf(pname, sizeof pname, bf ? "wgrad_h3d_kernel<%d, %d, bf16>" : "wgrad_h3d_kernel<%d, %d>", l.taps, l.h3w_mrep);
        prof_begin(st, pname, 2.0 * posn * l.cout * l.cin * l.taps, 4.0 * posn * (l.cout + l.cin));
        // two blocks per CU with a single buffer where the registers allow it (two independent blocks hide each other's
        // waits: +18-28 % on those kernels), else one block with double-buffered tiles
        const bool db = !((l.taps == 5 && l.h3w_mrep <= 4) || (l.taps == 15 && l.h3w_mrep <= 3));
        const int nyz = l.h3w_nblocks * l.h3w_mblocks;
        const dim3 grid_x((unsigned)(((l.h3w_ksplit + 7) / 8) * 8 * nyz));
        rc = wunet_launch_wgrad_h3d(a, l.taps, l.h3w_mrep, db, a.xcd_walk ? grid_x : grid, db ? smem_d : smem_d / 2, st, bf != 0, tp);
    } else {
        WgradH3Args a{};
        a.xh = xh; a.xl = xl; a.gh = gh; a.gl = gl; a.sc = sc; a.sc2 = sc2; a.part = part; a.B = B; a.Cin = l.cin; a.Cout = l.cout;
        a.XC8 = (l.cin + 7) / 8; a.GC8 = (l.cout + 7) / 8; a.L = l.L; a.logL = l.logL;
        a.chunks_per_split = l.h3w_cps; a.part_stride = h3w_part_stride(l);
        snprintf(pname, sizeof pname, bf ? "wgrad_h3_kernel<%d, %d, bf16>" : "wgrad_h3_kernel<%d, %d>", l.taps, l.h3w_mrep);
        prof_begin(st, pname, 2.0 * posn * l.cout * l.cin * l.taps, 4.0 * posn * (l.cout + l.cin));
        const int xrows = nseg == 1 ? tp + 20 : nseg * (tp / nseg + 16), xpos = ((xrows + 11) / 16) * 16 + 4;
        const size_t smem = ((size_t)npl * (l.h3w_mrep * 2) * (tp + 4) + (size_t)npl * xg * xpos + 8) * 16;
        rc = wunet_launch_wgrad_h3(a, l.taps, l.h3w_mrep, nseg, tp, grid, smem, st, bf != 0);
    }
    prof_end(st);
    if (rc != 0) return fail(WUNET_E_ARG, "no wgrad_h3 kernel for taps=%d mrep=%d (rc %d)", l.taps, l.h3w_mrep, rc);
    return 0;
}

// Flipped / transposed weight packs of every data gradient (fp32 fragment packs of the fp32 layers, hi / lo packs of the split
// layers): two launches that read only the weights (and the forward's weight maxima)
int launch_backward_packs(wunet_ctx* c, const float* const* params, float* ws, hipStream_t st)
{
    const int NL = c->NL;
    PackTable tab{};
    int nd = 0;
    for (int i = 1; i < NL; ++i) {
        const LayerPlan& l = c->ly[i];
        if (l.h3d) continue;                       // data gradient on the split pack
        PackDesc& d = tab.d[nd++];
        d.w = params[4 * i]; d.dst = ws + c->wpkb_off + l.d_wpk;
        d.Cout = l.cout; d.Cin = l.cin; d.taps = l.taps; d.M = l.cin; d.CP = l.d.cp; d.mtiles = l.d.mtiles_p; d.transposed = 1;
    }
    if (nd > 0) {
        WUNET_LAUNCH(pack_weights_kernel, dim3(pack_gx(), nd), dim3(WUNET_THREADS), 0, st, tab);
        WUNET_CHECK_LAUNCH();
    }
    if (c->h3) {
        PackH3Table t3{};
        int n3 = 0;
        wunet_half* wh = reinterpret_cast<wunet_half*>(ws + c->h3_wb_hi);
        wunet_half* wl = reinterpret_cast<wunet_half*>(ws + c->h3_wb_lo);
        for (int i = 1; i < NL; ++i) {
            const LayerPlan& l = c->ly[i];
            if (!l.h3d) continue;
            PackH3Desc& d = t3.d[n3++];
            d.w = params[4 * i]; d.hi = wh + l.h3d_wpk; d.lo = wl + l.h3d_wpk;
            d.Cout = l.cout; d.Cin = l.cin; d.taps = l.taps; d.rows = l.cin; d.kch = l.cout; d.mtiles = l.h3d_mtp; d.nch = l.h3d_nch; d.transposed = 1;
            d.ntt = l.h3d_ntt; d.nfull = ((l.cout + 7) / 8) / 4; d.ns = h3_stage_count(l.cout, l.taps, l.h3d_ntt);
            d.wmax = ws + c->wmax_off + (size_t)WUNET_WMAX_PARTS * i;      // the forward's maxima: the weights have not changed since
            d.wsc = l.h3f ? nullptr : ws + c->fslot_off + (size_t)WUNET_SLOT_FLOATS * i + 2;
            d.bf = c->bf;
        }
        if (n3 > 0) {
            WUNET_LAUNCH(pack_h3_kernel, dim3(pack_gx(), n3), dim3(WUNET_THREADS), 0, st, t3);
            WUNET_CHECK_LAUNCH();
        }
    }
    return 0;
}

// the side stream + fork / join events of the CURRENT device (nullptr + error text on failure)
wunet_ctx::Side* side_for_current_device(wunet_ctx* c)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { fail(WUNET_E_RUNTIME, "hipGetDevice failed"); return nullptr; }
    std::lock_guard<std::mutex> g(c->side_lock);
    auto it = c->side.find(dev);
    if (it != c->side.end()) return &it->second;
    wunet_ctx::Side sd;
    // both streams live on one device: device-scope release at the fork / join events is enough (-1 % per step against the default
    // system-scope fence)
    const unsigned evf = hipEventDisableTiming | (unsigned)hipEventDisableSystemFence;
    if (hipStreamCreateWithFlags(&sd.stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&sd.ev_fork, evf) != hipSuccess ||
        hipEventCreateWithFlags(&sd.ev_join, evf) != hipSuccess ||
        hipEventCreateWithFlags(&sd.ev_pack, evf) != hipSuccess ||
        hipEventCreateWithFlags(&sd.ev_fpack, evf) != hipSuccess ||
        hipEventCreateWithFlags(&sd.ev_fpack2, evf) != hipSuccess) {
        fail(WUNET_E_RUNTIME, "cannot create the side stream of device %d", dev);
        return nullptr;
    }
    return &(c->side[dev] = sd);
}

}  // namespace wunet_host
