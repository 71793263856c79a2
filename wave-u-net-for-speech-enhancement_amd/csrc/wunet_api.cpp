// Host side of libwunet_hip.so: shape planning, workspace layout and kernel launches behind the
// C ABI declared in include/wunet_hip.h.  No torch types, no hidden device allocations on the hot
// path (the caller owns the workspace), nothing synchronises the stream.
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "wunet_elementwise.h"
#include "wunet_kernels.h"
#include "wunet_launch.h"
#include "wunet_hip.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define WUNET_CHECK_LAUNCH()                                                                   \
    do {                                                                                       \
        hipError_t e_ = hipGetLastError();                                                     \
        if (e_ != hipSuccess) return fail(WUNET_E_RUNTIME, "%s:%d HIP error: %s", __FILE__, __LINE__, hipGetErrorString(e_)); \
    } while (0)

// ---- optional per-launch profiler (HIP events on the launch stream), used by bench.py's roofline leg
struct ProfRec { std::string name; double flops; double bytes; hipEvent_t e0, e1; };
bool g_prof_on = false;
std::vector<ProfRec> g_prof;

#ifdef WUNET_EMU
inline void prof_begin(hipStream_t, const char*, double, double) {}
inline void prof_end(hipStream_t) {}
#else
inline void prof_begin(hipStream_t st, const char* name, double flops, double bytes)
{
    if (!g_prof_on) return;
    ProfRec r;
    r.name = name; r.flops = flops; r.bytes = bytes;
    hipEventCreate(&r.e0);
    hipEventCreate(&r.e1);
    hipEventRecord(r.e0, st);
    g_prof.push_back(r);
}
inline void prof_end(hipStream_t st)
{
    if (!g_prof_on) return;
    hipEventRecord(g_prof.back().e1, st);
}
#endif

int ilog2(long long v) { int r = 0; while ((1LL << r) < v) ++r; return r; }
bool is_pow2(long long v) { return v > 0 && (v & (v - 1)) == 0; }
size_t align64(size_t v) { return (v + 63) & ~(size_t)63; }
int round_up(int v, int m) { return (v + m - 1) / m * m; }

// m-tiles (16 output channels each) handled per block: minimise zero padding, prefer bigger blocks.
int pick_mrep(int mtiles, int max_rep)
{
    int best = 2, best_pad = 1 << 30;
    for (int r = max_rep; r >= 2; --r) {
        const int pad = round_up(mtiles, r) - mtiles;
        if (pad < best_pad) { best_pad = pad; best = r; }
    }
    return best;
}

TileGeom make_geom(int L, int TN, int pad, int extra_cols, int row_mod)
{
    TileGeom g;
    g.seg = L < TN ? L : TN;
    g.seg_shift = ilog2(g.seg);
    g.nseg = TN / g.seg;
    g.segw = g.seg + 2 * pad;
    g.rowlen = g.nseg * g.segw + extra_cols;
    int rp = g.rowlen;
    while (rp % 32 != row_mod) ++rp;
    g.rowp = rp;
    g.segw_magic = (unsigned)(((1u << 20) + g.segw - 1) / g.segw);
    return g;
}

enum { LK_RAW = 0, LK_DECIM = 1, LK_UPCAT = 2 };

struct WgradCfg { int mrep, nw, xit, wsplit, mblocks, nblocks, ksplit, cps, rows; };

struct LayerPlan {
    int cin, cout, taps, L, logL, kind;
    int src0, src1;      // producer layers (src0 = -1: network input)
    int c0;              // UPCAT: channels from the upsampled branch
    // forward conv
    int f_mrep, f_nrep, f_mblocks, f_cinp, f_mtiles_p, f_grid_x, f_rows, f_ksplit, f_kcps;
    size_t f_wpk;
    // data gradient (rows = cin, K-channels = cout)
    int d_mrep, d_nrep, d_mblocks, d_cp, d_mtiles_p, d_grid_x, d_ksplit, d_kcps;
    size_t d_wpk;
    // weight gradient
    WgradCfg w;
    size_t xin;          // materialised activated conv input [B][cin][L] (layers >= 1), float offset
    // pass A
    int a_split;
    // workspace (float offsets)
    size_t z, a, s, mean, rstd, g, dx, k1, k2, k3;
};

}  // namespace

struct wunet_ctx {
    int n, ci, B, T, NL;
    std::vector<LayerPlan> ly;
    size_t stats_off, wpkf_off, spart_off, fwd_floats;
    size_t bpart_off, wgpart_off, wpkb_off, gh_off, hpart_off, gz_off, total_floats;
    int head_blocks;
};

namespace {

int kc_of(int taps) { return taps == 15 ? 4 : 12; }

int conv_nrep(int B, int L, int mblocks)
{
    const long long pos = (long long)B * L;
    return (L >= 256 && (pos / 256) * mblocks >= 256) ? 4 : 1;
}

// split-K over input-channel chunks when a layer has too few (position, m-block) tiles to fill 256 CUs.
// Returns the number of z-slices; *kcps = padded input channels per slice.
int plan_ksplit(int blocks, int cinp, int kc, int* kcps)
{
    const int nchunks = cinp / kc;
    *kcps = cinp;
    if (blocks >= 192 || nchunks < 4) return 1;
    const int want = (512 + blocks - 1) / blocks;
    int cps = (nchunks + want - 1) / want;
    if (cps < 2) cps = 2;
    const int ks = (nchunks + cps - 1) / cps;
    if (ks <= 1) return 1;
    *kcps = cps * kc;
    return ks;
}

SrcDesc layer_src(const wunet_ctx* c, int i, float* ws, const float* noisy)
{
    const LayerPlan& l = c->ly[i];
    SrcDesc d{};
    d.C = l.cin; d.L = l.L; d.logL = l.logL; d.C0 = l.cin; d.Lsrc0 = l.L; d.up_scale = 0.f;
    if (l.kind == LK_RAW) {
        d.p0 = noisy;
    } else if (l.kind == LK_DECIM) {
        const LayerPlan& p = c->ly[l.src0];
        d.p0 = ws + p.z; d.a0 = ws + p.a; d.s0 = ws + p.s; d.Lsrc0 = 2 * l.L;
    } else {
        const LayerPlan& p = c->ly[l.src0];
        const LayerPlan& k = c->ly[l.src1];
        d.p0 = ws + p.z; d.a0 = ws + p.a; d.s0 = ws + p.s;
        d.p1 = ws + k.z; d.a1 = ws + k.a; d.s1 = ws + k.s;
        d.C0 = l.c0; d.Lsrc0 = l.L / 2;
        d.up_scale = l.L > 1 ? (float)(d.Lsrc0 - 1) / (float)(l.L - 1) : 0.f;
    }
    return d;
}

SrcDesc gz_src(const float* g, const float* z, const float* k1, const float* k2, const float* k3, int C, int L)
{
    SrcDesc d{};
    d.p0 = g; d.a0 = k1; d.s0 = k2; d.p1 = z; d.a1 = k3;
    d.C = C; d.C0 = C; d.L = L; d.Lsrc0 = L; d.logL = ilog2(L);
    return d;
}

int launch_conv_any(int taps, int mode, const ConvArgs& a, int mrep, int nrep, dim3 grid, hipStream_t st)
{
    const int kc = kc_of(taps);
    char pname[96];
    snprintf(pname, sizeof pname, "conv_mfma_kernel<%d, %d, %d, %d>", taps, mode, mrep, nrep);
    // algorithmic work: 2*B*L*Cout*Cin*taps flops; the virtual input read once + the output written once
    const double posn = (double)a.B * a.src.L;
    prof_begin(st, pname, 2.0 * posn * a.Cout * a.src.C * taps, 4.0 * posn * (a.Cout + a.src.C));
    const size_t smem = ((size_t)kc * a.geo.rowp + (size_t)mrep * kc * taps * 16) * sizeof(float);
    int rc = -1;
    if (taps == 15 && mode == SRC_RAW) rc = wunet_launch_conv_15_0(a, mrep, nrep, grid, smem, st);
    else if (taps == 15 && mode == SRC_DECIM) rc = wunet_launch_conv_15_1(a, mrep, nrep, grid, smem, st);
    else if (taps == 5 && mode == SRC_UPCAT) rc = wunet_launch_conv_5_2(a, mrep, nrep, grid, smem, st);
    else if (taps == 15 && mode == SRC_GZ) rc = wunet_launch_conv_15_3(a, mrep, nrep, grid, smem, st);
    else if (taps == 5 && mode == SRC_GZ) rc = wunet_launch_conv_5_3(a, mrep, nrep, grid, smem, st);
    else if (taps == 5 && mode == SRC_RAW) rc = wunet_launch_conv_5_0(a, mrep, nrep, grid, smem, st);
    prof_end(st);
    if (rc != 0) return fail(WUNET_E_ARG, "no conv kernel for taps=%d mode=%d mrep=%d nrep=%d", taps, mode, mrep, nrep);
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

int mode_of(int kind) { return kind == LK_RAW ? SRC_RAW : (kind == LK_DECIM ? SRC_DECIM : SRC_UPCAT); }

WgradCfg plan_wgrad(int B, int L, int cin, int cout, int taps)
{
    WgradCfg w{};
    const int mt = (cout + 15) / 16;
    const int nt = taps == 15 ? cin : (cin + 2) / 3;         // n-tiles of 16 (ci,tap) columns
    const long long chunks = ((long long)B * L + 63) / 64;
    const bool big = L >= 64;
    if (taps == 15 && cin == 1) {
        // encoder[0]: a single n-tile - the four waves split the K steps instead
        w.wsplit = 1; w.nw = 1; w.xit = 1;
        w.mrep = pick_mrep(mt, 6);
        w.mblocks = round_up(mt, w.mrep) / w.mrep;
        w.nblocks = 1;
    } else {
        const int nws[2] = {taps == 15 ? 6 : 2, (taps == 5 && big) ? 6 : 0};
        long long best = -1; int best_area = 0;
        for (int k = 0; k < 2; ++k) {
            const int nw = nws[k];
            if (!nw) continue;
            for (int mr = 2; mr <= 6; ++mr) {
                if (mr * nw > 36) continue;
                const long long padded = (long long)round_up(mt, mr) * round_up(nt, 4 * nw);
                if (best < 0 || padded < best || (padded == best && mr * nw > best_area)) {
                    best = padded; best_area = mr * nw; w.mrep = mr; w.nw = nw;
                }
            }
        }
        w.wsplit = 0;
        w.mblocks = round_up(mt, w.mrep) / w.mrep;
        w.nblocks = round_up(nt, 4 * w.nw) / (4 * w.nw);
        w.xit = big ? (w.nw == 6 && taps == 5 ? 6 : 2) : 8;
    }
    long long want = 1024 / ((long long)w.mblocks * w.nblocks);
    if (want < 1) want = 1;
    long long ks = 1;
    while (ks * 2 <= want && ks * 2 <= chunks) ks *= 2;
    w.ksplit = (int)ks;
    w.cps = (int)((chunks + ks - 1) / ks);
    w.rows = w.ksplit * (w.wsplit ? 4 : 1);
    return w;
}

}  // namespace

extern "C" {

const char* wunet_last_error(void) { return g_err.c_str(); }

int wunet_create(int n_layers, int channels_interval, int batch, int length, wunet_ctx** out)
{
    if (!out) return fail(WUNET_E_ARG, "out is null");
    if (n_layers < 1 || 2 * n_layers + 1 > WUNET_MAX_CONV_LAYERS) return fail(WUNET_E_ARG, "n_layers=%d unsupported (1..16)", n_layers);
    if (channels_interval < 1 || batch < 1) return fail(WUNET_E_ARG, "bad channels_interval/batch");
    if (!is_pow2(length) || (length >> n_layers) < 4)
        return fail(WUNET_E_ARG, "length=%d unsupported: must be a power of two with length >> n_layers >= 4", length);
    const int n = n_layers, ci = channels_interval, B = batch, T = length;
    if ((long long)B * (2 * n + 1) * ci * T >= (1LL << 32)) return fail(WUNET_E_ARG, "tensor too large for 32-bit offsets");

    wunet_ctx* c = new wunet_ctx();
    c->n = n; c->ci = ci; c->B = B; c->T = T; c->NL = 2 * n + 1;
    c->ly.resize(c->NL);
    for (int i = 0; i < c->NL; ++i) {
        LayerPlan& l = c->ly[i];
        if (i < n) {            // model/unet_basic.py:38-39
            l.cin = i == 0 ? 1 : i * ci; l.cout = (i + 1) * ci; l.taps = 15; l.L = T >> i;
            l.kind = i == 0 ? LK_RAW : LK_DECIM; l.src0 = i - 1; l.src1 = -1; l.c0 = l.cin;
        } else if (i == n) {    // :52-57
            l.cin = l.cout = n * ci; l.taps = 15; l.L = T >> n; l.kind = LK_DECIM; l.src0 = n - 1; l.src1 = -1; l.c0 = l.cin;
        } else {                // :59-70
            const int j = i - n - 1;
            l.cout = (n - j) * ci; l.taps = 5; l.L = T >> (n - 1 - j); l.kind = LK_UPCAT;
            l.src0 = i - 1; l.src1 = n - 1 - j;
            l.c0 = c->ly[i - 1].cout;
            l.cin = l.c0 + c->ly[l.src1].cout;
        }
        l.logL = ilog2(l.L);
    }
    size_t off = 0, wpk = 0, stats_max = 0, spart_max = 0;
    for (int i = 0; i < c->NL; ++i) {
        LayerPlan& l = c->ly[i];
        const int kc = kc_of(l.taps);
        const int mt = (l.cout + 15) / 16;
        l.f_mrep = pick_mrep(mt, 6);
        l.f_mtiles_p = round_up(mt, l.f_mrep);
        l.f_mblocks = l.f_mtiles_p / l.f_mrep;
        l.f_nrep = conv_nrep(B, l.L, l.f_mblocks);
        l.f_cinp = round_up(l.cin, kc);
        const int tn = 64 * l.f_nrep;
        l.f_grid_x = (int)(((long long)B * l.L + tn - 1) / tn);
        l.f_rows = l.f_grid_x * WUNET_WAVES;
        l.f_ksplit = plan_ksplit(l.f_grid_x * l.f_mblocks, l.f_cinp, kc, &l.f_kcps);
        if (l.f_ksplit > 1 && (size_t)l.f_ksplit * B * l.cout * l.L > spart_max) spart_max = (size_t)l.f_ksplit * B * l.cout * l.L;
        l.f_wpk = wpk;
        wpk += align64((size_t)l.f_mtiles_p * l.f_cinp * l.taps * 16);
        if ((size_t)l.f_rows * l.cout * 2 > stats_max) stats_max = (size_t)l.f_rows * l.cout * 2;
        {   // data-gradient tiling (rows = cin, K-channels = cout); planned here so the split buffer covers it
            const int dmt = (l.cin + 15) / 16;
            l.d_mrep = pick_mrep(dmt, 6);
            l.d_mtiles_p = round_up(dmt, l.d_mrep);
            l.d_mblocks = l.d_mtiles_p / l.d_mrep;
            l.d_nrep = conv_nrep(B, l.L, l.d_mblocks);
            l.d_cp = round_up(l.cout, kc);
            l.d_grid_x = (int)(((long long)B * l.L + 64 * l.d_nrep - 1) / (64 * l.d_nrep));
            l.d_ksplit = plan_ksplit(l.d_grid_x * l.d_mblocks, l.d_cp, kc, &l.d_kcps);
            if (i > 0 && l.d_ksplit > 1 && (size_t)l.d_ksplit * B * l.cin * l.L > spart_max) spart_max = (size_t)l.d_ksplit * B * l.cin * l.L;
        }
        l.z = off; off += align64((size_t)B * l.cout * l.L);
        l.a = off; off += align64(l.cout);
        l.s = off; off += align64(l.cout);
        l.mean = off; off += align64(l.cout);
        l.rstd = off; off += align64(l.cout);
    }
    c->stats_off = off; off += align64(stats_max);
    c->wpkf_off = off; off += align64(wpk);
    c->spart_off = off; off += align64(spart_max);
    c->fwd_floats = off;

    size_t wpkb = 0, bpart_max = 0, wgpart_max = 0, gz_max = 0;
    for (int i = 0; i < c->NL; ++i) {
        LayerPlan& l = c->ly[i];
        const int kc = kc_of(l.taps);
        l.g = off; off += align64((size_t)B * l.cout * l.L);
        l.dx = off; if (i > 0) off += align64((size_t)B * l.cin * l.L);
        l.k1 = off; off += align64(l.cout);
        l.k2 = off; off += align64(l.cout);
        l.k3 = off; off += align64(l.cout);
        (void)kc;
        l.d_wpk = wpkb;
        if (i > 0) wpkb += align64((size_t)l.d_mtiles_p * l.d_cp * l.taps * 16);
        l.w = plan_wgrad(B, l.L, l.cin, l.cout, l.taps);
        l.xin = off; if (i > 0) off += align64((size_t)B * l.cin * l.L);
        if ((size_t)B * l.cout * l.L > gz_max) gz_max = (size_t)B * l.cout * l.L;
        const size_t wg = (size_t)l.w.rows * l.cout * l.cin * l.taps;
        if (wg > wgpart_max) wgpart_max = wg;
        long long sp = ((long long)B * l.L) / 4096;
        l.a_split = (int)(sp < 1 ? 1 : (sp > 64 ? 64 : sp));
        if ((size_t)l.a_split * l.cout * 2 > bpart_max) bpart_max = (size_t)l.a_split * l.cout * 2;
    }
    c->bpart_off = off; off += align64(bpart_max);
    c->wgpart_off = off; off += align64(wgpart_max);
    c->wpkb_off = off; off += align64(wpkb);
    c->gh_off = off; off += align64((size_t)B * T);
    {
        long long hb = ((long long)B * T) / 2048;
        c->head_blocks = (int)(hb < 1 ? 1 : (hb > 1024 ? 1024 : hb));
    }
    c->hpart_off = off; off += align64((size_t)c->head_blocks * (ci + 2));
    c->gz_off = off; off += align64(gz_max);
    c->total_floats = off;
    *out = c;
    return WUNET_OK;
}

void wunet_destroy(wunet_ctx* ctx) { delete ctx; }

size_t wunet_workspace_bytes(const wunet_ctx* ctx, int with_backward)
{
    if (!ctx) return 0;
    return (with_backward ? ctx->total_floats : ctx->fwd_floats) * sizeof(float);
}

int wunet_num_conv_layers(const wunet_ctx* ctx) { return ctx ? ctx->NL : 0; }

int wunet_layer_info(const wunet_ctx* ctx, int layer, size_t* z_offset_floats, int* channels, int* length)
{
    if (!ctx || layer < 0 || layer >= ctx->NL) return fail(WUNET_E_ARG, "bad layer");
    if (z_offset_floats) *z_offset_floats = ctx->ly[layer].z;
    if (channels) *channels = ctx->ly[layer].cout;
    if (length) *length = ctx->ly[layer].L;
    return WUNET_OK;
}

int wunet_forward(wunet_ctx* c, const float* noisy, const float* const* params, float* const* running,
                  long long* const* nbt, int training, int save_for_backward, void* workspace, float* enhanced, void* stream)
{
    if (!c || !noisy || !params || !running || !nbt || !workspace || !enhanced) return fail(WUNET_E_ARG, "null argument");
    hipStream_t st = (hipStream_t)stream;
    float* ws = (float*)workspace;
    // 1. pack all forward weights into MFMA-fragment order (one launch)
    {
        PackTable tab{};
        for (int i = 0; i < c->NL; ++i) {
            const LayerPlan& l = c->ly[i];
            PackDesc& d = tab.d[i];
            d.w = params[4 * i]; d.dst = ws + c->wpkf_off + l.f_wpk;
            d.Cout = l.cout; d.Cin = l.cin; d.taps = l.taps; d.M = l.cout; d.CP = l.f_cinp; d.mtiles = l.f_mtiles_p; d.transposed = 0;
        }
        WUNET_LAUNCH(pack_weights_kernel, dim3(128, c->NL), dim3(WUNET_THREADS), 0, st, tab);
        WUNET_CHECK_LAUNCH();
    }
    // 2. conv (+ fused BN/LeakyReLU/decimate/upsample/concat on load) and BN statistics per layer
    for (int i = 0; i < c->NL; ++i) {
        const LayerPlan& l = c->ly[i];
        ConvArgs a{};
        a.src = layer_src(c, i, ws, noisy);
        a.geo = make_geom(l.L, 64 * l.f_nrep, l.taps / 2, 0, 16);
        a.wpk = ws + c->wpkf_off + l.f_wpk;
        a.bias = params[4 * i + 1];
        a.out = ws + l.z;
        a.stats = training ? ws + c->stats_off : nullptr;
        a.B = c->B; a.Cout = l.cout; a.CinP = l.f_cinp;
        a.kc_per_split = l.f_kcps; a.split_stride = (size_t)c->B * l.cout * l.L;
        a.xout = (save_for_backward && i > 0) ? ws + l.xin : nullptr;   // activated conv input, kept for the weight gradient
        if (l.f_ksplit > 1) { a.bias = nullptr; a.out = ws + c->spart_off; a.stats = nullptr; }
        int rc = launch_conv_any(l.taps, mode_of(l.kind), a, l.f_mrep, l.f_nrep, dim3(l.f_grid_x, l.f_mblocks, l.f_ksplit), st);
        if (rc) return rc;
        WUNET_CHECK_LAUNCH();
        BnFwdArgs b{};
        b.stats = ws + c->stats_off; b.rows = l.f_rows; b.bias = params[4 * i + 1];
        b.gamma = params[4 * i + 2]; b.beta = params[4 * i + 3];
        b.running_mean = running[2 * i]; b.running_var = running[2 * i + 1]; b.nbt = nbt[i];
        b.a = ws + l.a; b.s = ws + l.s; b.mean = ws + l.mean; b.rstd = ws + l.rstd;
        b.C = l.cout; b.count = (double)c->B * l.L; b.training = training ? 1 : 0;
        if (l.f_ksplit > 1) {
            WUNET_LAUNCH(conv_reduce_bn_kernel, dim3(l.cout), dim3(WUNET_THREADS), 0, st, b, (const float*)(ws + c->spart_off),
                         l.f_ksplit, (size_t)c->B * l.cout * l.L, ws + l.z, c->B, l.L, l.logL);
        } else {
            WUNET_LAUNCH(bn_finalize_fwd_kernel, dim3(l.cout), dim3(WUNET_THREADS), 0, st, b);
        }
        WUNET_CHECK_LAUNCH();
    }
    // 3. head
    {
        const LayerPlan& l = c->ly[c->NL - 1];
        HeadFwdArgs h{};
        h.z = ws + l.z; h.a = ws + l.a; h.s = ws + l.s; h.in = noisy;
        h.wh = params[4 * c->NL]; h.bh = params[4 * c->NL + 1]; h.out = enhanced;
        h.B = c->B; h.C = c->ci; h.T = c->T; h.logT = ilog2(c->T);
        long long blocks = ((long long)c->B * c->T + WUNET_THREADS - 1) / WUNET_THREADS;
        if (blocks > 4096) blocks = 4096;
        WUNET_LAUNCH(head_fwd_kernel, dim3((unsigned)blocks), dim3(WUNET_THREADS), 0, st, h);
        WUNET_CHECK_LAUNCH();
    }
    return WUNET_OK;
}

int wunet_backward_range(wunet_ctx* c, const float* noisy, const float* const* params, const float* enhanced,
                         const float* grad_enhanced, void* workspace, float* const* grads,
                         int layer_begin, int layer_end, void* stream)
{
    if (!c || !noisy || !params || !enhanced || !grad_enhanced || !workspace || !grads) return fail(WUNET_E_ARG, "null argument");
    if (layer_begin < 0 || layer_end > c->NL || layer_begin >= layer_end) return fail(WUNET_E_ARG, "bad layer range");
    hipStream_t st = (hipStream_t)stream;
    float* ws = (float*)workspace;
    const int NL = c->NL, n = c->n;

    if (layer_end == NL) {
        // flipped/transposed weights for every data gradient (one launch)
        PackTable tab{};
        int nd = 0;
        for (int i = 1; i < NL; ++i) {
            const LayerPlan& l = c->ly[i];
            PackDesc& d = tab.d[nd++];
            d.w = params[4 * i]; d.dst = ws + c->wpkb_off + l.d_wpk;
            d.Cout = l.cout; d.Cin = l.cin; d.taps = l.taps; d.M = l.cin; d.CP = l.d_cp; d.mtiles = l.d_mtiles_p; d.transposed = 1;
        }
        if (nd > 0) {
            WUNET_LAUNCH(pack_weights_kernel, dim3(128, nd), dim3(WUNET_THREADS), 0, st, tab);
            WUNET_CHECK_LAUNCH();
        }
        // head backward: gh = gout * tanh', d wh, d bh
        const LayerPlan& l = c->ly[NL - 1];
        HeadBwdArgs h{};
        h.z = ws + l.z; h.a = ws + l.a; h.s = ws + l.s; h.in = noisy; h.out = enhanced; h.gout = grad_enhanced;
        h.gh = ws + c->gh_off; h.part = ws + c->hpart_off; h.B = c->B; h.C = c->ci; h.T = c->T; h.logT = ilog2(c->T);
        WUNET_LAUNCH(head_bwd_kernel, dim3(c->head_blocks), dim3(WUNET_THREADS), 0, st, h);
        WUNET_CHECK_LAUNCH();
        const int nh = c->ci + 2;
        WUNET_LAUNCH(rows_sum_kernel, dim3(nh), dim3(WUNET_THREADS), 0, st,
                     (const float*)(ws + c->hpart_off), c->head_blocks, nh, grads[4 * NL], c->ci + 1, grads[4 * NL + 1]);
        WUNET_CHECK_LAUNCH();
    }

    for (int i = layer_end - 1; i >= layer_begin; --i) {
        const LayerPlan& l = c->ly[i];
        // ---- pass A: assemble dL/d(BN output), LeakyReLU', BN-backward partial sums
        PassAArgs p{};
        p.z = ws + l.z; p.a = ws + l.a; p.s = ws + l.s; p.mean = ws + l.mean; p.rstd = ws + l.rstd;
        p.gpre = ws + l.g; p.part = ws + c->bpart_off; p.B = c->B; p.C = l.cout; p.L = l.L; p.logL = l.logL;
        const dim3 ga(l.cout, l.a_split);
        if (i == NL - 1) {
            p.g0 = ws + c->gh_off; p.g1 = params[4 * NL];
            WUNET_LAUNCH(pass_a_kernel<A_HEAD>, ga, dim3(WUNET_THREADS), 0, st, p);
        } else if (i >= n) {
            const LayerPlan& nx = c->ly[i + 1];
            p.g0 = ws + nx.dx; p.Cg0 = nx.cin;
            p.up_scale = (float)(l.L - 1) / (float)(2 * l.L - 1);
            WUNET_LAUNCH(pass_a_kernel<A_UP>, ga, dim3(WUNET_THREADS), 0, st, p);
        } else {
            const LayerPlan& dc = c->ly[2 * n - i];
            const LayerPlan& nx = c->ly[i + 1];
            p.g0 = ws + dc.dx; p.Cg0 = dc.cin; p.coff = dc.c0; p.g1 = ws + nx.dx;
            WUNET_LAUNCH(pass_a_kernel<A_ENC>, ga, dim3(WUNET_THREADS), 0, st, p);
        }
        WUNET_CHECK_LAUNCH();
        BnBwdArgs b{};
        b.part = ws + c->bpart_off; b.rows = l.a_split; b.gamma = params[4 * i + 2]; b.mean = ws + l.mean; b.rstd = ws + l.rstd;
        b.dgamma = grads[4 * i + 2]; b.dbeta = grads[4 * i + 3]; b.dbias = grads[4 * i + 1]; b.k1 = ws + l.k1; b.k2 = ws + l.k2; b.k3 = ws + l.k3;
        b.C = l.cout; b.count = (double)c->B * l.L;
        WUNET_LAUNCH(bn_finalize_bwd_kernel, dim3(l.cout), dim3(WUNET_THREADS), 0, st, b);
        WUNET_CHECK_LAUNCH();

        const SrcDesc gz = gz_src(ws + l.g, ws + l.z, ws + l.k1, ws + l.k2, ws + l.k3, l.cout, l.L);
        // ---- data gradient (not needed for the first layer)
        if (i > 0) {
            ConvArgs a{};
            a.src = gz;
            a.geo = make_geom(l.L, 64 * l.d_nrep, l.taps / 2, 0, 16);
            a.wpk = ws + c->wpkb_off + l.d_wpk; a.bias = nullptr; a.out = ws + l.dx; a.stats = nullptr;
            a.B = c->B; a.Cout = l.cin; a.CinP = l.d_cp;
            a.kc_per_split = l.d_kcps; a.split_stride = (size_t)c->B * l.cin * l.L;
            a.xout = ws + c->gz_off;          // g_z materialised for the weight gradient
            if (l.d_ksplit > 1) a.out = ws + c->spart_off;
            int rc = launch_conv_any(l.taps, SRC_GZ, a, l.d_mrep, l.d_nrep, dim3(l.d_grid_x, l.d_mblocks, l.d_ksplit), st);
            if (rc) return rc;
            WUNET_CHECK_LAUNCH();
            if (l.d_ksplit > 1) {
                const size_t nd = (size_t)c->B * l.cin * l.L;
                size_t blocks = (nd + WUNET_THREADS - 1) / WUNET_THREADS;
                if (blocks > 2048) blocks = 2048;
                WUNET_LAUNCH(split_sum_kernel, dim3((unsigned)blocks), dim3(WUNET_THREADS), 0, st, (const float*)(ws + c->spart_off), l.d_ksplit, nd, ws + l.dx);
                WUNET_CHECK_LAUNCH();
            }
        }
        else {
            const size_t ng = (size_t)c->B * l.cout * l.L;
            size_t blocks = (ng + WUNET_THREADS * 4 - 1) / (WUNET_THREADS * 4);
            if (blocks > 4096) blocks = 4096;
            WUNET_LAUNCH(gz_materialize_kernel, dim3((unsigned)blocks), dim3(WUNET_THREADS), 0, st, (const float*)(ws + l.g), (const float*)(ws + l.z),
                         (const float*)(ws + l.k1), (const float*)(ws + l.k2), (const float*)(ws + l.k3), l.cout, l.logL, ng, ws + c->gz_off);
            WUNET_CHECK_LAUNCH();
        }
        // ---- weight gradient: GEMM over positions on the materialised operands, split-K partials + deterministic reduce
        {
            const float* xin = i == 0 ? noisy : ws + l.xin;
            const WgradArgs w = make_wgrad_args(xin, ws + c->gz_off, ws + c->wgpart_off, c->B, l.cin, l.cout, l.L, l.taps, l.w.cps);
            int rc = launch_wgrad_any(l.taps, w, l.w, st);
            if (rc) return rc;
            WUNET_CHECK_LAUNCH();
            const size_t nw = (size_t)l.cout * l.cin * l.taps;
            size_t blocks = (nw + WUNET_THREADS - 1) / WUNET_THREADS;
            if (blocks > 2048) blocks = 2048;
            WUNET_LAUNCH(wgrad_reduce_kernel, dim3((unsigned)blocks), dim3(WUNET_THREADS), 0, st,
                         (const float*)(ws + c->wgpart_off), l.w.rows, nw, grads[4 * i]);
            WUNET_CHECK_LAUNCH();
        }
    }
    return WUNET_OK;
}

int wunet_backward(wunet_ctx* c, const float* noisy, const float* const* params, const float* enhanced,
                   const float* grad_enhanced, void* workspace, float* const* grads, void* stream)
{
    if (!c) return fail(WUNET_E_ARG, "null ctx");
    return wunet_backward_range(c, noisy, params, enhanced, grad_enhanced, workspace, grads, 0, c->NL, stream);
}

size_t wunet_loss_scratch_bytes(void) { return 256 * sizeof(double); }

int wunet_loss_forward(int kind, const float* clean, const float* enhanced, size_t n, float* loss_out, void* scratch, void* stream)
{
    if (kind < 0 || kind > 2 || !clean || !enhanced || !loss_out || !scratch || n == 0) return fail(WUNET_E_ARG, "bad loss argument");
    hipStream_t st = (hipStream_t)stream;
    size_t blocks = (n + WUNET_THREADS * 8 - 1) / (WUNET_THREADS * 8);
    if (blocks > 256) blocks = 256;
    WUNET_LAUNCH(loss_partial_kernel, dim3((unsigned)blocks), dim3(WUNET_THREADS), 0, st, kind, clean, enhanced, n, (double*)scratch);
    WUNET_CHECK_LAUNCH();
    WUNET_LAUNCH(loss_final_kernel, dim3(1), dim3(WUNET_THREADS), 0, st, (const double*)scratch, (int)blocks, n, loss_out);
    WUNET_CHECK_LAUNCH();
    return WUNET_OK;
}

int wunet_loss_backward(int kind, const float* clean, const float* enhanced, const float* grad_loss, size_t n, float* grad_enhanced, void* stream)
{
    if (kind < 0 || kind > 2 || !clean || !enhanced || !grad_loss || !grad_enhanced || n == 0) return fail(WUNET_E_ARG, "bad loss argument");
    hipStream_t st = (hipStream_t)stream;
    size_t blocks = (n + WUNET_THREADS * 4 - 1) / (WUNET_THREADS * 4);
    if (blocks > 2048) blocks = 2048;
    WUNET_LAUNCH(loss_bwd_kernel, dim3((unsigned)blocks), dim3(WUNET_THREADS), 0, st, kind, clean, enhanced, grad_loss, n, grad_enhanced);
    WUNET_CHECK_LAUNCH();
    return WUNET_OK;
}

// ---------------------------------------------------------------------------- profiler
int wunet_profile_enable(int on)
{
    g_prof_on = on != 0;
    return WUNET_OK;
}

// Writes one line per kernel name: "name\tlaunches\ttotal_ms\ttotal_flops\ttotal_bytes\n"; clears the records.
// Synchronises the device.  Returns the number of bytes written (<= cap-1) or a negative code.
long long wunet_profile_collect(char* buf, size_t cap)
{
#ifdef WUNET_EMU
    if (cap) buf[0] = 0;
    return 0;
#else
    if (hipDeviceSynchronize() != hipSuccess) return fail(WUNET_E_RUNTIME, "hipDeviceSynchronize");
    struct Agg { std::string name; long n; double ms, fl, by; };
    std::vector<Agg> agg;
    for (ProfRec& r : g_prof) {
        float ms = 0.f;
        hipEventElapsedTime(&ms, r.e0, r.e1);
        hipEventDestroy(r.e0);
        hipEventDestroy(r.e1);
        size_t k = 0;
        for (; k < agg.size(); ++k) if (agg[k].name == r.name) break;
        if (k == agg.size()) agg.push_back(Agg{r.name, 0, 0.0, 0.0, 0.0});
        agg[k].n += 1; agg[k].ms += ms; agg[k].fl += r.flops; agg[k].by += r.bytes;
    }
    g_prof.clear();
    std::string out;
    char line[256];
    for (const Agg& a : agg) {
        snprintf(line, sizeof line, "%s\t%ld\t%.6f\t%.6e\t%.6e\n", a.name.c_str(), a.n, a.ms, a.fl, a.by);
        out += line;
    }
    if (cap == 0) return 0;
    const size_t nb = out.size() < cap - 1 ? out.size() : cap - 1;
    memcpy(buf, out.data(), nb);
    buf[nb] = 0;
    return (long long)nb;
#endif
}

// ---------------------------------------------------------------------------- single-op entry points
static int op_check(int B, int Cin, int Cout, int L, int K)
{
    if (K != 5 && K != 15) return fail(WUNET_E_ARG, "K must be 5 or 15");
    if (B < 1 || Cin < 1 || Cout < 1 || !is_pow2(L) || L < 4) return fail(WUNET_E_ARG, "bad op shape (L must be a power of two >= 4)");
    return 0;
}

int wunet_op_conv1d(const float* x, const float* w, const float* bias, float* z, int B, int Cin, int Cout, int L, int K, void* stream)
{
    if (op_check(B, Cin, Cout, L, K)) return WUNET_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int kc = kc_of(K), mt = (Cout + 15) / 16;
    const int mrep = pick_mrep(mt, 6), mtp = round_up(mt, mrep), mblocks = mtp / mrep;
    const int nrep = conv_nrep(B, L, mblocks), cinp = round_up(Cin, kc);
    float* wpk = nullptr;
    if (hipMalloc((void**)&wpk, (size_t)mtp * cinp * K * 16 * sizeof(float)) != hipSuccess) return fail(WUNET_E_RUNTIME, "hipMalloc");
    PackTable tab{};
    PackDesc& d = tab.d[0];
    d.w = w; d.dst = wpk; d.Cout = Cout; d.Cin = Cin; d.taps = K; d.M = Cout; d.CP = cinp; d.mtiles = mtp; d.transposed = 0;
    WUNET_LAUNCH(pack_weights_kernel, dim3(64, 1), dim3(WUNET_THREADS), 0, st, tab);
    ConvArgs a{};
    a.src.p0 = x; a.src.C = Cin; a.src.C0 = Cin; a.src.L = L; a.src.Lsrc0 = L; a.src.logL = ilog2(L);
    a.geo = make_geom(L, 64 * nrep, K / 2, 0, 16);
    a.wpk = wpk; a.bias = bias; a.out = z; a.stats = nullptr; a.B = B; a.Cout = Cout; a.CinP = cinp;
    a.kc_per_split = cinp; a.split_stride = 0; a.xout = nullptr;
    const int gx = (int)(((long long)B * L + 64 * nrep - 1) / (64 * nrep));
    int rc = launch_conv_any(K, SRC_RAW, a, mrep, nrep, dim3(gx, mblocks), st);
    hipStreamSynchronize(st);
    hipFree(wpk);
    if (rc) return rc;
    WUNET_CHECK_LAUNCH();
    return WUNET_OK;
}

static int make_unit_gz(int C, float** k1, float** k0, hipStream_t st)
{
    if (hipMalloc((void**)k1, C * sizeof(float)) != hipSuccess || hipMalloc((void**)k0, C * sizeof(float)) != hipSuccess)
        return fail(WUNET_E_RUNTIME, "hipMalloc");
    WUNET_LAUNCH(fill_kernel, dim3(1), dim3(WUNET_THREADS), 0, st, *k1, (size_t)C, 1.0f);
    WUNET_LAUNCH(fill_kernel, dim3(1), dim3(WUNET_THREADS), 0, st, *k0, (size_t)C, 0.0f);
    return 0;
}

int wunet_op_conv1d_dgrad(const float* gz, const float* w, float* dx, int B, int Cin, int Cout, int L, int K, void* stream)
{
    if (op_check(B, Cin, Cout, L, K)) return WUNET_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int kc = kc_of(K), mt = (Cin + 15) / 16;
    const int mrep = pick_mrep(mt, 6), mtp = round_up(mt, mrep), mblocks = mtp / mrep;
    const int nrep = conv_nrep(B, L, mblocks), cp = round_up(Cout, kc);
    float *wpk = nullptr, *k1 = nullptr, *k0 = nullptr;
    if (hipMalloc((void**)&wpk, (size_t)mtp * cp * K * 16 * sizeof(float)) != hipSuccess) return fail(WUNET_E_RUNTIME, "hipMalloc");
    if (make_unit_gz(Cout, &k1, &k0, st)) return WUNET_E_RUNTIME;
    PackTable tab{};
    PackDesc& d = tab.d[0];
    d.w = w; d.dst = wpk; d.Cout = Cout; d.Cin = Cin; d.taps = K; d.M = Cin; d.CP = cp; d.mtiles = mtp; d.transposed = 1;
    WUNET_LAUNCH(pack_weights_kernel, dim3(64, 1), dim3(WUNET_THREADS), 0, st, tab);
    ConvArgs a{};
    a.src = gz_src(gz, gz, k1, k0, k0, Cout, L);
    a.geo = make_geom(L, 64 * nrep, K / 2, 0, 16);
    a.wpk = wpk; a.bias = nullptr; a.out = dx; a.stats = nullptr; a.B = B; a.Cout = Cin; a.CinP = cp;
    a.kc_per_split = cp; a.split_stride = 0; a.xout = nullptr;
    const int gx = (int)(((long long)B * L + 64 * nrep - 1) / (64 * nrep));
    int rc = launch_conv_any(K, SRC_GZ, a, mrep, nrep, dim3(gx, mblocks), st);
    hipStreamSynchronize(st);
    hipFree(wpk); hipFree(k1); hipFree(k0);
    if (rc) return rc;
    WUNET_CHECK_LAUNCH();
    return WUNET_OK;
}

int wunet_op_conv1d_wgrad(const float* gz, const float* x, float* dw, int B, int Cin, int Cout, int L, int K, void* stream)
{
    if (op_check(B, Cin, Cout, L, K)) return WUNET_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    const WgradCfg cfg = plan_wgrad(B, L, Cin, Cout, K);
    float* part = nullptr;
    const size_t nw = (size_t)Cout * Cin * K;
    if (hipMalloc((void**)&part, (size_t)cfg.rows * nw * sizeof(float)) != hipSuccess) return fail(WUNET_E_RUNTIME, "hipMalloc");
    const WgradArgs a = make_wgrad_args(x, gz, part, B, Cin, Cout, L, K, cfg.cps);
    int rc = launch_wgrad_any(K, a, cfg, st);
    if (!rc) {
        size_t blocks = (nw + WUNET_THREADS - 1) / WUNET_THREADS;
        if (blocks > 2048) blocks = 2048;
        WUNET_LAUNCH(wgrad_reduce_kernel, dim3((unsigned)blocks), dim3(WUNET_THREADS), 0, st, (const float*)part, cfg.rows, nw, dw);
    }
    hipStreamSynchronize(st);
    hipFree(part);
    if (rc) return rc;
    WUNET_CHECK_LAUNCH();
    return WUNET_OK;
}

}  // extern "C"
