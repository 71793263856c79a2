// Host side of libwunet_hip.so: shape planning, workspace layout and kernel launches behind the
// C ABI declared in include/wunet_hip.h.  No torch types, no hidden device allocations on the hot
// path (the caller owns the workspace), nothing synchronises the stream.
#include <cstdarg>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "wunet_elementwise.h"
#include "wunet_tiny.h"
#include "wunet_h3_elem.h"
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
int kc_of(int taps) { return taps == 15 ? 4 : 12; }

// m-tiles (16 output channels each) handled per block: minimise zero padding, prefer bigger blocks.
int pick_mrep(int mtiles, int max_rep)
{
    // minimise zero padding; among equals prefer 4 then 3 accumulator rows per wave: measured on MI355X the
    // smaller register footprint (3 waves per SIMD instead of 2) beats the extra re-staging of the x tile
    static const int order[5] = {4, 3, 5, 6, 2};
    int best = 2, best_pad = 1 << 30;
    for (int k = 0; k < 5; ++k) {
        const int r = order[k];
        if (r > max_rep) continue;
        const int pad = round_up(mtiles, r) - mtiles;
        if (pad < best_pad) { best_pad = pad; best = r; }
    }
    return best;
}

// Tiling of one implicit-GEMM conv: rows = output channels of the GEMM, kch = its K channels.
struct ConvCfg { int mrep, nrep, mtiles_p, mblocks, cp, grid_x, ksplit, kcps; };

ConvCfg plan_conv(int B, int L, int rows, int kch, int taps)
{
    ConvCfg c{};
    const int kc = kc_of(taps);
    const int mt = (rows + 15) / 16;
    c.mrep = pick_mrep(mt, 6);
    c.mtiles_p = round_up(mt, c.mrep);
    c.mblocks = c.mtiles_p / c.mrep;
    c.cp = round_up(kch, kc);
    const long long pos = (long long)B * L;
    const int nchunks = c.cp / kc;
    // 256-position tiles (N_REP=4) reuse every A fragment 4x; usable when the staged row fits the loader
    // (k15: L >= 16, k5: L >= 64) and - counting the split-K factor available - the grid still fills the chip
    const bool wide_ok = (taps == 15 ? L >= 16 : L >= 64) && pos >= 256;
    const long long tiles4 = ((pos + 255) / 256) * c.mblocks;
    const int ksmax = nchunks >= 4 ? (nchunks / 2 < 16 ? nchunks / 2 : 16) : 1;
    c.nrep = (wide_ok && tiles4 * ksmax >= 256) ? 4 : 1;
    const int tn = 64 * c.nrep;
    c.grid_x = (int)((pos + tn - 1) / tn);
    // split-K over input-channel chunks when there are too few (position, m-block) tiles to fill 256 CUs
    const int blocks = c.grid_x * c.mblocks;
    c.ksplit = 1; c.kcps = c.cp;
    if (blocks < 192 && nchunks >= 4) {
        const int want = (512 + blocks - 1) / blocks;
        int cps = (nchunks + want - 1) / want;
        if (cps < 2) cps = 2;
        const int ks = (nchunks + cps - 1) / cps;
        if (ks > 1) { c.ksplit = ks; c.kcps = cps * kc; }
    }
    return c;
}

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

int launch_conv(int taps, const char* what, const ConvArgs& a, const ConvCfg& c, hipStream_t st)
{
    const int kc = kc_of(taps);
    char pname[96];
    snprintf(pname, sizeof pname, "conv_mfma_kernel<%d, %d, %d>", taps, c.mrep, c.nrep);
    (void)what;
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

struct WgradCfg { int mrep, nw, xit, wsplit, mblocks, nblocks, ksplit, cps, rows; };

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
    // split-K: one resident wave of equal-work blocks.  Residency estimate: accumulators + staging registers
    // against the 512-entry register file per SIMD lane, and the LDS footprint against 160 KiB.
    {
        const int regs = 4 * w.mrep * w.nw + 72 + 4 * (w.mrep + w.xit);
        int occ = regs <= 168 ? 3 : (regs <= 256 ? 2 : 1);
        const int segw = (L < 64 ? L : 64) + 16;
        const int rowp = (64 / (L < 64 ? L : 64)) * segw + (taps == 5 ? 24 : 0);
        const int cib = w.wsplit ? 1 : 4 * w.nw * (taps == 15 ? 1 : 3);
        const long long lds = ((long long)w.mrep * 16 * 66 + (long long)cib * rowp) * 4;
        const int occ_lds = (int)(160 * 1024 / lds);
        if (occ_lds < occ) occ = occ_lds < 1 ? 1 : occ_lds;
        const long long slots = 256LL * occ;
        const long long mn = (long long)w.mblocks * w.nblocks;
        long long ks = slots / mn;
        if (ks < 1) ks = 1;
        if (ks > chunks) ks = chunks;
        w.ksplit = (int)ks;
        w.cps = (int)((chunks + ks - 1) / ks);
        // drop splits that would only run padding chunks
        w.ksplit = (int)((chunks + w.cps - 1) / w.cps);
    }
    w.rows = w.ksplit * (w.wsplit ? 4 : 1);
    return w;
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

enum { LK_RAW = 0, LK_DECIM = 1, LK_UPCAT = 2 };

struct LayerPlan {
    int cin, cout, taps, L, logL, kind;
    int Lt;              // samples of a row that exist: L is the power-of-two row stride, Lt <= L (lengths m*2^n; Lt == L otherwise)
    int src0, src1;      // producer layers (src0 = -1: network input)
    int c0;              // UPCAT: channels from the upsampled branch
    ConvCfg f;           // forward conv (rows = cout, K channels = cin)
    ConvCfg d;           // data gradient (rows = cin, K channels = cout)
    WgradCfg w;          // weight gradient
    int f_rows;          // BN statistics partial rows written by the forward conv
    int a_split;         // pass A position splits
    size_t f_wpk, d_wpk; // float offsets inside the forward / backward weight packs
    // workspace (float offsets)
    size_t z, a, s, mean, rstd, xin, g, dx, k1, k2, k3;
    // fp16-split path
    int h3f, h3d;                 // forward conv / data gradient use conv_h3_kernel
    int h3f_mrep, h3f_mtp, h3f_nch, h3d_mrep, h3d_mtp, h3d_nch;
    int h3f_ntt, h3d_ntt;    // K tail of conv_h3d_kernel: steps of a tail stage (0: the last chunk is padded to 32 channels)
    int h3f_sps, h3d_sps;         // K stages per split of conv_h3_kernel (the split count is f.ksplit / d.ksplit)
    int first;                    // encoder[0] (Cin = 1): direct fp32 kernel (conv_first_kernel)
    int h3x;                      // the conv input exists only in the split layout (no fp32 xin)
    int skip_from;                // decoder layer: > 0 = its skip half is written by the operand pass of encoder-side layer skip_from
    int h3w, h3w_mrep, h3w_mblocks, h3w_nblocks, h3w_ksplit, h3w_cps, h3w_tp;   // weight gradient uses wgrad_h3_kernel
    int feeds_h3;                 // the layer's activation is the (or a) source of a conv input that exists in the split layout
    size_t xh, xl;                // split activated input (float offsets)
    size_t gzh, gzl;              // split scaled g_z (float offsets)
    size_t h3f_wpk, h3d_wpk;      // half offsets inside the split weight packs
};

}  // namespace

struct wunet_ctx {
    int n, ci, B, T, NL;
    int Tt = 0;                   // the caller's frame length (T: rounded up to a power of two, the row stride of every level)
    bool padded = false;          // Tt < T
    size_t pad_in = 0, pad_out = 0, pad_gout = 0;      // padded copies of noisy / enhanced / grad_enhanced (float offsets; padded only)
    std::vector<LayerPlan> ly;
    size_t stats_off, wpkf_off, spart_off, fwd_floats;
    size_t bmax_off, bound_off;   // pass A maxima / per-channel |g_z| bounds (fp16-split scale)
    size_t bpart_off, wgpart_off, wpkb_off, gh_off, hpart_off, hpart2_off, total_floats;
    int head_blocks;
    int h3 = 0;                   // fp16-split GEMMs: 0 off, 1 where the planner wants them, 2 wherever they can run
    int bf = 0;                   // the split kernels run their bf16 mode (one bf16 word per operand value, one MFMA pass)
    size_t h3_wf_hi, h3_wf_lo, h3_wb_hi, h3_wb_lo, h3_slot;   // float offsets
    size_t fslot_off, wmax_off;   // forward segment: WUNET_SLOT_FLOATS per layer (x / weight scales, activation bound), partial max |W|
    size_t h3_wf_halfs, h3_wb_halfs;
    // side stream for the weight-gradient GEMMs (off the backward's critical chain): one per device the ctx is used on, created
    // lazily under the lock and never replaced, so replicas of one shape on several devices (or threads) do not disturb each other.
    // Everything else in the ctx is immutable after wunet_create / wunet_set_h3.
    struct Side {
        hipStream_t stream = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_pack = nullptr, ev_fpack = nullptr, ev_fpack2 = nullptr;
        const void* packed_ws = nullptr;      // the workspace whose backward weight packs the last training forward enqueued on `stream`
    };
    std::map<int, Side> side;
    std::mutex side_lock;
};

namespace {

// blocks per layer of the weight-pack launches (grid-stride loops inside)
unsigned pack_gx()
{
    static const int gx = getenv("WUNET_PACK_GX") ? atoi(getenv("WUNET_PACK_GX")) : 512;       // A/B switch
    return (unsigned)(gx > 0 ? gx : 512);
}

int pick_mrep_h3(int mtiles, const char* env, const char* dflt)
{
    int best = 2, best_pad = 1 << 30;
    const char* ord = env ? getenv(env) : nullptr;    // A/B switch for measurements
    if (!ord) ord = dflt;
    for (const char* p = ord; *p; ++p) {
        const int m = *p - '0';
        const int pad = round_up(mtiles, m) - mtiles;
        if (pad < best_pad) { best_pad = pad; best = m; }
    }
    return best;
}

// Preference order of the accumulator rows per wave (16 * M_REP output rows per block) of conv_h3_kernel.  3 and 2 keep the
// most blocks resident; 4 (no padding for 8 m-tiles, a third fewer re-reads of the x tile) wins on the 5-tap layers while
// the grid still has two blocks for every CU (measured per layer: decoder.7 forward 73.6 -> 65.9 us, decoder.10 data
// gradient 166.7 -> 141.4 us; at 256 samples the same choice leaves CUs idle: 35.7 -> 49.6 us) and is neutral on 15 taps.
const char* h3_order(int taps, int L, int ntiles, int mtiles)
{
    return (taps == 5 && L >= 256 && (long long)ntiles * ((mtiles + 3) / 4) >= 512) ? "432" : "32";
}

// floats of one split's tile-major partial dW (wgrad_h3_kernel epilogue): padded tiles
size_t h3w_part_stride(const LayerPlan& l)
{
    const int tw = l.taps == 15 ? 8 : 5;
    return (size_t)l.h3w_mblocks * l.h3w_nblocks * WUNET_WAVES * l.h3w_mrep * tw * 256;
}

// conv_h3_kernel split-K: with fewer than ~1.5 blocks per CU the K stages are split so that about two blocks per CU
// exist; returns the stages per split (== nstage: no split).
int h3_stages_per_split(int blocks, int nstage)
{
    if (blocks >= 384 || nstage <= 1 || getenv("WUNET_H3_NOSPLIT")) return nstage;      // (switch: tests reach the un-split epilogue on small shapes)
    int ks = (512 + blocks - 1) / blocks;
    if (ks > nstage) ks = nstage;
    return (nstage + ks - 1) / ks;
}

// Tiling of conv_h3_kernel for one GEMM (rows x B*L positions, K = kch channels x taps): accumulator rows per wave, padded
// m-tiles, chunks of 32 K channels, K stages per split and the split count.  Shared by the network planner and the
// single-op entry points, so a geometry gets the same kernel instantiation either way.
struct H3ConvPlan { int mrep, mtp, nch, sps, ksplit, ntiles, ntt; };
bool h3_conv_is_dma(int L);
bool h3_conv_is_paired(int B, int L);

// K tail of conv_h3d_kernel: with c8 = 4 nfull + t1 groups of 8 K channels, the t1 left-over groups run one tail stage each (the taps
// spread over the four K quarters of the MFMA: ceil(taps / 4) steps) instead of one chunk padded with zeros (taps steps).  Worth it
// for 15 taps with any tail (4 t1 steps instead of 15), for 5 taps with one left-over group (2 instead of 5).
// Returns the steps of a tail stage, 0 without a tail.  WUNET_H3_KTAIL=0: A/B switch (the padded chunk everywhere)
int h3_tail_steps(int B, int L, int kch, int taps, int bf)
{
    const char* kt = getenv("WUNET_H3_KTAIL");          // (read when a context is planned)
    const int on = kt ? atoi(kt) : 1;
    const char* ile = getenv("WUNET_H3_IL");
    const int t1 = ((kch + 7) / 8) & 3;
    if (!on || !h3_conv_is_dma(L) || (!bf && h3_conv_is_paired(B, L)) || (ile && atoi(ile) != 0)) return 0;
    return taps == 15 ? (t1 ? 4 : 0) : (t1 == 1 ? 2 : 0);
}
// stages of the K loop: full chunks * tap groups + tail stages
int h3_stage_count(int kch, int taps, int ntt)
{
    const int c8 = (kch + 7) / 8;
    return ntt ? (c8 / 4) * (taps / 5) + (c8 & 3) : ((c8 + 3) / 4) * (taps / 5);
}

H3ConvPlan plan_h3_conv(int B, int L, int rows, int kch, int taps, const char* order_env, int bf = 0)
{
    H3ConvPlan p{};
    const long long posn = (long long)B * L;
    p.ntiles = (int)((posn + 255) / 256);
    const int ntg = taps / 5, c8 = (kch + 7) / 8, mt = (rows + 15) / 16;
    // (4 accumulator rows per wave only exist un-segmented: L >= 256)
    p.mrep = pick_mrep_h3(mt, L >= 256 ? order_env : nullptr, h3_order(taps, L, p.ntiles, mt));
    p.mtp = round_up(mt, p.mrep);
    p.nch = (c8 + 3) / 4;
    p.ntt = (p.mrep < 4 && L >= 32) ? h3_tail_steps(B, L, kch, taps, bf) : 0;      // (WUNET_H3D_HAS_TAIL: the shapes at the register limit go without)
    const int ns = h3_stage_count(kch, taps, p.ntt);
    p.sps = h3_stages_per_split(p.ntiles * (p.mtp / p.mrep), ns);
    p.ksplit = (ns + p.sps - 1) / p.sps;
    (void)ntg;
    return p;
}

// Tiling of the split weight gradient of one layer (fills l.h3w_*)
void plan_h3_wgrad(LayerPlan& l, int B)
{
    const int mt = (l.cout + 15) / 16, cib = l.taps == 15 ? 32 : 64;
    // 3 or 2 m-tiles per block: those DMA-staged kernels fit two blocks per CU, and two independent blocks beat taller
    // single blocks (weight gradients 1.25 -> 1.14 ms per step; the register-staged kernel preferred 5-6 m-tiles)
    // per-layer sweep (profiles/r1_h3_rows_per_wave_sweep.txt): 4 where 5 taps divide evenly at >= 256 samples
    // (decoder.7: 79 -> 68 us); 2 on the short 15-tap levels (encoder.6/7/9: 44 -> 39, 32 -> 29, 21 -> 18 us)
    // (round-2 sweep, profiles/r2_wgrad_rows_sweep.txt: 5 m-tiles in one block - no padded rows - win on the 15-tap layer with
    // Cout = 80 .. 72 at >= 1024 samples (encoder.2: 118 -> 106 us) although that kernel runs one block per CU)
    const char* w_order = (l.taps == 5 && l.L >= 256 && mt % 4 == 0) ? "432" : (l.taps == 15 && l.L <= 256) ? "2" :
                          (l.taps == 15 && mt == 5 && l.L >= 1024) ? "532" : "32";
    l.h3w_mrep = pick_mrep_h3(mt, "WUNET_H3W_ORDER", w_order);
    l.h3w_mblocks = round_up(mt, l.h3w_mrep) / l.h3w_mrep;
    l.h3w_nblocks = (l.cin + cib - 1) / cib;
    // positions per K chunk.  256 doubles the time a chunk's prefetch has to land and is 5-8 % faster for the kernel
    // alone (one block per CU, it waits on its staging), but with its 87 KB of LDS only one conv_h3 block fits
    // beside it and the whole step (weight gradients run concurrently with the data-gradient chain) is slower:
    // 6.76 vs 6.68 ms.  128 unless WUNET_H3W_TP=256.
    const int tp_env = getenv("WUNET_H3W_TP") ? atoi(getenv("WUNET_H3W_TP")) : 128;       // (read per plan: tests / measurements toggle it)
    l.h3w_tp = tp_env == 256 ? 256 : 128;
    // 64: double-buffered chunks of 64 positions at two blocks per CU (wgrad_h3d_kernel<.., true, .., 64>) where that kernel exists
    const bool d64 = tp_env == 64 && !getenv("WUNET_NO_H3W_DMA") && l.L >= 128 && ((l.taps == 5 && l.h3w_mrep <= 4) || (l.taps == 15 && l.h3w_mrep <= 3));
    if (d64) l.h3w_tp = 64;
    // wgrad_h3d_kernel<.., false> (single LDS buffer, two blocks per CU) where its registers allow: k5 up to 4 m-tiles, k15 up to 3
    const bool sb2 = d64 || (!getenv("WUNET_NO_H3W_SB") && !getenv("WUNET_NO_H3W_DMA") && l.L >= 128 && l.h3w_tp == 128 &&
                             ((l.taps == 5 && l.h3w_mrep <= 4) || (l.taps == 15 && l.h3w_mrep <= 3)));
    const long long slots = 256LL * ((l.h3w_mrep <= 2 || sb2) ? 2 : 1);      // resident blocks: launch bounds of the wgrad kernels
    long long ks = slots / ((long long)l.h3w_mblocks * l.h3w_nblocks);
    if (ks < 1) ks = 1;
    const long long chunks = ((long long)B * l.L + l.h3w_tp - 1) / l.h3w_tp;
    if (ks > chunks) ks = chunks;
    l.h3w_cps = (int)((chunks + ks - 1) / ks);
    l.h3w_ksplit = (int)((chunks + l.h3w_cps - 1) / l.h3w_cps);
}

void layout_workspace(wunet_ctx* c)
{
    const int B = c->B, T = c->T, ci = c->ci;
    size_t off = 0, wpk = 0, stats_max = 0, spart_max = 0;
    for (int i = 0; i < c->NL; ++i) {
        LayerPlan& l = c->ly[i];
        l.f = plan_conv(B, l.L, l.cout, l.cin, l.taps);
        l.d = plan_conv(B, l.L, l.cin, l.cout, l.taps);
        l.w = plan_wgrad(B, l.L, l.cin, l.cout, l.taps);
        if (l.L < 4) { l.f.ksplit = 1; l.d.ksplit = 1; }    // levels of 1-2 samples run the scalar kernels of wunet_tiny.h
        {
            // auto: only where the fp32 planner would launch an un-split full-width grid (enough 256-position tiles to fill
            // the chip); forced (2): every level the kernels can run (tests of small shapes)
            l.first = (i == 0 && l.cin == 1 && l.taps == 15 && l.L >= 256) ? 1 : 0;
            // fp16-split kernels.  auto (1): levels >= 256 samples where the fp32 planner would launch an un-split
            // full-width grid (enough 256-position tiles to fill the chip) and the 128- to 16-sample levels of a large
            // batch (split-K fills the chip there); forced (2): every level the kernels can run (tests of small shapes)
            const long long posn = (long long)B * l.L;
            // (A/B switch.  The 16-sample level was slower on the register-staged split kernels (6.98 vs 6.94 ms per step, min 32
            // then); on conv_h3d_kernel<., ., 16> it is 20-38 us per step faster than on the fp32 kernels)
            static const int min_l = getenv("WUNET_H3_MINL") ? atoi(getenv("WUNET_H3_MINL")) : 16;
            const bool big = c->h3 && !l.first && l.L >= 16 && posn >= 256 &&
                             (c->h3 == 2 || (l.L >= 256 ? (l.f.nrep == 4 && l.f.ksplit == 1) : (l.L >= min_l && posn >= 1024)));
            l.h3f = big ? 1 : 0;
            // backward: data gradient AND weight gradient together (g_z then only exists in the split layout)
            l.h3d = (big && i > 0 && l.cin >= 16 && (c->h3 == 2 || l.L < 256 || (l.d.nrep == 4 && l.d.ksplit == 1))) ? 1 : 0;
            l.h3w = l.h3d;
            l.h3x = l.h3w;        // ... and so does the conv input (no fp32 xin)
            if (l.h3f) {
                const H3ConvPlan p = plan_h3_conv(B, l.L, l.cout, l.cin, l.taps, "WUNET_H3_ORDER", c->bf);
                l.h3f_mrep = p.mrep; l.h3f_mtp = p.mtp; l.h3f_nch = p.nch; l.h3f_sps = p.sps; l.h3f_ntt = p.ntt;
                l.f.ksplit = p.ksplit;
                l.f.grid_x = p.ntiles;                 // one statistics row per tile (f_rows below)
            }
            if (l.h3d) {
                const H3ConvPlan p = plan_h3_conv(B, l.L, l.cin, l.cout, l.taps, "WUNET_H3D_ORDER", c->bf);
                l.h3d_mrep = p.mrep; l.h3d_mtp = p.mtp; l.h3d_nch = p.nch; l.h3d_sps = p.sps; l.h3d_ntt = p.ntt;
                l.d.ksplit = p.ksplit;
            }
            if (l.first) { l.f.ksplit = 1; l.f.grid_x = (int)(((long long)B * l.L + 1023) / 1024); }   // one statistics row per wave
        }
        l.f_rows = l.h3f ? l.f.grid_x : l.f.grid_x * WUNET_WAVES;
        l.f_wpk = wpk;
        wpk += align64((size_t)l.f.mtiles_p * l.f.cp * l.taps * 16);
        if (l.f.ksplit == 1 && (size_t)l.f_rows * l.cout * 2 > stats_max) stats_max = (size_t)l.f_rows * l.cout * 2;
        // (a padded length takes every conv but the first through the bias-free buffer and conv_reduce_bn_kernel: its statistics skip
        // the row padding, the conv kernels need not know)
        if ((l.f.ksplit > 1 || c->padded) && (size_t)64 * l.cout * 2 > stats_max) stats_max = (size_t)64 * l.cout * 2;
        if ((l.f.ksplit > 1 || c->padded) && (size_t)l.f.ksplit * B * l.cout * l.L > spart_max) spart_max = (size_t)l.f.ksplit * B * l.cout * l.L;
        if (l.L < 4 && (size_t)B * l.cout * l.L > spart_max) spart_max = (size_t)B * l.cout * l.L;
        if (i > 0 && l.d.ksplit > 1 && (size_t)l.d.ksplit * B * l.cin * l.L > spart_max) spart_max = (size_t)l.d.ksplit * B * l.cin * l.L;
        l.z = off; off += align64((size_t)B * l.cout * l.L);
        l.a = off; off += align64(l.cout);
        l.s = off; off += align64(l.cout);
        l.mean = off; off += align64(l.cout);
        l.rstd = off; off += align64(l.cout);
        l.xin = off; if (i > 0 && !l.h3x) off += align64((size_t)B * l.cin * l.L);   // the conv's activated input, materialised once
    }
    // the operand pass of encoder-side layer i (decimation of its producer's activation) also writes that activation at full
    // resolution into the split input of the decoder layer that concatenates it: one read of the producer's z instead of two
    for (int i = 0; i < c->NL; ++i) c->ly[i].skip_from = 0;
    for (int i = 1; i <= c->n; ++i) {
        const int dj = 2 * c->n - i + 1;
        LayerPlan& e = c->ly[i];
        LayerPlan& d = c->ly[dj];
        if (e.kind == LK_DECIM && d.kind == LK_UPCAT && e.h3x && d.h3x && d.src1 == e.src0 && d.c0 % 8 == 0 && !getenv("WUNET_NO_SKIP_FUSE"))
            d.skip_from = i;
    }
    c->stats_off = off; off += align64(stats_max);
    c->wpkf_off = off; off += align64(wpk);
    c->spart_off = off; off += align64(spart_max);
    // ---- fp16-split path: split activations + forward weight packs live in the forward segment
    size_t wfh = 0;
    for (int i = 0; i < c->NL; ++i) {
        LayerPlan& l = c->ly[i];
        l.xh = l.xl = 0; l.h3f_wpk = l.h3d_wpk = 0;
        if (l.h3f) {
            const int c8 = (l.cin + 7) / 8;
            l.xh = off; off += align64((size_t)B * c8 * l.L * 4);
            l.xl = off; off += align64((size_t)B * c8 * l.L * 4);
            l.h3f_wpk = wfh; wfh += (size_t)l.h3f_mtp * l.h3f_nch * l.taps * 512;
        }
    }
    c->h3_wf_halfs = wfh;
    c->h3_wf_hi = off; off += align64((wfh + 1) / 2);
    c->h3_wf_lo = off; off += align64((wfh + 1) / 2);
    c->fslot_off = off; off += align64((size_t)WUNET_SLOT_FLOATS * c->NL);
    c->wmax_off = off; off += align64((size_t)WUNET_WMAX_PARTS * c->NL);
    for (int i = 0; i < c->NL; ++i) c->ly[i].feeds_h3 = 0;
    for (int i = 1; i < c->NL; ++i) {
        const LayerPlan& l = c->ly[i];
        if (!l.h3f) continue;
        c->ly[l.src0].feeds_h3 = 1;
        if (l.kind == LK_UPCAT) c->ly[l.src1].feeds_h3 = 1;
    }
    if (c->padded) {            // padded copies of the caller's tensors (rows of T floats, the caller's hold Tt)
        c->pad_in = off; off += align64((size_t)B * T);
        c->pad_out = off; off += align64((size_t)B * T);
    }
    c->fwd_floats = off;

    size_t wpkb = 0, bpart_max = 0, wgpart_max = 0;
    for (int i = 0; i < c->NL; ++i) {
        LayerPlan& l = c->ly[i];
        l.g = off; off += align64((size_t)B * l.cout * l.L);
        l.dx = off; if (i > 0) off += align64((size_t)B * l.cin * l.L);
        l.k1 = off; off += align64(l.cout);
        l.k2 = off; off += align64(l.cout);
        l.k3 = off; off += align64(l.cout);
        l.d_wpk = wpkb;
        if (i > 0) wpkb += align64((size_t)l.d.mtiles_p * l.d.cp * l.taps * 16);
        if (l.h3w) plan_h3_wgrad(l, B);
        const size_t wg = l.h3w ? (size_t)l.h3w_ksplit * h3w_part_stride(l) : (size_t)l.w.rows * l.cout * l.cin * l.taps;
        if (wg > wgpart_max) wgpart_max = wg;
        long long sp = ((long long)B * l.L) / 4096;
        l.a_split = (int)(sp < 1 ? 1 : (sp > 64 ? 64 : sp));
        if ((size_t)l.a_split * l.cout * 2 > bpart_max) bpart_max = (size_t)l.a_split * l.cout * 2;
    }
    c->bpart_off = off; off += align64(bpart_max);
    c->bmax_off = off; off += align64(bpart_max);
    c->bound_off = off; off += align64(4096);
    c->wgpart_off = off; off += align64(wgpart_max);
    c->wpkb_off = off; off += align64(wpkb);
    c->gh_off = off; off += align64((size_t)B * T);
    {
        long long hb = ((long long)B * T) / 2048;
        c->head_blocks = (int)(hb < 1 ? 1 : (hb > 1024 ? 1024 : hb));
    }
    c->hpart_off = off; off += align64((size_t)c->head_blocks * 2);
    c->hpart2_off = off; off += align64((size_t)64 * ci);          // pass A (head mode) partial head-weight gradients [a_split][ci]
    // ---- fp16-split data gradient: transposed packs, one shared split g_z buffer, scale slot
    size_t wbh = 0, gzs = 0;
    for (int i = 0; i < c->NL; ++i) {
        LayerPlan& l = c->ly[i];
        if (!l.h3d) continue;
        const int c8 = (l.cout + 7) / 8;
        l.h3d_wpk = wbh; wbh += (size_t)l.h3d_mtp * l.h3d_nch * l.taps * 512;
        l.gzh = off; off += align64((size_t)B * c8 * l.L * 4);      // per layer: the side stream reads it late
        l.gzl = off; off += align64((size_t)B * c8 * l.L * 4);
    }
    c->h3_wb_halfs = wbh;
    c->h3_wb_hi = off; off += align64((wbh + 1) / 2);
    c->h3_wb_lo = off; off += align64((wbh + 1) / 2);
    (void)gzs;
    c->h3_slot = off; off += align64(8 + 4 * (size_t)c->NL);      // 8 zero floats (DMA zero page), then {scale, 1/scale} of g_z per layer (offset 8 + 4*layer)
    if (c->padded) { c->pad_gout = off; off += align64((size_t)B * T); }
    c->total_floats = off;
}

// ---- fp16-split helpers
int launch_split(const float* x, wunet_half* hi, wunet_half* lo, const float* sc, const float* xb0, const float* xb1, float* xsc,
                 int B, int C, int L, hipStream_t st, int bf = 0)
{
    const int c8 = (C + 7) / 8;
    const size_t n = (size_t)B * c8 * (L / 4);
    size_t blocks = (n + WUNET_THREADS - 1) / WUNET_THREADS;
    if (blocks > 16384) blocks = 16384;
    WUNET_LAUNCH(split_act_kernel, dim3((unsigned)blocks), dim3(WUNET_THREADS), 0, st, x, hi, lo, sc, xb0, xb1, xsc, B, C, c8, L, ilog2(L), bf);
    return 0;
}

// conv_h3p_kernel (two position tiles per block) runs instead of conv_h3_kernel: L >= 256, whole tiles, an even number of them
bool h3_conv_is_paired(int B, int L)
{
    // A/B switch, off by default: measured on MI355X the paired kernel is 3 % SLOWER per step (6.31 vs 6.11 ms) - one 512-thread
    // block per CU runs its two halves in lockstep, so nothing overlaps their prologues, barriers and epilogues, which two
    // independent 256-thread blocks do for each other
    const char* pe = getenv("WUNET_H3_PAIR");
    const int pair_env = pe ? atoi(pe) : 0;
    const long long posn = (long long)B * L;
    return pair_env && L >= 256 && (posn & 511) == 0;
}

// conv_h3d_kernel (DMA-staged, pipelined, persistent) runs instead of conv_h3_kernel (the 16-sample level, 16 items per tile with a
// 64 KB x image, at one block per CU)
bool h3_conv_is_dma(int L)
{
    const char* e = getenv("WUNET_H3_XDMA");            // A/B switch (read per launch: tests toggle it): 0 = off, 2 = un-segmented tiles only
    const int v = e ? atoi(e) : 1;
    return v != 0 && L >= (v == 2 ? 256 : 16);
}

// resident conv_h3d blocks the grid is sized for: two per CU (its launch bounds); WUNET_H3_GRID overrides (tests: a few blocks walk
// many work items)
int h3_grid_cap()
{
    if (const char* e = getenv("WUNET_H3_GRID")) { const int v = atoi(e); if (v > 0) return v; }
#ifdef WUNET_EMU
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

unsigned long long* g_h3_trace = nullptr;       // wunet_debug_set_conv_trace

int launch_conv_h3(int taps, int mrep, int mtiles_p, int sps, const wunet_half* xh, const wunet_half* xl, const wunet_half* wh,
                   const wunet_half* wl, const float* bias, const float* sc, const float* sc2, float* out, float* stats, int B, int rows,
                   int kch, int nch, int L, hipStream_t st, const float* ev_a = nullptr, const float* ev_s = nullptr, float* xrows = nullptr,
                   int bf = 0, int ntt = 0)
{
    char pname[96];
    const double posn = (double)B * L;
    ConvH3Args a{};
    a.xh = xh; a.xl = xl; a.wh = wh; a.wl = wl; a.bias = bias; a.sc = sc; a.sc2 = sc2; a.out = out; a.stats = stats;
    a.B = B; a.Cout = rows; a.C8 = (kch + 7) / 8; a.NCH = nch; a.L = L; a.logL = ilog2(L);
    a.ev_a = ev_a; a.ev_s = ev_s; a.xrows = xrows;
    const int nseg = L >= 256 ? 1 : 256 / L, nstage = h3_stage_count(kch, taps, ntt);
    a.NS = nstage; a.NFS = ntt ? (a.C8 / 4) * (taps / 5) : nstage;
    const int ksplit = (nstage + sps - 1) / sps;
    a.stages_per_split = sps; a.split_stride = (size_t)B * rows * L;

    a.ntiles = (int)((posn + 255) / 256); a.mblocks = mtiles_p / mrep;
    a.trace = g_h3_trace;
    // paired tiles (conv_h3p_kernel: two tiles per 512-thread block, double-buffered shared W) for L >= 256 with an even tile count
    const bool paired = !ntt && !bf && h3_conv_is_paired(B, L);     // (A/B switch: WUNET_H3_PAIR=1)
    int rc;
    if (ntt || (!paired && h3_conv_is_dma(L))) {         // (a pack with a K tail was laid out for this kernel)
        // conv_h3d_kernel: x tile and W sub-tile by LDS-DMA, buffers re-filled under the MFMAs, persistent blocks (two per CU)
        snprintf(pname, sizeof pname, bf ? "conv_h3d_kernel<%d, %d, %d, bf16>" : "conv_h3d_kernel<%d, %d, %d>", taps, mrep, nseg);
        prof_begin(st, pname, 2.0 * posn * rows * kch * taps, (bf ? 2.0 * posn * kch + 4.0 * posn * rows : 4.0 * posn * (rows + kch)));
        const int npl = bf ? 1 : 2;
        const size_t smem = (size_t)(npl * 4 * nseg * (256 / nseg + 16) + npl * mrep * 5 * 64) * 16 + (size_t)(WUNET_WAVES * mrep * 32 + 4) * sizeof(float);
        const int nitems = a.ntiles * a.mblocks;
        int gx = h3_grid_cap() / ksplit;
        gx &= ~7;
        if (gx < 8) gx = 8;
        if (gx > nitems) gx = nitems;
        const dim3 grid((unsigned)gx, (unsigned)ksplit);
        const char* ile = getenv("WUNET_H3_IL");         // A/B switch: DMA pieces interleaved with the MFMA passes (un-segmented tiles)
        rc = wunet_launch_conv_h3d(a, taps, mrep, nseg, grid, smem, st, bf != 0, !ntt && ile && atoi(ile) != 0 && !g_h3_trace);
    } else if (paired) {
        snprintf(pname, sizeof pname, "conv_h3p_kernel<%d, %d>", taps, mrep);
        prof_begin(st, pname, 2.0 * posn * rows * kch * taps, 4.0 * posn * (rows + kch));
        const size_t smem = (size_t)(2 * 2 * 4 * 272 + 2 * 2 * mrep * 5 * 64) * 16;
        const dim3 grid((unsigned)((a.ntiles / 2) * a.mblocks), (unsigned)ksplit);
        rc = wunet_launch_conv_h3p(a, taps, mrep, grid, smem, st);
    } else {
        snprintf(pname, sizeof pname, bf ? "conv_h3_kernel<%d, %d, %d, bf16>" : "conv_h3_kernel<%d, %d, %d>", taps, mrep, nseg);
        prof_begin(st, pname, 2.0 * posn * rows * kch * taps, (bf ? 2.0 * posn * kch + 4.0 * posn * rows : 4.0 * posn * (rows + kch)));
        const int npl = bf ? 1 : 2;
        const size_t smem = (size_t)(npl * 4 * nseg * (256 / nseg + 16) + npl * mrep * 5 * 64) * 16;
        const dim3 grid((unsigned)(a.ntiles * a.mblocks), (unsigned)ksplit);
        rc = wunet_launch_conv_h3(a, taps, mrep, nseg, grid, smem, st, bf != 0);
    }
    prof_end(st);
    if (rc != 0) return fail(WUNET_E_ARG, "no conv_h3 kernel for taps=%d mrep=%d nseg=%d (rc %d)", taps, mrep, nseg, rc);
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
    static const bool dma = getenv("WUNET_NO_H3W_DMA") == nullptr;                // A/B switch
    const int tp = l.h3w_tp, nseg = l.L >= tp ? 1 : tp / l.L;
    return dma && nseg == 1 && (tp == 128 || tp == 64) && h3w_dma_smem(l, bf) <= 160 * 1024 && l.h3w_mrep <= (l.taps == 15 ? 6 : 5);
}
int launch_wgrad_h3(const LayerPlan& l, const wunet_half* xh, const wunet_half* xl, const wunet_half* gh, const wunet_half* gl,
                    const float* sc, const float* sc2, const float* zero, float* part, int B, hipStream_t st, int bf = 0)
{
    char pname[96];
    const double posn = (double)B * l.L;
    const int xg = l.taps == 15 ? 4 : 8, tp = l.h3w_tp, nseg = l.L >= tp ? 1 : tp / l.L;
    const dim3 grid(l.h3w_ksplit, l.h3w_nblocks, l.h3w_mblocks);
    // (blocks of two m-tiles keep the register-staged kernel: it runs two blocks per CU, which is worth more)
    const int npl = bf ? 1 : 2;
    const size_t smem_d = h3w_dma_smem(l, bf);
    int rc;
    if (h3w_is_dma(l, bf)) {
        WgradH3dArgs a{};
        a.xh = xh; a.xl = xl; a.gh = gh; a.gl = gl; a.sc = sc; a.sc2 = sc2; a.part = part; a.B = B; a.Cin = l.cin; a.Cout = l.cout;
        (void)zero;
        a.XC8 = (l.cin + 7) / 8; a.GC8 = (l.cout + 7) / 8; a.L = l.L; a.logL = l.logL;
        a.chunks_per_split = l.h3w_cps; a.part_stride = h3w_part_stride(l);
        { const char* e = getenv("WUNET_H3W_NOSKIP"); a.cin_active = (e && atoi(e) != 0) ? 0x7fffffff : l.cin; }     // A/B switch
        snprintf(pname, sizeof pname, bf ? "wgrad_h3d_kernel<%d, %d, bf16>" : (tp == 64 ? "wgrad_h3d_kernel<%d, %d, 64>" : "wgrad_h3d_kernel<%d, %d>"), l.taps, l.h3w_mrep);
        prof_begin(st, pname, 2.0 * posn * l.cout * l.cin * l.taps, 4.0 * posn * (l.cout + l.cin));
        // two blocks per CU with a single buffer where the registers allow it (two independent blocks hide each other's
        // waits: +18-28 % on those kernels), else one block with double-buffered tiles
        static const bool sb = getenv("WUNET_NO_H3W_SB") == nullptr;              // A/B switch
        const bool db = tp == 64 || !(sb && ((l.taps == 5 && l.h3w_mrep <= 4) || (l.taps == 15 && l.h3w_mrep <= 3)));
        rc = wunet_launch_wgrad_h3d(a, l.taps, l.h3w_mrep, db, grid, db ? smem_d : smem_d / 2, st, bf != 0, tp);
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
    // both streams live on one device: device-scope release at the fork / join events is enough (WUNET_EVENT_SYSFENCE=1: the default
    // system-scope fence, for A/B measurements)
    const unsigned evf = hipEventDisableTiming | (getenv("WUNET_EVENT_SYSFENCE") ? 0u : (unsigned)hipEventDisableSystemFence);
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
}  // namespace

extern "C" {

const char* wunet_last_error(void) { return g_err.c_str(); }

void wunet_debug_set_conv_trace(void* dev_buffer) { g_h3_trace = static_cast<unsigned long long*>(dev_buffer); }

int wunet_create(int n_layers, int channels_interval, int batch, int length, wunet_ctx** out)
{
    if (!out) return fail(WUNET_E_ARG, "out is null");
    if (n_layers < 1 || 2 * n_layers + 1 > WUNET_MAX_CONV_LAYERS) return fail(WUNET_E_ARG, "n_layers=%d unsupported (1..16)", n_layers);
    if (channels_interval < 1 || batch < 1) return fail(WUNET_E_ARG, "bad channels_interval/batch");
    // model/unet_basic.py:86,93 accepts any length divisible by 2^n_layers.  A length m*2^k (m odd > 1) is carried in rows padded to
    // the next power of two: the padding holds zeros wherever a conv reads it (== the conv's own zero padding), is left out of the
    // BatchNorm statistics and gets no gradient; the upsample uses the coordinates of the lengths that exist.
    if (length < 4 || (length % (1 << n_layers)) != 0 || (length >> n_layers) < 1)
        return fail(WUNET_E_ARG, "length=%d unsupported: must be >= 4 and divisible by 2^n_layers (n_layers=%d)", length, n_layers);
    int Tp = 4;
    while (Tp < length) Tp <<= 1;
    const int n = n_layers, ci = channels_interval, B = batch, T = Tp;
    if ((long long)B * (2 * n + 1) * ci * T >= (1LL << 32)) return fail(WUNET_E_ARG, "tensor too large for 32-bit offsets");

    wunet_ctx* c = new wunet_ctx();
    c->n = n; c->ci = ci; c->B = B; c->T = T; c->NL = 2 * n + 1;
    c->Tt = length; c->padded = length != T;
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
        l.Lt = (int)(((long long)l.L * length) / T);       // (T / L is the level's decimation factor, a power of two dividing length)
    }
    layout_workspace(c);
    *out = c;
    return WUNET_OK;
}

int wunet_set_h3(wunet_ctx* ctx, int enable)
{
    if (!ctx) return fail(WUNET_E_ARG, "null ctx");
    // 3 / 4: the planner's (3) or the forced (4) layer set on the bf16 mode of the same kernels
    ctx->bf = (enable == 3 || enable == 4) ? 1 : 0;
    ctx->h3 = (enable == 2 || enable == 4) ? 2 : (enable ? 1 : 0);
    layout_workspace(ctx);          // sizes and offsets change: call before wunet_workspace_bytes
    return WUNET_OK;
}

void wunet_destroy(wunet_ctx* ctx)
{
    if (!ctx) return;
    for (auto& kv : ctx->side) {
        hipStreamDestroy(kv.second.stream); hipEventDestroy(kv.second.ev_fork); hipEventDestroy(kv.second.ev_join); hipEventDestroy(kv.second.ev_pack);
        hipEventDestroy(kv.second.ev_fpack); hipEventDestroy(kv.second.ev_fpack2);
    }
    delete ctx;
}

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
    // The skip half of each decoder input only depends on an encoder level and could run on the side stream during
    // the encoder phase; measured on MI355X that is SLOWER (forward 4.24 vs 3.99 ms: the elementwise kernel steals
    // L2/HBM bandwidth and CU slots from the encoder GEMMs), so it stays on the caller's stream.
    hipStream_t sd = st;
    float* const fslot = ws + c->fslot_off;
    float* const enhanced_user = enhanced;
    if (c->padded) {
        // the caller's [B][1][Tt] rows into zero-padded rows of T floats; the result is cropped back at the end
        if (hipMemsetAsync(ws + c->pad_in, 0, (size_t)c->B * c->T * sizeof(float), st) != hipSuccess ||
            hipMemcpy2DAsync(ws + c->pad_in, (size_t)c->T * sizeof(float), noisy, (size_t)c->Tt * sizeof(float), (size_t)c->Tt * sizeof(float),
                             (size_t)c->B, hipMemcpyDeviceToDevice, st) != hipSuccess)
            return fail(WUNET_E_RUNTIME, "padding the input failed");
        noisy = ws + c->pad_in;
        enhanced = ws + c->pad_out;
    }
    // 1. pack all forward weights into MFMA-fragment order (one launch each for the fp32 and the split packs).  A training forward
    // whose first layer runs conv_first_kernel (it reads the raw weights) enqueues them - and the operand scales - on the side stream:
    // conv_first and its BatchNorm finalize run beside them instead of behind three launches that only read the weights
    // (WUNET_NO_EARLY_FPACK=1: A/B switch).  Order on the side stream: scales, split pack (first needed by layer 1), fp32 pack (the
    // levels of <= 8 samples), then the backward's packs.
    wunet_ctx::Side* fside = nullptr;
    if (training && c->NL > 1 && c->ly[0].first && !g_prof_on) {
        static const bool no_side = getenv("WUNET_NO_SIDE_STREAM") != nullptr || getenv("WUNET_NO_EARLY_FPACK") != nullptr;
        if (!no_side) fside = side_for_current_device(c);
        if (fside && (hipEventRecord(fside->ev_fork, st) != hipSuccess || hipStreamWaitEvent(fside->stream, fside->ev_fork, 0) != hipSuccess))
            return fail(WUNET_E_RUNTIME, "fork onto the side stream failed");
    }
    const hipStream_t pst = fside ? fside->stream : st;
    auto pack_fp32 = [&]() -> int {
        PackTable tab{};
        int nd = 0;
        for (int i = 0; i < c->NL; ++i) {
            const LayerPlan& l = c->ly[i];
            if (l.h3f || l.first) continue;            // those layers do not read the fp32 pack
            PackDesc& d = tab.d[nd++];
            d.w = params[4 * i]; d.dst = ws + c->wpkf_off + l.f_wpk;
            d.Cout = l.cout; d.Cin = l.cin; d.taps = l.taps; d.M = l.cout; d.CP = l.f.cp; d.mtiles = l.f.mtiles_p; d.transposed = 0;
        }
        if (nd > 0) {
            WUNET_LAUNCH(pack_weights_kernel, dim3(pack_gx(), nd), dim3(WUNET_THREADS), 0, pst, tab);
            WUNET_CHECK_LAUNCH();
        }
        return 0;
    };
    if (!fside) { const int rc = pack_fp32(); if (rc) return rc; }
    if (c->h3) {
        // power-of-two scales of the split operands (wunet_h3_elem.h): partial max |W| of every layer with a split pack, and
        // the activation bounds the x scales derive from (training: from gamma / beta; eval: cleared here, measured per layer)
        {
            ScaleTable T{};
            bool any = false;
            for (int i = 0; i < c->NL; ++i) {
                const LayerPlan& l = c->ly[i];
                ScaleDesc& d = T.d[i];
                d.w = (l.h3f || l.h3d) ? params[4 * i] : nullptr; d.wn = (unsigned)((size_t)l.cout * l.cin * l.taps);
                d.gamma = params[4 * i + 2]; d.beta = params[4 * i + 3]; d.C = l.cout;
                d.sqrtn = sqrtf((float)((double)c->B * l.Lt));
                any = any || l.h3f;
            }
            T.wmax = ws + c->wmax_off; T.slots = ws + c->fslot_off; T.training = training ? 1 : 0;
            if (any) {
                WUNET_LAUNCH(h3_scales_kernel, dim3(WUNET_WMAX_PARTS, c->NL), dim3(WUNET_THREADS), 0, pst, T);
                WUNET_CHECK_LAUNCH();
            }
        }
        PackH3Table tab{};
        int nd = 0;
        wunet_half* wh = reinterpret_cast<wunet_half*>(ws + c->h3_wf_hi);
        wunet_half* wl = reinterpret_cast<wunet_half*>(ws + c->h3_wf_lo);
        for (int i = 0; i < c->NL; ++i) {
            const LayerPlan& l = c->ly[i];
            if (!l.h3f) continue;
            PackH3Desc& d = tab.d[nd++];
            d.w = params[4 * i]; d.hi = wh + l.h3f_wpk; d.lo = wl + l.h3f_wpk;
            d.Cout = l.cout; d.Cin = l.cin; d.taps = l.taps; d.rows = l.cout; d.kch = l.cin; d.mtiles = l.h3f_mtp; d.nch = l.h3f_nch; d.transposed = 0;
            d.ntt = l.h3f_ntt; d.nfull = ((l.cin + 7) / 8) / 4; d.ns = h3_stage_count(l.cin, l.taps, l.h3f_ntt);
            d.wmax = ws + c->wmax_off + (size_t)WUNET_WMAX_PARTS * i; d.wsc = ws + c->fslot_off + (size_t)WUNET_SLOT_FLOATS * i + 2;
            d.bf = c->bf;
        }
        if (nd > 0) {
            WUNET_LAUNCH(pack_h3_kernel, dim3(pack_gx(), nd), dim3(WUNET_THREADS), 0, pst, tab);
            WUNET_CHECK_LAUNCH();
        }
    }
    if (fside) {
        if (hipEventRecord(fside->ev_fpack, pst) != hipSuccess) return fail(WUNET_E_RUNTIME, "recording the pack event failed");
        const int rc = pack_fp32();
        if (rc) return rc;
        if (hipEventRecord(fside->ev_fpack2, pst) != hipSuccess) return fail(WUNET_E_RUNTIME, "recording the pack event failed");
    }
    bool fpack_joined = fside == nullptr, fpack2_joined = fside == nullptr;
    // The backward's flipped / transposed weight packs only depend on the weights (and on the maxima h3_scales_kernel has just
    // taken): a training forward that will be followed by a backward enqueues them on the side stream now, beside the first convs,
    // instead of leaving two launches at the head of the backward's critical path (WUNET_NO_EARLY_BPACK=1: A/B switch).
    if (training && save_for_backward && c->NL > 1) {
        static const bool no_side = getenv("WUNET_NO_SIDE_STREAM") != nullptr || getenv("WUNET_NO_EARLY_BPACK") != nullptr;
        wunet_ctx::Side* side = (no_side || g_prof_on) ? nullptr : side_for_current_device(c);
        if (side) {
            // (behind the forward's packs on the side stream when they are there: no second fork)
            if (side != fside && (hipEventRecord(side->ev_fork, st) != hipSuccess || hipStreamWaitEvent(side->stream, side->ev_fork, 0) != hipSuccess))
                return fail(WUNET_E_RUNTIME, "fork onto the side stream failed");
            const int rc = launch_backward_packs(c, params, ws, side->stream);
            if (rc) return rc;
            if (hipEventRecord(side->ev_pack, side->stream) != hipSuccess) return fail(WUNET_E_RUNTIME, "recording the pack event failed");
            side->packed_ws = workspace;
        }
    }
    for (int i = 0; i < c->NL; ++i) {
        const LayerPlan& l = c->ly[i];
        // (packs on the side stream: the operand scales and the split pack are first read by layer 1, the fp32 pack by the first
        // layer on the fp32 kernels)
        if (!fpack_joined && i >= 1) {
            if (hipStreamWaitEvent(st, fside->ev_fpack, 0) != hipSuccess) return fail(WUNET_E_RUNTIME, "waiting for the weight packs failed");
            fpack_joined = true;
        }
        if (!fpack2_joined && i >= 1 && !l.h3f) {
            if (hipStreamWaitEvent(st, fside->ev_fpack2, 0) != hipSuccess) return fail(WUNET_E_RUNTIME, "waiting for the weight packs failed");
            fpack2_joined = true;
        }
        // 2a. materialise the conv input: BN scale/shift + LeakyReLU + decimation, or + x2 upsample + skip concat
        const float* xin = noisy;
        if (i > 0) {
            const LayerPlan& p = c->ly[l.src0];
            PrepArgs pa{};
            pa.z0 = ws + p.z; pa.a0 = ws + p.a; pa.s0 = ws + p.s; pa.x = ws + l.xin;
            pa.B = c->B; pa.C0 = l.c0; pa.C1 = l.cin - l.c0; pa.L = l.L; pa.logL = l.logL; pa.Lt = l.Lt;
            const size_t n4 = (size_t)c->B * l.cin * l.L / 4;
            size_t blocks = (n4 + WUNET_THREADS - 1) / WUNET_THREADS;
            if (blocks > 8192) blocks = 8192;
            if (l.h3x) {
                PrepH3Args ph{};
                ph.z0 = pa.z0; ph.a0 = pa.a0; ph.s0 = pa.s0;
                ph.xh = reinterpret_cast<wunet_half*>(ws + l.xh); ph.xl = reinterpret_cast<wunet_half*>(ws + l.xl);
                ph.B = c->B; ph.C0 = l.c0; ph.C1 = l.cin - l.c0; ph.C8 = (l.cin + 7) / 8; ph.L = l.L; ph.logL = l.logL; ph.Lt = l.Lt;
                ph.kind = l.kind == LK_UPCAT ? 1 : 0;
                // the encoder-side pass can only write the decoder's skip half when the decoder's x scale is known that early:
                // training mode (data-independent activation bounds); in eval mode the decoder-side pass reads the skip itself
                ph.up_only = (l.kind == LK_UPCAT && l.skip_from > 0 && training) ? 1 : 0;
                ph.xb0 = fslot + (size_t)WUNET_SLOT_FLOATS * l.src0 + 4;
                ph.xb1 = l.kind == LK_UPCAT ? fslot + (size_t)WUNET_SLOT_FLOATS * l.src1 + 4 : nullptr;
                ph.xsc = fslot + (size_t)WUNET_SLOT_FLOATS * i;
                ph.bf = c->bf;
                if (l.kind == LK_DECIM && training) {
                    const int dj = 2 * c->n - i + 1;             // the decoder layer that concatenates this pass's producer
                    if (dj < c->NL && c->ly[dj].skip_from == i) {
                        const LayerPlan& dl = c->ly[dj];
                        ph.sh = reinterpret_cast<wunet_half*>(ws + dl.xh); ph.sl = reinterpret_cast<wunet_half*>(ws + dl.xl);
                        ph.SC8 = (dl.cin + 7) / 8; ph.sc8off = dl.c0 / 8;
                        ph.ssb0 = fslot + (size_t)WUNET_SLOT_FLOATS * dl.src0 + 4;
                        ph.ssb1 = fslot + (size_t)WUNET_SLOT_FLOATS * dl.src1 + 4;
                    }
                }
                if (l.kind == LK_UPCAT) {
                    const LayerPlan& k = c->ly[l.src1];
                    ph.z1 = ws + k.z; ph.a1 = ws + k.a; ph.s1 = ws + k.s;
                    ph.up_scale = (float)(l.Lt / 2 - 1) / (float)(l.Lt - 1);
                }
                const size_t nt = (size_t)c->B * (ph.up_only ? ph.C0 / 8 : ph.C8) * (l.L / 4);
                size_t hb = (nt + WUNET_THREADS - 1) / WUNET_THREADS;
                if (hb > 16384) hb = 16384;
                const bool prep4 = getenv("WUNET_PREP4") != nullptr && !c->padded;                      // A/B switch (read per launch: tests toggle it)
                if (prep4) {
                    WUNET_LAUNCH(prep4_h3_kernel, dim3((unsigned)hb), dim3(WUNET_THREADS), 0, st, ph);
                } else {
                    size_t hb1 = (nt * 4 + WUNET_THREADS - 1) / WUNET_THREADS;            // one thread per sample and channel group
                    if (hb1 > 65536) hb1 = 65536;
                    const dim3 g1((unsigned)hb1), t1(WUNET_THREADS);
                    // algorithmic bytes of the operand pass (HBM-bound): fp32 sources read once, 2 + 2 (bf16: 2) bytes per value written
                    const double ob = c->bf ? 2.0 : 4.0, pe = (double)c->B * l.L;
                    const int mode = ph.up_only ? 3 : ph.kind ? 2 : ph.sh ? 1 : 0;
                    const double pbytes = mode == 0 ? pe * l.cin * (8.0 + ob) : mode == 1 ? pe * l.cin * (8.0 + 3.0 * ob)
                                        : mode == 2 ? pe * (ph.C0 * 2.0 + ph.C1 * 4.0 + l.cin * ob) : pe * ph.C0 * (2.0 + ob);
                    static const char* const pn[4] = {"prep_h3_kernel<0>", "prep_h3_kernel<1>", "prep_h3_kernel<2>", "prep_h3_kernel<3>"};
                    prof_begin(st, pn[mode], 0.0, pbytes);
                    if (ph.up_only) WUNET_LAUNCH((prep_h3_kernel<3>), g1, t1, 0, st, ph);
                    else if (ph.kind) WUNET_LAUNCH((prep_h3_kernel<2>), g1, t1, 0, st, ph);
                    else if (ph.sh) WUNET_LAUNCH((prep_h3_kernel<1>), g1, t1, 0, st, ph);
                    else WUNET_LAUNCH((prep_h3_kernel<0>), g1, t1, 0, st, ph);
                    prof_end(st);
                }
            } else if (l.L < 4) {
                if (l.kind == LK_UPCAT) {
                    const LayerPlan& k = c->ly[l.src1];
                    pa.z1 = ws + k.z; pa.a1 = ws + k.a; pa.s1 = ws + k.s;
                    pa.up_scale = l.Lt > 1 ? (float)(l.Lt / 2 - 1) / (float)(l.Lt - 1) : 0.f;
                }
                const size_t ne = (size_t)c->B * l.cin * l.L;
                WUNET_LAUNCH(prep_scalar_kernel, dim3((unsigned)((ne + WUNET_THREADS - 1) / WUNET_THREADS)), dim3(WUNET_THREADS), 0, st, pa,
                             l.kind == LK_UPCAT ? 1 : 0);
            } else if (l.kind == LK_DECIM) {
                WUNET_LAUNCH(prep_decim_kernel, dim3((unsigned)blocks), dim3(WUNET_THREADS), 0, st, pa);
            } else {
                const LayerPlan& k = c->ly[l.src1];
                pa.z1 = ws + k.z; pa.a1 = ws + k.a; pa.s1 = ws + k.s;
                pa.up_scale = l.Lt > 1 ? (float)(l.Lt / 2 - 1) / (float)(l.Lt - 1) : 0.f;
                WUNET_LAUNCH(prep_upcat_kernel, dim3((unsigned)blocks), dim3(WUNET_THREADS), 0, st, pa, 0, l.cin);
            }
            WUNET_CHECK_LAUNCH();
            xin = ws + l.xin;
        }
        // 2b. conv (+ bias, + per-wave BN statistics partials) on the matrix cores
        const bool tiny = l.L < 4;
        const bool split = l.f.ksplit > 1 || tiny || (c->padded && !l.first);      // all leave a bias-free result in the split buffer
        // BatchNorm statistics -> scale/shift for the consumers (+ running stats); eval mode: from the running statistics
        BnFwdArgs b{};
        b.stats = ws + c->stats_off; b.rows = l.f_rows; b.bias = params[4 * i + 1];
        b.gamma = params[4 * i + 2]; b.beta = params[4 * i + 3];
        b.running_mean = running[2 * i]; b.running_var = running[2 * i + 1]; b.nbt = nbt[i];
        b.a = ws + l.a; b.s = ws + l.s; b.mean = ws + l.mean; b.rstd = ws + l.rstd;
        b.C = l.cout; b.count = (double)c->B * l.Lt; b.training = training ? 1 : 0;
        // eval mode, a layer whose activation feeds split operands: its consumers' operand scale comes from the measured maximum
        // of |a z + s| (no batch statistics bound it).  The large producers (conv_first_kernel, un-split conv_h3_kernel) take it
        // in their epilogue - the BatchNorm coefficients only depend on the running statistics, so they are finalised BEFORE the
        // conv - the small ones get a pass of act_max_kernel over z.
        const bool ev_need = !training && c->h3 && l.feeds_h3;
        const bool ev_epi = ev_need && (l.first || (l.h3f && !split));
        float* const xrows = ws + c->stats_off;         // (eval: the statistics rows are free)
        if (ev_epi) {
            WUNET_LAUNCH(bn_finalize_fwd_kernel, dim3(l.cout), dim3(WUNET_THREADS), 0, st, b);
            WUNET_CHECK_LAUNCH();
        }
        if (l.first) {
            prof_begin(st, "conv_first_kernel<15>", 2.0 * c->B * l.L * l.cout * 15.0, 4.0 * c->B * l.L * (1.0 + l.cout));
            WUNET_LAUNCH(conv_first_kernel<15>, dim3((unsigned)l.f.grid_x), dim3(WUNET_THREADS), 0, st, xin, params[4 * i], params[4 * i + 1],
                         ws + l.z, training ? ws + c->stats_off : (float*)nullptr, c->B, l.cout, l.L, l.logL,
                         ev_epi ? ws + l.a : (const float*)nullptr, ev_epi ? ws + l.s : (const float*)nullptr, ev_epi ? xrows : (float*)nullptr, l.Lt);
            prof_end(st);
        } else if (l.h3f) {
            // fp16-split GEMM: split the materialised input, then 3 MFMA passes on the 2.5 PF pipe
            wunet_half* xh = reinterpret_cast<wunet_half*>(ws + l.xh);
            wunet_half* xl = reinterpret_cast<wunet_half*>(ws + l.xl);
            float* const sl = fslot + (size_t)WUNET_SLOT_FLOATS * i;      // [0..1] x scale, [2..3] weight scale
            if (!l.h3x) {
                launch_split(xin, xh, xl, nullptr, fslot + (size_t)WUNET_SLOT_FLOATS * l.src0 + 4,
                             l.kind == LK_UPCAT ? fslot + (size_t)WUNET_SLOT_FLOATS * l.src1 + 4 : nullptr, sl, c->B, l.cin, l.L, st, c->bf);
                WUNET_CHECK_LAUNCH();
            }
            int rc = launch_conv_h3(l.taps, l.h3f_mrep, l.h3f_mtp, l.h3f_sps, xh, xl,
                                    reinterpret_cast<const wunet_half*>(ws + c->h3_wf_hi) + l.h3f_wpk,
                                    reinterpret_cast<const wunet_half*>(ws + c->h3_wf_lo) + l.h3f_wpk, params[4 * i + 1], sl, sl + 2,
                                    split ? ws + c->spart_off : ws + l.z, (training && !split) ? ws + c->stats_off : nullptr, c->B, l.cout,
                                    l.cin, l.h3f_nch, l.L, st, ev_epi ? ws + l.a : nullptr, ev_epi ? ws + l.s : nullptr, ev_epi ? xrows : nullptr, c->bf, l.h3f_ntt);
            if (rc) return rc;
        } else if (tiny) {
            const size_t no = (size_t)c->B * l.cout * l.L;
            WUNET_LAUNCH(tiny_conv_kernel, dim3((unsigned)((no + WUNET_THREADS - 1) / WUNET_THREADS)), dim3(WUNET_THREADS), 0, st,
                         xin, params[4 * i], ws + c->spart_off, c->B, l.cin, l.cout, l.L, l.logL, l.taps, 0);
        } else {
            const ConvArgs a = make_conv_args(xin, l.cin, ws + c->wpkf_off + l.f_wpk, split ? nullptr : params[4 * i + 1],
                                              split ? ws + c->spart_off : ws + l.z, (training && !split) ? ws + c->stats_off : nullptr,
                                              c->B, l.cout, l.L, l.taps, l.f, (size_t)c->B * l.cout * l.L);
            int rc = launch_conv(l.taps, "fwd", a, l.f, st);
            if (rc) return rc;
        }
        WUNET_CHECK_LAUNCH();
        // 2c. BatchNorm statistics -> scale/shift for the consumers (+ running stats)
        if (ev_epi) {
            int nrows = l.first ? l.f.grid_x : (int)(((long long)c->B * l.L + 255) / 256) * (l.h3f_mtp / l.h3f_mrep);
            if (!l.first && !c->bf && h3_conv_is_paired(c->B, l.L)) nrows /= 2;            // conv_h3p_kernel: one row per tile pair
            WUNET_LAUNCH(xb_reduce_kernel, dim3(1), dim3(WUNET_THREADS), 0, st, (const float*)xrows, nrows, fslot + (size_t)WUNET_SLOT_FLOATS * i + 4);
        } else if (split) {
            // sum the z-slices (+bias -> z) and reduce the BN statistics; short levels finish BN in the same launch
            const long long pos = (long long)c->B * l.L;
            int rs = (int)(pos / 2048);
            if (rs < 1) rs = 1;
            if (rs > 64) rs = 64;
            b.rows = rs;
            WUNET_LAUNCH(conv_reduce_bn_kernel, dim3(l.cout, rs), dim3(WUNET_THREADS), 0, st, b, (const float*)(ws + c->spart_off),
                         tiny ? 1 : l.f.ksplit, (size_t)c->B * l.cout * l.L, ws + l.z, c->B, l.L, l.logL, ws + c->stats_off, l.Lt);
            if (rs > 1) {
                WUNET_CHECK_LAUNCH();
                WUNET_LAUNCH(bn_finalize_fwd_kernel, dim3(l.cout), dim3(WUNET_THREADS), 0, st, b);
            }
        } else {
            WUNET_LAUNCH(bn_finalize_fwd_kernel, dim3(l.cout), dim3(WUNET_THREADS), 0, st, b);
        }
        WUNET_CHECK_LAUNCH();
        if (ev_need && !ev_epi) {
            const size_t n4 = (size_t)c->B * l.cout * l.L / 4;
            size_t blocks = (n4 + WUNET_THREADS * 4 - 1) / (WUNET_THREADS * 4);
            if (blocks > 2048) blocks = 2048;
            if (blocks < 1) blocks = 1;
            WUNET_LAUNCH(act_max_kernel, dim3((unsigned)blocks), dim3(WUNET_THREADS), 0, st, (const float*)(ws + l.z), (const float*)(ws + l.a),
                         (const float*)(ws + l.s), l.cout, l.logL, n4, fslot + (size_t)WUNET_SLOT_FLOATS * i + 4);
            WUNET_CHECK_LAUNCH();
        }
    }
    if (!fpack_joined && hipStreamWaitEvent(st, fside->ev_fpack, 0) != hipSuccess) return fail(WUNET_E_RUNTIME, "waiting for the weight packs failed");
    if (!fpack2_joined && hipStreamWaitEvent(st, fside->ev_fpack2, 0) != hipSuccess) return fail(WUNET_E_RUNTIME, "waiting for the weight packs failed");
    // 3. head
    {
        const LayerPlan& l = c->ly[c->NL - 1];
        HeadFwdArgs h{};
        h.z = ws + l.z; h.a = ws + l.a; h.s = ws + l.s; h.in = noisy;
        h.wh = params[4 * c->NL]; h.bh = params[4 * c->NL + 1]; h.out = enhanced;
        h.B = c->B; h.C = c->ci; h.T = c->T; h.logT = ilog2(c->T);
        long long blocks = ((long long)c->B * c->T + WUNET_THREADS - 1) / WUNET_THREADS;
        if (blocks > 4096) blocks = 4096;
        prof_begin(st, "head_fwd_kernel", 2.0 * c->B * c->T * (c->ci + 1.0), 4.0 * c->B * c->T * (c->ci + 2.0));
        WUNET_LAUNCH(head_fwd_kernel, dim3((unsigned)blocks), dim3(WUNET_THREADS), 0, st, h);
        prof_end(st);
        WUNET_CHECK_LAUNCH();
    }
    if (c->padded && hipMemcpy2DAsync(enhanced_user, (size_t)c->Tt * sizeof(float), ws + c->pad_out, (size_t)c->T * sizeof(float),
                                      (size_t)c->Tt * sizeof(float), (size_t)c->B, hipMemcpyDeviceToDevice, st) != hipSuccess)
        return fail(WUNET_E_RUNTIME, "cropping the output failed");
    return WUNET_OK;
}

}  // extern "C"

namespace {
// dx of an encoder-side layer i (1 <= i <= n: encoder 1 .. middle) has ONE reader, pass A of layer i - 1, the next kernel on the
// stream: a split-K data gradient leaves its partials in the scratch and that pass adds them itself, in split order (same bits), instead
// of a split_sum_kernel launch in between.  (A decoder layer's dx is read again much later - its skip half by the encoder side - and is
// summed as before.)  WUNET_NO_SPLITSUM_FUSE=1: A/B switch
bool dx_stays_split(const wunet_ctx* c, int i)
{
    if (i < 1 || i > c->n || getenv("WUNET_NO_SPLITSUM_FUSE")) return false;
    const LayerPlan& l = c->ly[i];
    return l.d.ksplit > 1 && l.L >= 4 && c->ly[i - 1].L >= 4;       // (neither end on the scalar kernels of the 1-2-sample levels)
}

int backward_range_impl(wunet_ctx* c, const float* noisy, const float* const* params, const float* enhanced,
                        const float* grad_enhanced, void* workspace, float* const* grads,
                        int layer_begin, int layer_end, void* stream, bool join)
{
    if (!c || !noisy || !params || !enhanced || !grad_enhanced || !workspace || !grads) return fail(WUNET_E_ARG, "null argument");
    if (layer_begin < 0 || layer_end > c->NL || layer_begin >= layer_end) return fail(WUNET_E_ARG, "bad layer range");
    hipStream_t st = (hipStream_t)stream;
    float* ws = (float*)workspace;
    const int NL = c->NL, n = c->n;
    // weight gradients run on a side stream: they only depend on g_z and x of their own layer, so the HBM-bound
    // gradient-assembly kernels of the next layers overlap with them instead of idling the matrix cores
    wunet_ctx::Side* side = side_for_current_device(c);
    if (!side) return WUNET_E_RUNTIME;
    static const bool no_side = getenv("WUNET_NO_SIDE_STREAM") != nullptr;     // A/B switch for measurements
    hipStream_t sd = (g_prof_on || no_side) ? st : side->stream;  // the per-kernel profiler serialises everything on one stream
    if (c->padded) {
        // the forward's zero-padded copies of the input and of the result are still in the workspace; the incoming gradient is
        // padded with zeros here (the head's backward then gives the row padding a zero gradient)
        if (layer_end == NL &&
            (hipMemsetAsync(ws + c->pad_gout, 0, (size_t)c->B * c->T * sizeof(float), st) != hipSuccess ||
             hipMemcpy2DAsync(ws + c->pad_gout, (size_t)c->T * sizeof(float), grad_enhanced, (size_t)c->Tt * sizeof(float),
                              (size_t)c->Tt * sizeof(float), (size_t)c->B, hipMemcpyDeviceToDevice, st) != hipSuccess))
            return fail(WUNET_E_RUNTIME, "padding the output gradient failed");
        noisy = ws + c->pad_in; enhanced = ws + c->pad_out; grad_enhanced = ws + c->pad_gout;
    }

    if (layer_end == NL) {
        // flipped/transposed weights for every data gradient: already enqueued on the side stream by the training forward of this
        // workspace (they only depend on the weights: off the critical path), else packed here
        if (side->packed_ws == workspace && sd != st) {
            side->packed_ws = nullptr;
            if (hipStreamWaitEvent(st, side->ev_pack, 0) != hipSuccess) return fail(WUNET_E_RUNTIME, "waiting for the weight packs failed");
        } else {
            side->packed_ws = nullptr;
            const int rc = launch_backward_packs(c, params, ws, st);
            if (rc) return rc;
        }
        // head backward: gh = gout * tanh', d wh, d bh
        const LayerPlan& l = c->ly[NL - 1];
        HeadBwdArgs h{};
        h.z = ws + l.z; h.a = ws + l.a; h.s = ws + l.s; h.in = noisy; h.out = enhanced; h.gout = grad_enhanced;
        h.gh = ws + c->gh_off; h.part = ws + c->hpart_off; h.B = c->B; h.C = c->ci; h.T = c->T; h.logT = ilog2(c->T);
        WUNET_LAUNCH(head_bwd_kernel, dim3(c->head_blocks), dim3(WUNET_THREADS), 0, st, h);
        WUNET_CHECK_LAUNCH();
        // d(weight of the input channel), d bias; the ci per-channel weight gradients come out of the last layer's pass A
        WUNET_LAUNCH(rows_sum_kernel, dim3(2), dim3(WUNET_THREADS), 0, st,
                     (const float*)(ws + c->hpart_off), c->head_blocks, 2, grads[4 * NL] + c->ci, 1, grads[4 * NL + 1]);
        WUNET_CHECK_LAUNCH();
    }

    for (int i = layer_end - 1; i >= layer_begin; --i) {
        const LayerPlan& l = c->ly[i];
        // ---- pass A: assemble dL/d(BN output), LeakyReLU', BN-backward partial sums
        PassAArgs p{};
        p.z = ws + l.z; p.a = ws + l.a; p.s = ws + l.s; p.mean = ws + l.mean; p.rstd = ws + l.rstd;
        p.gpre = ws + l.g; p.part = ws + c->bpart_off; p.pmax = (i > 0 && l.h3d) ? ws + c->bmax_off : nullptr; p.B = c->B; p.C = l.cout; p.L = l.L; p.logL = l.logL; p.Lt = l.Lt;
        const dim3 ga(l.cout, l.a_split);
        const bool tiny = l.L < 4;
        // the first layer's g_z has one reader: its weight gradient forms it while it stages its chunks (WUNET_NO_GZ_FUSE: A/B switch)
        const bool gz_in_wgrad = i == 0 && !tiny && !l.h3d && !l.h3w && l.w.wsplit && !getenv("WUNET_NO_GZ_FUSE") &&
                                 !(l.a_split == 1 && (size_t)c->B * l.L <= 4 * WUNET_THREADS);       // (not where pass A finishes g_z itself)
        // a whole channel in one pass of one block (the levels of <= 16 samples at batch 64): BatchNorm-backward finalize and
        // g_z inside pass A - two launches of ~5 us (latency, not bandwidth) less per such level
        const bool fuse = !tiny && i < NL - 1 && !(i > 0 && l.h3d) && l.a_split == 1 && (size_t)c->B * l.L <= 4 * WUNET_THREADS &&
                          !getenv("WUNET_NO_PASSA_FUSE");
        if (fuse) {
            p.gamma = params[4 * i + 2]; p.dgamma = grads[4 * i + 2]; p.dbeta = grads[4 * i + 3]; p.dbias = grads[4 * i + 1];
            p.k1 = ws + l.k1; p.k2 = ws + l.k2; p.k3 = ws + l.k3; p.count = (double)c->B * l.Lt;
        }
        {   // algorithmic bytes of the gradient assembly (HBM-bound): z + the consumers' data gradients read, g written
            const double pe = (double)c->B * l.cout * l.L;
            const char* nm = i == NL - 1 ? "pass_a_kernel<HEAD>" : i >= n ? "pass_a_kernel<UP>" : "pass_a_kernel<ENC>";
            prof_begin(st, nm, 0.0, pe * (i == NL - 1 ? 8.0 : i >= n ? 16.0 : 14.0) + (i == NL - 1 ? 4.0 * c->B * l.L : 0.0));
        }
        if (i == NL - 1) {
            p.g0 = ws + c->gh_off; p.g1 = params[4 * NL]; p.hpart = ws + c->hpart2_off;
            WUNET_LAUNCH(pass_a_kernel<A_HEAD>, ga, dim3(WUNET_THREADS), 0, st, p);      // (the last layer has T >= 4 samples)
            prof_end(st);
            WUNET_CHECK_LAUNCH();
            WUNET_LAUNCH(rows_sum_kernel, dim3(c->ci), dim3(WUNET_THREADS), 0, st,
                         (const float*)(ws + c->hpart2_off), l.a_split, c->ci, grads[4 * NL], c->ci, grads[4 * NL + 1]);
        } else if (i >= n) {
            const LayerPlan& nx = c->ly[i + 1];
            p.g0 = ws + nx.dx; p.Cg0 = nx.cin;
            p.up_scale = (float)(l.Lt - 1) / (float)(2 * l.Lt - 1);
            p.no_fast = getenv("WUNET_NO_PASSA_FAST") ? 1 : 0;
            if (tiny) WUNET_LAUNCH(pass_a_scalar_kernel<A_UP>, ga, dim3(WUNET_THREADS), 0, st, p);
            else if (fuse) WUNET_LAUNCH((pass_a_kernel<A_UP, true>), ga, dim3(WUNET_THREADS), 0, st, p);
            else WUNET_LAUNCH(pass_a_kernel<A_UP>, ga, dim3(WUNET_THREADS), 0, st, p);
            prof_end(st);
        } else {
            const LayerPlan& dc = c->ly[2 * n - i];
            const LayerPlan& nx = c->ly[i + 1];
            p.g0 = ws + dc.dx; p.Cg0 = dc.cin; p.coff = dc.c0; p.g1 = ws + nx.dx;
            if (dx_stays_split(c, i + 1)) {
                p.g1 = ws + c->spart_off; p.g1_splits = nx.d.ksplit; p.g1_stride = (size_t)c->B * nx.cin * nx.L;
            }
            if (tiny) WUNET_LAUNCH(pass_a_scalar_kernel<A_ENC>, ga, dim3(WUNET_THREADS), 0, st, p);
            else if (fuse) WUNET_LAUNCH((pass_a_kernel<A_ENC, true>), ga, dim3(WUNET_THREADS), 0, st, p);
            else WUNET_LAUNCH(pass_a_kernel<A_ENC>, ga, dim3(WUNET_THREADS), 0, st, p);
            prof_end(st);
        }
        WUNET_CHECK_LAUNCH();
        if (!fuse) {
            BnBwdArgs b{};
            b.part = ws + c->bpart_off; b.rows = l.a_split; b.gamma = params[4 * i + 2]; b.mean = ws + l.mean; b.rstd = ws + l.rstd;
            b.dgamma = grads[4 * i + 2]; b.dbeta = grads[4 * i + 3]; b.dbias = grads[4 * i + 1]; b.k1 = ws + l.k1; b.k2 = ws + l.k2; b.k3 = ws + l.k3;
            b.C = l.cout; b.count = (double)c->B * l.Lt;
            b.pmax = (i > 0 && l.h3d) ? ws + c->bmax_off : nullptr; b.bound = ws + c->bound_off;
            // (short levels on the split kernels: the finalize runs in the prologue of gz_split_h3_kernel's blocks instead -
            //  WUNET_NO_BWDFIN_FUSE=1: A/B switch)
            static const int fin_loads = getenv("WUNET_GZ_FIN_LOADS") ? atoi(getenv("WUNET_GZ_FIN_LOADS")) : WUNET_GZ_FIN_LOADS;      // (sweep switch)
            const bool fin_in_gz = i > 0 && l.h3d && l.a_split * l.cout <= fin_loads && l.cout <= WUNET_GZ_FIN_C && !getenv("WUNET_NO_BWDFIN_FUSE");
            if (!fin_in_gz) {
                WUNET_LAUNCH(bn_finalize_bwd_kernel, dim3(l.cout), dim3(WUNET_THREADS), 0, st, b);
                WUNET_CHECK_LAUNCH();
            }

            // ---- g_z = k1*g + k2*z + k3, materialised once for both gradient GEMMs (fp32 in place, or scaled hi/lo halves)
            {
                const size_t n4 = (size_t)c->B * l.cout * l.L / 4;
                size_t blocks = (n4 + WUNET_THREADS - 1) / WUNET_THREADS;
                if (blocks > 8192) blocks = 8192;
                if (i > 0 && l.h3d) {
                    const int c8 = (l.cout + 7) / 8;
                    const size_t nt = (size_t)c->B * c8 * (l.L / 4);
                    size_t hb = (nt + WUNET_THREADS - 1) / WUNET_THREADS;
                    if (hb > 8192) hb = 8192;
                    prof_begin(st, "gz_split_h3_kernel", 0.0, (double)c->B * l.cout * l.L * (8.0 + (c->bf ? 2.0 : 4.0)));
                    if (fin_in_gz)
                        WUNET_LAUNCH(gz_split_h3_kernel<true>, dim3((unsigned)hb), dim3(WUNET_THREADS), 0, st, (const float*)(ws + l.g), (const float*)(ws + l.z),
                                     (const float*)(ws + l.k1), (const float*)(ws + l.k2), (const float*)(ws + l.k3), (const float*)(ws + c->bound_off),
                                     ws + c->h3_slot + 8 + 4 * i, reinterpret_cast<wunet_half*>(ws + l.gzh), reinterpret_cast<wunet_half*>(ws + l.gzl),
                                     c->B, l.cout, c8, l.L, l.logL, c->bf, l.Lt, b);
                    else
                        WUNET_LAUNCH(gz_split_h3_kernel<false>, dim3((unsigned)hb), dim3(WUNET_THREADS), 0, st, (const float*)(ws + l.g), (const float*)(ws + l.z),
                                     (const float*)(ws + l.k1), (const float*)(ws + l.k2), (const float*)(ws + l.k3), (const float*)(ws + c->bound_off),
                                     ws + c->h3_slot + 8 + 4 * i, reinterpret_cast<wunet_half*>(ws + l.gzh), reinterpret_cast<wunet_half*>(ws + l.gzl),
                                     c->B, l.cout, c8, l.L, l.logL, c->bf, l.Lt, b);
                    prof_end(st);
                } else if (tiny)
                    WUNET_LAUNCH(gz_scalar_kernel, dim3((unsigned)((n4 * 4 + WUNET_THREADS - 1) / WUNET_THREADS) + 1), dim3(WUNET_THREADS), 0, st, (const float*)(ws + l.g),
                                 (const float*)(ws + l.z), (const float*)(ws + l.k1), (const float*)(ws + l.k2), (const float*)(ws + l.k3), l.cout, l.logL,
                                 (size_t)c->B * l.cout * l.L, ws + l.g);
                else if (gz_in_wgrad) {
                } else
                    WUNET_LAUNCH(gz_materialize_kernel, dim3((unsigned)blocks), dim3(WUNET_THREADS), 0, st, (const float*)(ws + l.g), (const float*)(ws + l.z),
                                 (const float*)(ws + l.k1), (const float*)(ws + l.k2), (const float*)(ws + l.k3), l.cout, l.logL, n4, ws + l.g, l.Lt);   // in place
                WUNET_CHECK_LAUNCH();
            }
        }
        // ---- weight gradient on the side stream: GEMM over positions on the materialised operands, split-K partials
        //      + deterministic reduce
        auto weight_gradient = [&]() -> int {
            if (sd != st) {
                if (hipEventRecord(side->ev_fork, st) != hipSuccess || hipStreamWaitEvent(sd, side->ev_fork, 0) != hipSuccess)
                    return fail(WUNET_E_RUNTIME, "fork onto the weight-gradient stream failed");
            }
            const float* xin = i == 0 ? noisy : ws + l.xin;
            const size_t nw = (size_t)l.cout * l.cin * l.taps;
            if (tiny) {
                WUNET_LAUNCH(tiny_wgrad_kernel, dim3((unsigned)((nw + WUNET_THREADS - 1) / WUNET_THREADS)), dim3(WUNET_THREADS), 0, sd,
                             (const float*)(ws + l.g), xin, grads[4 * i], c->B, l.cin, l.cout, l.L, l.taps);
                WUNET_CHECK_LAUNCH();
            } else {
                int rc;
                if (l.h3w)
                    rc = launch_wgrad_h3(l, reinterpret_cast<const wunet_half*>(ws + l.xh), reinterpret_cast<const wunet_half*>(ws + l.xl),
                                         reinterpret_cast<const wunet_half*>(ws + l.gzh), reinterpret_cast<const wunet_half*>(ws + l.gzl),
                                         ws + c->h3_slot + 8 + 4 * i, ws + c->fslot_off + (size_t)WUNET_SLOT_FLOATS * i, ws + c->h3_slot,
                                         ws + c->wgpart_off, c->B, sd, c->bf);
                else {
                    WgradArgs w = make_wgrad_args(xin, ws + l.g, ws + c->wgpart_off, c->B, l.cin, l.cout, l.L, l.taps, l.w.cps);
                    if (gz_in_wgrad) {
                        w.z = ws + l.z; w.k1 = ws + l.k1; w.k2 = ws + l.k2; w.k3 = ws + l.k3; w.Lt = l.Lt;
                    }
                    rc = launch_wgrad_any(l.taps, w, l.w, sd);
                }
                if (rc) return rc;
                WUNET_CHECK_LAUNCH();
                if (l.h3w) {
                    WgradH3ReduceArgs ra{};
                    ra.part = ws + c->wgpart_off; ra.part_stride = h3w_part_stride(l); ra.splits = l.h3w_ksplit; ra.dw = grads[4 * i];
                    ra.Cout = l.cout; ra.Cin = l.cin; ra.taps = l.taps; ra.mrep = l.h3w_mrep; ra.tw = l.taps == 15 ? 8 : 5;
                    ra.nblocks = l.h3w_nblocks; ra.mblocks = l.h3w_mblocks; ra.cib = l.taps == 15 ? 32 : 64;
                    // A/B / test switch: WUNET_REDUCE_SERIAL = "<max splits>,<min float4 outputs>" (default 64,8192; measured
                    // 0,0 / 32,32768 / 64,8192: 6.19 / 6.17 / 6.13 ms per step, reduces 282 -> ~190 us per step on the side stream)
                    int serial_max = 64; long long serial_min_n4 = 8192;
                    if (const char* e = getenv("WUNET_REDUCE_SERIAL")) sscanf(e, "%d,%lld", &serial_max, &serial_min_n4);
                    const size_t n4 = ra.part_stride / 4;
                    if (ra.splits <= serial_max && (long long)n4 >= serial_min_n4) {
                        size_t blocks = (n4 + WUNET_THREADS - 1) / WUNET_THREADS;
                        if (blocks > 4096) blocks = 4096;
                        WUNET_LAUNCH(wgrad_h3_reduce_serial_kernel, dim3((unsigned)blocks), dim3(WUNET_THREADS), 0, sd, ra);
                    } else {
                        size_t blocks = (n4 + 15) / 16;
                        if (blocks > 4096) blocks = 4096;
                        WUNET_LAUNCH(wgrad_h3_reduce_kernel, dim3((unsigned)blocks), dim3(WUNET_THREADS), 0, sd, ra);
                    }
                } else {
                    size_t blocks = (nw / 4 + 15) / 16;                   // 16 float4 groups of outputs per block
                    if (blocks > 4096) blocks = 4096;
                    if (blocks < 1) blocks = 1;
                    WUNET_LAUNCH(wgrad_reduce_kernel, dim3((unsigned)blocks), dim3(WUNET_THREADS), 0, sd,
                                 (const float*)(ws + c->wgpart_off), l.w.rows, nw, grads[4 * i]);
                }
                WUNET_CHECK_LAUNCH();
            }
            return 0;
        };
        // fork point of the side stream (A/B switch WUNET_FORK_LATE=1: after this layer's data gradient is enqueued instead of
        // before it - the weight gradient then runs beside the next layer's HBM-bound gradient assembly, not beside the other GEMM)
        static const bool fork_late = getenv("WUNET_FORK_LATE") != nullptr;
        if (!fork_late || i == 0) { const int rc = weight_gradient(); if (rc) return rc; }
        // ---- data gradient (not needed for the first layer): the same conv kernel on the flipped/transposed pack
        if (i > 0 && l.h3d) {
            // fp16-split data gradient: scale g_z by a power of two into fp16's range, split, 3 MFMA passes, un-scale
            float* sc = ws + c->h3_slot + 8 + 4 * i;
            wunet_half* gh = reinterpret_cast<wunet_half*>(ws + l.gzh);
            wunet_half* gl = reinterpret_cast<wunet_half*>(ws + l.gzl);
            const bool split = l.d.ksplit > 1;
            int rc = launch_conv_h3(l.taps, l.h3d_mrep, l.h3d_mtp, l.h3d_sps, gh, gl,
                                    reinterpret_cast<const wunet_half*>(ws + c->h3_wb_hi) + l.h3d_wpk,
                                    reinterpret_cast<const wunet_half*>(ws + c->h3_wb_lo) + l.h3d_wpk, nullptr, sc,
                                    ws + c->fslot_off + (size_t)WUNET_SLOT_FLOATS * i + 2,
                                    split ? ws + c->spart_off : ws + l.dx, nullptr, c->B, l.cin, l.cout, l.h3d_nch, l.L, st, nullptr, nullptr,
                                    nullptr, c->bf, l.h3d_ntt);
            if (rc) return rc;
            WUNET_CHECK_LAUNCH();
            if (split && !dx_stays_split(c, i)) {
                const size_t nd = (size_t)c->B * l.cin * l.L;
                size_t blocks = (nd + WUNET_THREADS - 1) / WUNET_THREADS;
                if (blocks > 2048) blocks = 2048;
                WUNET_LAUNCH(split_sum_kernel, dim3((unsigned)blocks), dim3(WUNET_THREADS), 0, st, (const float*)(ws + c->spart_off), l.d.ksplit, nd, ws + l.dx, (const float*)nullptr, 1, 0);
                WUNET_CHECK_LAUNCH();
            }
        } else if (i > 0 && tiny) {
            const size_t nd = (size_t)c->B * l.cin * l.L;
            WUNET_LAUNCH(tiny_conv_kernel, dim3((unsigned)((nd + WUNET_THREADS - 1) / WUNET_THREADS)), dim3(WUNET_THREADS), 0, st,
                         (const float*)(ws + l.g), params[4 * i], ws + l.dx, c->B, l.cout, l.cin, l.L, l.logL, l.taps, 1);
            WUNET_CHECK_LAUNCH();
        } else if (i > 0) {
            const bool split = l.d.ksplit > 1;
            const size_t nd = (size_t)c->B * l.cin * l.L;
            const ConvArgs a = make_conv_args(ws + l.g, l.cout, ws + c->wpkb_off + l.d_wpk, nullptr,
                                              split ? ws + c->spart_off : ws + l.dx, nullptr, c->B, l.cin, l.L, l.taps, l.d, nd);
            int rc = launch_conv(l.taps, "dgrad", a, l.d, st);
            if (rc) return rc;
            WUNET_CHECK_LAUNCH();
            if (split && !dx_stays_split(c, i)) {
                size_t blocks = (nd + WUNET_THREADS - 1) / WUNET_THREADS;
                if (blocks > 2048) blocks = 2048;
                WUNET_LAUNCH(split_sum_kernel, dim3((unsigned)blocks), dim3(WUNET_THREADS), 0, st, (const float*)(ws + c->spart_off), l.d.ksplit, nd, ws + l.dx, (const float*)nullptr, 1, 0);
                WUNET_CHECK_LAUNCH();
            }
        }
        if (fork_late && i != 0) { const int rc = weight_gradient(); if (rc) return rc; }
    }
    // join: the caller's stream sees every weight gradient.  An un-joined range (wunet_backward_range_async) leaves them to
    // wunet_backward_join - except the range that ends the backward, which always joins: the next forward overwrites the
    // operands the side stream is still reading.
    if (sd != st && (join || layer_begin == 0)) {
        if (hipEventRecord(side->ev_join, sd) != hipSuccess || hipStreamWaitEvent(st, side->ev_join, 0) != hipSuccess)
            return fail(WUNET_E_RUNTIME, "join of the weight-gradient stream failed");
    }
    return WUNET_OK;
}
}  // namespace

extern "C" {

int wunet_backward_range(wunet_ctx* c, const float* noisy, const float* const* params, const float* enhanced,
                         const float* grad_enhanced, void* workspace, float* const* grads,
                         int layer_begin, int layer_end, void* stream)
{
    return backward_range_impl(c, noisy, params, enhanced, grad_enhanced, workspace, grads, layer_begin, layer_end, stream, true);
}

int wunet_backward_range_async(wunet_ctx* c, const float* noisy, const float* const* params, const float* enhanced,
                               const float* grad_enhanced, void* workspace, float* const* grads,
                               int layer_begin, int layer_end, void* stream)
{
    return backward_range_impl(c, noisy, params, enhanced, grad_enhanced, workspace, grads, layer_begin, layer_end, stream, false);
}

int wunet_backward_join(wunet_ctx* c, void* stream)
{
    if (!c) return fail(WUNET_E_ARG, "null ctx");
    wunet_ctx::Side* side = side_for_current_device(c);
    if (!side) return WUNET_E_RUNTIME;
    static const bool no_side = getenv("WUNET_NO_SIDE_STREAM") != nullptr;
    if (g_prof_on || no_side) return WUNET_OK;                   // the weight gradients ran on the caller's stream
    if (hipEventRecord(side->ev_join, side->stream) != hipSuccess ||
        hipStreamWaitEvent((hipStream_t)stream, side->ev_join, 0) != hipSuccess)
        return fail(WUNET_E_RUNTIME, "join of the weight-gradient stream failed");
    return WUNET_OK;
}

int wunet_backward(wunet_ctx* c, const float* noisy, const float* const* params, const float* enhanced,
                   const float* grad_enhanced, void* workspace, float* const* grads, void* stream)
{
    if (!c) return fail(WUNET_E_ARG, "null ctx");
    return backward_range_impl(c, noisy, params, enhanced, grad_enhanced, workspace, grads, 0, c->NL, stream, true);
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

// ---------------------------------------------------------------------------- fused Adam
int wunet_adam_step(int n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                    float* const* exp_avg_sq, const size_t* numels, double lr, double beta1, double beta2, double eps,
                    long long step, double grad_scale, long long* step_dev, float* hyper_dev, void* stream)
{
    if (n_tensors < 0 || (n_tensors > 0 && (!params || !grads || !exp_avg || !exp_avg_sq || !numels))) return fail(WUNET_E_ARG, "null argument");
    if (!step_dev && step < 1) return fail(WUNET_E_ARG, "step must be >= 1");
    if (step_dev && !hyper_dev) return fail(WUNET_E_ARG, "a device step counter needs the 2-float hyper buffer");
    hipStream_t st = (hipStream_t)stream;
    float step_size = 0.0f, bc2_sqrt = 1.0f;
    if (step_dev) {
        WUNET_LAUNCH(adam_hyper_kernel, dim3(1), dim3(1), 0, st, step_dev, lr, beta1, beta2, hyper_dev);
        WUNET_CHECK_LAUNCH();
    } else {
        const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
        step_size = (float)(lr / bc1); bc2_sqrt = (float)sqrt(bc2);
    }
    for (int base = 0; base < n_tensors; base += WUNET_ADAM_MAX) {
        AdamTable T{};
        const int cnt = n_tensors - base < WUNET_ADAM_MAX ? n_tensors - base : WUNET_ADAM_MAX;
        size_t nmax = 0;
        for (int k = 0; k < cnt; ++k) {
            if (numels[base + k] >= (1ull << 32)) return fail(WUNET_E_ARG, "tensor too large");
            T.p[k] = params[base + k]; T.g[k] = grads[base + k]; T.m[k] = exp_avg[base + k]; T.v[k] = exp_avg_sq[base + k];
            T.n[k] = (unsigned)numels[base + k];
            if (numels[base + k] > nmax) nmax = numels[base + k];
        }
        size_t bx = (nmax + WUNET_THREADS * 4 - 1) / (WUNET_THREADS * 4);
        if (bx < 1) bx = 1;
        if (bx > 256) bx = 256;
        WUNET_LAUNCH(adam_kernel, dim3((unsigned)bx, cnt), dim3(WUNET_THREADS), 0, st, T, (float)(1.0 - beta1), (float)beta2,
                     (float)(1.0 - beta2), bc2_sqrt, (float)eps, step_size, (float)grad_scale, (const float*)(step_dev ? hyper_dev : nullptr));
        WUNET_CHECK_LAUNCH();
    }
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

static int op_conv_common(const float* x, const float* w, const float* bias, float* out, int B, int kch, int rows,
                          int Cout, int Cin, int L, int K, int transposed, hipStream_t st)
{
    const ConvCfg cfg = plan_conv(B, L, rows, kch, K);
    float *wpk = nullptr, *part = nullptr;
    if (hipMalloc((void**)&wpk, (size_t)cfg.mtiles_p * cfg.cp * K * 16 * sizeof(float)) != hipSuccess) return fail(WUNET_E_RUNTIME, "hipMalloc");
    const size_t nout = (size_t)B * rows * L;
    if (cfg.ksplit > 1 && hipMalloc((void**)&part, (size_t)cfg.ksplit * nout * sizeof(float)) != hipSuccess) return fail(WUNET_E_RUNTIME, "hipMalloc");
    PackTable tab{};
    PackDesc& d = tab.d[0];
    d.w = w; d.dst = wpk; d.Cout = Cout; d.Cin = Cin; d.taps = K; d.M = rows; d.CP = cfg.cp; d.mtiles = cfg.mtiles_p; d.transposed = transposed;
    WUNET_LAUNCH(pack_weights_kernel, dim3(64, 1), dim3(WUNET_THREADS), 0, st, tab);
    const bool split = cfg.ksplit > 1;
    const ConvArgs a = make_conv_args(x, kch, wpk, split ? nullptr : bias, split ? part : out, nullptr, B, rows, L, K, cfg, nout);
    int rc = launch_conv(K, "op", a, cfg, st);
    if (!rc && split) {
        size_t blocks = (nout + WUNET_THREADS - 1) / WUNET_THREADS;
        if (blocks > 2048) blocks = 2048;
        WUNET_LAUNCH(split_sum_kernel, dim3((unsigned)blocks), dim3(WUNET_THREADS), 0, st, (const float*)part, cfg.ksplit, nout, out,
                     bias, rows, ilog2(L));
    }
    hipStreamSynchronize(st);
    hipFree(wpk);
    if (part) hipFree(part);
    if (rc) return rc;
    WUNET_CHECK_LAUNCH();
    return WUNET_OK;
}

int wunet_op_conv1d(const float* x, const float* w, const float* bias, float* z, int B, int Cin, int Cout, int L, int K, void* stream)
{
    if (op_check(B, Cin, Cout, L, K)) return WUNET_E_ARG;
    return op_conv_common(x, w, bias, z, B, Cin, Cout, Cout, Cin, L, K, 0, (hipStream_t)stream);
}

int wunet_op_conv1d_dgrad(const float* gz, const float* w, float* dx, int B, int Cin, int Cout, int L, int K, void* stream)
{
    if (op_check(B, Cin, Cout, L, K)) return WUNET_E_ARG;
    return op_conv_common(gz, w, nullptr, dx, B, Cout, Cin, Cout, Cin, L, K, 1, (hipStream_t)stream);
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

// ---- single-op entry points of the fp16-split kernels: the same planner, operand passes and GEMM kernels the network
//      uses for that geometry (scales from the measured maxima, as in eval mode)
namespace {
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) hipFree(p); }
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;            // (a kernel-launch lambda must capture the raw pointer, never the owner)
    bool alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16) == hipSuccess; }
    float* f() const { return (float*)p; }
    wunet_half* h() const { return (wunet_half*)p; }
};

int op_split_check(int B, int Cin, int Cout, int L, int K, bool backward)
{
    if (op_check(B, Cin, Cout, L, K)) return WUNET_E_ARG;
    if (L < 16 || (long long)B * L < 256) return fail(WUNET_E_ARG, "the split kernels need L >= 16 and B*L >= 256");
    if (backward && Cin < 16) return fail(WUNET_E_ARG, "the split data / weight gradient needs Cin >= 16");
    return 0;
}

// fp32 [B][C][L] -> scaled hi / lo in the split layout; slot: 8 floats, [0..1] receive {scale, 1/scale}, [4] the measured max
int op_split_operand(const float* x, int B, int C, int L, DevBuf& hi, DevBuf& lo, float* slot, const float* ones, const float* zeros, hipStream_t st)
{
    const int c8 = (C + 7) / 8;
    if (!hi.alloc((size_t)B * c8 * L * 16) || !lo.alloc((size_t)B * c8 * L * 16)) return fail(WUNET_E_RUNTIME, "hipMalloc");
    hipMemsetAsync(slot, 0, WUNET_SLOT_FLOATS * sizeof(float), st);
    const size_t n4 = (size_t)B * C * L / 4;
    size_t blocks = (n4 + WUNET_THREADS * 4 - 1) / (WUNET_THREADS * 4);
    if (blocks > 2048) blocks = 2048;
    WUNET_LAUNCH(act_max_kernel, dim3((unsigned)blocks), dim3(WUNET_THREADS), 0, st, x, ones, zeros, C, ilog2(L), n4, slot + 4);
    return launch_split(x, hi.h(), lo.h(), nullptr, slot + 4, nullptr, slot, B, C, L, st);
}

int op_conv_split_common(const float* x, const float* w, const float* bias, float* out, int B, int kch, int rows, int Cout, int Cin,
                         int L, int K, int transposed, hipStream_t st)
{
    const H3ConvPlan p = plan_h3_conv(B, L, rows, kch, K, transposed ? "WUNET_H3D_ORDER" : "WUNET_H3_ORDER");
    DevBuf xh, xl, wh, wl, misc, part;
    const size_t wh_halfs = (size_t)p.mtp * p.nch * K * 512, nout = (size_t)B * rows * L;
    // misc: slot of x (8 floats) | slot of w (8) | partial weight maxima | ones (kch) | zeros (kch)
    if (!wh.alloc(wh_halfs * 2) || !wl.alloc(wh_halfs * 2) || !misc.alloc((16 + WUNET_WMAX_PARTS + 2 * (size_t)kch) * sizeof(float)) ||
        (p.ksplit > 1 && !part.alloc((size_t)p.ksplit * nout * sizeof(float))))
        return fail(WUNET_E_RUNTIME, "hipMalloc");
    float* xslot = misc.f(), *wslot = misc.f() + 8, *wmax = misc.f() + 16, *oz = misc.f() + 16 + WUNET_WMAX_PARTS;
    WUNET_LAUNCH(fill_kernel, dim3(4), dim3(WUNET_THREADS), 0, st, oz, (size_t)kch, 1.0f);
    WUNET_LAUNCH(fill_kernel, dim3(4), dim3(WUNET_THREADS), 0, st, oz + kch, (size_t)kch, 0.0f);
    int rc = op_split_operand(x, B, kch, L, xh, xl, xslot, oz, oz + kch, st);
    if (rc) return rc;
    {
        ScaleTable T{};
        T.d[0].w = w; T.d[0].wn = (unsigned)((size_t)Cout * Cin * K); T.d[0].gamma = oz; T.d[0].beta = oz; T.d[0].C = 1; T.d[0].sqrtn = 0.0f;
        T.wmax = wmax; T.slots = wslot; T.training = 0;          // (clears wslot[4], which nothing reads)
        WUNET_LAUNCH(h3_scales_kernel, dim3(WUNET_WMAX_PARTS, 1), dim3(WUNET_THREADS), 0, st, T);
        PackH3Table tab{};
        PackH3Desc& d = tab.d[0];
        d.w = w; d.hi = wh.h(); d.lo = wl.h(); d.Cout = Cout; d.Cin = Cin; d.taps = K; d.rows = rows; d.kch = kch; d.mtiles = p.mtp; d.nch = p.nch;
        d.ntt = p.ntt; d.nfull = ((kch + 7) / 8) / 4; d.ns = h3_stage_count(kch, K, p.ntt);
        d.transposed = transposed; d.wmax = wmax; d.wsc = wslot + 2;
        WUNET_LAUNCH(pack_h3_kernel, dim3(64, 1), dim3(WUNET_THREADS), 0, st, tab);
    }
    const bool split = p.ksplit > 1;
    rc = launch_conv_h3(K, p.mrep, p.mtp, p.sps, xh.h(), xl.h(), wh.h(), wl.h(), split ? nullptr : bias, xslot, wslot + 2,
                        split ? part.f() : out, nullptr, B, rows, kch, p.nch, L, st, nullptr, nullptr, nullptr, 0, p.ntt);
    if (!rc && split) {
        size_t blocks = (nout + WUNET_THREADS - 1) / WUNET_THREADS;
        if (blocks > 2048) blocks = 2048;
        const float* partp = part.f();
        const int ks = p.ksplit;
        WUNET_LAUNCH(split_sum_kernel, dim3((unsigned)blocks), dim3(WUNET_THREADS), 0, st, partp, ks, nout, out, bias, rows, ilog2(L));
    }
    hipStreamSynchronize(st);
    if (rc) return rc;
    WUNET_CHECK_LAUNCH();
    return WUNET_OK;
}
}  // namespace

int wunet_op_conv1d_split(const float* x, const float* w, const float* bias, float* z, int B, int Cin, int Cout, int L, int K, void* stream)
{
    if (op_split_check(B, Cin, Cout, L, K, false)) return WUNET_E_ARG;
    return op_conv_split_common(x, w, bias, z, B, Cin, Cout, Cout, Cin, L, K, 0, (hipStream_t)stream);
}

int wunet_op_conv1d_dgrad_split(const float* gz, const float* w, float* dx, int B, int Cin, int Cout, int L, int K, void* stream)
{
    if (op_split_check(B, Cin, Cout, L, K, true)) return WUNET_E_ARG;
    return op_conv_split_common(gz, w, nullptr, dx, B, Cout, Cin, Cout, Cin, L, K, 1, (hipStream_t)stream);
}

int wunet_op_conv1d_wgrad_split(const float* gz, const float* x, float* dw, int B, int Cin, int Cout, int L, int K, void* stream)
{
    if (op_split_check(B, Cin, Cout, L, K, true)) return WUNET_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    LayerPlan l{};
    l.cin = Cin; l.cout = Cout; l.taps = K; l.L = L; l.logL = ilog2(L);
    plan_h3_wgrad(l, B);
    DevBuf xh, xl, gh, gl, misc, part;
    const int cmax = Cin > Cout ? Cin : Cout;
    // misc: 8 zero floats (DMA zero page) | slot of g_z | slot of x | ones | zeros
    if (!misc.alloc((24 + 2 * (size_t)cmax) * sizeof(float)) || !part.alloc((size_t)l.h3w_ksplit * h3w_part_stride(l) * sizeof(float)))
        return fail(WUNET_E_RUNTIME, "hipMalloc");
    float* zero = misc.f(), *gslot = misc.f() + 8, *xslot = misc.f() + 16, *oz = misc.f() + 24;
    hipMemsetAsync(zero, 0, 8 * sizeof(float), st);
    WUNET_LAUNCH(fill_kernel, dim3(4), dim3(WUNET_THREADS), 0, st, oz, (size_t)cmax, 1.0f);
    WUNET_LAUNCH(fill_kernel, dim3(4), dim3(WUNET_THREADS), 0, st, oz + cmax, (size_t)cmax, 0.0f);
    int rc = op_split_operand(gz, B, Cout, L, gh, gl, gslot, oz, oz + cmax, st);
    if (!rc) rc = op_split_operand(x, B, Cin, L, xh, xl, xslot, oz, oz + cmax, st);
    if (!rc) rc = launch_wgrad_h3(l, xh.h(), xl.h(), gh.h(), gl.h(), gslot, xslot, zero, part.f(), B, st);
    if (!rc) {
        WgradH3ReduceArgs ra{};
        ra.part = part.f(); ra.part_stride = h3w_part_stride(l); ra.splits = l.h3w_ksplit; ra.dw = dw;
        ra.Cout = Cout; ra.Cin = Cin; ra.taps = K; ra.mrep = l.h3w_mrep; ra.tw = K == 15 ? 8 : 5;
        ra.nblocks = l.h3w_nblocks; ra.mblocks = l.h3w_mblocks; ra.cib = K == 15 ? 32 : 64;
        size_t blocks = (ra.part_stride / 4 + 15) / 16;
        if (blocks > 4096) blocks = 4096;
        WUNET_LAUNCH(wgrad_h3_reduce_kernel, dim3((unsigned)blocks), dim3(WUNET_THREADS), 0, st, ra);
    }
    hipStreamSynchronize(st);
    if (rc) return rc;
    WUNET_CHECK_LAUNCH();
    return WUNET_OK;
}

}  // extern "C"
