// Shape planning, workspace layout, context life cycle, error text and the per-launch profiler of libwunet_hip.so
// (see wunet_host.h for the map of the host side).
#include "wunet_host.h"
#include "wunet_h3_elem.h"      // WUNET_SLOT_FLOATS, WUNET_WMAX_PARTS

namespace wunet_host {

namespace { thread_local std::string g_err; }

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

const char* last_error_text() { return g_err.c_str(); }

// ---- optional per-launch profiler (HIP events on the launch stream), used by bench.py's roofline leg
namespace {
struct ProfRec { std::string name; double flops; double bytes; hipEvent_t e0, e1; };
std::vector<ProfRec> g_prof;
}
bool g_prof_on = false;
unsigned long long* g_h3_trace = nullptr;

#ifdef WUNET_EMU
// (the emulator has no events: the records only say WHICH annotated kernels ran, how often - the tests ask that)
void prof_begin(hipStream_t, const char* name, double flops, double bytes)
{
    if (!g_prof_on) return;
    ProfRec r{};
    r.name = name; r.flops = flops; r.bytes = bytes;
    g_prof.push_back(r);
}
void prof_end(hipStream_t) {}
long long prof_collect(char* buf, size_t cap)
{
    std::vector<std::pair<std::string, long>> agg;
    for (const ProfRec& r : g_prof) {
        size_t k = 0;
        while (k < agg.size() && agg[k].first != r.name) ++k;
        if (k == agg.size()) agg.push_back({r.name, 0});
        ++agg[k].second;
    }
    g_prof.clear();
    std::string out;
    for (const auto& a : agg) out += a.first + "\t" + std::to_string(a.second) + "\t0\t0\t0\n";
    if (cap) { const size_t n = out.size() < cap - 1 ? out.size() : cap - 1; memcpy(buf, out.data(), n); buf[n] = 0; }
    return (long long)agg.size();
}
#else
void prof_begin(hipStream_t st, const char* name, double flops, double bytes)
{
    if (!g_prof_on) return;
    ProfRec r;
    r.name = name; r.flops = flops; r.bytes = bytes;
    hipEventCreate(&r.e0);
    hipEventCreate(&r.e1);
    hipEventRecord(r.e0, st);
    g_prof.push_back(r);
}
void prof_end(hipStream_t st)
{
    if (!g_prof_on) return;
    hipEventRecord(g_prof.back().e1, st);
}
// One line per kernel name: "name\tlaunches\ttotal_ms\ttotal_flops\ttotal_bytes\n"; clears the records.  Synchronises the device.
long long prof_collect(char* buf, size_t cap)
{
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
}
#endif

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
// blocks per layer of the weight-pack launches (grid-stride loops inside)
unsigned pack_gx() { return 512; }      // (swept 64 .. 1024 in round 1: 6.19 -> 6.13 ms per step at 512)

int pick_mrep_h3(int mtiles, const char* env, const char* dflt)
{
    int best = 2, best_pad = 1 << 30;
    const char* ord = env ? getenv(env) : nullptr;    // test hook (WUNET_H3_ORDER / _H3D_ORDER / _H3W_ORDER): tiny shapes reach every rows-per-wave instantiation
    if (!ord) ord = dflt;
    for (const char* p = ord; *p; ++p) {
        const int m = *p - '0';
        const int pad = round_up(mtiles, m) - mtiles;
        if (pad < best_pad) { best_pad = pad; best = m; }
    }
    return best;
}

// Preference order of the accumulator rows per wave (16 * M_REP output rows per block) of conv_h3d_kernel.  3 and 2 keep the
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

// dynamic LDS of one conv_h3d_kernel block: x tile [planes][4][NSEG * (256 / NSEG + 16)] + W sub-tile [planes][M_REP][5][64] pieces of 16 bytes
// + the statistics hand-over
// (un-segmented tiles: + the epilogue's per-row constants - bias, in eval mode also the BatchNorm scale / shift - of the mtp * 16 padded rows)
size_t h3d_smem(int nseg, int mrep, int bf, int mtp, int eval, int bsum)
{
    const int npl = bf ? 1 : 2;
    // (BSUM: four hand-over floats per row instead of two, no table of constants)
    return (size_t)(npl * 4 * nseg * (256 / nseg + 16) + npl * mrep * 5 * 64) * 16
         + (size_t)(WUNET_WAVES * mrep * (bsum ? 64 : 32) + 4 + ((nseg == 1 && !bsum) ? (eval ? 3 : 1) * mtp * 16 : 0)) * sizeof(float);
}
// blocks of that size a CU holds (launch bounds: two; the 16-segment tile's 96 KB: one) - the table of constants never costs the second block
// (launch_conv_h3 leaves the eval part in global memory where it would)
int h3d_blocks_per_cu(int nseg, int mrep, int bf) { return 2 * h3d_smem(nseg, mrep, bf, 0, 0) <= 160u * 1024u ? 2 : 1; }

// conv_h3d_kernel split-K of the levels with fewer work items than resident blocks: the K stages are split so that ONE round of
// resident blocks covers the layer, every block with (nearly) the same number of stages.  Round 4's block start / end times
// (tools/conv_bench.py --trace on the real-time counter) showed what the old rule (about 512 blocks, whatever fits) cost: the
// 16-sample levels hold one 96 KB block per CU, so 288 / 432 blocks ran in two rounds (19 us instead of 10); at 64 samples 96
// persistent blocks walked 112 items, 16 of them two (24 us instead of 13).  Cost of a candidate = rounds x (fixed part of a block:
// first fetch + epilogue, about 5 us, + 1.8 us per stage) + the partial sums the reduce kernel reads; returns the stages per split
// (== nstage: no split).
int h3_stages_per_split(int items, int nstage, int slots)
{
    if (items >= 384 || nstage <= 1 || getenv("WUNET_H3_NOSPLIT")) return nstage;      // (switch: tests reach the un-split epilogue on small shapes)
    if (const char* e = getenv("WUNET_H3_SPS")) {    // measurement hook (tools/conv_bench.py sweeps it per layer): the stages per split
        const int v = atoi(e);
        if (v >= 1) return v < nstage ? v : nstage;
    }
    if (getenv("WUNET_H3_OLDSPLIT")) {              // A/B switch: the rule of rounds 1 - 3
        int ks = (512 + items - 1) / items;
        if (ks > nstage) ks = nstage;
        return (nstage + ks - 1) / ks;
    }
    int best_sps = nstage;
    double best = 1e30;
    for (int sps = nstage; sps >= 1; --sps) {
        const int ks = (nstage + sps - 1) / sps;
        const long long blocks = (long long)items * ks;
        const long long rounds = (blocks + slots - 1) / slots;
        const double cost = (double)rounds * (5.0 + 1.8 * sps) + 0.12 * ks;
        if (cost < best - 1e-9) { best = cost; best_sps = sps; }
    }
    return best_sps;
}

// K tail of conv_h3d_kernel: with c8 = 4 nfull + t1 groups of 8 K channels, the t1 left-over groups run one tail stage each (the taps
// spread over the four K quarters of the MFMA: ceil(taps / 4) steps) instead of one chunk padded with zeros (taps steps).  Worth it
// for 15 taps with any tail (4 t1 steps instead of 15), for 5 taps with one left-over group (2 instead of 5).
// Returns the steps of a tail stage, 0 without a tail.  WUNET_H3_KTAIL=0 (read when a context is planned): the padded chunk
// everywhere - the other arm of tests/test_gpu_parity.py::test_k_tail_on_hardware; the two shapes at the register limit always run it
int h3_tail_steps(int kch, int taps)
{
    const char* kt = getenv("WUNET_H3_KTAIL");
    const int t1 = ((kch + 7) / 8) & 3;
    if (kt && atoi(kt) == 0) return 0;
    return taps == 15 ? (t1 ? 4 : 0) : (t1 == 1 ? 2 : 0);
}
// stages of the K loop: full chunks * tap groups + tail stages
int h3_stage_count(int kch, int taps, int ntt)
{
    const int c8 = (kch + 7) / 8;
    return ntt ? (c8 / 4) * (taps / 5) + (c8 & 3) : ((c8 + 3) / 4) * (taps / 5);
}

H3ConvPlan plan_h3_conv(int B, int L, int rows, int kch, int taps, const char* order_env, int bf)
{
    H3ConvPlan p{};
    const long long posn = (long long)B * L;
    p.ntiles = (int)((posn + 255) / 256);
    const int ntg = taps / 5, c8 = (kch + 7) / 8, mt = (rows + 15) / 16;
    // (4 accumulator rows per wave only exist un-segmented: L >= 256)
    p.mrep = pick_mrep_h3(mt, L >= 256 ? order_env : nullptr, h3_order(taps, L, p.ntiles, mt));
    p.mtp = round_up(mt, p.mrep);
    p.nch = (c8 + 3) / 4;
    p.ntt = (p.mrep < 4 && L >= 32) ? h3_tail_steps(kch, taps) : 0;      // (WUNET_H3D_HAS_TAIL: the shapes at the register limit go without)
    const int ns = h3_stage_count(kch, taps, p.ntt);
    const int nseg = L >= 256 ? 1 : 256 / L;
    p.sps = h3_stages_per_split(p.ntiles * (p.mtp / p.mrep), ns, 256 * h3d_blocks_per_cu(nseg, p.mrep, bf));
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
    // 128 positions per K chunk (256 measured 5-8 % faster for the kernel alone and slower in the concurrent step, 64-position
    // double-buffered chunks neutral: HISTORY.md sections 7, 8)
    l.h3w_tp = 128;
    // wgrad_h3d_kernel<.., false> (single LDS buffer, two blocks per CU) where its registers allow: k5 up to 4 m-tiles, k15 up to 3
    const bool sb2 = l.L >= 128 && ((l.taps == 5 && l.h3w_mrep <= 4) || (l.taps == 15 && l.h3w_mrep <= 3));
    const long long slots = 256LL * ((l.h3w_mrep <= 2 || sb2) ? 2 : 1);      // resident blocks: launch bounds of the wgrad kernels
    long long ks = slots / ((long long)l.h3w_mblocks * l.h3w_nblocks);
    if (ks < 1) ks = 1;
    const long long chunks = ((long long)B * l.L + l.h3w_tp - 1) / l.h3w_tp;
    if (ks > chunks) ks = chunks;
    // XCD-aware walk of wgrad_h3d_kernel (WgradH3dArgs::xcd_walk): the nyz blocks of one K split share an XCD, so an XCD hosts
    // ceil(ks / 8) * nyz blocks - they must fit its share of the resident slots, else some blocks wait for a second round (measured:
    // wgrad_h3d_kernel<5, 3> 404 -> 493 us per step with 66 / 72 blocks on an XCD of 64 slots).  The split count is trimmed to fit when
    // that costs at most 6 % of the blocks; otherwise the layer keeps the 3-D grid.  WUNET_WGRAD_XCD=0: off (A/B switch).
    {
        static const bool xcd_off = getenv("WUNET_WGRAD_XCD") != nullptr && atoi(getenv("WUNET_WGRAD_XCD")) == 0;
        const long long nyz = (long long)l.h3w_mblocks * l.h3w_nblocks, per_xcd = slots / 8;
        l.h3w_xcd = 0;
        if (!xcd_off && nyz > 1 && l.L >= 128) {
            if (((ks + 7) / 8) * nyz <= per_xcd) l.h3w_xcd = 1;
            else {
                const long long ks8 = (per_xcd / nyz) * 8;
                if (ks8 >= 8 && ks8 * 100 >= ks * 94) { ks = ks8; l.h3w_xcd = 1; }
            }
        }
    }
    l.h3w_cps = (int)((chunks + ks - 1) / ks);
    l.h3w_ksplit = (int)((chunks + l.h3w_cps - 1) / l.h3w_cps);
}

// The x2 upsample's source pairs as the fused loaders assume them (conv_h3u_kernel, and the gradient-side kernels that gather through
// the same pairs): output j >= 1 of a row of Lin source samples reads (i0, i1) = ((j-1) >> 1, i0 + 1) under ATen's fp32 coordinates
// (wunet_up_coord's arithmetic, restated here on the host with every product rounded to fp32), output 0 reads source 0 with weight
// 1, and an output whose i0 is the row's last sample has weight exactly 0 on the sample behind the row.  True for every power-of-two
// length up to 4 M samples; checked per length when a context is planned (cached), and a layer whose length fails it keeps the
// two-kernel path.
bool up_pairs_regular(int Lin)
{
    static std::mutex lock;
    static std::map<int, bool> seen;
    std::lock_guard<std::mutex> g(lock);
    auto it = seen.find(Lin);
    if (it != seen.end()) return it->second;
    bool ok = Lin >= 2;
    const int Lout = 2 * Lin;
    const float scale = (float)(Lin - 1) / (float)(Lout - 1);
    for (int j = 0; ok && j < Lout; ++j) {
        volatile float src = scale * (float)j;                 // (volatile: the product is rounded to fp32 before it is used)
        int a = (int)floorf(src);
        a = a > Lin - 1 ? Lin - 1 : a;
        float lam = src - (float)a;
        lam = lam < 0.0f ? 0.0f : (lam > 1.0f ? 1.0f : lam);
        if (j == 0) ok = a == 0 && lam == 0.0f;
        else ok = a == ((j - 1) >> 1) && (a < Lin - 1 || lam == 0.0f);
    }
    return seen[Lin] = ok;
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
            // (the 16-sample level: 20-38 us per step faster on conv_h3d_kernel<., ., 16> than on the fp32 kernels)
            const int min_l = 16;
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
            // conv_h3u_kernel (wunet_h3u.h: the operand pass inside the conv, loader waves beside the MFMA waves) for the decoder levels
            // whose conv is bound by operand bytes, not by the matrix pipe.  WUNET_H3U = "<eval min L>,<train min L>" (0: off), read
            // when the context is planned
            {
                int u_eval = 512, u_train = 2048;    // (batch 64, one box - eval: off 1.87 ms, from 2048 1.567, from 512 1.516; training step: off 5.12 / 5.20, from 2048 or 4096 5.08 / 5.15, from 512 5.17: profiles/r5_h3u_threshold_sweep.txt)
                if (const char* e = getenv("WUNET_H3U")) {
                    int a = 0, b = 0;
                    const int got = sscanf(e, "%d,%d", &a, &b);
                    if (got == 2) { u_eval = a; u_train = b; }
                    else if (got == 1) u_eval = u_train = a;      // one number: both thresholds (WUNET_H3U=0 turns the kernel off everywhere)
                }
                const bool can = l.h3f && l.h3x && l.kind == LK_UPCAT && l.taps == 5 && l.L >= 256 && l.c0 % 8 == 0 && l.cin % 8 == 0 &&
                                 l.h3f_mrep <= 4 && l.f.ksplit == 1 && !c->bf && !c->padded && up_pairs_regular(l.L / 2);
                l.h3u = (can && u_eval > 0 && l.L >= u_eval) ? 1 : 0;
                l.h3u_train = (can && u_train > 0 && l.L >= u_train) ? 1 : 0;
            }
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
        l.wl1 = off; off += align64(l.cout);
        l.cst = off; off += align64((size_t)4 * l.cout);
        l.xin = off; if (i > 0 && !l.h3x) off += align64((size_t)B * l.cin * l.L);   // the conv's activated input, materialised once
    }
    // the operand pass of encoder-side layer i (decimation of its producer's activation) also writes that activation at full
    // resolution into the split input of the decoder layer that concatenates it: one read of the producer's z instead of two
    for (int i = 0; i < c->NL; ++i) c->ly[i].skip_from = 0;
    for (int i = 1; i <= c->n; ++i) {
        const int dj = 2 * c->n - i + 1;
        LayerPlan& e = c->ly[i];
        LayerPlan& d = c->ly[dj];
        if (e.kind == LK_DECIM && d.kind == LK_UPCAT && e.h3x && d.h3x && d.src1 == e.src0 && d.c0 % 8 == 0 && !getenv("WUNET_NO_SKIP_FUSE"))      // (test hook: the decoder-side pass reads the skip itself, eval mode's path)
            d.skip_from = i;
    }
    c->stats_off = off; off += align64(stats_max);
    c->wpkf_off = off; off += align64(wpk);
    c->spart_off = off; off += align64(spart_max);
    // ---- fp16-split path: split activations + forward weight packs live in the forward segment
    size_t wfh = 0;
    for (int i = 0; i < c->NL; ++i) {
        LayerPlan& l = c->ly[i];
        l.xh = l.xl = l.xzp = 0; l.h3f_wpk = l.h3d_wpk = 0;
        if (l.h3f) {
            const int c8 = (l.cin + 7) / 8;
            l.xh = off; off += align64((size_t)B * c8 * l.L * 4);
            l.xl = off; off += align64((size_t)B * c8 * l.L * 4);
            l.xzp = off; off += 64;
            l.h3f_wpk = wfh; wfh += (size_t)l.h3f_mtp * l.h3f_nch * l.taps * 512;
        }
    }
    c->h3_wf_halfs = wfh;
    c->h3_wf_hi = off; off += align64((wfh + 1) / 2);
    c->h3_wf_lo = off; off += align64((wfh + 1) / 2);
    c->fslot_off = off; off += align64((size_t)WUNET_SLOT_FLOATS * c->NL);
    c->wmax_off = off; off += align64((size_t)WUNET_WMAX_PARTS * c->NL);
    // eval mode: an encoder level's un-split split conv (15 taps, whole-row tiles) writes the next encoder level's operand in its epilogue
    // (conv_h3d_kernel<.., EVOP>; WUNET_NO_EVOP=1, read when the context is planned: A/B switch)
    for (int i = 0; i < c->NL; ++i) c->ly[i].evop = 0;
    for (int i = 1; i + 1 <= c->n; ++i) {
        LayerPlan& p = c->ly[i];
        const LayerPlan& q = c->ly[i + 1];
        if (p.h3f && !p.first && p.taps == 15 && p.L >= 256 && p.f.ksplit == 1 && p.h3f_mrep <= 4 && q.kind == LK_DECIM && q.src0 == i && q.h3f && q.h3x &&
            !c->bf && !c->padded && !getenv("WUNET_NO_EVOP"))
            p.evop = 1;
    }
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
        // pieces of a channel's B*L positions, one block of the gradient-assembly pass each: at least 4096 positions, at most 128 pieces
        // (round 4, with the blocks of one channel adjacent in the grid: 32 / 64 / 128 / 256 pieces -> 5.174 / 5.136 / 5.116 / 5.130 ms
        // per step; WUNET_A_CAP: measurement hook)
        static const int a_cap_env = getenv("WUNET_A_CAP") ? atoi(getenv("WUNET_A_CAP")) : 128;
        const int a_cap = a_cap_env < 1 ? 1 : (a_cap_env > 256 ? 256 : a_cap_env);      // (hpart2 below holds 256 rows)
        l.a_split = (int)(sp < 1 ? 1 : (sp > a_cap ? a_cap : sp));
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
    c->hpart2_off = off; off += align64((size_t)256 * ci);          // pass A (head mode) partial head-weight gradients [a_split][ci]
    c->e0part_off = off; off += align64((size_t)256 * ci * 48);     // pass A of the first layer (E0): its weight gradient's partial sums [a_split][ci][48]
    c->e0 = getenv("WUNET_NO_E0") ? 0 : 1;
    // ---- fp16-split data gradient: transposed packs, one shared split g_z buffer, scale slot
    size_t wbh = 0, gzs = 0;
    for (int i = 0; i < c->NL; ++i) {
        LayerPlan& l = c->ly[i];
        if (!l.h3d) continue;
        const int c8 = (l.cout + 7) / 8;
        l.h3d_wpk = wbh; wbh += (size_t)l.h3d_mtp * l.h3d_nch * l.taps * 512;
        l.gzh = off; off += align64((size_t)B * c8 * l.L * 4);      // per layer: the side stream reads it late
        l.gzl = off; off += align64((size_t)B * c8 * l.L * 4);
        l.gzp = off; off += 64;
    }
    // ---- BatchNorm-backward sums in the epilogue of the data gradient that produces dL/d(activation) (conv_h3d_kernel<.., BSUM>): for the
    // layers whose consumers' data gradients run un-split on whole-row tiles of at least WUNET_BSUM samples (read when the context is
    // planned).  Such a layer has no pass_a_kernel launch; its g_z is formed by gz_split_h3_kernel from the data gradients.
    // OFF by default (0): measured at batch 64 x 16384 the step is 0.57 - 0.70 ms SLOWER with it (profiles/r6_bsum_ab.txt: pass A -0.48 ms,
    // but the epilogue's loads sit exposed behind the K loop - no registers, no LDS to prefetch them - +0.65 ms on the data gradients, and the
    // transposed upsample gathered inside gz_split_h3_kernel +0.47 ms).  Kept as a measured, tested alternative.
    {
        int bs_min = 0;
        if (const char* e = getenv("WUNET_BSUM")) bs_min = atoi(e);
        const int n = c->n, NL = c->NL;
        auto dgrad_can = [&](const LayerPlan& q) {
            return bs_min > 0 && q.h3d && q.L >= 256 && q.L >= bs_min && q.d.ksplit == 1 && (q.cin & 3) == 0 && !c->bf && !c->padded;
        };
        for (int p = 0; p < NL; ++p) { c->ly[p].bsum = 0; c->ly[p].bs_kind = 0; c->ly[p].bsp = 0; }
        for (int p = 1; p + 1 < NL; ++p) {
            LayerPlan& l = c->ly[p];
            if (!l.h3d || l.L < 4) continue;            // (its own g_z on the split kernels: gz_split_h3_kernel's recompute modes)
            if (p >= n) {                               // feeds the x2 upsample of the next decoder layer
                const LayerPlan& q = c->ly[p + 1];
                if (q.kind == LK_UPCAT && q.src0 == p && q.c0 == l.cout && (q.c0 & 3) == 0 && dgrad_can(q) && up_pairs_regular(l.L)) l.bsum = 1;
            } else if (2 * n - p < NL) {                // an encoder layer: skip of decoder layer 2n - p, decimated into layer p + 1
                const LayerPlan& dq = c->ly[2 * n - p];
                const LayerPlan& eq = c->ly[p + 1];
                if (dq.kind == LK_UPCAT && dq.src1 == p && (dq.c0 & 3) == 0 && dq.cin - dq.c0 == l.cout && eq.kind == LK_DECIM && eq.src0 == p &&
                    eq.cin == l.cout && dgrad_can(dq) && dgrad_can(eq))
                    l.bsum = 1;
            }
        }
        for (int i = 1; i < NL; ++i) {
            LayerPlan& q = c->ly[i];
            if (!dgrad_can(q)) continue;
            if (q.kind == LK_UPCAT && (c->ly[q.src0].bsum || c->ly[q.src1].bsum)) q.bs_kind = 1;
            else if (q.kind == LK_DECIM && q.src0 >= 0 && c->ly[q.src0].bsum) q.bs_kind = 2;
            if (q.bs_kind) { q.bsp = off; off += align64((size_t)q.cin * (((size_t)B * q.L + 255) / 256) * 4); }
        }
    }
    // ---- UPT: a decoder layer's data gradient stores the rows of the upsampled half of its input pulled back through the upsample
    // (conv_h3d_kernel<.., 3>, ConvH3Args::uh_*): un-split whole-row tiles, the producer on the split kernels.  WUNET_UPT=0: off (A/B switch).
    {
        const bool upt_off = getenv("WUNET_UPT") != nullptr && atoi(getenv("WUNET_UPT")) == 0;      // (read when a context is planned)
        for (int i = 0; i < c->NL; ++i) { c->ly[i].upt = 0; c->ly[i].dxh = c->ly[i].usp = 0; }
        for (int i = c->n + 1; i < c->NL && !upt_off; ++i) {
            LayerPlan& q = c->ly[i];
            const LayerPlan& p = c->ly[q.src0];
            if (q.kind != LK_UPCAT || !q.h3d || q.bs_kind || q.L < 256 || q.d.ksplit != 1 || (q.c0 & 3) || (q.cin & 3) || q.c0 != p.cout || c->bf || c->padded ||
                !p.h3d || p.L < 128 || !up_pairs_regular(q.L / 2))
                continue;
            q.upt = 1;
            q.dxh = off; off += align64((size_t)B * q.c0 * (q.L / 2));
            q.usp = off; off += align64((size_t)2 * q.c0 * (((size_t)B * q.L + 255) / 256));
        }
    }
    c->h3_wb_halfs = wbh;
    c->h3_wb_hi = off; off += align64((wbh + 1) / 2);
    c->h3_wb_lo = off; off += align64((wbh + 1) / 2);
    (void)gzs;
    c->h3_slot = off; off += align64(8 + 4 * (size_t)c->NL);      // 8 zero floats (DMA zero page), then {scale, 1/scale} of g_z per layer (offset 8 + 4*layer)
    if (c->padded) { c->pad_gout = off; off += align64((size_t)B * T); }
    c->total_floats = off;
}

}  // namespace wunet_host

using namespace wunet_host;

extern "C" {

const char* wunet_last_error(void) { return last_error_text(); }

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

}  // extern "C"
