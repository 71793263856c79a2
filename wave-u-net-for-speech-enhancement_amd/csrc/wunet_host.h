// Host side of libwunet_hip.so, shared declarations: the per-shape plan (tilings, workspace layout), the context, and the
// launch helpers the forward / backward / single-op translation units call.  Nothing here is part of the C ABI
// (include/wunet_hip.h); no torch types, no hidden device allocations on the hot path, nothing synchronises the stream.
//   wunet_plan.cpp      shape planning, workspace layout, context life cycle, error text, per-launch profiler
//   wunet_launchers.cpp argument blocks + launches of the GEMM kernels, weight packs, side stream
//   wunet_forward.cpp   wunet_forward
//   wunet_backward.cpp  wunet_backward[_range[_async]], wunet_backward_join
//   wunet_ops.cpp       loss, fused Adam, window crops, profiler read-out, single-op entry points (tests / tools)
#pragma once
#include <cstdarg>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "wunet_kernels.h"
#include "wunet_h3.h"
#include "wunet_h3u.h"
#include "wunet_launch.h"
#include "wunet_hip.h"

namespace wunet_host {

int fail(int code, const char* fmt, ...);
const char* last_error_text();

#define WUNET_CHECK_LAUNCH()                                                                   \
    do {                                                                                       \
        hipError_t e_ = hipGetLastError();                                                     \
        if (e_ != hipSuccess) return wunet_host::fail(WUNET_E_RUNTIME, "%s:%d HIP error: %s", __FILE__, __LINE__, hipGetErrorString(e_)); \
    } while (0)

// ---- optional per-launch profiler (HIP events on the launch stream), used by bench.py's roofline leg
extern bool g_prof_on;
void prof_begin(hipStream_t st, const char* name, double flops, double bytes);
void prof_end(hipStream_t st);
long long prof_collect(char* buf, size_t cap);
extern unsigned long long* g_h3_trace;       // wunet_debug_set_conv_trace

inline int ilog2(long long v) { int r = 0; while ((1LL << r) < v) ++r; return r; }
inline bool is_pow2(long long v) { return v > 0 && (v & (v - 1)) == 0; }
inline size_t align64(size_t v) { return (v + 63) & ~(size_t)63; }
inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
inline int kc_of(int taps) { return taps == 15 ? 4 : 12; }

// Tiling of one implicit-GEMM conv: rows = output channels of the GEMM, kch = its K channels.
struct ConvCfg { int mrep, nrep, mtiles_p, mblocks, cp, grid_x, ksplit, kcps; };
ConvCfg plan_conv(int B, int L, int rows, int kch, int taps);
ConvArgs make_conv_args(const float* x, int kch, const float* wpk, const float* bias, float* out, float* stats,
                        int B, int rows, int L, int taps, const ConvCfg& c, size_t split_stride);
int launch_conv(int taps, const ConvArgs& a, const ConvCfg& c, hipStream_t st);

struct WgradCfg { int mrep, nw, xit, wsplit, mblocks, nblocks, ksplit, cps, rows; };
WgradCfg plan_wgrad(int B, int L, int cin, int cout, int taps);
WgradArgs make_wgrad_args(const float* x, const float* g, float* part, int B, int Cin, int Cout, int L, int taps, int cps);
int launch_wgrad_any(int taps, const WgradArgs& a, const WgradCfg& w, hipStream_t st);

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
    size_t wl1;          // cout absolute row sums of the conv weight (eval mode: the bound conv_h3d_kernel<.., EVOP> scales the next operand by)
    int evop;            // eval mode: this encoder layer's conv also writes the next layer's operand (no prep_h3_kernel<0> for that one)
    // fp16-split path
    int h3f, h3d;                 // forward conv / data gradient use conv_h3_kernel
    int h3f_mrep, h3f_mtp, h3f_nch, h3d_mrep, h3d_mtp, h3d_nch;
    int h3f_ntt, h3d_ntt;    // K tail of conv_h3d_kernel: steps of a tail stage (0: the last chunk is padded to 32 channels)
    int h3f_sps, h3d_sps;         // K stages per split of conv_h3_kernel (the split count is f.ksplit / d.ksplit)
    int first;                    // encoder[0] (Cin = 1): direct fp32 kernel (conv_first_kernel)
    int h3x;                      // the conv input exists only in the split layout (no fp32 xin)
    int skip_from;                // decoder layer: > 0 = its skip half is written by the operand pass of encoder-side layer skip_from
    int h3w, h3w_mrep, h3w_mblocks, h3w_nblocks, h3w_ksplit, h3w_cps, h3w_tp;   // weight gradient uses wgrad_h3_kernel
    int h3w_xcd;                  // wgrad_h3d_kernel walks its blocks XCD-aware (the blocks of one K split on one XCD)
    int h3u, h3u_train;           // conv_h3u_kernel (operand pass fused into the conv's loader waves) runs the eval / also the training forward
    int feeds_h3;                 // the layer's activation is the (or a) source of a conv input that exists in the split layout
    size_t xh, xl, xzp;           // split activated input (float offsets): hi plane, lo plane right behind it, then 16 zero bytes (DMA pad)
    size_t gzh, gzl, gzp;         // split scaled g_z (float offsets), likewise
    size_t h3f_wpk, h3d_wpk;      // half offsets inside the split weight packs
    // BatchNorm-backward sums out of the consumers' data-gradient epilogues (conv_h3d_kernel<.., BSUM>, ConvH3Args::bs_*)
    size_t cst;                   // [cout][4] {a, s, mean, rstd}: the layer's BatchNorm constants as one 16-byte row per channel (training forward)
    int upt;                      // a decoder layer whose data gradient stores its upsampled rows at the producer's resolution (conv_h3d_kernel<.., 3>)
    size_t dxh, usp;              // ... [B][c0][L/2] and the tile-edge terms [2][c0][tiles]
    int bsum;                     // THIS layer's sums come from its consumers' epilogues: no pass_a_kernel, g_z from the data gradients
    int bs_kind;                  // this layer's own data gradient takes sums for (some of) its producers: 1 decoder form, 2 encoder form
    size_t bsp;                   // ... and writes them here: [cin][tiles][4]
};

// Tiling of conv_h3d_kernel for one GEMM (rows x B*L positions, K = kch channels x taps): accumulator rows per wave, padded
// m-tiles, chunks of 32 K channels, K stages per split and the split count.  Shared by the network planner and the
// single-op entry points, so a geometry gets the same kernel instantiation either way.
struct H3ConvPlan { int mrep, mtp, nch, sps, ksplit, ntiles, ntt; };
H3ConvPlan plan_h3_conv(int B, int L, int rows, int kch, int taps, const char* order_env, int bf = 0);
size_t h3d_smem(int nseg, int mrep, int bf, int mtp, int eval, int bsum = 0);          // dynamic LDS of a conv_h3d_kernel block
int h3d_blocks_per_cu(int nseg, int mrep, int bf);   // resident blocks per CU at that size
int h3_stage_count(int kch, int taps, int ntt);
void plan_h3_wgrad(LayerPlan& l, int B);
size_t h3w_part_stride(const LayerPlan& l);
bool up_pairs_regular(int Lin);       // the x2 upsample reads the source pairs ((j-1) >> 1, +1) at this length (fp32 coordinates, host check)
unsigned pack_gx();

}  // namespace wunet_host

struct wunet_ctx {
    int n, ci, B, T, NL;
    int Tt = 0;                   // the caller's frame length (T: rounded up to a power of two, the row stride of every level)
    bool padded = false;          // Tt < T
    size_t pad_in = 0, pad_out = 0, pad_gout = 0;      // padded copies of noisy / enhanced / grad_enhanced (float offsets; padded only)
    std::vector<wunet_host::LayerPlan> ly;
    size_t stats_off, wpkf_off, spart_off, fwd_floats;
    size_t bmax_off, bound_off;   // pass A maxima / per-channel |g_z| bounds (fp16-split scale)
    size_t bpart_off, wgpart_off, wpkb_off, gh_off, hpart_off, hpart2_off, e0part_off, total_floats;
    int e0 = 1;                   // the first layer's weight gradient from pass A's sums (pass_a_kernel<.., E0>); WUNET_NO_E0=1, read when the context is planned: off
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

namespace wunet_host {

void layout_workspace(wunet_ctx* c);
int launch_split(const float* x, wunet_half* hi, wunet_half* lo, const float* sc, const float* xb0, const float* xb1, float* xsc,
                 int B, int C, int L, hipStream_t st, int bf = 0);
// (op: eval mode, the conv also writes the next encoder layer's operand - ConvH3Args::op_*, conv_h3d_kernel<.., EVOP>)
struct ConvH3OpOut { wunet_half* h; wunet_half* l; const float* wl1; const float* xmax; float* xsc; int C8; };
// (bs: training backward, the data gradient also takes the BatchNorm-backward sums of the layers that produced its rows - ConvH3Args::bs_*,
//  conv_h3d_kernel<.., BSUM = kind>; a producer with z == nullptr keeps pass_a_kernel)
struct ConvH3Bsum { int kind; const float* z[2]; const float* cst[2]; int C[2]; int c0; float up_scale; float* part;
                    float* uh_out; float* uh_spill; };     // kind 3 (UPT): the rows < c0 stored pulled back through the upsample (ConvH3Args::uh_*)
int launch_conv_h3(int taps, int mrep, int mtiles_p, int sps, const wunet_half* xh, const wunet_half* xl, const wunet_half* wh,
                   const wunet_half* wl, const float* bias, const float* sc, const float* sc2, float* out, float* stats, int B, int rows,
                   int kch, int nch, int L, hipStream_t st, const void* zpad, const float* ev_a = nullptr, const float* ev_s = nullptr,
                   float* xrows = nullptr, int bf = 0, int ntt = 0, const ConvH3OpOut* op = nullptr, const ConvH3Bsum* bs = nullptr);
int launch_conv_h3u(const ConvH3uArgs& a, int mrep, int mtiles_p, int kch, hipStream_t st);
int launch_wgrad_h3(const LayerPlan& l, const wunet_half* xh, const wunet_half* xl, const wunet_half* gh, const wunet_half* gl,
                    const float* sc, const float* sc2, float* part, int B, hipStream_t st, int bf = 0);
int launch_backward_packs(wunet_ctx* c, const float* const* params, float* ws, hipStream_t st);
// the side stream + fork / join events of the CURRENT device (nullptr + error text on failure)
wunet_ctx::Side* side_for_current_device(wunet_ctx* c);

}  // namespace wunet_host
