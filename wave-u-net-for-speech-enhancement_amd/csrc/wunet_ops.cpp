// Loss, fused Adam, window crops, profiler read-out and the single-op entry points (tests / tools) of libwunet_hip.so
// (see wunet_host.h).
#include "wunet_host.h"
#include "wunet_elementwise.h"
#include "wunet_h3_elem.h"

using namespace wunet_host;

extern "C" {

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
long long wunet_profile_collect(char* buf, size_t cap) { return prof_collect(buf, cap); }

void wunet_debug_set_conv_trace(void* dev_buffer) { g_h3_trace = static_cast<unsigned long long*>(dev_buffer); }

// ---------------------------------------------------------------------------- data input
// Aligned crops of the (mixture, clean) pair out of two flat float32 arrays resident in HBM: window b is samples
// starts[b] .. starts[b] + length - 1 of both arrays - the reference's per-item crop (util/utils.py:101-113,
// dataset/waveform_dataset.py:56-67) for a whole batch in one launch.  The window starts are device-side int64 (drawn on the
// host, uploaded with the batch's other 8 B x batch of metadata); out rows are [batch][1][length].
int wunet_crop_windows(const float* mixture_flat, const float* clean_flat, const long long* starts, long long total,
                       int batch, int length, float* mixture, float* clean, void* stream)
{
    if (!mixture_flat || !clean_flat || !starts || !mixture || !clean) return fail(WUNET_E_ARG, "null argument");
    if (batch < 1 || length < 1 || total < length) return fail(WUNET_E_ARG, "bad crop shape (batch=%d length=%d total=%lld)", batch, length, total);
    hipStream_t st = (hipStream_t)stream;
    const unsigned bx = (unsigned)((length + 4 * WUNET_THREADS - 1) / (4 * WUNET_THREADS));
    WUNET_LAUNCH(crop_windows_kernel, dim3(bx, (unsigned)batch), dim3(WUNET_THREADS), 0, st, mixture_flat, clean_flat, starts, total, length,
                 mixture, clean);
    WUNET_CHECK_LAUNCH();
    return WUNET_OK;
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
    int rc = launch_conv(K, a, cfg, st);
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
// (one allocation: hi plane | lo plane | 64 zero bytes - the layout conv_h3d_kernel's DMA addressing wants, launch_conv_h3)
struct SplitBuf {
    DevBuf buf;
    size_t plane = 0;
    wunet_half* h() const { return buf.h(); }
    wunet_half* l() const { return reinterpret_cast<wunet_half*>(static_cast<char*>(buf.p) + plane); }
    const void* zpad() const { return static_cast<char*>(buf.p) + 2 * plane; }
};
int op_split_operand(const float* x, int B, int C, int L, SplitBuf& sb, float* slot, const float* ones, const float* zeros, hipStream_t st)
{
    const int c8 = (C + 7) / 8;
    sb.plane = (size_t)B * c8 * L * 16;
    if (!sb.buf.alloc(2 * sb.plane + 64)) return fail(WUNET_E_RUNTIME, "hipMalloc");
    hipMemsetAsync(static_cast<char*>(sb.buf.p) + 2 * sb.plane, 0, 64, st);
    hipMemsetAsync(slot, 0, WUNET_SLOT_FLOATS * sizeof(float), st);
    const size_t n4 = (size_t)B * C * L / 4;
    size_t blocks = (n4 + WUNET_THREADS * 4 - 1) / (WUNET_THREADS * 4);
    if (blocks > 2048) blocks = 2048;
    WUNET_LAUNCH(act_max_kernel, dim3((unsigned)blocks), dim3(WUNET_THREADS), 0, st, x, ones, zeros, C, ilog2(L), n4, slot + 4);
    return launch_split(x, sb.h(), sb.l(), nullptr, slot + 4, nullptr, slot, B, C, L, st);
}

int op_conv_split_common(const float* x, const float* w, const float* bias, float* out, int B, int kch, int rows, int Cout, int Cin,
                         int L, int K, int transposed, hipStream_t st)
{
    const H3ConvPlan p = plan_h3_conv(B, L, rows, kch, K, transposed ? "WUNET_H3D_ORDER" : "WUNET_H3_ORDER");
    SplitBuf xs;
    DevBuf wpk, misc, part;
    const size_t wh_halfs = (size_t)p.mtp * p.nch * K * 512, nout = (size_t)B * rows * L;
    // misc: slot of x (8 floats) | slot of w (8) | partial weight maxima | ones (kch) | zeros (kch)
    if (!wpk.alloc(wh_halfs * 4) || !misc.alloc((16 + WUNET_WMAX_PARTS + 2 * (size_t)kch) * sizeof(float)) ||
        (p.ksplit > 1 && !part.alloc((size_t)p.ksplit * nout * sizeof(float))))
        return fail(WUNET_E_RUNTIME, "hipMalloc");
    float* xslot = misc.f(), *wslot = misc.f() + 8, *wmax = misc.f() + 16, *oz = misc.f() + 16 + WUNET_WMAX_PARTS;
    WUNET_LAUNCH(fill_kernel, dim3(4), dim3(WUNET_THREADS), 0, st, oz, (size_t)kch, 1.0f);
    WUNET_LAUNCH(fill_kernel, dim3(4), dim3(WUNET_THREADS), 0, st, oz + kch, (size_t)kch, 0.0f);
    wunet_half* const whp = wpk.h();
    wunet_half* const wlp = wpk.h() + wh_halfs;
    int rc = op_split_operand(x, B, kch, L, xs, xslot, oz, oz + kch, st);
    if (rc) return rc;
    {
        ScaleTable T{};
        T.d[0].w = w; T.d[0].wn = (unsigned)((size_t)Cout * Cin * K); T.d[0].gamma = oz; T.d[0].beta = oz; T.d[0].C = 1; T.d[0].sqrtn = 0.0f;
        T.wmax = wmax; T.slots = wslot; T.training = 0;          // (clears wslot[4], which nothing reads)
        WUNET_LAUNCH(h3_scales_kernel, dim3(WUNET_WMAX_PARTS, 1), dim3(WUNET_THREADS), 0, st, T);
        PackH3Table tab{};
        PackH3Desc& d = tab.d[0];
        d.w = w; d.hi = whp; d.lo = wlp; d.Cout = Cout; d.Cin = Cin; d.taps = K; d.rows = rows; d.kch = kch; d.mtiles = p.mtp; d.nch = p.nch;
        d.ntt = p.ntt; d.nfull = ((kch + 7) / 8) / 4; d.ns = h3_stage_count(kch, K, p.ntt);
        d.transposed = transposed; d.wmax = wmax; d.wsc = wslot + 2;
        WUNET_LAUNCH(pack_h3_kernel, dim3(64, 1), dim3(WUNET_THREADS), 0, st, tab);
    }
    const bool split = p.ksplit > 1;
    rc = launch_conv_h3(K, p.mrep, p.mtp, p.sps, xs.h(), xs.l(), whp, wlp, split ? nullptr : bias, xslot, wslot + 2,
                        split ? part.f() : out, nullptr, B, rows, kch, p.nch, L, st, xs.zpad(), nullptr, nullptr, nullptr, 0, p.ntt);
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
    SplitBuf xs, gs;
    DevBuf misc, part;
    const int cmax = Cin > Cout ? Cin : Cout;
    // misc: slot of g_z | slot of x | ones | zeros
    if (!misc.alloc((16 + 2 * (size_t)cmax) * sizeof(float)) || !part.alloc((size_t)l.h3w_ksplit * h3w_part_stride(l) * sizeof(float)))
        return fail(WUNET_E_RUNTIME, "hipMalloc");
    float* gslot = misc.f(), *xslot = misc.f() + 8, *oz = misc.f() + 16;
    WUNET_LAUNCH(fill_kernel, dim3(4), dim3(WUNET_THREADS), 0, st, oz, (size_t)cmax, 1.0f);
    WUNET_LAUNCH(fill_kernel, dim3(4), dim3(WUNET_THREADS), 0, st, oz + cmax, (size_t)cmax, 0.0f);
    int rc = op_split_operand(gz, B, Cout, L, gs, gslot, oz, oz + cmax, st);
    if (!rc) rc = op_split_operand(x, B, Cin, L, xs, xslot, oz, oz + cmax, st);
    if (!rc) rc = launch_wgrad_h3(l, xs.h(), xs.l(), gs.h(), gs.l(), gslot, xslot, part.f(), B, st);
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
