// wunet_backward[_range[_async]] / wunet_backward_join: the backward of the network, whole or by layer ranges (see wunet_host.h).
#include "wunet_host.h"
#include "wunet_elementwise.h"
#include "wunet_tiny.h"
#include "wunet_h3_elem.h"

using namespace wunet_host;

namespace {
// ---- measurement hook (WUNET_STAMP=1; bench.py prints the slots): one-thread kernels that write the device's 100 MHz wall clock into a slot, enqueued
// beside the backward's launches - in FRONT of a layer's data gradient on the caller's stream (slot 2 i), in FRONT of its weight gradient on
// the side stream (slot 2 i + 1), at the join (slots 126, 127).  They are captured into a step graph like any launch, so a REPLAY can be asked
// when its two chains really ran - a tracer changes how a graph is submitted.  Off: no launches, no allocation.  (The buffer is allocated by
// the first backward that runs with the switch on: an eager warm-up step, never inside a capture.)
#ifndef WUNET_EMU
__global__ void stamp_kernel(unsigned long long* p) { *p = wall_clock64(); }
unsigned long long* g_stamps = nullptr;
bool stamps_on()
{
    static const bool on = getenv("WUNET_STAMP") != nullptr;
    if (on && !g_stamps && (hipMalloc(&g_stamps, 128 * sizeof(unsigned long long)) != hipSuccess || hipMemset(g_stamps, 0, 128 * sizeof(unsigned long long)) != hipSuccess)) g_stamps = nullptr;
    return on && g_stamps;
}
#define WUNET_STAMP(stream_, slot_) { if (stamps_on()) hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, stream_, g_stamps + (slot_)); }
#else
#define WUNET_STAMP(stream_, slot_) {}
#endif

// dx of an encoder-side layer i (1 <= i <= n: encoder 1 .. middle) has ONE reader, pass A of layer i - 1, the next kernel on the
// stream: a split-K data gradient leaves its partials in the scratch and that pass adds them itself, in split order (same bits), instead
// of a split_sum_kernel launch in between.  (A decoder layer's dx is read again much later - its skip half by the encoder side - and is
// summed as before.)
bool dx_stays_split(const wunet_ctx* c, int i)
{
    if (i < 1 || i > c->n) return false;
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
    static const bool no_side = getenv("WUNET_NO_SIDE_STREAM") != nullptr;     // profiling switch: one stream, serial kernels
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
        // (d(weight of the input channel), d bias and the ci per-channel head-weight gradients - sums of head_bwd_kernel's and of the last layer's
        //  pass A's partial rows - are parameter gradients nothing on the chain reads: they are taken on the weight-gradient stream, below)
    }

    for (int i = layer_end - 1; i >= layer_begin; --i) {
        const LayerPlan& l = c->ly[i];
        // ---- a layer whose BatchNorm-backward sums came out of its consumers' data-gradient epilogues (conv_h3d_kernel<.., BSUM>, planned by
        // layout_workspace): no pass A; finalize from the per-tile rows; g_z formed by gz_split_h3_kernel from the data gradients themselves
        static const bool no_enc_gz_ = getenv("WUNET_NO_ENC_GZ") != nullptr;
        const bool bsum = l.bsum && !(i < n && no_enc_gz_);
        // ---- pass A: assemble dL/d(BN output), LeakyReLU', BN-backward partial sums
        PassAArgs p{};
        p.z = ws + l.z; p.a = ws + l.a; p.s = ws + l.s; p.mean = ws + l.mean; p.rstd = ws + l.rstd;
        p.gpre = ws + l.g; p.part = ws + c->bpart_off; p.pmax = (i > 0 && l.h3d) ? ws + c->bmax_off : nullptr; p.B = c->B; p.C = l.cout; p.L = l.L; p.logL = l.logL; p.Lt = l.Lt;
        // grid (pieces, channels): consecutive blocks walk consecutive pieces of ONE channel row (rounds 1 - 3 had (channels, pieces):
        // neighbours 4 L bytes apart; 5.156 -> 5.139 ms per step; WUNET_PA_SWAP=0: A/B switch)
        static const int pa_swap = getenv("WUNET_PA_SWAP") ? atoi(getenv("WUNET_PA_SWAP")) : 1;
        const bool tiny = l.L < 4;
        p.swap = (pa_swap && l.a_split > 1 && !tiny) ? 1 : 0;        // (pass_a_scalar_kernel reads its channel from blockIdx.x)
        const dim3 ga(p.swap ? l.a_split : l.cout, p.swap ? l.cout : l.a_split);
        // the first layer's g_z has one reader: its weight gradient forms it while it stages its chunks (WUNET_NO_GZ_FUSE: A/B switch)
        const bool gz_in_wgrad = i == 0 && !tiny && !l.h3d && !l.h3w && l.w.wsplit &&
                                 !(l.a_split == 1 && (size_t)c->B * l.L <= 4 * WUNET_THREADS);       // (not where pass A finishes g_z itself)
        // a whole channel in one pass of one block (the levels of <= 16 samples at batch 64): BatchNorm-backward finalize and
        // g_z inside pass A - two launches of ~5 us (latency, not bandwidth) less per such level
        const bool fuse = !tiny && i < NL - 1 && !(i > 0 && l.h3d) && l.a_split == 1 && (size_t)c->B * l.L <= 4 * WUNET_THREADS;
        // the first layer (Cin = 1, 15 taps): its weight gradient's sums are taken by this pass and finished by the finalize - no g written, no
        // weight-gradient GEMM, no split reduce (the last 66 us of the backward, alone on the chip; WUNET_NO_E0=1 when the context is planned: A/B switch)
        const bool e0 = c->e0 && gz_in_wgrad && !fuse && l.cin == 1 && l.taps == 15 && l.L >= 16 && !dx_stays_split(c, 1) && n >= 1;
        if (fuse) {
            p.gamma = params[4 * i + 2]; p.dgamma = grads[4 * i + 2]; p.dbeta = grads[4 * i + 3]; p.dbias = grads[4 * i + 1];
            p.k1 = ws + l.k1; p.k2 = ws + l.k2; p.k3 = ws + l.k3; p.count = (double)c->B * l.Lt;
        }
        // the last layer on the split kernels: its g is recomputed from the head gradient by gz_split_h3_kernel (HEAD mode), not stored
        const bool head_in_gz = i == NL - 1 && i > 0 && l.h3d && !fuse && !tiny;
        // an encoder layer on the split kernels whose two data gradients are whole tensors: g is recomputed by gz_split_h3_kernel (ENC
        // mode) from them instead of written here and read back (WUNET_NO_ENC_GZ=1: A/B switch)
        static const bool no_enc_gz = getenv("WUNET_NO_ENC_GZ") != nullptr;
        const bool enc_in_gz = i > 0 && i < n && l.h3d && !fuse && !tiny && !dx_stays_split(c, i + 1) && !no_enc_gz;
        // a layer behind an upsample whose consumer's data gradient arrives at this layer's resolution (UPT, planned by layout_workspace)
        const bool uph = !bsum && i >= n && i + 1 < NL && c->ly[i + 1].upt && l.h3d && !fuse && !tiny;
        if (!bsum) {   // algorithmic bytes of the gradient assembly (HBM-bound): z + the consumers' data gradients read, g written
            const double pe = (double)c->B * l.cout * l.L;
            const char* nm = i == NL - 1 ? "pass_a_kernel<HEAD>" : i >= n ? "pass_a_kernel<UP>" : e0 ? "pass_a_kernel<ENC, e0>" : "pass_a_kernel<ENC>";
            prof_begin(st, uph ? "pass_a_kernel<UPH>" : nm, 0.0, pe * (i == NL - 1 ? (head_in_gz ? 4.0 : 8.0) : uph ? 8.0 : i >= n ? 16.0 : ((enc_in_gz || e0) ? 10.0 : 14.0)) + (i == NL - 1 ? 4.0 * c->B * l.L : 0.0));
        }
        if (bsum) {
        } else if (i == NL - 1) {
            if (head_in_gz) p.gpre = nullptr;
            p.g0 = ws + c->gh_off; p.g1 = params[4 * NL]; p.hpart = ws + c->hpart2_off;
            WUNET_LAUNCH(pass_a_kernel<A_HEAD>, ga, dim3(WUNET_THREADS), 0, st, p);      // (the last layer has T >= 4 samples)
            prof_end(st);
        } else if (i >= n && uph) {
            // the next decoder layer's data gradient arrived at THIS resolution (conv_h3d_kernel<.., 3>): elementwise; g is not stored -
            // gz_split_h3_kernel forms it again from the same two arrays (UPH mode)
            const LayerPlan& nx = c->ly[i + 1];
            p.g0 = ws + nx.dxh; p.sp = ws + nx.usp; p.ntiles = (int)(((size_t)c->B * nx.L + 255) / 256); p.tpr = l.L >> 7;
            p.gpre = nullptr;
            WUNET_LAUNCH(pass_a_kernel<A_UPH>, ga, dim3(WUNET_THREADS), 0, st, p);
            prof_end(st);
        } else if (i >= n) {
            const LayerPlan& nx = c->ly[i + 1];
            p.g0 = ws + nx.dx; p.Cg0 = nx.cin;
            p.up_scale = (float)(l.Lt - 1) / (float)(2 * l.Lt - 1);
            if (tiny) WUNET_LAUNCH(pass_a_scalar_kernel<A_UP>, ga, dim3(WUNET_THREADS), 0, st, p);
            else if (fuse) WUNET_LAUNCH((pass_a_kernel<A_UP, true>), ga, dim3(WUNET_THREADS), 0, st, p);
            else WUNET_LAUNCH(pass_a_kernel<A_UP>, ga, dim3(WUNET_THREADS), 0, st, p);
            prof_end(st);
        } else {
            const LayerPlan& dc = c->ly[2 * n - i];
            const LayerPlan& nx = c->ly[i + 1];
            p.g0 = ws + dc.dx; p.Cg0 = dc.cin; p.coff = dc.c0; p.g1 = ws + nx.dx;
            if (enc_in_gz) p.gpre = nullptr;
            if (dx_stays_split(c, i + 1)) {
                p.g1 = ws + c->spart_off; p.g1_splits = nx.d.ksplit; p.g1_stride = (size_t)c->B * nx.cin * nx.L;
            }
            if (e0) { p.gpre = nullptr; p.x0 = noisy; p.e0rows = ws + c->e0part_off; }
            if (tiny) WUNET_LAUNCH(pass_a_scalar_kernel<A_ENC>, ga, dim3(WUNET_THREADS), 0, st, p);
            else if (e0) WUNET_LAUNCH((pass_a_kernel<A_ENC, false, false, true>), ga, dim3(WUNET_THREADS), 0, st, p);
            else if (fuse && p.g1_splits > 1) WUNET_LAUNCH((pass_a_kernel<A_ENC, true, true>), ga, dim3(WUNET_THREADS), 0, st, p);
            else if (fuse) WUNET_LAUNCH((pass_a_kernel<A_ENC, true>), ga, dim3(WUNET_THREADS), 0, st, p);
            else if (p.g1_splits > 1) WUNET_LAUNCH((pass_a_kernel<A_ENC, false, true>), ga, dim3(WUNET_THREADS), 0, st, p);
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
            if (e0) { b.e0rows = ws + c->e0part_off; b.dw0 = grads[0]; }
            // (short levels on the split kernels: the finalize runs in the prologue of gz_split_h3_kernel's blocks instead -
            //  WUNET_NO_BWDFIN_FUSE=1: A/B switch)
            const bool fin_in_gz = !bsum && i > 0 && l.h3d && l.a_split * l.cout <= WUNET_GZ_FIN_LOADS && l.cout <= WUNET_GZ_FIN_C;
            if (bsum) {
                BnBwdTilesArgs t{};
                if (i >= n) {            // one set: rows 0 .. c0 - 1 of the next decoder layer's data gradient
                    const LayerPlan& q = c->ly[i + 1];
                    t.part0 = ws + q.bsp; t.tiles0 = (int)(((size_t)c->B * q.L + 255) / 256);
                } else {                 // the skip rows of decoder layer 2n - i, then the decimated rows of layer i + 1
                    const LayerPlan& dq = c->ly[2 * n - i];
                    const LayerPlan& eq = c->ly[i + 1];
                    t.tiles0 = (int)(((size_t)c->B * dq.L + 255) / 256); t.part0 = ws + dq.bsp + (size_t)dq.c0 * t.tiles0 * 4;
                    t.tiles1 = (int)(((size_t)c->B * eq.L + 255) / 256); t.part1 = ws + eq.bsp;
                }
                t.gamma = b.gamma; t.mean = b.mean; t.rstd = b.rstd; t.dgamma = b.dgamma; t.dbeta = b.dbeta; t.dbias = b.dbias;
                t.k1 = b.k1; t.k2 = b.k2; t.k3 = b.k3; t.bound = b.bound; t.C = l.cout; t.count = b.count;
                WUNET_LAUNCH(bn_finalize_bwd_tiles_kernel, dim3(l.cout), dim3(WUNET_THREADS), 0, st, t);
                WUNET_CHECK_LAUNCH();
            } else if (!fin_in_gz) {
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
                    const bool up_in_gz = bsum && i >= n;
                    prof_begin(st, "gz_split_h3_kernel", 0.0, (double)c->B * l.cout * l.L * ((head_in_gz ? 4.0 : enc_in_gz ? 10.0 : up_in_gz ? 12.0 : 8.0) + (c->bf ? 2.0 : 4.0)));
                    GzHeadArgs hd{};
                    if (head_in_gz) { hd.gh = ws + c->gh_off; hd.wh = params[4 * NL]; hd.a = ws + l.a; hd.s = ws + l.s; }
                    if (uph) {
                        const LayerPlan& q = c->ly[i + 1];
                        hd.gq = ws + q.dxh; hd.a = ws + l.a; hd.s = ws + l.s;
                    }
                    if (up_in_gz) {
                        const LayerPlan& q = c->ly[i + 1];
                        hd.gu = ws + q.dx; hd.Cg0 = q.cin; hd.up_scale = (float)(l.Lt - 1) / (float)(2 * l.Lt - 1); hd.a = ws + l.a; hd.s = ws + l.s;
                    }
                    if (enc_in_gz) {
                        const LayerPlan& dc = c->ly[2 * n - i];
                        hd.gd = ws + dc.dx; hd.Cg0 = dc.cin; hd.coff = dc.c0; hd.ge = ws + c->ly[i + 1].dx; hd.a = ws + l.a; hd.s = ws + l.s;
                    }
                    // (the source of g is a compile-time mode of the kernel: every load of a thread is issued before its first use)
                    const int gm = head_in_gz ? GZ_HEAD : enc_in_gz ? GZ_ENC : uph ? GZ_UPH : up_in_gz ? GZ_UP : GZ_G;
#define WUNET_GZ_LAUNCH(FIN_, GM_, BF_)                                                                                                               \
    WUNET_LAUNCH((gz_split_h3_kernel<FIN_, GM_, BF_>), dim3((unsigned)hb), dim3(WUNET_THREADS), 0, st, (const float*)(ws + l.g), (const float*)(ws + l.z), \
                 (const float*)(ws + l.k1), (const float*)(ws + l.k2), (const float*)(ws + l.k3), (const float*)(ws + c->bound_off),                 \
                 ws + c->h3_slot + 8 + 4 * i, reinterpret_cast<wunet_half*>(ws + l.gzh), reinterpret_cast<wunet_half*>(ws + l.gzl),                  \
                 c->B, l.cout, c8, l.L, l.logL, l.Lt, b, hd)
#define WUNET_GZ_MODE(GM_)                                                                                                                            \
    case GM_:                                                                                                                                         \
        if (fin_in_gz) { if (c->bf) WUNET_GZ_LAUNCH(true, GM_, true); else WUNET_GZ_LAUNCH(true, GM_, false); }                                       \
        else { if (c->bf) WUNET_GZ_LAUNCH(false, GM_, true); else WUNET_GZ_LAUNCH(false, GM_, false); }                                               \
        break;
                    switch (gm) { WUNET_GZ_MODE(GZ_G) WUNET_GZ_MODE(GZ_HEAD) WUNET_GZ_MODE(GZ_ENC) WUNET_GZ_MODE(GZ_UPH) WUNET_GZ_MODE(GZ_UP) }
#undef WUNET_GZ_MODE
#undef WUNET_GZ_LAUNCH
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
        // The fork EVENT is recorded here, right behind g_z; the side stream's launches are issued after the data gradient's (below), so
        // that the kernel of the critical chain is enqueued - and, in a captured graph, created - FIRST: its persistent blocks take the
        // CUs and the weight gradient fills in behind it and beside the next layer's memory-bound passes.  Eager launches got that order
        // anyway (the event wait costs the side stream ~6 us); a captured graph released both kernels at once, the weight gradient
        // won the race (data gradient 204 us instead of 110) and the replay was 10 % slower than the eager step (5.93 -> 5.25 ms with
        // this order, eager 5.26: profiles/r4_graph_vs_eager.txt).
        if (sd != st && hipEventRecord(side->ev_fork, st) != hipSuccess) return fail(WUNET_E_RUNTIME, "fork onto the weight-gradient stream failed");
        auto weight_gradient = [&]() -> int {
            if (sd != st) {
                if (hipStreamWaitEvent(sd, side->ev_fork, 0) != hipSuccess)
                    return fail(WUNET_E_RUNTIME, "fork onto the weight-gradient stream failed");
            }
            WUNET_STAMP(sd, 2 * i + 1)
            if (i == NL - 1) {
                // the head's parameter gradients (rounds 1 - 6: two 4.7 us launches between the kernels of the main chain, which never reads them):
                // d(weight of the input channel), d bias from head_bwd_kernel's rows; the ci per-channel weight gradients from pass A's
                WUNET_LAUNCH(rows_sum_kernel, dim3(2), dim3(WUNET_THREADS), 0, sd,
                             (const float*)(ws + c->hpart_off), c->head_blocks, 2, grads[4 * NL] + c->ci, 1, grads[4 * NL + 1]);
                WUNET_CHECK_LAUNCH();
                if (!bsum) {
                    WUNET_LAUNCH(rows_sum_kernel, dim3(c->ci), dim3(WUNET_THREADS), 0, sd,
                                 (const float*)(ws + c->hpart2_off), l.a_split, c->ci, grads[4 * NL], c->ci, grads[4 * NL + 1]);
                    WUNET_CHECK_LAUNCH();
                }
            }
            if (e0) return 0;                          // (finished by bn_finalize_bwd_kernel from pass A's sums)
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
                                         ws + c->h3_slot + 8 + 4 * i, ws + c->fslot_off + (size_t)WUNET_SLOT_FLOATS * i,
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
                    // few splits over many outputs: the barrier-free serial form (64, 8192: 6.19 -> 6.13 ms per step in round 1).
                    // Emulator-build test hook WUNET_REDUCE_SERIAL = "<max splits>,<min float4 outputs>": tiny shapes reach that kernel
                    int serial_max = 64; long long serial_min_n4 = 8192;
#ifdef WUNET_EMU
                    if (const char* e = getenv("WUNET_REDUCE_SERIAL")) sscanf(e, "%d,%lld", &serial_max, &serial_min_n4);
#endif
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
        // ---- data gradient (not needed for the first layer): the same conv kernel on the flipped/transposed pack
        WUNET_STAMP(st, 2 * i)
        if (i > 0 && l.h3d) {
            // fp16-split data gradient: scale g_z by a power of two into fp16's range, split, 3 MFMA passes, un-scale
            float* sc = ws + c->h3_slot + 8 + 4 * i;
            wunet_half* gh = reinterpret_cast<wunet_half*>(ws + l.gzh);
            wunet_half* gl = reinterpret_cast<wunet_half*>(ws + l.gzl);
            const bool split = l.d.ksplit > 1;
            // (BSUM: the epilogue also takes the BatchNorm-backward sums of the layers that produced this layer's input rows)
            ConvH3Bsum bs{};
            // (UPT: the rows of the upsampled half leave at the producer's resolution - the producer's gradient assembly is then elementwise)
            const bool upt = l.upt && !split && c->ly[l.src0].h3d;
            if (upt) {
                bs.kind = 3; bs.c0 = l.c0; bs.up_scale = (float)(l.Lt / 2 - 1) / (float)(l.Lt - 1);
                bs.uh_out = ws + l.dxh; bs.uh_spill = ws + l.usp;
            }
            if (l.bs_kind && !split) {
                const LayerPlan& pa = c->ly[l.src0];
                bs.kind = l.bs_kind; bs.c0 = l.c0; bs.up_scale = (float)(l.Lt / 2 - 1) / (float)(l.Lt - 1); bs.part = ws + l.bsp;
                const bool on_a = pa.bsum && !(l.src0 < n && no_enc_gz_);
                bs.z[0] = on_a ? ws + pa.z : nullptr; bs.cst[0] = ws + pa.cst; bs.C[0] = pa.cout;
                if (l.bs_kind == 1) {
                    const LayerPlan& pb = c->ly[l.src1];
                    bs.z[1] = (pb.bsum && !no_enc_gz_) ? ws + pb.z : nullptr; bs.cst[1] = ws + pb.cst; bs.C[1] = pb.cout;
                }
            }
            int rc = launch_conv_h3(l.taps, l.h3d_mrep, l.h3d_mtp, l.h3d_sps, gh, gl,
                                    reinterpret_cast<const wunet_half*>(ws + c->h3_wb_hi) + l.h3d_wpk,
                                    reinterpret_cast<const wunet_half*>(ws + c->h3_wb_lo) + l.h3d_wpk, nullptr, sc,
                                    ws + c->fslot_off + (size_t)WUNET_SLOT_FLOATS * i + 2,
                                    split ? ws + c->spart_off : ws + l.dx, nullptr, c->B, l.cin, l.cout, l.h3d_nch, l.L, st, ws + l.gzp, nullptr, nullptr,
                                    nullptr, c->bf, l.h3d_ntt, nullptr, (upt || bs.z[0] || bs.z[1]) ? &bs : nullptr);
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
            int rc = launch_conv(l.taps, a, l.d, st);
            if (rc) return rc;
            WUNET_CHECK_LAUNCH();
            if (split && !dx_stays_split(c, i)) {
                size_t blocks = (nd + WUNET_THREADS - 1) / WUNET_THREADS;
                if (blocks > 2048) blocks = 2048;
                WUNET_LAUNCH(split_sum_kernel, dim3((unsigned)blocks), dim3(WUNET_THREADS), 0, st, (const float*)(ws + c->spart_off), l.d.ksplit, nd, ws + l.dx, (const float*)nullptr, 1, 0);
                WUNET_CHECK_LAUNCH();
            }
        }
        { const int rc = weight_gradient(); if (rc) return rc; }
    }
    // join: the caller's stream sees every weight gradient.  An un-joined range (wunet_backward_range_async) leaves them to
    // wunet_backward_join - except the range that ends the backward, which always joins: the next forward overwrites the
    // operands the side stream is still reading.
    if (sd != st && (join || layer_begin == 0)) {
        WUNET_STAMP(st, 126)
        WUNET_STAMP(sd, 127)
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

// measurement hook: the stamps of the last backward (WUNET_STAMP=1), 128 device wall-clock values (100 MHz; 0 = slot not written)
int wunet_debug_stamps(unsigned long long* out, int n)
{
#ifndef WUNET_EMU
    if (!out || n < 0 || n > 128 || !g_stamps) return fail(WUNET_E_ARG, "no stamps (WUNET_STAMP unset, or no backward ran)");
    return hipMemcpy(out, g_stamps, (size_t)n * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess ? WUNET_OK : fail(WUNET_E_RUNTIME, "copy of the stamps failed");
#else
    (void)out; (void)n; return fail(WUNET_E_ARG, "no stamps in the emulator build");
#endif
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

}  // extern "C"
