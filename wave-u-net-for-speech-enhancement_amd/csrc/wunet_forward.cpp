// wunet_forward: the training / eval forward of the network as one enqueue (see wunet_host.h).
#include "wunet_host.h"
#include "wunet_elementwise.h"
#include "wunet_tiny.h"
#include "wunet_h3_elem.h"

using namespace wunet_host;

extern "C" {

int wunet_forward(wunet_ctx* c, const float* noisy, const float* const* params, float* const* running,
                  long long* const* nbt, int training, int save_for_backward, void* workspace, float* enhanced, void* stream)
{
    if (!c || !noisy || !params || !running || !nbt || !workspace || !enhanced) return fail(WUNET_E_ARG, "null argument");
    hipStream_t st = (hipStream_t)stream;
    float* ws = (float*)workspace;
    // (eval mode, the caller's workspace still holds this ctx's packs of these weights: nothing to pack)
    const bool packs_valid = !training && (save_for_backward & WUNET_FWD_PACKS_VALID) != 0;
    save_for_backward &= WUNET_FWD_SAVE;
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
        static const bool no_side = getenv("WUNET_NO_SIDE_STREAM") != nullptr;       // (profiling switch: everything on the caller's stream)
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
    if (!fside && !packs_valid) { const int rc = pack_fp32(); if (rc) return rc; }
    if (c->h3 && packs_valid) {
        ScaleTable T{};
        for (int i = 0; i < c->NL; ++i) T.d[i].zp0 = c->ly[i].h3f ? ws + c->ly[i].xzp : nullptr;
        T.slots = ws + c->fslot_off;
        WUNET_LAUNCH(h3_slots_clear_kernel, dim3(1), dim3(WUNET_THREADS), 0, st, T, c->NL);
        WUNET_CHECK_LAUNCH();
    } else if (c->h3) {
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
                d.zp0 = l.h3f ? ws + l.xzp : nullptr;
                d.zp1 = (l.h3d && training && save_for_backward) ? ws + l.gzp : nullptr;      // (the backward segment only exists then)
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
            // (conv_h3u_kernel walks whole chunks: the layers it runs in this mode get a pack without K tail - never longer than the one with)
            d.ntt = (training ? l.h3u_train : l.h3u) ? 0 : l.h3f_ntt; d.nfull = ((l.cin + 7) / 8) / 4; d.ns = h3_stage_count(l.cin, l.taps, d.ntt);
            d.wmax = ws + c->wmax_off + (size_t)WUNET_WMAX_PARTS * i; d.wsc = ws + c->fslot_off + (size_t)WUNET_SLOT_FLOATS * i + 2;
            d.bf = c->bf;
        }
        if (nd > 0) {
            WUNET_LAUNCH(pack_h3_kernel, dim3(pack_gx(), nd), dim3(WUNET_THREADS), 0, pst, tab);
            WUNET_CHECK_LAUNCH();
        }
        if (!training) {
            RowL1Table rt{};
            int nr = 0;
            for (int i = 0; i < c->NL; ++i) {
                const LayerPlan& l = c->ly[i];
                if (!l.evop) continue;
                RowL1Desc& d = rt.d[nr++];
                d.w = params[4 * i]; d.dst = ws + l.wl1; d.rows = l.cout; d.rowlen = l.cin * l.taps;
            }
            if (nr > 0) {
                WUNET_LAUNCH(w_rowl1_kernel, dim3(8, nr), dim3(WUNET_THREADS), 0, pst, rt);
                WUNET_CHECK_LAUNCH();
            }
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
    wunet_ctx::Side* bside = nullptr;          // the side stream holds this forward's backward packs
    if (training && save_for_backward && c->NL > 1) {
        static const bool no_side = getenv("WUNET_NO_SIDE_STREAM") != nullptr;
        wunet_ctx::Side* side = (no_side || g_prof_on) ? nullptr : side_for_current_device(c);
        if (side) {
            // (behind the forward's packs on the side stream when they are there: no second fork)
            if (side != fside && (hipEventRecord(side->ev_fork, st) != hipSuccess || hipStreamWaitEvent(side->stream, side->ev_fork, 0) != hipSuccess))
                return fail(WUNET_E_RUNTIME, "fork onto the side stream failed");
            const int rc = launch_backward_packs(c, params, ws, side->stream);
            if (rc) return rc;
            if (hipEventRecord(side->ev_pack, side->stream) != hipSuccess) return fail(WUNET_E_RUNTIME, "recording the pack event failed");
            side->packed_ws = workspace;
            bside = side;
        }
    }
    if (!training) {
        // eval mode: every layer's BatchNorm scale / shift from the running statistics, one launch ahead of the chain
        BnEvalTable T{};
        int maxc = 1;
        for (int i = 0; i < c->NL; ++i) {
            const LayerPlan& l = c->ly[i];
            BnEvalDesc& d = T.d[i];
            d.gamma = params[4 * i + 2]; d.beta = params[4 * i + 3]; d.running_mean = running[2 * i]; d.running_var = running[2 * i + 1];
            d.a = ws + l.a; d.s = ws + l.s; d.C = l.cout;
            if (l.cout > maxc) maxc = l.cout;
        }
        WUNET_LAUNCH(bn_eval_all_kernel, dim3((unsigned)((maxc + WUNET_THREADS - 1) / WUNET_THREADS), (unsigned)c->NL), dim3(WUNET_THREADS), 0, st, T);
        WUNET_CHECK_LAUNCH();
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
        // conv_h3u_kernel builds its operand itself (wunet_h3u.h): no operand pass for this layer
        const bool use_u = i > 0 && (training ? l.h3u_train : l.h3u);
        // eval mode: the producing encoder conv wrote this layer's operand in its epilogue (conv_h3d_kernel<.., EVOP>)
        const bool op_by_producer = !training && i > 0 && l.kind == LK_DECIM && c->ly[l.src0].evop;
        if (i > 0 && !use_u && !op_by_producer) {
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
                    if (dj < c->NL && c->ly[dj].skip_from == i && !c->ly[dj].h3u_train) {      // (conv_h3u_kernel writes its whole operand)
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
                {
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
        b.cst = (training && l.bsum) ? ws + l.cst : nullptr;      // (read by the consumers' data-gradient epilogues, conv_h3d_kernel<.., BSUM>)
        b.C = l.cout; b.count = (double)c->B * l.Lt; b.training = training ? 1 : 0;
        // eval mode, a layer whose activation feeds split operands: its consumers' operand scale comes from the measured maximum
        // of |a z + s| (no batch statistics bound it).  The large producers (conv_first_kernel, un-split conv_h3_kernel) take it
        // in their epilogue - the BatchNorm coefficients only depend on the running statistics, so they are finalised BEFORE the
        // conv - the small ones get a pass of act_max_kernel over z.
        const bool ev_need = !training && c->h3 && l.feeds_h3;
        const bool ev_epi = ev_need && (l.first || (l.h3f && !split));
        float* const xrows = fslot + (size_t)WUNET_SLOT_FLOATS * i + 4;     // the layer's xb slot (cleared by h3_scales_kernel): blocks fold their maxima into it
        // (eval: a / s of every layer are already there, bn_eval_all_kernel)
        if (l.first) {
            prof_begin(st, "conv_first_kernel<15>", 2.0 * c->B * l.L * l.cout * 15.0, 4.0 * c->B * l.L * (1.0 + l.cout));
            WUNET_LAUNCH(conv_first_kernel<15>, dim3((unsigned)l.f.grid_x), dim3(WUNET_THREADS), 0, st, xin, params[4 * i], params[4 * i + 1],
                         ws + l.z, training ? ws + c->stats_off : (float*)nullptr, c->B, l.cout, l.L, l.logL,
                         ev_epi ? ws + l.a : (const float*)nullptr, ev_epi ? ws + l.s : (const float*)nullptr, ev_epi ? xrows : (float*)nullptr, l.Lt);
            prof_end(st);
        } else if (use_u) {
            const LayerPlan& p = c->ly[l.src0];
            const LayerPlan& k = c->ly[l.src1];
            ConvH3uArgs u{};
            u.z0 = ws + p.z; u.a0 = ws + p.a; u.s0 = ws + p.s; u.z1 = ws + k.z; u.a1 = ws + k.a; u.s1 = ws + k.s;
            u.xb0 = fslot + (size_t)WUNET_SLOT_FLOATS * l.src0 + 4; u.xb1 = fslot + (size_t)WUNET_SLOT_FLOATS * l.src1 + 4;
            u.xsc = fslot + (size_t)WUNET_SLOT_FLOATS * i;
            u.up_scale = (float)(l.Lt / 2 - 1) / (float)(l.Lt - 1); u.Lt = l.Lt; u.C0 = l.c0; u.C1 = l.cin - l.c0;
            u.wh = reinterpret_cast<const wunet_half*>(ws + c->h3_wf_hi) + l.h3f_wpk;
            u.wdelta = (unsigned)((const char*)(ws + c->h3_wf_lo) - (const char*)(ws + c->h3_wf_hi));
            u.bias = params[4 * i + 1]; u.sc2 = fslot + (size_t)WUNET_SLOT_FLOATS * i + 2;
            u.out = ws + l.z; u.stats = training ? ws + c->stats_off : nullptr;
            u.ev_a = ev_epi ? ws + l.a : nullptr; u.ev_s = ev_epi ? ws + l.s : nullptr; u.xrows = ev_epi ? xrows : nullptr;
            if (training && save_for_backward) {        // the weight gradient reads the split operand
                u.oxh = reinterpret_cast<wunet_half*>(ws + l.xh); u.oxl = reinterpret_cast<wunet_half*>(ws + l.xl);
            }
            u.B = c->B; u.Cout = l.cout; u.L = l.L;
            const int rc = launch_conv_h3u(u, l.h3f_mrep, l.h3f_mtp, l.cin, st);
            if (rc) return rc;
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
            ConvH3OpOut opo{};
            const bool evop = !training && l.evop && ev_epi;
            if (evop) {
                const LayerPlan& nx = c->ly[i + 1];
                opo.h = reinterpret_cast<wunet_half*>(ws + nx.xh); opo.l = reinterpret_cast<wunet_half*>(ws + nx.xl);
                opo.wl1 = ws + l.wl1; opo.xmax = fslot + (size_t)WUNET_SLOT_FLOATS * l.src0 + 4;
                opo.xsc = fslot + (size_t)WUNET_SLOT_FLOATS * (i + 1); opo.C8 = (nx.cin + 7) / 8;
            }
            int rc = launch_conv_h3(l.taps, l.h3f_mrep, l.h3f_mtp, l.h3f_sps, xh, xl,
                                    reinterpret_cast<const wunet_half*>(ws + c->h3_wf_hi) + l.h3f_wpk,
                                    reinterpret_cast<const wunet_half*>(ws + c->h3_wf_lo) + l.h3f_wpk,
                                    // (a result that goes through conv_reduce_bn_kernel gets its bias THERE: an un-split launch into the split
                                    //  buffer - a padded length - added it twice; invisible behind training-mode BatchNorm, 0.03 off in eval mode)
                                    split ? nullptr : params[4 * i + 1], sl, sl + 2,
                                    split ? ws + c->spart_off : ws + l.z, (training && !split) ? ws + c->stats_off : nullptr, c->B, l.cout,
                                    l.cin, l.h3f_nch, l.L, st, ws + l.xzp, ev_epi ? ws + l.a : nullptr, ev_epi ? ws + l.s : nullptr, ev_epi ? xrows : nullptr, c->bf, l.h3f_ntt,
                                    evop ? &opo : nullptr);
            if (rc) return rc;
        } else if (tiny) {
            const size_t no = (size_t)c->B * l.cout * l.L;
            WUNET_LAUNCH(tiny_conv_kernel, dim3((unsigned)((no + WUNET_THREADS - 1) / WUNET_THREADS)), dim3(WUNET_THREADS), 0, st,
                         xin, params[4 * i], ws + c->spart_off, c->B, l.cin, l.cout, l.L, l.logL, l.taps, 0);
        } else {
            const ConvArgs a = make_conv_args(xin, l.cin, ws + c->wpkf_off + l.f_wpk, split ? nullptr : params[4 * i + 1],
                                              split ? ws + c->spart_off : ws + l.z, (training && !split) ? ws + c->stats_off : nullptr,
                                              c->B, l.cout, l.L, l.taps, l.f, (size_t)c->B * l.cout * l.L);
            int rc = launch_conv(l.taps, a, l.f, st);
            if (rc) return rc;
        }
        WUNET_CHECK_LAUNCH();
        // 2c. BatchNorm statistics -> scale/shift for the consumers (+ running stats)
        if (ev_epi) {
            // (eval, un-split conv: its epilogue has folded the activation bound into the xb slot; nothing to launch)
        } else if (split) {
            // sum the z-slices (+bias -> z) and reduce the BN statistics; short levels finish BN in the same launch
            const long long pos = (long long)c->B * l.L;
            // (eval: every block ends in a block-wide maximum and an atomic on one address - half as many blocks: eval forward 1.470 -> 1.445 ms,
            //  one block per channel the same, twice as many 1.53 -> 1.57; training 2048 / 4096 / 8192 positions per block: 5.166 / 5.152 /
            //  5.148 ms, within the noise and another summation order - left alone; profiles/r5_small_ab.txt)
            int rs = (int)(pos / (training ? 2048 : 4096));
            if (rs < 1) rs = 1;
            if (rs > 64) rs = 64;
            b.rows = rs;
            WUNET_LAUNCH(conv_reduce_bn_kernel, dim3(l.cout, rs), dim3(WUNET_THREADS), 0, st, b, (const float*)(ws + c->spart_off),
                         tiny ? 1 : l.f.ksplit, (size_t)c->B * l.cout * l.L, ws + l.z, c->B, l.L, l.logL, ws + c->stats_off, l.Lt,
                         ev_need ? xrows : (float*)nullptr);
            if (rs > 1 && training) {
                WUNET_CHECK_LAUNCH();
                WUNET_LAUNCH(bn_finalize_fwd_kernel, dim3(l.cout), dim3(WUNET_THREADS), 0, st, b);
            }
        } else if (training) {
            WUNET_LAUNCH(bn_finalize_fwd_kernel, dim3(l.cout), dim3(WUNET_THREADS), 0, st, b);
        }
        WUNET_CHECK_LAUNCH();
        if (ev_need && !ev_epi && !split) {          // (a split layer's reduce kernel has taken the bound)
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
        long long blocks = ((long long)c->B * c->T / 4 + WUNET_THREADS - 1) / WUNET_THREADS;      // four samples per thread
        if (blocks > 4096) blocks = 4096;
        prof_begin(st, "head_fwd_kernel", 2.0 * c->B * c->T * (c->ci + 1.0), 4.0 * c->B * c->T * (c->ci + 2.0));
        WUNET_LAUNCH(head_fwd_kernel, dim3((unsigned)blocks), dim3(WUNET_THREADS), 0, st, h);
        prof_end(st);
        WUNET_CHECK_LAUNCH();
    }
    // The backward packs were written into the caller's workspace from the side stream: the caller's stream joins them here (they
    // finished long ago - the wait costs nothing), so a forward whose backward never runs (loss discarded, an exception) leaves no
    // write in flight on a stream the owner of the workspace does not know about when it frees or re-uses the block.
    if (bside && hipStreamWaitEvent(st, bside->ev_pack, 0) != hipSuccess) return fail(WUNET_E_RUNTIME, "waiting for the weight packs failed");
    if (c->padded && hipMemcpy2DAsync(enhanced_user, (size_t)c->Tt * sizeof(float), ws + c->pad_out, (size_t)c->T * sizeof(float),
                                      (size_t)c->Tt * sizeof(float), (size_t)c->B, hipMemcpyDeviceToDevice, st) != hipSuccess)
        return fail(WUNET_E_RUNTIME, "cropping the output failed");
    return WUNET_OK;
}

}  // extern "C"
