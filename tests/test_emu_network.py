"""CPU tests of the product's kernel + host sources (csrc/) executed by the fiber emulator
(tests/emu): the whole forward / backward through the C ABI and the Python module, against the
C oracle on tiny networks.  This validates index math, the MFMA lane layout as documented in the
CDNA4 guide, workspace planning and the autograd plumbing without a GPU; the `-m gpu` tests repeat
the comparison on real hardware."""
import ctypes
import importlib

import numpy as np
import pytest
import torch

import emu_lib
from conftest import PKG_NAME
from oracle import c_oracle, plan


@pytest.fixture(scope="module")
def emu_engine():
    eng_mod = importlib.import_module(PKG_NAME + ".engine")
    lib_mod = importlib.import_module(PKG_NAME + "._lib")
    return eng_mod.Engine(lib=lib_mod.declare(emu_lib.lib()), host_memory=True)


def _build(n, ci, emu_engine):
    pkg_model = importlib.import_module(PKG_NAME + ".model")
    pkg_loss = importlib.import_module(PKG_NAME + ".loss")
    sd = plan.golden_state(n, ci, 0)
    m = pkg_model.Model(n_layers=n, channels_interval=ci)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    m._engine_override = emu_engine
    return m, sd, pkg_loss


@pytest.mark.parametrize("n,ci,B,T,loss", [(2, 4, 2, 32, "mse"), (3, 8, 3, 64, "l1"), (2, 24, 1, 1024, "smooth_l1"),
                                            (1, 5, 1, 8, "mse"), (3, 10, 3, 128, "smooth_l1"),
                                            (4, 4, 2, 16, "mse"), (5, 6, 3, 64, "l1"), (3, 8, 2, 16, "smooth_l1"),    # middle of 1 / 2 samples
                                            # lengths m * 2^n (model/unet_basic.py:86,93 accepts them): rows padded to the next power of two
                                            (3, 8, 2, 96, "mse"), (2, 4, 3, 24, "l1"), (4, 6, 2, 80, "smooth_l1"), (2, 12, 1, 1536, "mse")])
def test_train_step_matches_oracle(emu_engine, n, ci, B, T, loss):
    m, sd, pkg_loss = _build(n, ci, emu_engine)
    noisy, clean = plan.golden_batch(B, T, 0)
    ref = c_oracle.step(sd, noisy, clean, n, ci, True, loss, want_acts=True, precision="f64")
    crit = {"mse": pkg_loss.mse_loss, "l1": pkg_loss.l1_loss, "smooth_l1": pkg_loss.smooth_l1_loss}[loss]()
    crit._engine_override = emu_engine
    m.train()
    out = m(torch.from_numpy(noisy))
    lv = crit(torch.from_numpy(clean), out)
    lv.backward()
    assert np.abs(out.detach().numpy() - ref["out"]).max() < 2e-5
    assert abs(lv.item() - ref["loss"]) < 1e-5
    for k, p in m.named_parameters():
        g, r = p.grad.numpy(), ref["grads"][k]
        if k.endswith(".0.bias") and not k.startswith("out"):
            assert np.all(g == 0.0), k          # exact zero (true gradient), reference holds fp32 noise
            continue
        scale = max(np.abs(r).max(), 1e-6)
        assert np.abs(g - r).max() < 3e-4 * scale + 1e-6, (k, np.abs(g - r).max(), scale)
    post = m.state_dict()
    for k in plan.buffer_names(n, ci):
        assert np.abs(post[k].numpy().astype(np.float64) - sd[k]).max() < 1e-5, k


@pytest.fixture(scope="module")
def emu_engine_h3():
    """fp16-split GEMMs forced onto every level the kernels can run (wunet_set_h3(ctx, 2)): conv_h3 / wgrad_h3 /
    prep_h3 / gz_split_h3, with the emulator's v_mfma_f32_16x16x32_f16, ds_read_b64_tr_b16 and v_alignbit models."""
    eng_mod = importlib.import_module(PKG_NAME + ".engine")
    lib_mod = importlib.import_module(PKG_NAME + "._lib")
    return eng_mod.Engine(lib=lib_mod.declare(emu_lib.lib()), host_memory=True, h3=2)


@pytest.mark.parametrize("n,ci,B,T,loss", [(2, 24, 1, 1024, "smooth_l1"),     # every level on the split path (L = 1024, 512, 256)
                                            (3, 16, 3, 1024, "mse"),           # 128-sample level: two-item tiles (odd batch: a half-empty tile), split-K
                                            (1, 20, 3, 512, "l1"),             # channel counts that are not multiples of 8
                                            (4, 16, 5, 1024, "smooth_l1"),     # 64-sample level: four-item conv tiles, two-item wgrad chunks, odd batch
                                            (6, 16, 17, 1024, "mse"),          # 32- and 16-sample levels: 2 / 4 items under one wave
                                            (2, 24, 3, 1536, "smooth_l1"),     # 3 * 512 samples: levels of 1536 / 768 / 384 in rows of 2048 / 1024 / 512
                                            # (batch 3: at batch 2 one activation of decoder.0 sits within 1e-7 of 0 - see below - and
                                            #  its slope follows the order of the sums: exactly one channel's d beta then moves by 3e-3)
                                            (3, 16, 3, 768, "mse")])           # ... 768 / 384 / 192 / 96: segmented tiles with padded items
def test_fp16_split_train_step_matches_oracle(emu_engine_h3, n, ci, B, T, loss):
    """Same comparison as test_train_step_matches_oracle with the fp16-split path forced on: output within 2e-5, every
    gradient within 3e-4 of its tensor's largest entry (the f32-vs-f64 noise floor of these nets is 1.5e-4).  The
    6-level net is the exception: LeakyReLU' is discontinuous, the split path's forward noise (~2e-6) flips the slope of
    the odd activation that sits within that distance of 0, and one flipped high-gradient element moves the BatchNorm
    sums of a few-thousand-position level by up to a percent (checked element-wise against the fp32 path: data
    gradients agree to 1e-6, single elements of g differ by the slope factor 10) - its bar is 2 %, which an indexing
    mistake in the two/four-items-per-wave paths (errors of order one) cannot pass."""
    gtol = 2e-2 if n >= 6 else 3e-4
    eng = emu_engine_h3
    m, sd, pkg_loss = _build(n, ci, eng)
    noisy, clean = plan.golden_batch(B, T, 0)
    ref = c_oracle.step(sd, noisy, clean, n, ci, True, loss, precision="f64")
    crit = {"mse": pkg_loss.mse_loss, "l1": pkg_loss.l1_loss, "smooth_l1": pkg_loss.smooth_l1_loss}[loss]()
    crit._engine_override = eng
    m.train()
    out = m(torch.from_numpy(noisy))
    lv = crit(torch.from_numpy(clean), out)
    lv.backward()
    # the split path must really have been planned: it needs extra workspace
    import ctypes
    sizes = []
    for mode in (0, 2):
        h = ctypes.c_void_p()
        assert eng.lib.wunet_create(n, ci, B, T, ctypes.byref(h)) == 0
        assert eng.lib.wunet_set_h3(h, mode) == 0
        sizes.append(eng.lib.wunet_workspace_bytes(h, 0))
        eng.lib.wunet_destroy(h)
    assert sizes[1] != sizes[0]
    assert np.abs(out.detach().numpy() - ref["out"]).max() < 2e-5
    assert abs(lv.item() - ref["loss"]) < 1e-5
    for k, p in m.named_parameters():
        g, r = p.grad.numpy(), ref["grads"][k]
        if k.endswith(".0.bias") and not k.startswith("out"):
            assert np.all(g == 0.0), k
            continue
        scale = max(np.abs(r).max(), 1e-6)
        assert np.abs(g - r).max() < gtol * scale + 1e-6, (k, np.abs(g - r).max(), scale)
    post = m.state_dict()
    for k in plan.buffer_names(n, ci):
        assert np.abs(post[k].numpy().astype(np.float64) - sd[k]).max() < 1e-5, k


def test_fp16_split_four_rows_per_wave(emu_engine_h3, monkeypatch):
    """conv_h3d_kernel<TAPS, 4, 1> (64 output rows per block; the planner picks it for the large 5-tap layers of the
    12-level net, too large for the emulator) forced through the A/B switches on a net whose m-tile counts are 4 and 8."""
    monkeypatch.setenv("WUNET_H3_ORDER", "432")
    monkeypatch.setenv("WUNET_H3D_ORDER", "432")
    test_fp16_split_train_step_matches_oracle(emu_engine_h3, 2, 32, 2, 1024, "mse")


def test_fp16_split_serial_weight_gradient_reduce(emu_engine_h3, monkeypatch):
    """wgrad_h3_reduce_serial_kernel (one thread per float4 of dW walks the splits; the planner uses it for the deep levels
    of the 12-level net: <= 64 splits of >= 8192 float4) forced onto every layer of a small net."""
    monkeypatch.setenv("WUNET_REDUCE_SERIAL", "100000,0")
    test_fp16_split_train_step_matches_oracle(emu_engine_h3, 3, 16, 3, 1024, "mse")


@pytest.mark.parametrize("switch,h3,cfg", [("WUNET_NO_SKIP_FUSE", 2, (4, 16, 3, 1024)),   # decoder-side pass reads the skip itself (eval mode's path) vs written by the encoder side
                                           # conv_h3d_kernel: 8 persistent blocks walk all work items (the emulator's default grid is one item per block), split-K stages
                                           ("WUNET_H3_GRID=8", 2, (2, 24, 2, 1024)),
                                           # ... un-split (bias + BatchNorm statistics in the epilogue), 64-row blocks, edge tiles of every item
                                           ("WUNET_H3_GRID=8 WUNET_H3_NOSPLIT=1 WUNET_H3_ORDER=432 WUNET_H3D_ORDER=432", 2, (2, 32, 3, 1024)),
                                           # ... channel counts off the 8 / 32 grid (zero-page pieces in the last chunk), 16 blocks
                                           ("WUNET_H3_GRID=16 WUNET_H3_NOSPLIT=1", 2, (1, 20, 3, 2048)),
                                           # ... tiles of 2 .. 8 whole items (128 .. 32 samples), odd batch: items beyond the batch in the last tile
                                           ("WUNET_H3_GRID=8", 2, (6, 16, 5, 1024)),
                                           ("WUNET_H3_GRID=8 WUNET_H3_NOSPLIT=1", 2, (5, 12, 3, 1024))])
def test_planner_hooks_are_bit_identical(switch, h3, cfg, monkeypatch):
    """The result must not depend on how the work is dealt out: one training step with and without the hook (persistent blocks
    walking many work items vs one item per block; the skip half written by either operand pass) gives the same output and the same
    gradients, bit for bit.  ("A=v B=w ...": A=v is the hook, the other settings hold for both runs.)"""
    n, ci, B, T = cfg
    eng_mod = importlib.import_module(PKG_NAME + ".engine")
    lib_mod = importlib.import_module(PKG_NAME + "._lib")
    settings = [kv.split("=") if "=" in kv else [kv, "1"] for kv in switch.split()]
    for k, v in settings[1:]:
        monkeypatch.setenv(k, v)
    results = []
    for on in (False, True):
        if on:
            monkeypatch.setenv(settings[0][0], settings[0][1])
        eng = eng_mod.Engine(lib=lib_mod.declare(emu_lib.lib()), host_memory=True, h3=h3)
        m, sd, pkg_loss = _build(n, ci, eng)
        noisy, clean = plan.golden_batch(B, T, 0)
        crit = pkg_loss.mse_loss()
        crit._engine_override = eng
        m.train()
        out = m(torch.from_numpy(noisy))
        crit(torch.from_numpy(clean), out).backward()
        results.append([out.detach().numpy().copy()] + [p.grad.numpy().copy() for p in m.parameters()])
    for a, b in zip(*results):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("cfg", [(2, 24, 2, 1024), (3, 16, 3, 768), (4, 40, 5, 512)])
def test_k_tail_equals_the_padded_chunk_up_to_rounding(cfg, monkeypatch):
    """conv_h3d_kernel's K tail sums the same products as the zero-padded last chunk in another order: one training step with and
    without it (WUNET_H3_KTAIL=0) agrees to fp32 rounding in the output and in every gradient - and is not the same run twice
    (the tail really is planned for these channel counts)."""
    n, ci, B, T = cfg
    eng_mod = importlib.import_module(PKG_NAME + ".engine")
    lib_mod = importlib.import_module(PKG_NAME + "._lib")
    results = []
    for kt in ("1", "0"):
        monkeypatch.setenv("WUNET_H3_KTAIL", kt)
        eng = eng_mod.Engine(lib=lib_mod.declare(emu_lib.lib()), host_memory=True, h3=2)
        m, sd, pkg_loss = _build(n, ci, eng)
        noisy, clean = plan.golden_batch(B, T, 0)
        crit = pkg_loss.mse_loss()
        crit._engine_override = eng
        m.train()
        out = m(torch.from_numpy(noisy))
        crit(torch.from_numpy(clean), out).backward()
        results.append([out.detach().numpy().copy()] + [p.grad.numpy().copy() for p in m.parameters()])
    same = True
    for a, b in zip(*results):
        assert np.abs(a - b).max() <= 3e-4 * max(np.abs(b).max(), 1e-6) + 1e-6
        same = same and np.array_equal(a, b)
    assert not same


@pytest.mark.parametrize("T", [1024, 768])        # 768 = 3 * 2^8: padded rows, the padded copies of input / result / gradient
def test_fp16_split_backward_in_buckets_equals_whole(emu_engine_h3, T):
    """wunet_backward_range in three buckets == one wunet_backward, bit for bit, with the split kernels forced on (per-layer
    gradient scales and split weight packs have to survive the bucket boundaries) - the path GradSync drives."""
    eng = emu_engine_h3
    n, ci, B = 2, 24, 2
    sd = plan.golden_state(n, ci, 0)
    noisy, _ = plan.golden_batch(B, T, 0)
    names, bnames = plan.param_names(n, ci), plan.buffer_names(n, ci)
    params = [torch.from_numpy(sd[k].copy()) for k in names]
    running = [torch.from_numpy(sd[k].copy()) for k in bnames if k.endswith("running_mean") or k.endswith("running_var")]
    nbt = [torch.from_numpy(np.asarray(sd[k]).copy()).reshape(()).to(torch.int64) for k in bnames if k.endswith("num_batches_tracked")]
    x = torch.from_numpy(noisy)
    out, ws = eng.forward(n, ci, x, params, running, nbt, True, True)
    gout = torch.from_numpy(np.random.default_rng(1).standard_normal(out.shape).astype(np.float32))
    g1 = [torch.zeros_like(p) for p in params]
    g2 = [torch.zeros_like(p) for p in params]
    eng.backward(n, ci, x, params, out, gout, ws, g1)
    nl = 2 * n + 1
    for lb, le in [(3, nl), (1, 3), (0, 1)]:
        eng.backward(n, ci, x, params, out, gout, ws, g2, layer_range=(lb, le))
    for k, a, b in zip(names, g1, g2):
        assert torch.equal(a, b), k


@pytest.mark.parametrize("T", [64, 96])          # 96 = 3 * 2^5: rows padded to 128 inside
def test_eval_forward_matches_oracle(emu_engine, T):
    n, ci, B = 3, 4, 2
    m, sd, _ = _build(n, ci, emu_engine)
    noisy, _ = plan.golden_batch(B, T, 0)
    ref = c_oracle.step(sd, noisy, None, n, ci, False, want_grads=False, precision="f64")
    m.eval()
    with torch.no_grad():
        out = m(torch.from_numpy(noisy))
    assert np.abs(out.numpy() - ref["out"]).max() < 2e-5
    before = plan.golden_state(n, ci, 0)
    for k in plan.buffer_names(n, ci):
        assert np.array_equal(m.state_dict()[k].numpy(), before[k]), k


@pytest.mark.parametrize("n,ci,B,T", [(2, 8, 2, 96), (4, 8, 2, 384), (3, 8, 2, 768), (2, 8, 4, 192)])
def test_fp16_split_on_padded_lengths_eval_and_running_statistics(emu_engine_h3, n, ci, B, T):
    """Lengths m * 2^n on the split path where a conv runs UN-split into the split buffer (conv_reduce_bn_kernel then adds the bias
    and skips the row padding in the statistics): the conv itself must not add the bias as well.  Behind training-mode BatchNorm a
    doubled bias cancels in the output and in every gradient - it shows in the eval-mode output (0.03 off before the fix, found on
    the hardware at 12 levels x 12288 samples) and in the running means the training forward leaves behind."""
    eng = emu_engine_h3
    m, sd, _ = _build(n, ci, eng)
    noisy, clean = plan.golden_batch(B, T, 0)
    ref = c_oracle.step({k: v.copy() for k, v in sd.items()}, noisy, None, n, ci, False, want_grads=False, precision="f64")
    m.eval()
    with torch.no_grad():
        out = m(torch.from_numpy(noisy))
    assert np.abs(out.numpy() - ref["out"]).max() < 2e-5
    sd2 = {k: v.copy() for k, v in plan.golden_state(n, ci, 0).items()}
    c_oracle.step(sd2, noisy, clean, n, ci, True, "mse", precision="f64")          # (updates the running statistics in sd2)
    m.train()
    m(torch.from_numpy(noisy))
    post = m.state_dict()
    for k in plan.buffer_names(n, ci):
        assert np.abs(post[k].numpy().astype(np.float64) - sd2[k]).max() < 1e-5, k


@pytest.mark.parametrize("n,ci,B,T,order", [(2, 24, 2, 1024, ""),        # decoder form (upsampled + skip rows) twice, encoder form once
                                              (3, 24, 3, 2048, ""),        # c0 = 72: an m-tile whose rows belong to two producers; odd batch
                                              (3, 20, 2, 1024, "432")])    # four accumulator rows per wave, channel counts that are not multiples of 8
def test_bn_backward_sums_in_the_data_gradient_epilogue(monkeypatch, n, ci, B, T, order):
    """conv_h3d_kernel<.., BSUM>: the data gradient's epilogue takes sum g, sum g xhat and the bounds of the layers that produced its rows
    (through the x2 upsample: against the upsampled mask - the sums are linear in the data gradient; skip rows; decimated rows), those layers
    run no pass_a_kernel, bn_finalize_bwd_tiles_kernel adds the per-tile rows, gz_split_h3_kernel forms g_z from the data gradients (UP mode:
    the transposed upsample moved there).  Against the float64 oracle at the bars of test_fp16_split_train_step_matches_oracle, and against
    the same step with the epilogue off (WUNET_BSUM=0, read when a context is planned): the two take the same sums in another order."""
    import ctypes
    from test_scale_robustness import errors, run_step
    eng_mod = importlib.import_module(PKG_NAME + ".engine")
    lib_mod = importlib.import_module(PKG_NAME + "._lib")
    monkeypatch.setenv("WUNET_H3_NOSPLIT", "1")          # (small shapes: un-split data gradients, as at the BASELINE size)
    monkeypatch.setenv("WUNET_UPT", "0")                 # (against the classic gradient assembly: pass_a_kernel<UP> on full-resolution rows)
    if order:
        monkeypatch.setenv("WUNET_H3D_ORDER", order)
    sd = plan.golden_state(n, ci, 0)
    noisy, clean = plan.golden_batch(B, T, 0)
    ref = c_oracle.step({k: v.copy() for k, v in sd.items()}, noisy, clean, n, ci, True, "mse", precision="f64")
    got = {}
    for bs in ("256", "0"):
        monkeypatch.setenv("WUNET_BSUM", bs)
        eng = eng_mod.Engine(lib=lib_mod.declare(emu_lib.lib()), host_memory=True, h3=2)
        eng.lib.wunet_profile_enable(1)
        out, grads = run_step(eng, sd, n, ci, noisy, clean)
        buf = ctypes.create_string_buffer(1 << 16)
        eng.lib.wunet_profile_collect(buf, len(buf))
        eng.lib.wunet_profile_enable(0)
        names = {ln.split("\t")[0]: int(ln.split("\t")[1]) for ln in buf.value.decode().strip().splitlines()}
        got[bs] = (out, grads, names)
        oe, ge = errors(out, grads, ref)
        assert oe < 2e-5 and ge < 3e-4, (bs, oe, ge)
    on, off = got["256"][2], got["0"][2]
    assert sum(v for k, v in on.items() if k.endswith(", bsum>")) >= 3 and not any(k.endswith(", bsum>") for k in off)
    assert "pass_a_kernel<UP>" not in on and off["pass_a_kernel<UP>"] == n          # middle + decoders 0 .. n-2
    # (the first layer keeps its pass - its g feeds the fp32 weight gradient - and so does an encoder layer whose decimating consumer runs
    #  below 256 samples)
    enc = lambda d: d.get("pass_a_kernel<ENC>", 0) + d.get("pass_a_kernel<ENC, e0>", 0)      # (the first layer's pass may carry its weight-gradient sums)
    assert 1 <= enc(on) < enc(off) == n
    for k, g in got["256"][1].items():
        r = got["0"][1][k]
        assert np.abs(g - r).max() <= 2e-5 * max(np.abs(r).max(), 1e-12), k


@pytest.mark.parametrize("n,ci,B,T,order", [(2, 24, 2, 1024, ""), (3, 24, 3, 2048, ""), (3, 20, 2, 1024, "432")])
def test_upsample_transpose_in_the_data_gradient_epilogue(monkeypatch, n, ci, B, T, order):
    """conv_h3d_kernel<.., 3> (UPT): a decoder layer's data gradient stores the rows of the upsampled half of its input already pulled back
    through the x2 upsample (the transposed upsample on the accumulators: neighbour lanes by DPP, neighbour waves through LDS, neighbour
    tiles through the edge terms the reader adds), the producer's gradient assembly is the elementwise pass_a_kernel<UPH> and its g is
    re-formed by gz_split_h3_kernel from the same half-resolution array.  Against the float64 oracle and against the same step with the
    full-resolution rows + pass_a_kernel<UP> (WUNET_UPT=0): the same multiply-adds in the same order except at the two edge inputs of a
    128-input tile."""
    import ctypes
    from test_scale_robustness import errors, run_step
    eng_mod = importlib.import_module(PKG_NAME + ".engine")
    lib_mod = importlib.import_module(PKG_NAME + "._lib")
    monkeypatch.setenv("WUNET_H3_NOSPLIT", "1")          # (small shapes: un-split data gradients, as at the BASELINE size)
    if order:
        monkeypatch.setenv("WUNET_H3D_ORDER", order)
    sd = plan.golden_state(n, ci, 0)
    noisy, clean = plan.golden_batch(B, T, 0)
    ref = c_oracle.step({k: v.copy() for k, v in sd.items()}, noisy, clean, n, ci, True, "mse", precision="f64")
    got = {}
    for v in ("1", "0"):
        monkeypatch.setenv("WUNET_UPT", v)
        eng = eng_mod.Engine(lib=lib_mod.declare(emu_lib.lib()), host_memory=True, h3=2)
        eng.lib.wunet_profile_enable(1)
        out, grads = run_step(eng, sd, n, ci, noisy, clean)
        buf = ctypes.create_string_buffer(1 << 16)
        eng.lib.wunet_profile_collect(buf, len(buf))
        eng.lib.wunet_profile_enable(0)
        names = {ln.split("\t")[0]: int(ln.split("\t")[1]) for ln in buf.value.decode().strip().splitlines()}
        got[v] = (out, grads, names)
        oe, ge = errors(out, grads, ref)
        assert oe < 2e-5 and ge < 3e-4, (v, oe, ge)
    on, off = got["1"][2], got["0"][2]
    n_upt = sum(c for k, c in on.items() if k.endswith(", upt>"))
    assert n_upt >= 2 and on.get("pass_a_kernel<UPH>", 0) == n_upt and not any(k.endswith(", upt>") or k == "pass_a_kernel<UPH>" for k in off)
    assert on.get("pass_a_kernel<UP>", 0) + n_upt == off["pass_a_kernel<UP>"]
    for k, g in got["1"][1].items():
        r = got["0"][1][k]
        assert np.abs(g - r).max() <= 1e-5 * max(np.abs(r).max(), 1e-12), k


def test_fused_adam_matches_torch(emu_engine):
    """the fused Adam launch against torch.optim.Adam (train.py:31-35), three steps."""
    optim_mod = importlib.import_module(PKG_NAME + ".optim")
    torch.manual_seed(0)
    shapes = [(7, 3, 15), (7,), (33,), (1, 5, 1), (300,)]
    ref_p = [torch.randn(s, requires_grad=True) for s in shapes]
    our_p = [p.detach().clone().requires_grad_(True) for p in ref_p]
    ref = torch.optim.Adam(ref_p, lr=1e-3, betas=(0.9, 0.999))
    ours = optim_mod.FusedAdam(our_p, lr=1e-3, betas=(0.9, 0.999))
    ours._engine_override = emu_engine
    for it in range(3):
        for a, b in zip(ref_p, our_p):
            g = torch.randn_like(a) * (10.0 ** (it - 1))
            a.grad = g.clone()
            b.grad = g.clone()
        ref.step()
        ours.step()
        for a, b in zip(ref_p, our_p):
            assert (a - b).abs().max().item() < 2e-7, it
    sd_ref, sd_our = ref.state_dict(), ours.state_dict()
    assert sd_ref["state"].keys() == sd_our["state"].keys()
    for k in sd_ref["state"]:
        assert set(sd_ref["state"][k].keys()) == set(sd_our["state"][k].keys())
        assert (sd_ref["state"][k]["exp_avg_sq"] - sd_our["state"][k]["exp_avg_sq"]).abs().max().item() < 1e-7


@pytest.mark.parametrize("n,ci,B,T,grid,order", [(2, 24, 2, 1024, "8", ""),      # persistent blocks walk two items each; chunks with both branches
                                                 (3, 24, 1, 2048, "4", ""),      # three decoder levels on the kernel, 8 / 4 / 2 tiles per row
                                                 (2, 16, 3, 512, "", ""),        # 16-channel groups: one branch per chunk pair, a half-empty last chunk
                                                 (2, 24, 2, 1024, "8", "4")])    # four accumulator rows per wave (conv_h3u_kernel<4>: eight runs per W sub-tile)
def test_fused_operand_conv_matches_the_two_kernel_path(monkeypatch, n, ci, B, T, grid, order):
    """conv_h3u_kernel (wunet_h3u.h: the decoder conv whose loader waves build the operand from the producers' raw conv outputs - BatchNorm
    scale / shift, LeakyReLU, ATen's upsample coordinates, concat, split - instead of reading what prep_h3_kernel wrote) against the oracle
    and against prep_h3_kernel + conv_h3d_kernel, in eval mode (its product use) and in training mode (statistics rows, the operand written
    out for the weight gradient).  WUNET_H3U = "<eval min L>,<train min L>" is read when a context is planned: every arm its own engine."""
    eng_mod = importlib.import_module(PKG_NAME + ".engine")
    lib_mod = importlib.import_module(PKG_NAME + "._lib")
    pkg_loss = importlib.import_module(PKG_NAME + ".loss")
    monkeypatch.setenv("WUNET_H3_NOSPLIT", "1")
    if grid:
        monkeypatch.setenv("WUNET_H3_GRID", grid)
    if order:
        monkeypatch.setenv("WUNET_H3_ORDER", order)
    noisy, clean = plan.golden_batch(B, T, 0)
    sd = plan.golden_state(n, ci, 0)
    ref_e = c_oracle.step(sd, noisy, clean, n, ci, False, "mse", want_grads=False, precision="f64")
    ref_t = c_oracle.step(sd, noisy, clean, n, ci, True, "mse", precision="f64")

    ran = {}

    def run(h3u, training):
        monkeypatch.setenv("WUNET_H3U", h3u)
        eng = eng_mod.Engine(lib=lib_mod.declare(emu_lib.lib()), host_memory=True, h3=2)
        m, _, _ = _build(n, ci, eng)
        eng.lib.wunet_profile_enable(1)                  # (emulator: the names of the annotated kernels that ran)
        try:
            if not training:
                m.eval()
                with torch.no_grad():
                    return m(torch.from_numpy(noisy)).numpy(), None
            m.train()
            crit = pkg_loss.mse_loss()
            crit._engine_override = eng
            out = m(torch.from_numpy(noisy))
            crit(torch.from_numpy(clean), out).backward()
            return out.detach().numpy(), {k: p.grad.numpy().copy() for k, p in m.named_parameters()}
        finally:
            buf = ctypes.create_string_buffer(1 << 16)
            eng.lib.wunet_profile_collect(buf, len(buf))
            eng.lib.wunet_profile_enable(0)
            ran[(h3u, training)] = [ln.split("\t")[0] for ln in buf.value.decode().splitlines()]

    e_old, _ = run("0,0", False)
    e_new, _ = run("256,256", False)
    assert np.abs(e_new - ref_e["out"]).max() < 2e-6 and np.abs(e_new - e_old).max() < 1e-6
    want = "conv_h3u_kernel<4>" if order == "4" else "conv_h3u_kernel<"
    assert any(k.startswith(want) for k in ran[("256,256", False)]) and not any("h3u" in k for k in ran[("0,0", False)]), ran
    if ci == 24 and not order:
        assert not np.array_equal(e_new, e_old)     # (these layers' K tails vs whole chunks add in another order)
    t_old, g_old = run("0,0", True)
    t_new, g_new = run("256,256", True)
    assert np.abs(t_new - ref_t["out"]).max() < 2e-5
    for k, r in ref_t["grads"].items():
        if k.endswith(".0.bias") and not k.startswith("out"):
            assert np.all(g_new[k] == 0.0), k
            continue
        scale = max(np.abs(r).max(), 1e-6)
        assert np.abs(g_new[k] - r).max() < 3e-4 * scale + 1e-6, (k, np.abs(g_new[k] - r).max(), scale)
        assert np.abs(g_new[k] - g_old[k]).max() < 1e-4 * scale + 1e-7, k
    assert any(k.startswith(want) for k in ran[("256,256", True)]), ran


@pytest.mark.parametrize("n,ci,B,T,grid", [(3, 24, 2, 2048, ""), (4, 16, 3, 4096, "8"), (3, 20, 2, 1024, "")])
def test_eval_encoder_conv_writes_the_next_operand(monkeypatch, n, ci, B, T, grid):
    """Eval mode (enhancement.py:43,65-66): the BatchNorm coefficients are known before the chain starts, so an encoder level's conv
    (conv_h3d_kernel<.., EVOP>) applies them, LeakyReLU and the `[:, :, ::2]` of unet_basic.py:86 in its epilogue and writes the next
    level's split operand itself - scaled by a rigorous bound of its activation (absolute weight row sums x the measured maximum of its
    input), since the measured maximum only exists when the launch ends.  Against the oracle and against the two-kernel path
    (WUNET_NO_EVOP=1: prep_h3_kernel<0> + the measured scale); channel counts that are no multiple of 8; persistent blocks walking
    several tiles; a second forward on the cached weight packs (and row sums) gives the same bits."""
    eng_mod = importlib.import_module(PKG_NAME + ".engine")
    lib_mod = importlib.import_module(PKG_NAME + "._lib")
    monkeypatch.setenv("WUNET_H3_NOSPLIT", "1")
    if grid:
        monkeypatch.setenv("WUNET_H3_GRID", grid)
    noisy, clean = plan.golden_batch(B, T, 0)
    sd = plan.golden_state(n, ci, 0)
    ref = c_oracle.step(sd, noisy, clean, n, ci, False, "mse", want_grads=False, precision="f64")["out"]

    def run(no_evop):
        if no_evop:
            monkeypatch.setenv("WUNET_NO_EVOP", "1")
        else:
            monkeypatch.delenv("WUNET_NO_EVOP", raising=False)
        eng = eng_mod.Engine(lib=lib_mod.declare(emu_lib.lib()), host_memory=True, h3=2)
        m, _, _ = _build(n, ci, eng)
        m.eval()
        with torch.no_grad():
            return m(torch.from_numpy(noisy)).numpy(), m(torch.from_numpy(noisy)).numpy()

    old, _ = run(True)
    new, again = run(False)
    assert np.abs(new - ref).max() < 2e-6 and np.abs(new - old).max() < 1e-6
    assert not np.array_equal(new, old)             # (another scale, another rounding of the lo halves: really the other path)
    assert np.array_equal(new, again)


@pytest.mark.parametrize("n,ci,B,T", [(2, 24, 2, 1024), (3, 24, 3, 2048), (2, 16, 5, 512), (2, 24, 2, 768), (3, 24, 1, 1536)])
def test_first_layer_weight_gradient_from_the_gradient_assembly_sums(monkeypatch, n, ci, B, T):
    """The first layer (Cin = 1, model/unet_basic.py:44-50): dW[c][k] = sum g_z[b,c,l] x[b,l+k-7] with g_z = k1 g + k2 z + k3 is three sums per tap that
    need no BatchNorm constant; pass_a_kernel<.., E0> takes them while it forms g (which it then does not write), bn_finalize_bwd_kernel combines them - no
    weight-gradient GEMM, no split reduce for that layer.  Against the float64 oracle and against the same step with the GEMM (WUNET_NO_E0=1)."""
    import ctypes
    from test_scale_robustness import errors, run_step
    eng_mod = importlib.import_module(PKG_NAME + ".engine")
    lib_mod = importlib.import_module(PKG_NAME + "._lib")
    monkeypatch.setenv("WUNET_H3_NOSPLIT", "1")          # (small shapes: encoder.1's data gradient un-split, as at the BASELINE size)
    sd = plan.golden_state(n, ci, 0)
    noisy, clean = plan.golden_batch(B, T, 0)
    ref = c_oracle.step({k: v.copy() for k, v in sd.items()}, noisy, clean, n, ci, True, "mse", precision="f64")
    got = {}
    for v in ("", "1"):
        if v:
            monkeypatch.setenv("WUNET_NO_E0", v)
        else:
            monkeypatch.delenv("WUNET_NO_E0", raising=False)
        eng = eng_mod.Engine(lib=lib_mod.declare(emu_lib.lib()), host_memory=True, h3=2)
        eng.lib.wunet_profile_enable(1)
        out, grads = run_step(eng, sd, n, ci, noisy, clean)
        buf = ctypes.create_string_buffer(1 << 16)
        eng.lib.wunet_profile_collect(buf, len(buf))
        eng.lib.wunet_profile_enable(0)
        names = {ln.split("\t")[0]: int(ln.split("\t")[1]) for ln in buf.value.decode().strip().splitlines()}
        got[v] = (out, grads, names)
        oe, ge = errors(out, grads, ref)
        assert oe < 2e-5 and ge < 3e-4, (v, oe, ge)
    assert got[""][2].get("pass_a_kernel<ENC, e0>", 0) == 1 and "pass_a_kernel<ENC, e0>" not in got["1"][2]
    k = "encoder.0.main.0.weight"
    a, b = got[""][1][k], got["1"][1][k]
    assert np.abs(a - b).max() <= 2e-5 * np.abs(b).max(), np.abs(a - b).max() / np.abs(b).max()
    for kk, g in got[""][1].items():            # nothing else moves
        if kk != k:
            assert np.array_equal(g, got["1"][1][kk]), kk
