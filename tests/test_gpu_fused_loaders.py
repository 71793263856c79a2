"""-m gpu: the kernels that carry an operand pass inside a conv - conv_h3u_kernel<2|3|4> (decoder levels: BatchNorm + LeakyReLU +
x2 upsample + concat in the conv's loader waves, model/unet_basic.py:91-96) and conv_h3d_kernel<.., evop> (eval encoder levels: the
next level's operand written by the conv's epilogue, :83-86) - pinned on the hardware at the BASELINE geometry:
  (i)   the default planner really launches them (wunet_profile_collect names), so a threshold regression cannot pass silently;
  (ii)  the raw conv outputs of decoder.7-11 and encoder.2-6 against the reference's own ATen CPU ops (oracle/torch_port.py's op
        sequence, here with the per-layer tensors kept) within 2e-5 of the layer's maximum - the bar of test_layer_activations_vs_oracle;
  (iii) the same layers against the two-kernel path (WUNET_H3U=0,0 / WUNET_NO_EVOP=1, read when a context is planned): same MFMA order
        on the same operand values -> the same z up to the operand scale's binade (EVOP scales by a rigorous bound instead of the
        measured maximum: 22-bit operands either way, so 'equal' means 2e-6 of the layer's maximum, and bit-equal for conv_h3u in training);
  (iv)  a length that is not a power of two (12288 = 3 * 2^12: padded rows - the planner must NOT take the fused loaders there) and a
        ragged batch (B = 5: the last tile rows of the grid are partly outside the batch)."""
import ctypes
import importlib
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import PKG_NAME
from oracle import plan, torch_port

pytestmark = pytest.mark.gpu

N, CI = 12, 24
DEC = [N + 1 + j for j in (7, 8, 9, 10, 11)]        # conv layer index of decoder.7 .. decoder.11
ENC = [2, 3, 4, 5, 6]


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need an MI355X"
    return torch.device("cuda:0")


def _engine(env):
    """A private engine whose contexts are planned under `env` (the planner reads WUNET_H3U / WUNET_NO_EVOP when a context is created)."""
    eng_mod = importlib.import_module(PKG_NAME + ".engine")
    e = eng_mod.Engine()
    e._plan_env = dict(env)
    return e


class _planned_under:
    def __init__(self, env):
        self.env, self.old = env, {}

    def __enter__(self):
        for k, v in self.env.items():
            self.old[k] = os.environ.get(k)
            os.environ[k] = v

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _reference_layers(sd_np, noisy, training):
    """Raw conv outputs z_i of every conv layer from the reference's op sequence on the CPU (torch_port.forward's, with the tensors kept)."""
    sd = torch_port.state_to_torch(sd_np)
    layers = plan.conv_layers(N, CI)
    acts, skips = [], []
    h = torch.from_numpy(noisy)

    def block(x, prefix, k):
        z = F.conv1d(x, sd[f"{prefix}.0.weight"], sd[f"{prefix}.0.bias"], padding=k // 2)
        acts.append(z)
        y = F.batch_norm(z, sd[f"{prefix}.1.running_mean"].clone(), sd[f"{prefix}.1.running_var"].clone(), sd[f"{prefix}.1.weight"],
                         sd[f"{prefix}.1.bias"], training, 0.1, 1e-5)
        return F.leaky_relu(y, 0.1)
    with torch.no_grad():
        for prefix, _, _, k in layers[:N]:
            h = block(h, prefix, k)
            skips.append(h)
            h = h[:, :, ::2]
        h = block(h, layers[N][0], layers[N][3])
        for j, (prefix, _, _, k) in enumerate(layers[N + 1:]):
            h = F.interpolate(h, scale_factor=2, mode="linear", align_corners=True)
            h = block(torch.cat([h, skips[N - 1 - j]], dim=1), prefix, k)
    return acts


def _run(engine, env, pkg, dev, noisy, training):
    """One forward on `engine` (contexts planned under env): (kernel names, {layer: z})."""
    sd = plan.golden_state(N, CI, 0)
    m = pkg.Model(n_layers=N, channels_interval=CI)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    m.to(dev).train(training)
    running, nbt = m._wunet_buffers()
    B, _, T = noisy.shape
    x = torch.from_numpy(noisy).to(dev)
    params = [p.detach() for p in m._wunet_params()]
    lib = engine.lib
    with _planned_under(env):
        lib.wunet_profile_enable(1)
        try:
            out, ws = engine.forward(N, CI, x, params, running, nbt, training, False)
            torch.cuda.synchronize()
            buf = ctypes.create_string_buffer(1 << 16)
            lib.wunet_profile_collect(buf, len(buf))
        finally:
            lib.wunet_profile_enable(0)
        names = [ln.split("\t")[0] for ln in buf.value.decode().strip().splitlines()]
        zs = {i: engine.layer_output(N, CI, B, T, ws, i).clone() for i in DEC + ENC}
    return names, zs, out


@pytest.mark.parametrize("training", [False, True], ids=["eval", "train"])
def test_fused_loader_kernels_run_and_match_at_batch_64(pkg, dev, training):
    B, T = 64, 16384
    noisy, _ = plan.golden_batch(B, T, 5)
    names, zs, out = _run(_engine({}), {}, pkg, dev, noisy, training)
    # (i) the default planner launches them
    for mrep in (2, 3, 4) if not training else (2, 3):
        assert any(nm.startswith("conv_h3u_kernel<%d>" % mrep) for nm in names), (mrep, sorted(set(names)))
    if not training:
        assert sum(nm.endswith(", evop>") for nm in names) >= 1, sorted(set(names))
    else:
        assert not any(nm.endswith(", evop>") for nm in names)
    # (ii) against the reference's ops on the host
    ref = _reference_layers(plan.golden_state(N, CI, 0), noisy, training)
    for i in DEC + ENC:
        r = ref[i]
        err = (zs[i].cpu() - r).abs().max().item()
        assert err < 2e-5 * max(1.0, r.abs().max().item()), (i, err)
    # (iii) against the two-kernel path
    off = {"WUNET_H3U": "0,0", "WUNET_NO_EVOP": "1"}
    names0, zs0, out0 = _run(_engine(off), off, pkg, dev, noisy, training)
    assert not any(nm.startswith("conv_h3u_kernel") or nm.endswith(", evop>") for nm in names0), sorted(set(names0))
    for i in DEC + ENC:
        d = (zs[i] - zs0[i]).abs().max().item()
        scale = max(1.0, zs0[i].abs().max().item())
        assert d < 2e-6 * scale, (i, d)
        if training and i in DEC and d != 0.0:
            # (same operand values - training scales derive from the BatchNorm bound either way - and the same MFMA order: expected
            #  bit-equal; reported, not asserted, because an fma contraction in the loaders' conversion is allowed to differ)
            print(f"decoder layer {i}: conv_h3u vs prep_h3 + conv_h3d not bit-equal, max |diff| {d:.2e}")
    assert (out - out0).abs().max().item() < 2e-6


@pytest.mark.parametrize("B,T", [(5, 16384), (8, 12288)], ids=["ragged_batch", "padded_rows"])
def test_fused_loaders_on_ragged_and_padded_shapes(pkg, dev, B, T):
    noisy, _ = plan.golden_batch(B, T, 7)
    names, zs, out = _run(_engine({}), {}, pkg, dev, noisy, False)
    padded = T & (T - 1) != 0
    if padded:       # rows padded to the next power of two: the fused loaders know no row padding - the planner must not use them
        assert not any(nm.startswith("conv_h3u_kernel") or nm.endswith(", evop>") for nm in names), sorted(set(names))
    else:
        assert any(nm.startswith("conv_h3u_kernel") for nm in names), sorted(set(names))
    sd = torch_port.state_to_torch(plan.golden_state(N, CI, 0))
    with torch.no_grad():
        ref = torch_port.forward(sd, torch.from_numpy(noisy), N, CI, False)
    assert (out.cpu() - ref).abs().max().item() < 1e-4
    off = {"WUNET_H3U": "0,0", "WUNET_NO_EVOP": "1"}
    _, _, out0 = _run(_engine(off), off, pkg, dev, noisy, False)
    assert (out - out0).abs().max().item() < 2e-6


def test_bn_backward_sums_in_the_data_gradient_epilogue(pkg, dev):
    """conv_h3d_kernel<.., BSUM> on the hardware (off by default - profiles/r6_bsum_ab.txt - so the planner switch WUNET_BSUM is set here): a
    training step at batch 64 x 16384 with the BatchNorm-backward sums taken in the data gradients' epilogues from 1024 samples up equals the
    same step with pass_a_kernel to the rounding of another summation order (1e-5 of each gradient tensor's maximum), the kernels really run
    (profile names), and pass A is gone for those layers (unet_basic.py:12-13,25-26,86,93-95 backwards)."""
    B, T = 64, 16384
    noisy, clean = plan.golden_batch(B, T, 9)
    sd = plan.golden_state(N, CI, 0)
    eng_mod = importlib.import_module(PKG_NAME + ".engine")
    res = {}
    for bs in ("1024", "0"):
        env = {"WUNET_BSUM": bs, "WUNET_UPT": "0"}       # (both arms on full-resolution rows: the comparison is against pass_a_kernel<UP>)
        with _planned_under(env):
            eng = eng_mod.Engine()
            m = pkg.Model(n_layers=N, channels_interval=CI)
            m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
            m.to(dev).train()
            crit = pkg.smooth_l1_loss()
            m._engine_override = crit._engine_override = eng
            eng.lib.wunet_profile_enable(1)
            try:
                out = m(torch.from_numpy(noisy).to(dev))
                crit(torch.from_numpy(clean).to(dev), out).backward()
                torch.cuda.synchronize()
                buf = ctypes.create_string_buffer(1 << 16)
                eng.lib.wunet_profile_collect(buf, len(buf))
            finally:
                eng.lib.wunet_profile_enable(0)
            names = {ln.split("\t")[0]: int(ln.split("\t")[1]) for ln in buf.value.decode().strip().splitlines()}
            res[bs] = (names, {k: p.grad.clone() for k, p in m.named_parameters()})
    on, off = res["1024"][0], res["0"][0]
    assert sum(v for k, v in on.items() if k.endswith(", bsum>")) >= 2 and not any(k.endswith(", bsum>") for k in off), sorted(on)
    assert on.get("pass_a_kernel<UP>", 0) < off["pass_a_kernel<UP>"] and on.get("pass_a_kernel<ENC>", 0) < off["pass_a_kernel<ENC>"]
    for k, g in res["1024"][1].items():
        r = res["0"][1][k]
        assert (g - r).abs().max().item() <= 1e-5 * max(r.abs().max().item(), 1e-12) + 1e-9, k


def test_upsample_transpose_in_the_data_gradient_epilogue(pkg, dev):
    """conv_h3d_kernel<.., 3> (UPT, the default planner at the BASELINE size) on the hardware - DPP row shifts, the LDS hand-over between the
    four waves, the tile-edge terms: a training step at batch 64 x 16384 equals the step with full-resolution rows + pass_a_kernel<UP>
    (WUNET_UPT=0) to the rounding of the edge inputs' other summation order, the kernels run, pass_a_kernel<UP> is gone for those layers
    (model/unet_basic.py:93 backwards)."""
    B, T = 64, 16384
    noisy, clean = plan.golden_batch(B, T, 9)
    sd = plan.golden_state(N, CI, 0)
    eng_mod = importlib.import_module(PKG_NAME + ".engine")
    res = {}
    for v in ("1", "0"):
        with _planned_under({"WUNET_UPT": v}):
            eng = eng_mod.Engine()
            m = pkg.Model(n_layers=N, channels_interval=CI)
            m.load_state_dict({k: torch.from_numpy(a.copy()) for k, a in sd.items()})
            m.to(dev).train()
            crit = pkg.smooth_l1_loss()
            m._engine_override = crit._engine_override = eng
            eng.lib.wunet_profile_enable(1)
            try:
                out = m(torch.from_numpy(noisy).to(dev))
                crit(torch.from_numpy(clean).to(dev), out).backward()
                torch.cuda.synchronize()
                buf = ctypes.create_string_buffer(1 << 16)
                eng.lib.wunet_profile_collect(buf, len(buf))
            finally:
                eng.lib.wunet_profile_enable(0)
            names = {ln.split("\t")[0]: int(ln.split("\t")[1]) for ln in buf.value.decode().strip().splitlines()}
            res[v] = (names, {k: p.grad.clone() for k, p in m.named_parameters()})
    on, off = res["1"][0], res["0"][0]
    n_upt = sum(c for k, c in on.items() if k.endswith(", upt>"))
    assert n_upt >= 5 and on.get("pass_a_kernel<UPH>", 0) == n_upt, sorted(on)
    assert not any(k.endswith(", upt>") or k == "pass_a_kernel<UPH>" for k in off)
    assert on.get("pass_a_kernel<UP>", 0) + n_upt == off["pass_a_kernel<UP>"]
    for k, g in res["1"][1].items():
        r = res["0"][1][k]
        assert (g - r).abs().max().item() <= 1e-5 * max(r.abs().max().item(), 1e-12) + 1e-9, k
