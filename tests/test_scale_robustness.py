"""The conv -> BatchNorm pair makes the conv-weight scale a free gauge (reference model/unet_basic.py:9-14, 22-27): multiplying
`main.0.weight` of any layer by s > 0 leaves the training-mode network function unchanged, and BatchNorm's gamma / beta set the
magnitude of every activation.  A checkpoint may therefore sit at any scale, and the fp16-split GEMM path (csrc/wunet_h3.h) must
not depend on it: weights, activations and gradients are carried with power-of-two scales derived on the device.  These tests run
the product kernels in the CPU fiber emulator (split path forced onto every level) against the f64 oracle on re-scaled
networks; tests/test_gpu_parity.py repeats them on the hardware."""
import importlib

import numpy as np
import pytest
import torch

import emu_lib
from conftest import PKG_NAME
from oracle import c_oracle, plan


def _engine(h3):
    eng_mod = importlib.import_module(PKG_NAME + ".engine")
    lib_mod = importlib.import_module(PKG_NAME + "._lib")
    return eng_mod.Engine(lib=lib_mod.declare(emu_lib.lib()), host_memory=True, h3=h3)


def rescaled_state(n, ci, wscale=1.0, gscale=1.0, bscale=None, per_layer=False, seed=0):
    """golden_state with every conv weight feeding a BatchNorm multiplied by wscale (per_layer: wscale ** u, u ~ U(-1,1) per
    layer), BN gamma by gscale and beta by bscale (default: gscale)."""
    sd = plan.golden_state(n, ci, seed)
    rng = np.random.Generator(np.random.PCG64(77))
    for prefix, _, _, _ in plan.conv_layers(n, ci):
        s = wscale ** rng.uniform(-1, 1) if per_layer else wscale
        sd[prefix + ".0.weight"] = (sd[prefix + ".0.weight"] * np.float32(s)).astype(np.float32)
        sd[prefix + ".0.bias"] = (sd[prefix + ".0.bias"] * np.float32(s)).astype(np.float32)
        sd[prefix + ".1.weight"] = (sd[prefix + ".1.weight"] * np.float32(gscale)).astype(np.float32)
        sd[prefix + ".1.bias"] = (sd[prefix + ".1.bias"] * np.float32(gscale if bscale is None else bscale)).astype(np.float32)
    return sd


def run_step(eng, sd, n, ci, noisy, clean, loss="mse", train=True):
    pkg_model = importlib.import_module(PKG_NAME + ".model")
    pkg_loss = importlib.import_module(PKG_NAME + ".loss")
    m = pkg_model.Model(n_layers=n, channels_interval=ci)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    m._engine_override = eng
    crit = {"mse": pkg_loss.mse_loss, "l1": pkg_loss.l1_loss, "smooth_l1": pkg_loss.smooth_l1_loss}[loss]()
    crit._engine_override = eng
    if not train:
        m.eval()
        with torch.no_grad():
            return m(torch.from_numpy(noisy)).numpy(), None
    m.train()
    out = m(torch.from_numpy(noisy))
    crit(torch.from_numpy(clean), out).backward()
    return out.detach().numpy(), {k: p.grad.numpy() for k, p in m.named_parameters()}


def errors(out, grads, ref):
    """max |output error|, worst per-tensor gradient error relative to the tensor's norm (conv biases in front of BN excluded:
    their true gradient is 0)."""
    oe = float(np.abs(out - ref["out"]).max())
    ge = 0.0
    if grads is not None:
        for k, g in grads.items():
            if k.endswith(".0.bias") and not k.startswith("out"):
                continue
            r = ref["grads"][k]
            ge = max(ge, float(np.linalg.norm((g - r).ravel()) / max(np.linalg.norm(r.ravel()), 1e-30)))
    return oe, ge


CASES = [dict(wscale=1e-3), dict(wscale=1e-2), dict(wscale=0.05), dict(wscale=1e2), dict(wscale=1e4),
         dict(gscale=0.05), dict(gscale=20.0), dict(wscale=1e-3, gscale=20.0), dict(wscale=1e4, gscale=0.05),
         dict(wscale=1e3, per_layer=True), dict(gscale=1e-3), dict(gscale=2e3, bscale=1.0)]


@pytest.mark.parametrize("case", CASES, ids=lambda c: ",".join(f"{k}={v}" for k, v in c.items()))
@pytest.mark.parametrize("net", [(3, 16, 3, 1024), (2, 24, 2, 1024)], ids=["n3ci16", "n2ci24"])
def test_split_path_is_scale_invariant(net, case):
    """Bars: within 3x of the exact-fp32 MFMA path (+ 2e-6: both sit at the fp32 noise floor) at every scale, and output error
    <= 1e-5 / worst gradient relative-norm error <= 1e-4 wherever the fp32 path itself is 3x inside those bars (gamma x 20
    puts 20x larger values in front of the tanh: fp32 MFMA itself is then 1.2e-5 away from the f64 oracle; before the scales
    existed the split path was at 1e-4 .. 6e-2 on the small-scale cases and NaN at 1e4)."""
    n, ci, B, T = net
    sd = rescaled_state(n, ci, **case)
    noisy, clean = plan.golden_batch(B, T, 0)
    ref = c_oracle.step({k: v.copy() for k, v in sd.items()}, noisy, clean, n, ci, True, "mse", precision="f64")
    out3, g3 = run_step(_engine(2), sd, n, ci, noisy, clean)
    out0, g0 = run_step(_engine(0), sd, n, ci, noisy, clean)
    oe3, ge3 = errors(out3, g3, ref)
    oe0, ge0 = errors(out0, g0, ref)
    assert np.isfinite(out3).all()
    assert oe3 <= max(1e-5, 3 * oe0), (oe3, oe0)
    assert ge3 <= max(1e-4, 3 * ge0), (ge3, ge0)
    assert oe3 <= 3 * oe0 + 2e-6, (oe3, oe0)
    assert ge3 <= 3 * ge0 + 2e-6, (ge3, ge0)


@pytest.mark.parametrize("case", [dict(wscale=1e-3), dict(wscale=1e4), dict(gscale=20.0), dict(gscale=1e-3)],
                         ids=lambda c: ",".join(f"{k}={v}" for k, v in c.items()))
def test_split_path_eval_mode_is_scale_robust(case):
    """Eval mode (enhancement.py:66): BatchNorm uses the running statistics, so activations are NOT normalised by the batch and
    no a-priori bound exists; the split path takes the activation scale from the measured maxima."""
    n, ci, B, T = 3, 16, 3, 1024
    sd = rescaled_state(n, ci, **case)
    if "wscale" in case:      # running statistics consistent with the re-scaled conv output (z scales with the weights)
        for prefix, _, _, _ in plan.conv_layers(n, ci):
            sd[prefix + ".1.running_mean"] = (sd[prefix + ".1.running_mean"] * np.float32(case["wscale"])).astype(np.float32)
            sd[prefix + ".1.running_var"] = (sd[prefix + ".1.running_var"] * np.float32(case["wscale"]) ** 2).astype(np.float32)
    noisy, clean = plan.golden_batch(B, T, 0)
    ref = c_oracle.step({k: v.copy() for k, v in sd.items()}, noisy, clean, n, ci, False, "mse", precision="f64")
    out3, _ = run_step(_engine(2), sd, n, ci, noisy, clean, train=False)
    out0, _ = run_step(_engine(0), sd, n, ci, noisy, clean, train=False)
    e3, e0 = np.abs(out3 - ref["out"]).max(), np.abs(out0 - ref["out"]).max()
    assert np.isfinite(out3).all()
    assert e3 <= 1e-5 and e3 <= 3 * e0 + 2e-6, (e3, e0)


def test_split_path_eval_mode_mismatched_running_statistics():
    """Running statistics that do not describe the data (a fresh model evaluated before training: mean 0, var 1, while the conv
    output is 1000x larger): activations of order 1e3..1e5 must neither overflow fp16 nor lose precision."""
    n, ci, B, T = 3, 16, 2, 1024
    sd = rescaled_state(n, ci, wscale=300.0)
    noisy, clean = plan.golden_batch(B, T, 0)
    ref = c_oracle.step({k: v.copy() for k, v in sd.items()}, noisy, clean, n, ci, False, "mse", precision="f64")
    out3, _ = run_step(_engine(2), sd, n, ci, noisy, clean, train=False)
    out0, _ = run_step(_engine(0), sd, n, ci, noisy, clean, train=False)
    e3, e0 = np.abs(out3 - ref["out"]).max(), np.abs(out0 - ref["out"]).max()
    assert np.isfinite(out3).all()
    assert e3 <= 1e-5 and e3 <= 3 * e0 + 2e-6, (e3, e0)


def trained_like_state(n, ci, B, T, steps=50, lr=3e-3):
    """A state as training leaves it: `steps` Adam steps of the reference's loop (trainer/trainer.py:34-38) from the default-init
    law on the golden batch, run by the CPU restatement (ATen ops).  Rows, layers and BatchNorm parameters drift apart in scale -
    the situation one uniform fixture law (golden_state) never produces."""
    from oracle import torch_port
    sd = torch_port.state_to_torch(plan.golden_state(n, ci, 0), requires_grad=True)
    params = [v for v in sd.values() if v.requires_grad]
    opt = torch.optim.Adam(params, lr=lr, betas=(0.9, 0.999))
    noisy, clean = (torch.from_numpy(a) for a in plan.golden_batch(B, T, 0))
    for _ in range(steps):
        opt.zero_grad()
        torch_port.loss_value("mse", clean, torch_port.forward(sd, noisy, n, ci, True)).backward()
        opt.step()
    return {k: v.detach().numpy().copy() for k, v in sd.items()}


def test_split_path_on_a_trained_like_state():
    n, ci, B, T = 3, 16, 3, 1024
    sd = trained_like_state(n, ci, B, T)
    noisy, clean = plan.golden_batch(B, T, 1)
    ref = c_oracle.step({k: v.copy() for k, v in sd.items()}, noisy, clean, n, ci, True, "mse", precision="f64")
    out3, g3 = run_step(_engine(2), sd, n, ci, noisy, clean)
    out0, g0 = run_step(_engine(0), sd, n, ci, noisy, clean)
    oe3, ge3 = errors(out3, g3, ref)
    oe0, ge0 = errors(out0, g0, ref)
    assert oe3 <= 1e-5 and ge3 <= 1e-4, (oe3, ge3, oe0, ge0)
    assert oe3 <= 3 * oe0 + 2e-6 and ge3 <= 3 * ge0 + 2e-6, (oe3, ge3, oe0, ge0)


def test_split_path_on_a_length_that_is_not_a_power_of_two():
    """model/unet_basic.py:86,93 accepts any length divisible by 2^n_layers.  768 = 3 * 2^8 samples (levels of 768 / 384 / 192 / 96 in
    rows of 1024 / 512 / 256 / 128): training step and eval forward of the split path and of the fp32 path against the f64 oracle,
    including the bucketed backward GradSync drives."""
    n, ci, B, T = 3, 16, 2, 768
    sd = rescaled_state(n, ci)
    noisy, clean = plan.golden_batch(B, T, 0)
    ref = c_oracle.step({k: v.copy() for k, v in sd.items()}, noisy, clean, n, ci, True, "mse", precision="f64")
    for h3 in (2, 0):
        out, grads = run_step(_engine(h3), sd, n, ci, noisy, clean, train=True)
        assert out.shape == (B, 1, T)
        assert np.abs(out - ref["out"]).max() < 1e-5
        for k, g in grads.items():
            if k.endswith(".0.bias") and not k.startswith("out"):
                continue
            r = ref["grads"][k]
            assert np.linalg.norm((g - r).ravel()) <= 1e-4 * np.linalg.norm(r.ravel()) + 1e-9, (h3, k)
    refe = c_oracle.step({k: v.copy() for k, v in sd.items()}, noisy, clean, n, ci, False, "mse", precision="f64")
    oute, _ = run_step(_engine(2), sd, n, ci, noisy, clean, train=False)
    assert np.abs(oute - refe["out"]).max() < 1e-5
