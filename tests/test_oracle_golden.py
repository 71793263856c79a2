"""The CPU restatement (oracle/wunet_oracle.c) against fixtures produced by the imported
reference (tests/golden/make_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest

from conftest import golden
from oracle import c_oracle, plan

# fp32 rounding noise of the 2n+1-layer train-mode forward is ~2-4e-6; the
# north_star budget is 1e-4.  The oracle is held to a tighter bar than the HIP path.
OUT_TOL = 2e-5
GRAD_RTOL = 2e-4


def _run(name, precision, training=True):
    fx = golden(name)
    n, ci, B, T = (int(v) for v in fx["meta"])
    sd = plan.golden_state(n, ci, 0)
    noisy, clean = plan.golden_batch(B, T, 0)
    r = c_oracle.step(sd, noisy, clean, n, ci, training, str(fx["loss_kind"]), want_grads=training,
                      precision=precision)
    return fx, sd, r, (n, ci, B, T)


@pytest.mark.parametrize("precision", ["f32", "f64"])
@pytest.mark.parametrize("name", ["tiny_mse", "small_l1", "small_smoothl1"])
def test_small_cases_full_dump(name, precision):
    fx, sd, r, (n, ci, B, T) = _run(name, precision)
    assert np.abs(r["out"] - fx["out_train"]).max() < OUT_TOL
    assert abs(r["loss"] - float(fx["loss"])) < 1e-5
    for k in plan.param_names(n, ci):
        ref = fx["grad/" + k]
        scale = max(np.abs(ref).max(), 1e-6)
        err = np.abs(r["grads"][k] - ref).max()
        if k.endswith(".0.bias") and not k.startswith("out"):
            # conv bias before training-mode BN: true gradient is 0, reference holds fp32 noise
            assert err < 1e-6, (k, err)
        else:
            assert err < GRAD_RTOL * scale + 1e-7, (k, err, scale)
    for k in plan.buffer_names(n, ci):
        ref = fx["buf/" + k]
        assert np.abs(sd[k].astype(np.float64) - ref).max() < 1e-5, k


@pytest.mark.parametrize("name", ["tiny_mse", "small_l1"])
def test_eval_forward(name):
    fx, sd, r, _ = _run(name, "f64", training=False)
    assert np.abs(r["out"] - fx["out_eval"]).max() < OUT_TOL
    # eval must not touch the running statistics
    n, ci = int(fx["meta"][0]), int(fx["meta"][1])
    ref_sd = plan.golden_state(n, ci, 0)
    for k in plan.buffer_names(n, ci):
        assert np.array_equal(sd[k], ref_sd[k]), k


def test_full_12_level_digest():
    """12-level / 16384-sample / B=2 train step (BASELINE.json configs[0] shape at B=2)."""
    from golden.make_digest import digest
    fx, sd, r, (n, ci, B, T) = _run("full12_mse", "f32")
    assert np.abs(r["out"] - fx["out_train"]).max() < OUT_TOL
    assert abs(r["loss"] - float(fx["loss"])) < 1e-5
    names = plan.param_names(n, ci)
    for i, k in enumerate(names):
        ref = fx["grad_digest"][i]
        got = digest(r["grads"][k])
        if k.endswith(".0.bias") and not k.startswith("out"):
            assert got[0] < 1e-5, (k, got[0])
            continue
        assert abs(got[0] - ref[0]) < 5e-3 * ref[0] + 1e-7, (k, got[0], ref[0])
        # element-level: the fp32 reference itself is only reproducible to a few % of the tensor's
        # rms (f32-vs-f64 restatement differ by up to 0.16 rms through 25 BN backward passes), while
        # every absolute difference stays far inside the 1e-4 north_star budget.
        rms = ref[0] / np.sqrt(r["grads"][k].size)
        err = np.abs(got[2:] - ref[2:]).max()
        assert err < 1e-4 and err < 0.05 * rms + 1e-7, (k, err, rms)
    for i, k in enumerate(plan.buffer_names(n, ci)):
        ref = fx["buf_digest"][i]
        got = digest(sd[k])
        assert abs(got[0] - ref[0]) < 1e-4 * abs(ref[0]) + 1e-6, k


# ---- the travelling port against the imported reference, every run in the build container (VERDICT r4 missing #5): the full-size GPU
# tests and bench.py's cpu_baseline use oracle/torch_port.py, so its equality with /root/reference/model/unet_basic.py + model/loss.py
# is checked here and not only when somebody regenerates the fixtures
REF = "/root/reference"


@pytest.mark.skipif(not __import__("os").path.isdir(REF), reason="the reference tree only exists in the build container")
@pytest.mark.parametrize("n,ci,B,T,loss_kind", [(3, 4, 2, 64, "mse"), (5, 8, 3, 512, "l1"), (4, 6, 2, 256, "smooth_l1"), (12, 24, 2, 4096, "mse")])
def test_torch_port_equals_the_imported_reference(monkeypatch, n, ci, B, T, loss_kind):
    import sys
    import torch
    from oracle import torch_port
    sys.dont_write_bytecode = True
    monkeypatch.syspath_prepend(REF)
    for name in [m for m in sys.modules if m.split(".")[0] == "model"]:
        monkeypatch.delitem(sys.modules, name, raising=False)
    try:
        from model.unet_basic import Model as RefModel          # the reference, unmodified
        from model import loss as ref_loss
        sd_np = plan.golden_state(n, ci, seed=0)
        noisy_np, clean_np = plan.golden_batch(B, T, seed=0)
        noisy, clean = torch.from_numpy(noisy_np), torch.from_numpy(clean_np)
        model = RefModel(n_layers=n, channels_interval=ci)
        model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()})
        crit = {"mse": ref_loss.mse_loss, "l1": ref_loss.l1_loss, "smooth_l1": torch.nn.SmoothL1Loss}[loss_kind]()
        # eval path (enhancement.py:43,65-66)
        model.eval()
        with torch.no_grad():
            ref_eval = model(noisy)
            port_eval = torch_port.forward(torch_port.state_to_torch(sd_np), noisy, n, ci, False)
        assert torch.equal(ref_eval, port_eval)
        # training step (trainer/trainer.py:34-37): output, loss, every gradient, the running statistics
        model.train()
        loss = crit(clean, model(noisy))
        loss.backward()
        tsd = torch_port.state_to_torch(sd_np, requires_grad=True)
        out = torch_port.forward(tsd, noisy, n, ci, True)
        l2 = torch_port.loss_value(loss_kind, clean, out)
        l2.backward()
        assert loss.item() == l2.item()
        for k, p in model.named_parameters():
            assert torch.equal(p.grad, tsd[k].grad), k
        post = model.state_dict()
        for k in plan.buffer_names(n, ci):
            assert torch.equal(post[k], tsd[k]), k
    finally:
        for name in [m for m in sys.modules if m.split(".")[0] == "model"]:
            sys.modules.pop(name, None)
