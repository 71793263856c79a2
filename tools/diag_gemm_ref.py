#!/usr/bin/env python
"""usage (GPU box): python tools/diag_gemm_ref.py   - 12 levels, batch 16: per-tensor gradient errors of the HIP path (exact-fp32
and split GEMMs) and of the reference's own fp32 CPU arithmetic against a float64 run of the same network (ATen's fp32
upsample coordinates kept): whose noise is the 1e-2 seen on encoder.10's weight gradient?"""
import importlib
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import plan, torch_port  # noqa: E402

PKG = "wave-u-net-for-speech-enhancement_amd"


def up_fp32_coords(h):
    lin = h.shape[-1]
    lout = 2 * lin
    scale = np.float32(lin - 1) / np.float32(lout - 1) if lout > 1 else np.float32(0)
    j = np.arange(lout, dtype=np.float32)
    src = (scale * j).astype(np.float32)
    i0 = np.minimum(np.floor(src).astype(np.int64), lin - 1)
    lam = np.clip((src - i0.astype(np.float32)).astype(np.float32), 0, 1)
    i1 = i0 + (i0 < lin - 1)
    l1 = torch.from_numpy(lam.astype(np.float64))
    l0 = torch.from_numpy((np.float32(1) - lam).astype(np.float64))
    return h[..., torch.from_numpy(i0)] * l0 + h[..., torch.from_numpy(i1)] * l1


def main():
    n, ci, B, T = 12, 24, 16, 16384
    noisy, clean = plan.golden_batch(B, T, 0)
    res = {}
    for name, dt in (("ref32", torch.float32), ("ref64", torch.float64)):
        tsd = torch_port.state_to_torch(plan.golden_state(n, ci, 0), dtype=dt, requires_grad=True)
        if dt == torch.float64:
            orig = F.interpolate
            F.interpolate = lambda h, **kw: up_fp32_coords(h)
        o = torch_port.forward(tsd, torch.from_numpy(noisy).to(dt), n, ci, True)
        l = torch_port.loss_value("smooth_l1", torch.from_numpy(clean).to(dt), o)
        l.backward()
        if dt == torch.float64:
            F.interpolate = orig
        res[name] = {k: v.grad.double() for k, v in tsd.items() if v.requires_grad}
        res[name]["__out"] = o.detach().double()
    pkg = importlib.import_module(PKG)
    eng_mod = importlib.import_module(PKG + ".engine")
    dev = torch.device("cuda:0")
    for mode in (0, 1):
        eng = eng_mod.Engine(h3=mode)
        m = pkg.Model(n_layers=n, channels_interval=ci)
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in plan.golden_state(n, ci, 0).items()})
        m.to(dev).train()
        m._engine_override = eng
        crit = pkg.smooth_l1_loss()
        crit._engine_override = eng
        out = m(torch.from_numpy(noisy).to(dev))
        crit(torch.from_numpy(clean).to(dev), out).backward()
        torch.cuda.synchronize()
        res[f"hip{mode}"] = {k: p.grad.cpu().double() for k, p in m.named_parameters()}
        res[f"hip{mode}"]["__out"] = out.detach().cpu().double()
    r64 = res["ref64"]
    rel = lambda a, b: ((a - b).norm() / (b.norm() + 1e-30)).item()
    print("%-28s %10s %10s %10s %10s" % ("tensor", "ref32/64", "hip0/64", "hip1/64", "hip1/ref32"))
    worst = {"ref32": 0, "hip0": 0, "hip1": 0}
    for k in r64:
        if k.endswith(".0.bias") and not k.startswith("out"):
            continue
        a, b, c, d = rel(res["ref32"][k], r64[k]), rel(res["hip0"][k], r64[k]), rel(res["hip1"][k], r64[k]), rel(res["hip1"][k], res["ref32"][k])
        worst["ref32"] = max(worst["ref32"], a); worst["hip0"] = max(worst["hip0"], b); worst["hip1"] = max(worst["hip1"], c)
        if k.endswith(".0.weight") or k == "__out":
            print("%-28s %10.2e %10.2e %10.2e %10.2e" % (k, a, b, c, d))
    print("worst over all tensors:", worst)


if __name__ == "__main__":
    main()
