"""Random-shape check of the forced fp16-split path in the CPU emulator against the oracle (not part of the test suite):
   python tools/fuzz_emu.py [seed] [trials]      # from the repo root; prints one line per configuration, "bad: 0" at the end
Bars: output 5e-5 absolute; worst gradient 5e-2 of that tensor's maximum (small nets sit on LeakyReLU kinks, HISTORY.md section 7)."""
import sys, importlib, os, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch
import emu_lib
from conftest import PKG_NAME
from oracle import c_oracle, plan
eng_mod = importlib.import_module(PKG_NAME + ".engine"); lib_mod = importlib.import_module(PKG_NAME + "._lib")
model_mod = importlib.import_module(PKG_NAME + ".model"); loss_mod = importlib.import_module(PKG_NAME + ".loss")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
eng = eng_mod.Engine(lib=lib_mod.declare(emu_lib.lib()), host_memory=True, h3=2)
bad = 0
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
    n = int(rng.integers(1, 6)); ci = int(rng.choice([8, 12, 16, 20, 24, 40])); B = int(rng.integers(1, 7))
    T = int(2 ** rng.integers(max(n + 2, 6), 12)); loss = str(rng.choice(["mse", "l1", "smooth_l1"]))
    sd = plan.golden_state(n, ci, trial); noisy, clean = plan.golden_batch(B, T, trial)
    ref = c_oracle.step({k: v.copy() for k, v in sd.items()}, noisy, clean, n, ci, True, loss=loss, want_grads=True)
    m = model_mod.Model(n_layers=n, channels_interval=ci)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}); m._engine_override = eng; m.train()
    crit = {"mse": loss_mod.mse_loss, "l1": loss_mod.l1_loss, "smooth_l1": loss_mod.smooth_l1_loss}[loss](); crit._engine_override = eng
    t = time.time()
    out = m(torch.from_numpy(noisy)); crit(torch.from_numpy(clean), out).backward()
    eo = np.abs(out.detach().numpy() - ref["out"]).max(); worst = 0; wk = ""
    for k, p in m.named_parameters():
        if k.endswith(".0.bias") and not k.startswith("out"): continue
        g = ref["grads"][k]; e = np.abs(p.grad.numpy() - g).max() / max(np.abs(g).max(), 1e-12)
        if e > worst: worst, wk = e, k
    flag = "" if (eo < 5e-5 and worst < 5e-2) else "   <<<<<< BAD"
    bad += bool(flag)
    print("n=%d ci=%2d B=%d T=%4d %-9s out %.1e worst rel grad %.1e (%s) %.1fs%s" % (n, ci, B, T, loss, eo, worst, wk, time.time() - t, flag), flush=True)
print("bad:", bad)
