"""Where does a training step that is synchronised every step (the reference's `loss.item()` per step, trainer/trainer.py:40) lose time against the
back-to-back step?  Host time stamps and device events at the phase boundaries of 30 steps, both ways (GPU box)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
pkg = importlib.import_module("wave-u-net-for-speech-enhancement_amd")
optim = importlib.import_module("wave-u-net-for-speech-enhancement_amd.optim")
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = pkg.Model().to(dev).train()
crit = pkg.smooth_l1_loss()
opt = optim.FusedAdam(m.parameters(), lr=1e-3)
x = torch.randn(64, 1, 16384, device=dev); y = torch.randn(64, 1, 16384, device=dev)
names = ["forward", "loss", "backward", "adam"]


def step(sync, rec):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
    h = [time.perf_counter()]
    ev[0].record()
    opt.zero_grad(set_to_none=True)
    out = m(x); h.append(time.perf_counter()); ev[1].record()
    loss = crit(y, out); h.append(time.perf_counter()); ev[2].record()
    loss.backward(); h.append(time.perf_counter()); ev[3].record()
    opt.step(); h.append(time.perf_counter()); ev[4].record()
    if sync:
        ev[4].synchronize()
    h.append(time.perf_counter())
    rec.append((h, ev))


for sync in (False, True):
    for _ in range(5):
        step(sync, [])
    torch.cuda.synchronize()
    rec = []
    t0 = time.perf_counter()
    for _ in range(30):
        step(sync, rec)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 30 * 1e3
    host = np.median(np.array([[(h[i + 1] - h[i]) * 1e3 for i in range(5)] for h, _ in rec]), axis=0)
    gpu = np.median(np.array([[ev[i].elapsed_time(ev[i + 1]) for i in range(4)] for _, ev in rec]), axis=0)
    print(("synchronised every step" if sync else "back to back") + ": %.3f ms per step" % wall)
    print("   host ms  " + "  ".join("%s %.3f" % (n, v) for n, v in zip(names + ["wait"], host)) + "   (sum of the four calls %.3f)" % host[:4].sum())
    print("   GPU ms between the events after  " + "  ".join("%s %.3f" % (n, v) for n, v in zip(names, gpu)) + "   (sum %.3f)" % gpu.sum())
