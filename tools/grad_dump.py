#!/usr/bin/env python
"""One training-mode forward + backward at the BASELINE geometry (seeded weights and batch), every gradient written to an .npz - two builds of the
library (WUNET_LIB_PATH) are compared bit for bit, tensor by tensor, in backward order:
    WUNET_LIB_PATH=$PWD/tools/_lib_base.so python tools/grad_dump.py /tmp/a.npz;  python tools/grad_dump.py /tmp/b.npz;  python tools/grad_dump.py --cmp /tmp/a.npz /tmp/b.npz
(measurement tool; the parity tests are tests/test_gpu_parity.py)"""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "wave-u-net-for-speech-enhancement_amd"


def dump(path, batch=64, frame=16384):
    pkg = importlib.import_module(PKG)
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = pkg.Model(n_layers=12, channels_interval=24).to(dev).train()
    crit = pkg.smooth_l1_loss()
    g = torch.Generator().manual_seed(1)
    clean = (torch.rand(batch, 1, frame, generator=g) * 2 - 1).to(dev)
    noisy = clean + 0.1 * torch.randn(batch, 1, frame, generator=g).to(dev)
    out = model(noisy)
    loss = crit(clean, out)
    loss.backward()
    torch.cuda.synchronize()
    d = {"__out__": out.detach().cpu().numpy(), "__loss__": loss.detach().cpu().numpy()}
    for n, p in model.named_parameters():
        d[n] = p.grad.detach().cpu().numpy()
    np.savez(path, **d)
    print("wrote", path, "loss %.9g" % float(loss))


def cmp(a, b):
    A, B = np.load(a), np.load(b)
    bad = 0
    for k in A.files:
        x, y = A[k], B[k]
        same = x.tobytes() == y.tobytes()
        if not same:
            bad += 1
            den = np.abs(x).max() + 1e-30
            print("%-44s differs: max |d| %.3e  (%.3e of max |x|), %d of %d values" % (k, np.abs(x - y).max(), np.abs(x - y).max() / den, int((x != y).sum()), x.size))
    print("%d of %d tensors differ" % (bad, len(A.files)))


if __name__ == "__main__":
    if sys.argv[1] == "--cmp":
        cmp(sys.argv[2], sys.argv[3])
    else:
        dump(sys.argv[1])
