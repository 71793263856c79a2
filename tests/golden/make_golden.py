"""Generate the golden fixtures in this directory from the REAL reference implementation.

Run in the build container only (needs /root/reference; the GPU box does not have it):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

It imports /root/reference/model/unet_basic.py:Model and model/loss.py unmodified, loads the
deterministic parameters of oracle/plan.py:golden_state into it, runs forward / loss / backward
on CPU in fp32 (torch 2.10.0, CUDA_VISIBLE_DEVICES=-1 path of the reference, README.md:68-70)
and stores the results.  Small cases store everything; the full 12-level / 16384-sample case
stores the output, the loss and per-tensor gradient digests (norm, sum, first/last 8 values)
because the 40 MB of weights/gradients are regenerated from the seed instead of committed.
It also asserts that oracle/torch_port.py (the travelling functional restatement) reproduces
the imported reference, which is what lets GPU-box tests use the port at full size.
"""
import os
import sys

sys.dont_write_bytecode = True
os.environ.setdefault("CUDA_VISIBLE_DEVICES", "-1")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

import numpy as np
import torch

from model.unet_basic import Model as RefModel          # noqa: E402  (the reference)
from model import loss as ref_loss                      # noqa: E402
from oracle import plan, torch_port                     # noqa: E402
from make_digest import digest                          # noqa: E402

CASES = [
    # name, n_layers, ci, B, T, loss, full_dump
    ("tiny_mse", 3, 4, 2, 64, "mse", True),
    ("small_l1", 5, 8, 3, 512, "l1", True),
    ("small_smoothl1", 4, 6, 2, 256, "smooth_l1", True),
    ("full12_mse", 12, 24, 2, 16384, "mse", False),
]


def run_case(name, n, ci, B, T, loss_kind, full):
    sd_np = plan.golden_state(n, ci, seed=0)
    noisy_np, clean_np = plan.golden_batch(B, T, seed=0)
    noisy, clean = torch.from_numpy(noisy_np), torch.from_numpy(clean_np)

    model = RefModel(n_layers=n, channels_interval=ci)
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()})
    if loss_kind == "mse":
        crit = ref_loss.mse_loss()
    elif loss_kind == "l1":
        crit = ref_loss.l1_loss()
    else:
        crit = torch.nn.SmoothL1Loss()       # not in model/loss.py, same factory convention

    # eval-mode forward with the initial running statistics (enhancement.py:43,66)
    model.eval()
    with torch.no_grad():
        out_eval = model(noisy).numpy().copy()

    # training step forward/backward (trainer/trainer.py:34-37)
    model.train()
    enhanced = model(noisy)
    loss = crit(clean, enhanced)
    loss.backward()

    # the travelling restatement must reproduce the reference
    tsd = torch_port.state_to_torch(sd_np, requires_grad=True)
    o2 = torch_port.forward(tsd, noisy, n, ci, True)
    l2 = torch_port.loss_value(loss_kind, clean, o2)
    l2.backward()
    port_out_diff = float((o2 - enhanced).abs().max())
    port_grad_diff = max(float((tsd[k].grad - p.grad).abs().max()) for k, p in model.named_parameters())
    assert port_out_diff <= 1e-6 and port_grad_diff <= 1e-6, (port_out_diff, port_grad_diff)

    fix = {
        "meta": np.array([n, ci, B, T], np.int64),
        "loss_kind": np.array(loss_kind),
        "torch_version": np.array(torch.__version__),
        "out_eval": out_eval,
        "out_train": enhanced.detach().numpy(),
        "loss": np.array(loss.item(), np.float64),
        "port_vs_ref": np.array([port_out_diff, port_grad_diff]),
    }
    names = plan.param_names(n, ci)
    assert names == [k for k, _ in model.named_parameters()]
    post = model.state_dict()
    if full:
        for k, p in model.named_parameters():
            fix["grad/" + k] = p.grad.numpy()
        for k in plan.buffer_names(n, ci):
            fix["buf/" + k] = post[k].numpy()
    else:
        fix["grad_digest"] = np.stack([digest(p.grad.numpy()) for _, p in model.named_parameters()])
        fix["buf_digest"] = np.stack([digest(post[k].numpy()) for k in plan.buffer_names(n, ci)])
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **fix)
    print(f"{name}: loss={loss.item():.8f} port-vs-ref out {port_out_diff:.1e} grad {port_grad_diff:.1e} "
          f"-> {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    torch.manual_seed(0)
    for case in CASES:
        run_case(*case)
