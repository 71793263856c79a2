"""TEST INFRASTRUCTURE ONLY (oracle/). Never imported by the product package.

Functional restatement of the reference Wave-U-Net on top of the same ATen ops the
reference dispatches to (conv1d / batch_norm / leaky_relu / interpolate / cat / tanh).
It exists because /root/reference cannot travel to the GPU box: this is what the
`cpu_baseline` leg of bench.py times (it executes exactly the ATen CPU kernels the
reference's CUDA_VISIBLE_DEVICES=-1 path executes) and what the full-size parity tests
compare against.  Pinned against the imported reference by tests/golden/make_golden.py
(max |diff| == 0 on CPU: same ops, same order).

Follows /root/reference/model/unet_basic.py:77-100 (forward), :6-30 (layer bodies),
/root/reference/model/loss.py:3-7 (losses).
"""
import torch
import torch.nn.functional as F

from .plan import conv_layers

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
SLOPE = 0.1


def _conv_bn_act(x, sd, prefix, taps, training):
    # Conv1d(k, stride 1, pad k//2) -> BatchNorm1d -> LeakyReLU(0.1)   (unet_basic.py:9-14, :22-27, :52-57)
    z = F.conv1d(x, sd[f"{prefix}.0.weight"], sd[f"{prefix}.0.bias"], padding=taps // 2)
    y = F.batch_norm(z, sd[f"{prefix}.1.running_mean"], sd[f"{prefix}.1.running_var"],
                     sd[f"{prefix}.1.weight"], sd[f"{prefix}.1.bias"],
                     training, BN_MOMENTUM, BN_EPS)
    if training:
        sd[f"{prefix}.1.num_batches_tracked"] += 1
    return F.leaky_relu(y, SLOPE)


def upsample_fp32_coords(h):
    """x2 linear upsample, align_corners=True, of a float64 tensor with the source indices and weights ATen computes in FLOAT32
    (UpSample.h area_pixel_compute_scale / _source_index, guard_index_and_lambda; model/unet_basic.py:93): the float64 arbiter of
    the parity tests must interpolate at the reference's coordinates, otherwise it is itself 3.8e-4 away from the reference
    (SURVEY.md section 7)."""
    import numpy as np
    lin = h.shape[-1]
    lout = 2 * lin
    scale = np.float32(lin - 1) / np.float32(lout - 1) if lout > 1 else np.float32(0)
    src = (scale * np.arange(lout, dtype=np.float32)).astype(np.float32)
    i0 = np.minimum(np.floor(src).astype(np.int64), lin - 1)
    lam = np.clip((src - i0.astype(np.float32)).astype(np.float32), np.float32(0), np.float32(1))
    i1 = i0 + (i0 < lin - 1)
    l1 = torch.from_numpy(lam.astype(np.float64))
    l0 = torch.from_numpy((np.float32(1) - lam).astype(np.float64))
    return h[..., torch.from_numpy(i0)] * l0 + h[..., torch.from_numpy(i1)] * l1


def forward(sd, noisy, n_layers=12, ci=24, training=True):
    """sd: dict name -> torch tensor (running stats are updated in place when training).  float64 tensors: the arbiter run
    (same ops in double precision, the upsample at ATen's float32 coordinates)."""
    layers = conv_layers(n_layers, ci)
    enc, mid, dec = layers[:n_layers], layers[n_layers], layers[n_layers + 1:]
    skips = []
    h = noisy
    for prefix, _, _, k in enc:                         # unet_basic.py:82-86
        h = _conv_bn_act(h, sd, prefix, k, training)
        skips.append(h)
        h = h[:, :, ::2]
    h = _conv_bn_act(h, sd, mid[0], mid[3], training)   # :88
    for j, (prefix, _, _, k) in enumerate(dec):         # :91-96
        h = upsample_fp32_coords(h) if h.dtype == torch.float64 else F.interpolate(h, scale_factor=2, mode="linear", align_corners=True)
        h = torch.cat([h, skips[n_layers - 1 - j]], dim=1)
        h = _conv_bn_act(h, sd, prefix, k, training)
    h = torch.cat([h, noisy], dim=1)                    # :98
    return torch.tanh(F.conv1d(h, sd["out.0.weight"], sd["out.0.bias"]))   # :99


def loss_value(kind, clean, enhanced):
    """loss(clean, enhanced) as called at trainer/trainer.py:36 (prediction is the 2nd argument)."""
    if kind == "mse":
        return F.mse_loss(clean, enhanced)
    if kind == "l1":
        return F.l1_loss(clean, enhanced)
    if kind == "smooth_l1":
        return F.smooth_l1_loss(clean, enhanced)
    raise ValueError(kind)


def state_to_torch(sd_np, device="cpu", dtype=torch.float32, requires_grad=False):
    out = {}
    for k, v in sd_np.items():
        t = torch.from_numpy(v.copy()).to(device)
        if t.is_floating_point():
            t = t.to(dtype)
            if requires_grad and ("running" not in k):
                t.requires_grad_(True)
        out[k] = t
    return out
