"""Loss plugins, same factory convention as the reference (/root/reference/model/loss.py:3-7):
zero-argument factories returning a module called as `loss(clean, enhanced)`
(trainer/trainer.py:36 - the prediction is the second argument).

    "loss_function": {"module": "wave-u-net-for-speech-enhancement_amd.loss", "main": "mse_loss", "args": {}}

`smooth_l1_loss` (torch.nn.SmoothL1Loss semantics, beta = 1) is the loss BASELINE.json's config 3
names; the reference only mentions it in its README.
"""
import torch
import torch.nn as nn

from .engine import default_engine


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, kind, engine, clean, enhanced):
        clean_c, enh_c = clean.detach().contiguous(), enhanced.detach().contiguous()
        ctx.kind, ctx.engine = kind, engine
        ctx.save_for_backward(clean_c, enh_c)
        return engine.loss_forward(kind, clean_c, enh_c)

    @staticmethod
    def backward(ctx, grad_loss):
        clean, enh = ctx.saved_tensors
        g = ctx.engine.loss_backward(ctx.kind, clean, enh, grad_loss)
        g_clean = -g if ctx.needs_input_grad[2] else None
        return None, None, g_clean, (g if ctx.needs_input_grad[3] else None)


class HipLoss(nn.Module):
    def __init__(self, kind):
        super().__init__()
        self.kind = kind
        self._engine_override = None

    def forward(self, clean, enhanced):
        engine = self._engine_override if self._engine_override is not None else default_engine()
        return _LossFn.apply(self.kind, engine, clean, enhanced)


def mse_loss():
    return HipLoss("mse")


def l1_loss():
    return HipLoss("l1")


def smooth_l1_loss():
    return HipLoss("smooth_l1")
