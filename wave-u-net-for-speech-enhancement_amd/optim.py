"""fused Adam for the hot loop.

Drop-in for the optimiser the reference builds in train.py:31-35
(`torch.optim.Adam(params=model.parameters(), lr=..., betas=(beta1, beta2))`): same constructor arguments the
reference uses, same update rule and operation order as torch's single-tensor path (eps=1e-8, weight_decay=0,
amsgrad=False), same `state_dict()` layout ("step", "exp_avg", "exp_avg_sq" per parameter) so the reference's
checkpoints (trainer/base_trainer.py:62-124) load both ways - but one or two HIP launches over all 102 tensors
instead of ~10 multi-tensor ATen kernels.
"""
import torch

from .engine import default_engine


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        if lr < 0.0 or eps < 0.0 or not (0.0 <= betas[0] < 1.0) or not (0.0 <= betas[1] < 1.0):
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._engine_override = None

    def _engine(self):
        return self._engine_override if self._engine_override is not None else default_engine()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            ps, gs, ms, vs = [], [], [], []
            step = None
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)      # host scalar, like torch's default
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                s = int(st["step"].item())
                if step is None:
                    step = s
                elif s != step:
                    raise RuntimeError("FusedAdam: parameters of one group must share the step count")
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                ps.append(p); gs.append(g); ms.append(st["exp_avg"]); vs.append(st["exp_avg_sq"])
            if ps:
                b1, b2 = group["betas"]
                self._engine().adam_step(ps, gs, ms, vs, group["lr"], b1, b2, group["eps"], step)
        return loss
