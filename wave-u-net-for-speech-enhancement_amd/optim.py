"""fused Adam for the hot loop.

Drop-in for the optimiser the reference builds in train.py:31-35
(`torch.optim.Adam(params=model.parameters(), lr=..., betas=(beta1, beta2))`): same constructor arguments the
reference uses, same update rule and operation order as torch's single-tensor path (eps=1e-8, weight_decay=0,
amsgrad=False), same `state_dict()` layout ("step", "exp_avg", "exp_avg_sq" per parameter; param_groups carry every key
torch.optim.Adam's do) so the reference's checkpoints (trainer/base_trainer.py:62-124) load both ways - but one or two HIP
launches over all 102 tensors instead of ~10 multi-tensor ATen kernels.

`grad_scale` (attribute): every gradient is multiplied by it inside the step - the 1/world_size of the data-parallel average
(parallel.GradSync(scale_in_optimizer=True)).
`device_step=True`: the step counter lives on the device (incremented by the step's own kernel, bias corrections computed there
in double), so forward + loss + backward + step can be captured in one hipGraph and replayed (trainer.Trainer(graph=True));
the host-side `state["step"]` is kept in step by `advance_host_step()` after every replay.
"""
import torch

from .engine import bump_versions as _bump_versions, default_engine

_ADAM_DEFAULTS = dict(weight_decay=0, amsgrad=False, maximize=False, foreach=None, capturable=False, differentiable=False,
                      fused=None, decoupled_weight_decay=False)


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, device_step=False):
        if lr < 0.0 or eps < 0.0 or not (0.0 <= betas[0] < 1.0) or not (0.0 <= betas[1] < 1.0):
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, **_ADAM_DEFAULTS))
        self._engine_override = None
        self.grad_scale = 1.0
        self.device_step = device_step
        self._dev = {}            # group index -> (int64 step counter, 2-float hyper buffer) on the parameters' device

    def hyper_signature(self):
        """Everything a captured step() bakes into its kernel arguments (lr, betas, eps per group, grad_scale): a driver that
        replays a graph compares it with the signature at capture time and re-captures when it changed (trainer.Trainer._step)."""
        return tuple((g["lr"], tuple(g["betas"]), g["eps"]) for g in self.param_groups) + (self.grad_scale,)

    def _engine(self):
        return self._engine_override if self._engine_override is not None else default_engine()

    def state_dict(self):
        """torch.optim.Adam's layout, with a "step" tensor of ITS OWN per parameter (inside this class the parameters of a group
        share one counter object, see step(): an optimiser that increments per parameter - torch.optim.Adam after loading this
        checkpoint, trainer/base_trainer.py:74 - must not see them aliased)."""
        sd = super().state_dict()
        sd["state"] = {k: ({**v, "step": v["step"].clone()} if "step" in v else v) for k, v in sd["state"].items()}
        return sd

    def load_state_dict(self, state_dict):
        for g in state_dict["param_groups"]:
            if g.get("weight_decay", 0) != 0 or g.get("amsgrad", False) or g.get("maximize", False):
                raise ValueError("FusedAdam implements the reference's Adam (train.py:31-35: weight_decay=0, amsgrad=False, "
                                 "maximize=False); the checkpoint asks for something else")
        super().load_state_dict(state_dict)
        for g in self.param_groups:
            for k, v in _ADAM_DEFAULTS.items():
                g.setdefault(k, v)
        # "step" as this class keeps it: a float32 scalar on the HOST.  A checkpoint loaded with map_location=<gpu>
        # (trainer.Trainer._resume_checkpoint, the reference's base_trainer.py:70-74) leaves it on the device - step() would then
        # synchronise once per parameter and abort a graph capture - and the torch 1.2 the reference was tested with stored a
        # Python int (README.md:27).
        for st in self.state.values():
            if "step" in st:
                v = st["step"]
                st["step"] = torch.tensor(float(v.item() if torch.is_tensor(v) else v), dtype=torch.float32, device="cpu")
        self._dev = {}            # re-seeded from the loaded host step at the next step()

    def advance_host_step(self, n=1):
        """After replaying a captured graph that contains step(): the device counter advanced, bring state["step"] along."""
        seen = {}
        for group in self.param_groups:
            for p in group["params"]:
                st = self.state.get(p)
                if st:
                    seen[id(st["step"])] = st["step"]           # (the parameters of a group share one counter object, see step())
        for t in seen.values():
            t += n
        # the replay rewrote the weights on the device; no Python version counter moved (engine.py: the eval weight-pack cache)
        self._engine().note_weights_changed()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            ps, gs, ms, vs, steps, skipped = [], [], [], [], [], []
            state = self.state
            for p in group["params"]:
                g = p.grad
                if g is None:
                    skipped.append(p)
                    continue
                st = state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)      # host scalar, like torch's default
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                ps.append(p); gs.append(g if g.is_contiguous() else g.contiguous())
                ms.append(st["exp_avg"]); vs.append(st["exp_avg_sq"]); steps.append(st["step"])
            if not ps:
                continue
            # The host-side step counters.  torch.optim.Adam's layout wants one "step" tensor per parameter; 102 scalar `+= 1`
            # (or one _foreach_add_ over 102 CPU scalars) cost the host 0.2 - 0.3 ms per step, so parameters whose counts agree
            # share ONE tensor object (state_dict() and checkpoints look the same: the value per key).
            s0 = steps[0]
            for p in skipped:               # a parameter without a gradient does not step (torch.optim.Adam): it leaves the shared counter
                st = state.get(p)
                if st and st.get("step") is s0:
                    st["step"] = s0.clone()
            if all(t is s0 for t in steps):
                s0 += 1
                launches = [(int(s0.item()), ps, gs, ms, vs)]
            else:
                # counts that differ (a parameter that sat out some steps, a checkpoint just loaded): one launch per distinct count -
                # a launch has one bias correction - and one shared counter per such sub-group from here on
                by = {}
                for k, t in enumerate(steps):
                    by.setdefault(float(t.item()), []).append(k)
                if self.device_step and len(by) != 1:
                    # (checked BEFORE any counter is rewritten: a caller that catches this finds host counters, device counter and
                    # moments as they were)
                    raise RuntimeError("FusedAdam(device_step=True): the parameters of one group must share the step count "
                                       "(the group has ONE counter on the device)")
                launches = []
                for v, idx in by.items():
                    sv = torch.tensor(v + 1.0, dtype=torch.float32)
                    for k in idx:
                        state[ps[k]]["step"] = sv
                    launches.append((int(v) + 1, [ps[k] for k in idx], [gs[k] for k in idx], [ms[k] for k in idx], [vs[k] for k in idx]))
            b1, b2 = group["betas"]
            step_dev = hyper_dev = None
            if self.device_step:
                if len(launches) != 1:
                    raise RuntimeError("FusedAdam(device_step=True): the parameters of one group must share the step count "
                                       "(the group has ONE counter on the device)")
                step = launches[0][0]
                if gi not in self._dev:              # (outside any capture: the first eager step creates and seeds it)
                    self._dev[gi] = (torch.full((), step - 1, dtype=torch.int64, device=ps[0].device),
                                     torch.zeros(2, dtype=torch.float32, device=ps[0].device))
                step_dev, hyper_dev = self._dev[gi]
            for step, lp, lg, lm, lv in launches:
                self._engine().adam_step(lp, lg, lm, lv, group["lr"], b1, b2, group["eps"], step, self.grad_scale, step_dev, hyper_dev)
                # the kernel wrote p, exp_avg and exp_avg_sq through raw pointers: tell autograd's version counters, as every in-place
                # torch op would (a graph that saved a parameter and is back-propagated after this step must fail loudly, and anything
                # keyed on `_version` must see the change)
                _bump_versions(lp + lm + lv)
        return loss
