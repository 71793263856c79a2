"""-m gpu: what round 4 left unproven.

* DESIGN.md section 3 says "the step is bit-reproducible": checked here at BASELINE.json's full size - 50 backward passes over ONE
  batch-64 x 16384 training forward give 102 bitwise-equal gradient tensors, and 50 replays of the captured step graph from one
  saved state give bitwise-equal parameters and moments (reference: trainer/trainer.py:34-38 - the loop these replace).
* The eval-mode weight-pack cache (engine.Engine.forward) against the two ways it could serve stale packs: a captured step graph
  replayed between two eval forwards (the reference's validate-every-N-epochs loop, trainer/base_trainer.py:199-203), and a second
  model whose freshly allocated weights land on the freed addresses of the first.
"""
import importlib

import numpy as np
import pytest
import torch

from conftest import PKG_NAME
from oracle import plan

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need an MI355X"
    return torch.device("cuda:0")


def _model(pkg, n, ci, dev, seed=0, engine=None):
    m = pkg.Model(n_layers=n, channels_interval=ci)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in plan.golden_state(n, ci, seed).items()})
    if engine is not None:
        m._engine_override = engine
    return m.to(dev)


@pytest.mark.timeout(900)
def test_fifty_backwards_over_one_batch64_forward_are_bitwise_equal(pkg, dev):
    """Every gradient of the 12-level net at batch 64 x 16384, 50 times over the same saved activations: no tensor may differ in a
    single bit (split-K partials, statistics rows, BatchNorm-backward sums and the weight-gradient reduction all run in a fixed order;
    the only atomics are maxima of bit patterns)."""
    eng = importlib.import_module(PKG_NAME + ".engine").default_engine()
    n, ci, B, T = 12, 24, 64, 16384
    m = _model(pkg, n, ci, dev).train()
    g = torch.Generator().manual_seed(7)
    clean = torch.rand(B, 1, T, generator=g) * 2 - 1
    noisy = (clean + 0.1 * torch.randn(B, 1, T, generator=g)).to(dev)
    clean = clean.to(dev)
    params = m._wunet_params()
    running, nbt = m._wunet_buffers()
    with torch.no_grad():
        out, ws = eng.forward(n, ci, noisy, params, running, nbt, True, True)
    gout = ((out - clean) * (2.0 / out.numel())).contiguous()
    sizes = [p.numel() for p in params]
    offsets = np.concatenate([[0], np.cumsum(sizes)[:-1]]).tolist()
    FlatGrads = importlib.import_module(PKG_NAME + ".engine").FlatGrads
    ref = None
    for it in range(50):
        flat = torch.full((sum(sizes),), float("nan"), device=dev)
        eng.backward(n, ci, noisy, params, out, gout, ws, FlatGrads(flat, offsets))
        torch.cuda.synchronize()
        if ref is None:
            ref = flat
            assert torch.isfinite(ref).all()
            continue
        if not torch.equal(flat, ref):
            bad = [(k, int((flat[o:o + s] != ref[o:o + s]).sum())) for k, (o, s) in enumerate(zip(offsets, sizes))
                   if not torch.equal(flat[o:o + s], ref[o:o + s])]
            pytest.fail(f"backward #{it} differs from #0 in parameter tensors (index, elements): {bad[:10]}")


@pytest.mark.timeout(900)
def test_fifty_graph_replays_from_one_state_are_bitwise_equal(pkg, dev):
    """The whole captured step (forward, smooth-L1, backward on both streams, fused Adam) replayed 50 times, each time from the same
    restored parameters / moments / running statistics / step counter: the state after the replay is the same 50 times, bit for bit,
    and so is the loss."""
    optim_mod = importlib.import_module(PKG_NAME + ".optim")
    n, ci, B, T = 12, 24, 64, 16384
    m = _model(pkg, n, ci, dev).train()
    crit = pkg.smooth_l1_loss()
    opt = optim_mod.FusedAdam(m.parameters(), lr=1e-3, betas=(0.9, 0.999), device_step=True)
    g = torch.Generator().manual_seed(11)
    clean = torch.rand(B, 1, T, generator=g) * 2 - 1
    noisy = (clean + 0.1 * torch.randn(B, 1, T, generator=g)).to(dev)
    clean = clean.to(dev)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = crit(clean, m(noisy))
        loss.backward()
        opt.step()
        return loss

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    opt.zero_grad(set_to_none=True)
    with torch.cuda.graph(graph):
        loss_g = step()
    opt.advance_host_step(-1)
    torch.cuda.synchronize()
    state = [t for t in m.state_dict().values()]
    for st in opt.state.values():
        state += [st["exp_avg"], st["exp_avg_sq"]]
    state += [t for pair in opt._dev.values() for t in pair]
    saved = [t.clone() for t in state]
    ref = None
    for it in range(50):
        with torch.no_grad():
            for t, s in zip(state, saved):
                t.copy_(s)
        graph.replay()
        torch.cuda.synchronize()
        now = [t.clone() for t in state] + [loss_g.clone()]
        if ref is None:
            ref = now
            assert all(torch.isfinite(t.float()).all() for t in ref)
            continue
        bad = [k for k, (a, b) in enumerate(zip(now, ref)) if not torch.equal(a, b)]
        assert not bad, f"replay #{it} differs from #0 in state tensors {bad[:10]}"


def test_eval_after_graph_replays_sees_the_new_weights(pkg, dev):
    """eval forward, N replays of the captured training step (they rewrite the weights on the device; no Python
    version counter moves), eval forward again on the same shape and stream - the second one must use the NEW weights' packs: it
    equals a fresh engine's forward of the same parameters bit for bit and differs from the first."""
    eng_mod = importlib.import_module(PKG_NAME + ".engine")
    optim_mod = importlib.import_module(PKG_NAME + ".optim")
    n, ci, B, T = 5, 16, 4, 4096
    eng = eng_mod.Engine()
    m = _model(pkg, n, ci, dev, engine=eng)
    crit = pkg.smooth_l1_loss()
    crit._engine_override = eng
    opt = optim_mod.FusedAdam(m.parameters(), lr=1e-2, betas=(0.9, 0.999), device_step=True)
    opt._engine_override = eng
    noisy, clean = plan.golden_batch(B, T, 3)
    x, y = torch.from_numpy(noisy).to(dev), torch.from_numpy(clean).to(dev)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = crit(y, m(x))
        loss.backward()
        opt.step()

    m.train()
    for _ in range(3):
        step()
    m.eval()
    with torch.no_grad():
        e1 = m(x).clone()
        assert torch.equal(m(x), e1)                           # (cached packs, same weights: same bits)
    assert len(eng._eval_ws) == 1
    m.train()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    opt.zero_grad(set_to_none=True)
    with torch.cuda.graph(graph):
        step()
    opt.advance_host_step(-1)
    m.eval()
    with torch.no_grad():
        e1b = m(x).clone()                                     # capture executes nothing: still the old weights (and a cache entry again)
    assert torch.equal(e1b, e1)
    for _ in range(4):
        graph.replay()
        opt.advance_host_step(1)
    with torch.no_grad():
        e2 = m(x).clone()
    torch.cuda.synchronize()
    m2 = pkg.Model(n_layers=n, channels_interval=ci)
    m2.load_state_dict({k: v.detach().cpu().clone() for k, v in m.state_dict().items()})
    m2._engine_override = eng_mod.Engine()
    m2.to(dev).eval()
    with torch.no_grad():
        ref = m2(x)
    assert torch.equal(e2, ref)
    assert not torch.equal(e2, e1)


def test_eval_cache_is_not_shared_by_a_second_model_on_recycled_addresses(pkg, dev):
    """checkpoints evaluated one after the other with freshly built models - the caching allocator hands the freed
    weights' blocks to the next model, version counters start equal: (address, version) alone would match the first model's packs."""
    eng_mod = importlib.import_module(PKG_NAME + ".engine")
    n, ci, B, T = 5, 16, 2, 4096
    eng = eng_mod.Engine()
    noisy, _ = plan.golden_batch(B, T, 4)
    x = torch.from_numpy(noisy).to(dev)
    outs, ptrs = [], []
    for seed in (0, 1):
        m = _model(pkg, n, ci, dev, seed=seed, engine=eng).eval()
        ptrs.append([p.data_ptr() for p in m.parameters()])
        with torch.no_grad():
            outs.append(m(x).clone())
            assert torch.equal(m(x), outs[-1])
        ref = _model(pkg, n, ci, dev, seed=seed, engine=eng_mod.Engine()).eval()
        with torch.no_grad():
            assert torch.equal(ref(x), outs[-1]), f"model {seed}: served with another model's weight packs"
        del m, ref
        torch.cuda.synchronize()
    assert not torch.equal(outs[0], outs[1])
