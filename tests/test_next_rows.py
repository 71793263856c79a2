"""CPU tests (emulator engine) of the SURVEY.md §8(f) rows built so far: f2 train-step driver, f3 chunked inference,
and the synthetic dataset plugin."""
import copy
import importlib

import numpy as np
import pytest
import torch

import emu_lib
from conftest import PKG_NAME
from oracle import c_oracle, plan


@pytest.fixture(scope="module")
def emu_engine():
    eng_mod = importlib.import_module(PKG_NAME + ".engine")
    lib_mod = importlib.import_module(PKG_NAME + "._lib")
    return eng_mod.Engine(lib=lib_mod.declare(emu_lib.lib()), host_memory=True)


def _model(n, ci, eng):
    m = importlib.import_module(PKG_NAME + ".model").Model(n_layers=n, channels_interval=ci)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in plan.golden_state(n, ci, 0).items()})
    m._engine_override = eng
    return m


def test_chunked_inference_equals_reference_loop(emu_engine):
    """f3: one batched forward over the chunks == the reference's sequential batch-1 loop (enhancement.py:57-69),
    here run through the oracle chunk by chunk."""
    inference = importlib.import_module(PKG_NAME + ".inference")
    n, ci, sl = 2, 4, 64
    m = _model(n, ci, emu_engine).eval()
    T = 3 * sl + 17                                          # ragged: needs 47 samples of zero padding
    rng = np.random.default_rng(5)
    mix = rng.standard_normal((1, 1, T)).astype(np.float32)
    out = inference.enhance(m, torch.from_numpy(mix), sample_length=sl, max_batch=3).numpy()
    assert torch.equal(inference.enhance(m, torch.from_numpy(mix), sample_length=sl, max_batch=3, to_host=True), torch.from_numpy(out))
    assert out.shape == (1, 1, T)
    padded = np.concatenate([mix, np.zeros((1, 1, (-T) % sl), np.float32)], axis=-1)
    ref = []
    for k in range(padded.shape[-1] // sl):
        sd = plan.golden_state(n, ci, 0)
        ref.append(c_oracle.step(sd, padded[:, :, k * sl:(k + 1) * sl], None, n, ci, False, want_grads=False)["out"])
    ref = np.concatenate(ref, axis=-1)[:, :, :T]
    assert np.abs(out - ref).max() < 2e-5
    with pytest.raises(ValueError):
        inference.enhance(m, torch.zeros(2, 1, sl))         # batch 1 only, like the reference
    with pytest.raises(RuntimeError):
        inference.enhance(m.train(), torch.zeros(1, 1, sl))


def test_trainer_plugin_runs_the_hot_loop(emu_engine, tmp_path):
    """f2: Trainer(config, resume, model, loss, optimizer, loaders).train() == the hand-written reference loop."""
    trainer_mod = importlib.import_module(PKG_NAME + ".trainer")
    dataset_mod = importlib.import_module(PKG_NAME + ".dataset")
    loss_mod = importlib.import_module(PKG_NAME + ".loss")
    optim_mod = importlib.import_module(PKG_NAME + ".optim")
    n, ci, sl = 2, 4, 64
    ds = dataset_mod.Dataset(n_items=6, sample_length=sl, seed=1)
    mixture, clean, name = ds[0]
    assert mixture.shape == (1, sl) and clean.shape == (1, sl) and mixture.dtype == torch.float32 and isinstance(name, str)
    loader = torch.utils.data.DataLoader(ds, batch_size=3, shuffle=False)

    def make():
        m = _model(n, ci, emu_engine)
        crit = loss_mod.mse_loss()
        crit._engine_override = emu_engine
        opt = optim_mod.FusedAdam(m.parameters(), lr=1e-3, betas=(0.9, 0.999))
        opt._engine_override = emu_engine
        return m, crit, opt

    cfg = {"root_dir": str(tmp_path), "experiment_name": "t", "trainer": {"epochs": 2, "save_checkpoint_interval": 2}}
    m1, crit1, opt1 = make()
    tr = trainer_mod.Trainer(cfg, False, m1, crit1, opt1, loader, None)
    tr.device = torch.device("cpu")                          # the emulator engine takes host tensors
    tr.model = m1.to("cpu")
    tr.train()
    m2, crit2, opt2 = make()
    m2.train()
    for _ in range(2):                                        # trainer/trainer.py:30-38 written out
        for mix, cl, _ in loader:
            opt2.zero_grad()
            crit2(cl, m2(mix)).backward()
            opt2.step()
    for (k, a), (_, b) in zip(m1.named_parameters(), m2.named_parameters()):
        assert torch.equal(a, b), k
    assert len(tr.epoch_losses) == 2 and tr.epoch_losses[1] < tr.epoch_losses[0]
    ck = torch.load((tmp_path / "t" / "checkpoints" / "latest_model.tar").as_posix())
    assert set(ck) == {"epoch", "best_score", "optimizer", "model"} and ck["epoch"] == 2
    assert set(ck["model"]) == set(plan.golden_state(n, ci, 0))          # the reference's 7*(2n+1)+2 keys
    m3, crit3, opt3 = make()
    tr3 = trainer_mod.Trainer(cfg, True, m3, crit3, opt3, loader, None)   # resume (base_trainer.py:62-81)
    assert tr3.start_epoch == 3
    for (k, a), (_, b) in zip(m1.named_parameters(), m3.named_parameters()):
        assert torch.equal(a, b), k


def test_fused_adam_state_round_trips_with_torch_adam(emu_engine):
    """The reference resumes with torch.optim.Adam.load_state_dict (trainer/base_trainer.py:74): FusedAdam's state_dict loads into
    torch's Adam and steps there, torch's loads into FusedAdam; options the reference never sets are refused, not ignored."""
    optim_mod = importlib.import_module(PKG_NAME + ".optim")

    def params():
        torch.manual_seed(3)
        return [torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.randn(7))]

    def grads(ps, k):
        g = torch.Generator().manual_seed(10 + k)
        for p in ps:
            p.grad = torch.randn(p.shape, generator=g)

    pa, pb = params(), params()
    fa = optim_mod.FusedAdam(pa, lr=2e-3, betas=(0.8, 0.99))
    fa._engine_override = emu_engine
    ta = torch.optim.Adam(pb, lr=2e-3, betas=(0.8, 0.99))
    assert set(fa.param_groups[0]) == set(ta.param_groups[0])
    for k in range(2):
        grads(pa, k); grads(pb, k)
        fa.step(); ta.step()
    # FusedAdam -> torch.optim.Adam: load and keep stepping
    pc = params()
    tc = torch.optim.Adam(pc, lr=2e-3, betas=(0.8, 0.99))
    tc.load_state_dict(copy.deepcopy(fa.state_dict()))       # (a checkpoint round trip copies; load_state_dict alone aliases the moments)
    with torch.no_grad():
        for p, q in zip(pc, pa):
            p.copy_(q)
    grads(pa, 2); grads(pb, 2); grads(pc, 2)
    fa.step(); ta.step(); tc.step()
    for a, b, c in zip(pa, pb, pc):
        assert (a - b).abs().max() < 1e-6 and (a - c).abs().max() < 1e-6
    # torch.optim.Adam -> FusedAdam
    pd = params()
    fd = optim_mod.FusedAdam(pd, lr=2e-3, betas=(0.8, 0.99))
    fd._engine_override = emu_engine
    fd.load_state_dict(copy.deepcopy(ta.state_dict()))
    with torch.no_grad():
        for p, q in zip(pd, pb):
            p.copy_(q)
    grads(pb, 3); grads(pd, 3)
    ta.step(); fd.step()
    for b, d in zip(pb, pd):
        assert (b - d).abs().max() < 1e-6
    bad = ta.state_dict()
    bad["param_groups"][0]["weight_decay"] = 0.01
    with pytest.raises(ValueError, match="weight_decay"):
        fd.load_state_dict(bad)


def test_fused_adam_device_step_counter_and_grad_scale(emu_engine):
    """device_step=True (the hipGraph-capturable form: the step count and bias corrections live on the device) and grad_scale
    (1/world folded into the step) give the numbers of the host-counted step on pre-scaled gradients."""
    optim_mod = importlib.import_module(PKG_NAME + ".optim")
    torch.manual_seed(4)
    base = [torch.randn(33, 5), torch.randn(9)]
    pa = [torch.nn.Parameter(t.clone()) for t in base]
    pb = [torch.nn.Parameter(t.clone()) for t in base]
    fa = optim_mod.FusedAdam(pa, lr=1e-2, device_step=True)
    fb = optim_mod.FusedAdam(pb, lr=1e-2)
    fa._engine_override = fb._engine_override = emu_engine
    fa.grad_scale = 0.25
    for k in range(4):
        g = torch.Generator().manual_seed(k)
        for a, b in zip(pa, pb):
            a.grad = torch.randn(a.shape, generator=g)
            b.grad = a.grad * 0.25
        fa.step(); fb.step()
    for a, b in zip(pa, pb):
        assert torch.equal(a, b)
    assert int(fa._dev[0][0]) == 4 and int(fa.state[pa[0]]["step"]) == 4
    # the step writes through raw pointers: autograd's version counters are told (a stale graph fails loudly)
    w = torch.nn.Parameter(torch.ones(3))
    fo = optim_mod.FusedAdam([w], lr=1e-2)
    fo._engine_override = emu_engine
    y = (w * w).sum()                        # saves w for its backward
    w.grad = torch.ones(3)
    v0 = w._version
    fo.step()
    assert w._version > v0
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        y.backward()


def test_fused_adam_shared_step_counter_is_invisible(emu_engine):
    """Inside FusedAdam the parameters of a group share ONE host step counter object (102 scalar increments per step were a tenth of
    the host's time per step); outside it must look like torch.optim.Adam's per-parameter counters: state_dict() hands out separate
    tensors, a parameter that has no gradient in a step does not advance (and leaves the shared counter), torch's Adam takes over."""
    optim_mod = importlib.import_module(PKG_NAME + ".optim")
    torch.manual_seed(5)
    base = [torch.randn(6, 4), torch.randn(11), torch.randn(3)]
    pa = [torch.nn.Parameter(t.clone()) for t in base]
    pb = [torch.nn.Parameter(t.clone()) for t in base]
    fa = optim_mod.FusedAdam(pa, lr=1e-2)
    fa._engine_override = emu_engine
    tb = torch.optim.Adam(pb, lr=1e-2)
    for k in range(5):
        g = torch.Generator().manual_seed(k)
        for i, (a, b) in enumerate(zip(pa, pb)):
            gr = torch.randn(a.shape, generator=g)
            skip = i == 1 and k in (2, 3)                    # the middle parameter sits two steps out
            a.grad = None if skip else gr.clone()
            b.grad = None if skip else gr.clone()
        fa.step(); tb.step()
    assert [int(fa.state[p]["step"]) for p in pa] == [int(tb.state[p]["step"]) for p in pb] == [5, 3, 5]
    for a, b in zip(pa, pb):
        assert (a - b).abs().max() < 1e-6
    assert fa.state[pa[0]]["step"] is fa.state[pa[2]]["step"] and fa.state[pa[1]]["step"] is not fa.state[pa[0]]["step"]
    sd = fa.state_dict()
    steps = [sd["state"][i]["step"] for i in range(3)]
    assert steps[0] is not steps[2] and steps[0].data_ptr() != steps[2].data_ptr()
    steps[0] += 10                                           # a holder of the checkpoint cannot move the live counters
    assert int(fa.state[pa[0]]["step"]) == 5
    # (counters that disagree: one launch per distinct count - checked above against torch's Adam; the device-resident counter of
    # the graph-capturable form is ONE per group and refuses)
    fd = optim_mod.FusedAdam(pa, lr=1e-2, device_step=True)
    fd._engine_override = emu_engine
    fd.load_state_dict(fa.state_dict())
    before = [(int(fd.state[p]["step"]), fd.state[p]["step"], fd.state[p]["exp_avg"].clone(), p.detach().clone()) for p in pa]
    with pytest.raises(RuntimeError, match="share the step count"):
        fd.step()
    # ... and refuses BEFORE it touches anything: a caller that catches the error finds counters (the same objects), moments and
    # parameters as they were
    for p, (n0, obj, m0, p0) in zip(pa, before):
        assert int(fd.state[p]["step"]) == n0 and fd.state[p]["step"] is obj
        assert torch.equal(fd.state[p]["exp_avg"], m0) and torch.equal(p.detach(), p0)


def test_bump_versions_on_a_torch_that_takes_one_tensor(monkeypatch):
    """torch 2.1 - 2.4: torch.autograd.graph.increment_version takes ONE tensor (a list raises TypeError); engine.bump_versions then
    goes tensor by tensor instead of failing every training forward and optimiser step."""
    engine_mod = importlib.import_module(PKG_NAME + ".engine")
    real = torch.autograd.graph.increment_version
    calls = []

    def one_tensor_only(t):
        if not isinstance(t, torch.Tensor):
            raise TypeError("increment_version(): argument 'tensor' must be Tensor, not list")
        calls.append(t)
        real(t)

    monkeypatch.setattr(torch.autograd.graph, "increment_version", one_tensor_only)
    a, b = torch.zeros(3), torch.zeros(2)
    va, vb = a._version, b._version
    engine_mod.bump_versions([a, b])
    assert a._version == va + 1 and b._version == vb + 1 and len(calls) == 2


def test_native_comm_wants_rank_with_world():
    par = importlib.import_module(PKG_NAME + ".parallel")
    with pytest.raises(ValueError, match="rank"):
        par.NativeComm(world=2)


def test_batched_validation_names_the_offending_tensor(emu_engine):
    """The host validates the 102 parameters / 75 buffers of a call in one pass (engine._require_all); a bad tensor is still refused
    before anything is enqueued, by name and index."""
    pkg = importlib.import_module(PKG_NAME)
    engine_mod = importlib.import_module(PKG_NAME + ".engine")
    m = pkg.Model(n_layers=2, channels_interval=4)
    m._engine_override = emu_engine
    torch.manual_seed(7)
    x = torch.randn(2, 1, 64)
    m(x)                                                            # (fine as built)
    m.encoder[1].main[0].weight.data = m.encoder[1].main[0].weight.data.double()
    with pytest.raises(engine_mod.WunetError, match=r"param\[4\]: expected float32"):
        m(x)
    m.encoder[1].main[0].weight.data = m.encoder[1].main[0].weight.data.float().transpose(0, 1).contiguous().transpose(0, 1)
    with pytest.raises(engine_mod.WunetError, match=r"param\[4\]: tensor must be contiguous"):
        m(x)
    # a replaced parameter is seen by the next call (the lists are read from the modules every time, nothing is cached)
    m.encoder[1].main[0].weight = torch.nn.Parameter(torch.randn(m.encoder[1].main[0].weight.shape))
    before = m(x)
    m.encoder[1].main[0].weight = torch.nn.Parameter(torch.randn(m.encoder[1].main[0].weight.shape))
    assert not torch.equal(m(x), before)


def test_second_backward_and_input_gradient_are_refused_with_a_message(emu_engine):
    m = _model(2, 4, emu_engine).train()
    x = torch.zeros(2, 1, 64)
    out = m(x)
    out.sum().backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="already run"):
        out.sum().backward()
    before = m.encoder[0].main[1].num_batches_tracked.clone()
    with pytest.raises(NotImplementedError, match="waveform input"):
        m(x.clone().requires_grad_(True))
    assert torch.equal(m.encoder[0].main[1].num_batches_tracked, before)      # refused before the forward touched the buffers


def test_fused_adam_load_state_dict_normalises_the_step(emu_engine):
    """A resumed state must hold "step" the way FusedAdam keeps it - a float32 scalar on the host - whatever the checkpoint held:
    the Python int of the torch 1.2 the reference was tested with (README.md:27), or a tensor that map_location moved elsewhere
    (trainer._resume_checkpoint loads with map_location=device: a device-side step would cost one blocking read-back per
    parameter per step and abort a graph capture)."""
    optim_mod = importlib.import_module(PKG_NAME + ".optim")
    torch.manual_seed(5)
    base = [torch.randn(6, 4), torch.randn(3)]

    def fresh(lr=1e-2):
        ps = [torch.nn.Parameter(t.clone()) for t in base]
        o = optim_mod.FusedAdam(ps, lr=lr)
        o._engine_override = emu_engine
        return ps, o

    def grads(ps, k):
        g = torch.Generator().manual_seed(20 + k)
        for p in ps:
            p.grad = torch.randn(p.shape, generator=g)

    pa, fa = fresh()
    for k in range(3):
        grads(pa, k); fa.step()
    for form in ("int", "f64-tensor", "f32-1d"):
        sd = copy.deepcopy(fa.state_dict())
        for st in sd["state"].values():
            st["step"] = {"int": 3, "f64-tensor": torch.tensor(3.0, dtype=torch.float64), "f32-1d": torch.tensor([3.0])[0]}[form]
        pb, fb = fresh()
        fb.load_state_dict(sd)
        with torch.no_grad():
            for p, q in zip(pb, pa):
                p.copy_(q)
        for st in fb.state.values():
            assert torch.is_tensor(st["step"]) and st["step"].device.type == "cpu" and st["step"].dtype == torch.float32
            assert st["step"].dim() == 0 and float(st["step"]) == 3.0
        pc = [torch.nn.Parameter(p.detach().clone()) for p in pa]
        fc = optim_mod.FusedAdam(pc, lr=1e-2)
        fc._engine_override = emu_engine
        fc.load_state_dict(copy.deepcopy(fa.state_dict()))
        grads(pb, 9); grads(pc, 9)
        fb.step(); fc.step()
        for b, c in zip(pb, pc):
            assert torch.equal(b, c), form
    # the signature a graph-replaying driver compares (trainer.Trainer._step re-captures when it changes)
    sig = fa.hyper_signature()
    fa.param_groups[0]["lr"] = 5e-3
    assert fa.hyper_signature() != sig
    sig = fa.hyper_signature()
    fa.grad_scale = 0.5
    assert fa.hyper_signature() != sig


def test_engine_never_evicts_a_context_in_use(emu_engine):
    """engine.Engine keeps at most MAX_CONTEXTS shapes, least recently used first - but never destroys a context some caller still
    holds (stock nn.DataParallel drives one thread per replica through one engine, trainer/base_trainer.py:26-27)."""
    eng_mod = importlib.import_module(PKG_NAME + ".engine")
    lib_mod = importlib.import_module(PKG_NAME + "._lib")
    eng = eng_mod.Engine(lib=lib_mod.declare(emu_lib.lib()), host_memory=True)
    eng.MAX_CONTEXTS = 2
    with eng._using(1, 4, 1, 8) as held:
        handles = []
        for b in (2, 3, 4, 5):
            with eng._using(1, 4, b, 8) as h:
                handles.append(h.value)
        assert (1, 4, 1, 8, "None") in eng._ctx                       # still there: it is held
        assert len(eng._ctx) <= 3
        assert eng.lib.wunet_num_conv_layers(held) == 3                # and alive
    with eng._using(1, 4, 6, 8):
        pass
    assert (1, 4, 1, 8, "None") not in eng._ctx and len(eng._ctx) <= 2  # released -> evictable
