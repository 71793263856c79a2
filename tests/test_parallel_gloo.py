"""world_size 2 / 4 / 8 gloo tests (CPU) of the data-parallel path: parallel.GradSync buckets the flat
gradient buffer in backward order, all-reduces each finished bucket and averages - and the result
equals the average of the per-shard oracle gradients with per-shard BatchNorm (the DataParallel
semantics of the reference, trainer/base_trainer.py:26-27)."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG_NAME, ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bucket_ranges_cover_everything_in_backward_order():
    parallel = importlib.import_module(PKG_NAME + ".parallel")
    plan = importlib.import_module(PKG_NAME + ".plan")
    n, ci = 12, 24
    numels = []
    for c_in, c_out, k in plan.conv_layer_shapes(n, ci):
        numels += [c_out * c_in * k, c_out, c_out, c_out]
    numels += [ci + 1, 1]
    nl = 2 * n + 1
    for nb in (1, 2, 4, 7):
        ranges = parallel.bucket_ranges(numels, nl, nb)
        assert 1 <= len(ranges) <= nb
        assert ranges[0][1] == nl and ranges[-1][0] == 0           # starts at the head, ends at encoder[0]
        for (lb, le, fb, fe), nxt in zip(ranges, ranges[1:] + [None]):
            assert lb < le and fb < fe
            if nxt is not None:
                assert nxt[1] == lb and nxt[3] == fb                  # contiguous, walking backwards
        assert ranges[-1][2] == 0 and ranges[0][3] == sum(numels)    # whole flat buffer, head params included
        if nb == 4:
            sizes = [fe - fb for _, _, fb, fe in ranges]
            assert max(sizes) < 0.6 * sum(numels)                    # balanced enough to overlap


def _worker(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import emu_lib
        from oracle import plan
        eng_mod = importlib.import_module(PKG_NAME + ".engine")
        lib_mod = importlib.import_module(PKG_NAME + "._lib")
        model_mod = importlib.import_module(PKG_NAME + ".model")
        loss_mod = importlib.import_module(PKG_NAME + ".loss")
        parallel = importlib.import_module(PKG_NAME + ".parallel")
        eng = eng_mod.Engine(lib=lib_mod.declare(emu_lib.lib()), host_memory=True)
        n, ci, B, T = 3, 8, 2, 64
        sd = plan.golden_state(n, ci, 0)
        m = model_mod.Model(n_layers=n, channels_interval=ci)
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
        m._engine_override = eng
        m.grad_sync = parallel.GradSync(n_buckets=3)
        noisy, clean = plan.golden_batch(B * world, T, 0)
        sl = slice(rank * B, (rank + 1) * B)                          # this rank's shard of the global batch
        crit = loss_mod.mse_loss()
        crit._engine_override = eng
        m.train()
        out = m(torch.from_numpy(noisy[sl].copy()))
        crit(torch.from_numpy(clean[sl].copy()), out).backward()
        grads = {k: p.grad.numpy().copy() for k, p in m.named_parameters()}
        np.savez(os.path.join(tmpdir, f"rank{rank}.npz"), **grads)
        # every rank must hold identical (averaged) gradients
        flat = m.last_flat_grad.clone()
        ref = flat.clone()
        dist.broadcast(ref, src=0)
        assert torch.equal(flat, ref)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_grad_sync_matches_average_of_shard_oracles(tmp_path, world):
    """with the global batch split into `world` equal shards (nn.DataParallel's scatter,
    trainer/base_trainer.py:26-27), every rank ends the backward holding the MEAN of the per-shard oracle gradients - each shard
    back-propagated on its own with its own BatchNorm statistics - at world 2, 4 and the node's 8."""
    from oracle import c_oracle, plan
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    n, ci, B, T = 3, 8, 2, 64
    noisy, clean = plan.golden_batch(B * world, T, 0)
    refs = []
    for r in range(world):
        sd = plan.golden_state(n, ci, 0)
        refs.append(c_oracle.step(sd, noisy[r * B:(r + 1) * B], clean[r * B:(r + 1) * B], n, ci, True, "mse", precision="f64")["grads"])
    got = [np.load(os.path.join(str(tmp_path), f"rank{r}.npz")) for r in range(world)]
    for k in refs[0]:
        avg = sum(ref[k].astype(np.float64) for ref in refs) / world
        for r in range(1, world):
            assert np.array_equal(got[0][k], got[r][k]), (k, r)
        if k.endswith(".0.bias") and not k.startswith("out"):
            assert np.all(got[0][k] == 0.0)
            continue
        scale = max(np.abs(avg).max(), 1e-6)
        assert np.abs(got[0][k] - avg).max() < 3e-4 * scale + 1e-6, (k, np.abs(got[0][k] - avg).max(), scale)
        # the per-shard BatchNorm matters: the gradient of the whole batch as ONE shard is a different number
    whole = c_oracle.step(plan.golden_state(n, ci, 0), noisy, clean, n, ci, True, "mse", precision="f64")["grads"]
    k = "encoder.1.main.0.weight"
    avg = sum(ref[k].astype(np.float64) for ref in refs) / world
    assert np.abs(whole[k] - avg).max() > 1e-3 * np.abs(avg).max()


# ---------------------------------------------------------------------------------------------------------------------
# the whole data-parallel training path through the trainer plugin: torchrun environment -> Trainer joins the process group,
# shards the DataLoader, attaches GradSync (1/world folded into FusedAdam) - both ranks must hold identical parameters, equal to
# a one-process simulation of the two shards (trainer/base_trainer.py:26-27 semantics: per-replica BatchNorm, averaged gradients)
def _trainer_worker(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import emu_lib
    from oracle import plan
    eng_mod = importlib.import_module(PKG_NAME + ".engine")
    lib_mod = importlib.import_module(PKG_NAME + "._lib")
    eng = eng_mod.Engine(lib=lib_mod.declare(emu_lib.lib()), host_memory=True)
    n, ci, sl = 2, 4, 64
    m = importlib.import_module(PKG_NAME + ".model").Model(n_layers=n, channels_interval=ci)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in plan.golden_state(n, ci, 0).items()})
    m._engine_override = eng
    crit = importlib.import_module(PKG_NAME + ".loss").mse_loss()
    crit._engine_override = eng
    opt = importlib.import_module(PKG_NAME + ".optim").FusedAdam(m.parameters(), lr=1e-3, betas=(0.9, 0.999))
    opt._engine_override = eng
    ds = importlib.import_module(PKG_NAME + ".dataset").Dataset(n_items=12, sample_length=sl, seed=1)
    loader = torch.utils.data.DataLoader(ds, batch_size=4, shuffle=False)          # the GLOBAL batch, as train.py builds it
    cfg = {"root_dir": tmpdir, "experiment_name": "dp", "trainer": {"epochs": 1, "save_checkpoint_interval": 1}}
    tr = importlib.import_module(PKG_NAME + ".trainer").Trainer(cfg, False, m, crit, opt, loader, None)
    assert dist.is_initialized() and tr.world == world and tr.rank == rank
    assert tr.train_data_loader.batch_size == 2 and len(tr.train_data_loader) == 3
    assert m.grad_sync is not None and m.grad_sync.scale_in_optimizer and opt.grad_scale == 0.5
    tr.train()
    torch.save({k: v.detach().clone() for k, v in m.named_parameters()}, os.path.join(tmpdir, f"params{rank}.pt"))


def test_trainer_under_torchrun_env_trains_data_parallel(tmp_path):
    world = 2
    mp.spawn(_trainer_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    p0 = torch.load(os.path.join(str(tmp_path), "params0.pt"))
    p1 = torch.load(os.path.join(str(tmp_path), "params1.pt"))
    for k in p0:
        assert torch.equal(p0[k], p1[k]), k                                        # identical replicas after 3 steps
    assert os.path.exists(os.path.join(str(tmp_path), "dp", "checkpoints", "latest_model.tar"))     # written once, by rank 0
    # one-process simulation: per step, each shard's gradients from the same parameters (its own BatchNorm statistics),
    # summed in fp32 (the all-reduce), 1/2 applied inside the Adam step
    import emu_lib
    from oracle import plan
    eng_mod = importlib.import_module(PKG_NAME + ".engine")
    lib_mod = importlib.import_module(PKG_NAME + "._lib")
    eng = eng_mod.Engine(lib=lib_mod.declare(emu_lib.lib()), host_memory=True)
    n, ci, sl = 2, 4, 64
    m = importlib.import_module(PKG_NAME + ".model").Model(n_layers=n, channels_interval=ci)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in plan.golden_state(n, ci, 0).items()})
    m._engine_override = eng
    m.train()
    crit = importlib.import_module(PKG_NAME + ".loss").mse_loss()
    crit._engine_override = eng
    opt = importlib.import_module(PKG_NAME + ".optim").FusedAdam(m.parameters(), lr=1e-3, betas=(0.9, 0.999))
    opt._engine_override = eng
    opt.grad_scale = 0.5
    ds = importlib.import_module(PKG_NAME + ".dataset").Dataset(n_items=12, sample_length=sl, seed=1)
    for step in range(3):
        total = None
        for r in range(world):                       # DistributedSampler(shuffle=False): rank r holds items r, r+2, r+4, ...
            items = [ds[r + world * (2 * step + j)] for j in range(2)]
            mix = torch.stack([it[0] for it in items])
            cl = torch.stack([it[1] for it in items])
            opt.zero_grad(set_to_none=True)
            crit(cl, m(mix)).backward()
            g = [p.grad.clone() for p in m.parameters()]
            total = g if total is None else [a + b for a, b in zip(total, g)]
        for p, g in zip(m.parameters(), total):
            p.grad = g
        opt.step()
    for k, p in m.named_parameters():
        assert torch.equal(p.detach(), p0[k]), k


def test_native_comm_entry_at_world_one():
    """include/wunet_hip.h wunet_comm_*: the library's own all-reduce entry.  The CPU test build has no RCCL: world size 1 must work
    without it (the sum over one rank is the identity, GradSync's bucketed schedule runs unchanged), a larger world must fail with a
    message, not crash."""
    import ctypes
    import emu_lib
    from oracle import plan
    eng_mod = importlib.import_module(PKG_NAME + ".engine")
    lib_mod = importlib.import_module(PKG_NAME + "._lib")
    parallel = importlib.import_module(PKG_NAME + ".parallel")
    eng = eng_mod.Engine(lib=lib_mod.declare(emu_lib.lib()), host_memory=True)
    comm = parallel.NativeComm(engine=eng, world=1, rank=0)
    assert eng.lib.wunet_comm_world(comm.handle) == 1
    t = torch.arange(10, dtype=torch.float32)
    assert torch.equal(comm.all_reduce_(t.clone()), t)
    h = ctypes.c_void_p()
    assert eng.lib.wunet_comm_create((ctypes.c_ubyte * 128)(), 2, 0, ctypes.byref(h)) != 0
    assert b"RCCL" in eng.lib.wunet_last_error()
    n, ci, B, T = 3, 8, 2, 64
    noisy, clean = plan.golden_batch(B, T, 0)

    def grads(sync):
        m = importlib.import_module(PKG_NAME + ".model").Model(n_layers=n, channels_interval=ci)
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in plan.golden_state(n, ci, 0).items()})
        m._engine_override = eng
        m.grad_sync = sync
        crit = importlib.import_module(PKG_NAME + ".loss").mse_loss()
        crit._engine_override = eng
        m.train()
        crit(torch.from_numpy(clean), m(torch.from_numpy(noisy))).backward()
        return [p.grad.clone() for p in m.parameters()]

    a = grads(None)
    b = grads(parallel.GradSync(n_buckets=3, always_reduce=True, comm=comm))
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    comm.close()
