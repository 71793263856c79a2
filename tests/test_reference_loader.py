"""The drop-in boundary exercised by the REFERENCE's own code: `util/utils.py:55-72 initialize_config` resolves
the config strings of INTEGRATION.md §1 to this package, the product Model is seed-for-seed identical to the reference's at
construction (same 177 state_dict entries, same default init), and the reference's unmodified `train.py:main` runs an epoch
through the product's trainer / dataset / model plugins.  Needs /root/reference (build container only; the GPU box does not
have it - the seed-for-seed init is additionally pinned by a committed digest, tests/golden/init_digest.npz) and `shims/`
first on sys.path for the reference's third-party imports that this image lacks."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch

from conftest import PKG_NAME, ROOT

REF = "/root/reference"
needs_reference = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree only exists in the build container")


@pytest.fixture
def reference_on_path(monkeypatch):
    sys.dont_write_bytecode = True
    monkeypatch.syspath_prepend(REF)
    monkeypatch.syspath_prepend(os.path.join(ROOT, "shims"))
    for name in [m for m in sys.modules if m.split(".")[0] in ("util", "model", "trainer", "dataset", "train")]:
        monkeypatch.delitem(sys.modules, name, raising=False)
    yield
    for name in [m for m in sys.modules if m.split(".")[0] in ("util", "model", "trainer", "dataset", "train")]:
        sys.modules.pop(name, None)


def _init_digest(state_dict):
    out = []
    for k, v in state_dict.items():
        a = v.detach().double().numpy().ravel()
        out.append([a.size, a.sum(), np.abs(a).sum(), a[0], a[-1]])
    return np.array(out, np.float64)


def test_default_init_matches_committed_digest(pkg):
    """torch.manual_seed(0); Model() -> the digest recorded from the reference's Model() under the same seed."""
    fx = np.load(os.path.join(ROOT, "tests", "golden", "init_digest.npz"))
    torch.manual_seed(0)
    m = pkg.Model()
    sd = m.state_dict()
    assert list(sd.keys()) == [str(k) for k in fx["keys"]]
    assert sum(p.numel() for p in m.parameters()) == 10132802
    np.testing.assert_array_equal(_init_digest(sd), fx["digest"])


@needs_reference
def test_default_init_is_seed_for_seed_the_references(pkg, reference_on_path):
    from model.unet_basic import Model as RefModel
    for kw in ({}, {"n_layers": 5, "channels_interval": 8}):
        torch.manual_seed(0)
        ref = RefModel(**kw)
        torch.manual_seed(0)
        ours = pkg.Model(**kw)
        rs, os_ = ref.state_dict(), ours.state_dict()
        assert list(rs.keys()) == list(os_.keys())
        for k in rs:
            assert rs[k].shape == os_[k].shape and torch.equal(rs[k], os_[k]), k
        assert [tuple(p.shape) for p in ref.parameters()] == [tuple(p.shape) for p in ours.parameters()]
        ours.load_state_dict(rs)                  # reference checkpoints load both ways
        ref.load_state_dict(os_)
    if not os.path.exists(os.path.join(ROOT, "tests", "golden", "init_digest.npz")) or os.environ.get("WUNET_REGEN_GOLDEN"):
        torch.manual_seed(0)
        sd = RefModel().state_dict()
        np.savez(os.path.join(ROOT, "tests", "golden", "init_digest.npz"), keys=np.array(list(sd.keys())), digest=_init_digest(sd))


@needs_reference
def test_reference_initialize_config_resolves_the_integration_strings(reference_on_path):
    from util.utils import initialize_config              # the reference's loader, unmodified
    model = initialize_config({"module": PKG_NAME + ".model", "main": "Model", "args": {}})
    assert isinstance(model, torch.nn.Module) and model.n_layers == 12 and model.channels_interval == 24
    assert len(model.state_dict()) == 177
    small = initialize_config({"module": PKG_NAME + ".model", "main": "Model", "args": {"n_layers": 3, "channels_interval": 4}})
    assert small.n_layers == 3
    for name in ("mse_loss", "l1_loss", "smooth_l1_loss"):
        crit = initialize_config({"module": PKG_NAME + ".loss", "main": name, "args": {}})
        assert callable(crit)
    trainer_cls = initialize_config({"module": PKG_NAME + ".trainer", "main": "Trainer"}, pass_args=False)
    assert trainer_cls is importlib.import_module(PKG_NAME + ".trainer").Trainer
    ds = initialize_config({"module": PKG_NAME + ".dataset", "main": "Dataset", "args": {"n_items": 5, "sample_length": 64}})
    mix, clean, name = ds[0]
    assert mix.shape == (1, 64) and len(ds) == 5


@needs_reference
def test_reference_train_py_runs_an_epoch_through_the_plugins(reference_on_path, tmp_path):
    """/root/reference/train.py:main(config, resume) unmodified: it builds the DataLoaders, torch.optim.Adam and calls our
    trainer plugin's .train() - with the model / loss plugins bound to the CPU emulator engine (tests/emu_plugins.py), since the
    product path has no CPU fallback.  Result == the written-out loop of trainer/trainer.py:30-38."""
    import train as ref_train
    cfg = {
        "seed": 0, "root_dir": str(tmp_path), "experiment_name": "ref_main",
        "train_dataset": {"module": PKG_NAME + ".dataset", "main": "Dataset", "args": {"n_items": 6, "sample_length": 64, "seed": 1}},
        "validation_dataset": {"module": PKG_NAME + ".dataset", "main": "Dataset", "args": {"n_items": 1, "sample_length": 64}},
        "train_dataloader": {"batch_size": 3, "num_workers": 0, "shuffle": False, "pin_memory": False},
        "model": {"module": "emu_plugins", "main": "Model", "args": {"n_layers": 2, "channels_interval": 4}},
        "optimizer": {"lr": 1e-3, "beta1": 0.9, "beta2": 0.999},
        "loss_function": {"module": "emu_plugins", "main": "mse_loss", "args": {}},
        "trainer": {"module": PKG_NAME + ".trainer", "main": "Trainer", "epochs": 1, "save_checkpoint_interval": 1},
    }
    ref_train.main(cfg, resume=False)
    ck = torch.load((tmp_path / "ref_main" / "checkpoints" / "latest_model.tar").as_posix())
    assert ck["epoch"] == 1 and len(ck["model"]) == 7 * 5 + 2
    # the same two steps written out (train.py:12-13 seeds, :29 model, :31-35 Adam; trainer/trainer.py:30-38)
    import emu_plugins
    torch.manual_seed(0)
    np.random.seed(0)
    ds = importlib.import_module(PKG_NAME + ".dataset").Dataset(n_items=6, sample_length=64, seed=1)
    m = emu_plugins.Model(n_layers=2, channels_interval=4)
    opt = torch.optim.Adam(params=m.parameters(), lr=1e-3, betas=(0.9, 0.999))
    crit = emu_plugins.mse_loss()
    m.train()
    for mix, cl, _ in torch.utils.data.DataLoader(ds, batch_size=3, shuffle=False):
        opt.zero_grad()
        crit(cl, m(mix)).backward()
        opt.step()
    for k, v in m.state_dict().items():
        assert torch.equal(v, ck["model"][k]), k
