"""TEST INFRASTRUCTURE ONLY.  Plugin module in the reference's config convention (util/utils.py:55-72) whose objects are the
product's Model / losses bound to the CPU emulator engine, so the reference's own train.py can run a real epoch in the GPU-less
build container (tests/test_reference_loader.py)."""
import importlib

import emu_lib
from conftest import PKG_NAME

_ENGINE = None


def _engine():
    global _ENGINE
    if _ENGINE is None:
        eng_mod = importlib.import_module(PKG_NAME + ".engine")
        lib_mod = importlib.import_module(PKG_NAME + "._lib")
        _ENGINE = eng_mod.Engine(lib=lib_mod.declare(emu_lib.lib()), host_memory=True)
    return _ENGINE


def Model(**kw):
    m = importlib.import_module(PKG_NAME + ".model").Model(**kw)
    m._engine_override = _engine()
    return m


def mse_loss():
    crit = importlib.import_module(PKG_NAME + ".loss").mse_loss()
    crit._engine_override = _engine()
    return crit
