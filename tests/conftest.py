import importlib
import os
import sys

import pytest

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PKG_NAME = "wave-u-net-for-speech-enhancement_amd"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    """The product package (its directory name has hyphens, so it is importlib-only -
    exactly how the reference's initialize_config resolves plugins, util/utils.py:55-72)."""
    return importlib.import_module(PKG_NAME)


def golden(name):
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"), allow_pickle=False)
