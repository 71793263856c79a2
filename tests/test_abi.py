"""-m "not gpu": the C-ABI library builds, loads on a GPU-less box and exports every symbol that
include/wunet_hip.h declares; the product path refuses to run without it and without a GPU."""
import ctypes
import importlib
import os
import re

import pytest
import torch

from conftest import PKG_NAME, ROOT


def _declared():
    src = open(os.path.join(ROOT, "include", "wunet_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(wunet_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported():
    lib_mod = importlib.import_module(PKG_NAME + "._lib")
    if not os.path.exists(lib_mod.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(lib_mod.LIB_PATH)          # HIP runtime initialises lazily: loading needs no GPU
    names = _declared()
    assert len(names) >= 18
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/wunet_hip.h but not exported"
    assert sorted(lib_mod.EXPORTS) == names, "engine binding list and header disagree"


def test_shape_errors_come_back_as_codes_not_crashes():
    lib_mod = importlib.import_module(PKG_NAME + "._lib")
    lib = lib_mod.declare(ctypes.CDLL(lib_mod.LIB_PATH))
    h = ctypes.c_void_p()
    assert lib.wunet_create(12, 24, 4, 16000, ctypes.byref(h)) == -1          # not divisible by 2^12 (reference: 16000 fails too, unet_basic.py:86,93)
    assert b"divisible by 2^n_layers" in lib.wunet_last_error()
    assert lib.wunet_create(12, 24, 4, 12288, ctypes.byref(h)) == 0           # 3 * 2^12: rows padded to 16384 inside
    lib.wunet_destroy(h)
    assert lib.wunet_create(12, 24, 4, 2048, ctypes.byref(h)) == -1           # 2048 >> 12 = 0: deeper than log2(T)
    assert lib.wunet_create(0, 24, 4, 16384, ctypes.byref(h)) == -1
    assert lib.wunet_create(12, 24, 4, 16384, ctypes.byref(h)) == 0
    assert lib.wunet_num_conv_layers(h) == 25
    fwd, tot = lib.wunet_workspace_bytes(h, 0), lib.wunet_workspace_bytes(h, 1)
    assert 0 < fwd < tot
    lib.wunet_destroy(h)


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_no_cpu_fallback(pkg):
    m = pkg.Model(n_layers=2, channels_interval=4)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 1, 64))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pkg.mse_loss()(torch.zeros(1, 1, 64), torch.zeros(1, 1, 64))


def test_torch_extension_is_optional_and_exposes_the_three_calls():
    """torch_ext/wunet_torch.cpp (built by __graft_entry__.build()): when it is there it loads without a GPU and exposes the three
    per-step calls; the engine of an injected library (the emulator tests) never uses it."""
    eng_mod = importlib.import_module(PKG_NAME + ".engine")
    ext = eng_mod._load_torch_ext()
    path = os.path.join(ROOT, PKG_NAME, "torch_ext", "_wunet_torch.so")
    assert (ext is not None) == (os.path.exists(path) and not os.environ.get("WUNET_LIB_PATH") and not os.environ.get("WUNET_NO_TORCH_EXT"))
    if ext is not None:
        for name in ("forward", "backward_range", "adam_step"):
            assert callable(getattr(ext, name))
        # a CPU tensor is refused by the extension with the product path's message, before anything is enqueued
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            ext.adam_step([torch.zeros(3)], [torch.zeros(3)], [torch.zeros(3)], [torch.zeros(3)], 1e-3, 0.9, 0.999, 1e-8, 1, 1.0, None, None)


def test_loader_prefetch_registers_are_untouched_until_their_wait():
    """conv_h3u_kernel's loader waves prefetch from inline asm and order the use of the loaded registers with hand-placed waits (wunet_h3u.h):
    tools/check_h3u_isa.py compiles the kernel to gfx950 ISA (hipcc cross-compiles here) and proves for this build that no instruction
    touches a prefetch destination between its load and the wait that covers it."""
    import subprocess
    import sys
    from conftest import ROOT
    import os
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_h3u_isa.py")], capture_output=True, text=True, timeout=600)
    # conv_h3u_kernel<NTG, COPY>: three tap groupings x (eval: no operand copy | training: the loaders' stores counted in their waits)
    assert p.returncode == 0 and "6 kernels checked, 0 problems" in p.stdout, p.stdout[-2000:] + p.stderr[-500:]


def test_the_isa_checker_sees_a_touched_prefetch_register():
    """The checker must FAIL on what it guards against: the same ISA with one instruction planted that (a) overwrites a prefetch load's destination
    right behind the load, (b) reads it just before the stage barrier that does not cover it yet - the compiler-inserted copy / temporary re-use
    that tools/check_h3u_isa.py exists to catch."""
    import importlib.util
    import os
    import re
    from conftest import ROOT
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    spec = importlib.util.spec_from_file_location("check_h3u_isa", os.path.join(ROOT, "tools", "check_h3u_isa.py"))
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    name, lines = next(iter(chk.kernels(chk.compile_isa())))          # (one instantiation is enough)
    assert chk.check(name, lines) == []
    barriers = [i for i, ln in enumerate(lines) if re.search(r"s_waitcnt vmcnt\(10\) lgkmcnt\(0\)", ln)]
    # a prefetch load of the steady loop: an asm global_load_dwordx4 behind the first stage barrier
    load = next(i for i in range(barriers[0], len(lines)) if re.match(r"\s*global_load_dwordx4 v\[(\d+):\d+\], v\d+, s\[", lines[i]))
    reg = int(re.match(r"\s*global_load_dwordx4 v\[(\d+):", lines[load]).group(1))
    overwritten = lines[:load + 1] + [f"\tv_mov_b32_e32 v{reg}, 0"] + lines[load + 1:]
    errs = chk.check(name, overwritten)
    assert errs and f"v_mov_b32_e32 v{reg}, 0" in errs[0], errs[:2]
    nxt = next(b for b in barriers if b > load)                        # the barrier that ends the load's own stage: the data need not be there yet
    read_early = lines[:nxt] + [f"\tv_add_f32_e32 v{reg + 1}, v{reg + 1}, v{reg + 1}"] + lines[nxt:]
    errs = chk.check(name, read_early)
    assert errs and "v_add_f32_e32" in errs[0], errs[:2]
