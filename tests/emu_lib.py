"""TEST INFRASTRUCTURE ONLY.  ctypes access to tests/emu/libwunet_emu.so: the product's kernel and
host sources compiled against the CPU fiber emulator, with numpy arrays standing in for device memory."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
FP = ctypes.POINTER(ctypes.c_float)


def lib():
    global _LIB
    if _LIB is None:
        import fcntl
        with open(os.path.join(_HERE, "emu", ".build.lock"), "w") as lk:      # pytest-xdist workers must not rebuild concurrently
            fcntl.flock(lk, fcntl.LOCK_EX)
            subprocess.run(["make", "-C", os.path.join(_HERE, "emu"), "-s", "-j8"], check=True)
        L = ctypes.CDLL(os.path.join(_HERE, "emu", "libwunet_emu.so"))
        L.wunet_last_error.restype = ctypes.c_char_p
        L.wunet_workspace_bytes.restype = ctypes.c_size_t
        L.wunet_loss_scratch_bytes.restype = ctypes.c_size_t
        _LIB = L
    return _LIB


def fp(a):
    return a.ctypes.data_as(FP) if a is not None else None


def check(rc):
    if rc != 0:
        raise RuntimeError(f"wunet rc={rc}: {lib().wunet_last_error().decode()}")


def op_conv1d(x, w, bias):
    B, Cin, L = x.shape
    Cout, _, K = w.shape
    z = np.full((B, Cout, L), np.nan, np.float32)
    check(lib().wunet_op_conv1d(fp(x), fp(w), fp(bias), fp(z), B, Cin, Cout, L, K, None))
    return z


def op_dgrad(gz, w, Cin):
    B, Cout, L = gz.shape
    K = w.shape[2]
    dx = np.full((B, Cin, L), np.nan, np.float32)
    check(lib().wunet_op_conv1d_dgrad(fp(gz), fp(w), fp(dx), B, Cin, Cout, L, K, None))
    return dx


def op_wgrad(gz, x, K):
    B, Cout, L = gz.shape
    Cin = x.shape[1]
    dw = np.full((Cout, Cin, K), np.nan, np.float32)
    check(lib().wunet_op_conv1d_wgrad(fp(gz), fp(x), fp(dw), B, Cin, Cout, L, K, None))
    return dw


def op_conv1d_split(x, w, bias):
    B, Cin, L = x.shape
    Cout, _, K = w.shape
    z = np.full((B, Cout, L), np.nan, np.float32)
    check(lib().wunet_op_conv1d_split(fp(x), fp(w), fp(bias), fp(z), B, Cin, Cout, L, K, None))
    return z


def op_dgrad_split(gz, w, Cin):
    B, Cout, L = gz.shape
    K = w.shape[2]
    dx = np.full((B, Cin, L), np.nan, np.float32)
    check(lib().wunet_op_conv1d_dgrad_split(fp(gz), fp(w), fp(dx), B, Cin, Cout, L, K, None))
    return dx


def op_wgrad_split(gz, x, K):
    B, Cout, L = gz.shape
    Cin = x.shape[1]
    dw = np.full((Cout, Cin, K), np.nan, np.float32)
    check(lib().wunet_op_conv1d_wgrad_split(fp(gz), fp(x), fp(dw), B, Cin, Cout, L, K, None))
    return dw
