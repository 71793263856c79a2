"""TEST INFRASTRUCTURE ONLY (oracle/).  ctypes binding of oracle/wunet_oracle.c."""
import ctypes
import os
import subprocess

import numpy as np

from .plan import conv_layers, param_names

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}
_FP = ctypes.POINTER(ctypes.c_float)


def build():
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)


def lib(precision="f64"):
    if precision not in _LIBS:
        path = os.path.join(_HERE, f"libwunet_oracle_{precision}.so")
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        L.wuo_step.restype = ctypes.c_int
        _LIBS[precision] = L
    return _LIBS[precision]


def _fp(a):
    return a.ctypes.data_as(_FP) if a is not None else None


def conv1d_fwd(x, w, bias, precision="f64"):
    B, Cin, L = x.shape
    Cout, _, K = w.shape
    x = np.ascontiguousarray(x, np.float32); w = np.ascontiguousarray(w, np.float32)
    b = None if bias is None else np.ascontiguousarray(bias, np.float32)
    z = np.empty((B, Cout, L), np.float32)
    lib(precision).wuo_conv1d_fwd(_fp(x), _fp(w), _fp(b), _fp(z), B, Cin, Cout, L, K)
    return z


def conv1d_bwd(gz, x, w, precision="f64"):
    B, Cin, L = x.shape
    Cout, _, K = w.shape
    gz = np.ascontiguousarray(gz, np.float32); x = np.ascontiguousarray(x, np.float32)
    w = np.ascontiguousarray(w, np.float32)
    dx = np.empty_like(x); dw = np.empty_like(w); db = np.empty((Cout,), np.float32)
    lib(precision).wuo_conv1d_bwd(_fp(gz), _fp(x), _fp(w), _fp(dx), _fp(dw), _fp(db), B, Cin, Cout, L, K)
    return dx, dw, db


def upsample2x_fwd(x, precision="f64"):
    B, C, Lin = x.shape
    x = np.ascontiguousarray(x, np.float32)
    y = np.empty((B, C, 2 * Lin), np.float32)
    lib(precision).wuo_upsample2x_fwd(_fp(x), _fp(y), B, C, Lin)
    return y


def upsample2x_bwd(gy, precision="f64"):
    B, C, Lout = gy.shape
    gy = np.ascontiguousarray(gy, np.float32)
    gx = np.empty((B, C, Lout // 2), np.float32)
    lib(precision).wuo_upsample2x_bwd(_fp(gy), _fp(gx), B, C, Lout // 2)
    return gx


def step(sd, noisy, clean=None, n_layers=12, ci=24, training=True, loss="mse", want_grads=True,
         want_acts=False, precision="f64"):
    """Run the restated network.  sd: name -> numpy array (running stats / num_batches_tracked are
    updated IN PLACE when training).  Returns dict(out, loss, grads{name: array}, acts[list])."""
    names = param_names(n_layers, ci)
    layers = conv_layers(n_layers, ci)
    NL = len(layers)
    B, _, T = noisy.shape
    noisy = np.ascontiguousarray(noisy, np.float32)
    params = [np.ascontiguousarray(sd[n], np.float32) for n in names]
    running = []
    for prefix, *_ in layers:
        for s in ("running_mean", "running_var"):
            a = sd[f"{prefix}.1.{s}"]
            assert a.dtype == np.float32 and a.flags.c_contiguous
            running.append(a)
    nbt = np.array([int(sd[f"{p}.1.num_batches_tracked"]) for p, *_ in layers], dtype=np.int64)
    out = np.empty((B, 1, T), np.float32)
    loss_out = ctypes.c_float(float("nan"))
    kind = {"mse": 0, "l1": 1, "smooth_l1": 2}[loss]
    do_grads = want_grads and clean is not None
    grads = [np.zeros_like(p) for p in params] if do_grads else None
    acts = [np.empty((B, c_out, T >> (i if i <= n_layers else 2 * n_layers - i)), np.float32)
            for i, (_, _, c_out, _) in enumerate(layers)] if want_acts else None
    PA = (_FP * len(params))(*[_fp(p) for p in params])
    RA = (_FP * len(running))(*[_fp(r) for r in running])
    GA = (_FP * len(params))(*[_fp(g) for g in grads]) if do_grads else None
    AA = (_FP * NL)(*[_fp(a) for a in acts]) if want_acts else None
    cl = np.ascontiguousarray(clean, np.float32) if clean is not None else None
    rc = lib(precision).wuo_step(n_layers, ci, B, T, PA, RA, nbt.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong)),
                                 _fp(noisy), _fp(cl), int(training), kind, _fp(out), ctypes.byref(loss_out),
                                 GA, AA)
    if rc != 0:
        raise RuntimeError(f"wuo_step failed rc={rc}")
    for i, (prefix, *_) in enumerate(layers):
        sd[f"{prefix}.1.num_batches_tracked"] = np.array(nbt[i], dtype=np.int64)
    return {"out": out, "loss": float(loss_out.value),
            "grads": dict(zip(names, grads)) if do_grads else None, "acts": acts}
