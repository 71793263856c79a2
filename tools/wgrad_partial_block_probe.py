#!/usr/bin/env python
"""What the partly filled input-channel block of wgrad_h3d_kernel costs: the weight gradient of the top decoder geometries with Cin as it is
(72 / 144 / 216 / 288: the last block of 64 input channels holds 8 / 16 / 24 / 32) and with Cin cut to whole blocks (64 / 128 / 192 / 256), through
the single-op C ABI entry and the library's HIP-event profiler.  Measurement only."""
import ctypes
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "wave-u-net-for-speech-enhancement_amd"
lib = importlib.import_module(PKG + "._lib").load_hip()
dev = torch.device("cuda:0")
B = 64


def run(cin, cout, L, K, reps=6):
    x = torch.randn(B, cin, L, device=dev)
    gz = torch.randn(B, cout, L, device=dev)
    dw = torch.empty(cout, cin, K, device=dev)
    f = lambda: lib.wunet_op_conv1d_wgrad_split(gz.data_ptr(), x.data_ptr(), dw.data_ptr(), B, cin, cout, L, K, None)
    assert f() == 0
    torch.cuda.synchronize()
    lib.wunet_profile_enable(1)
    for _ in range(reps):
        f()
    buf = ctypes.create_string_buffer(1 << 14)
    lib.wunet_profile_collect(buf, len(buf))
    lib.wunet_profile_enable(0)
    out = []
    for ln in buf.value.decode().strip().splitlines():
        name, n, ms, fl, by = ln.split("\t")
        if name.startswith("wgrad_h3") and "reduce" not in name:
            out.append((name, float(ms) / int(n) * 1e3))
    return out


for cout, L, K, cins in [(24, 16384, 5, (72, 64, 128)), (48, 8192, 5, (144, 128, 192)), (72, 4096, 5, (216, 192, 256)), (96, 2048, 5, (288, 256, 320)),
                         (48, 8192, 15, (24, 32)), (72, 4096, 15, (48, 32, 64)), (96, 2048, 15, (72, 64, 96))]:
    for cin in cins:
        r = run(cin, cout, L, K)
        us = sum(v for _, v in r)
        fl = 2.0 * B * L * cin * cout * K
        gb = 4.0 * B * L * (cin + cout) / 1e9
        print("k%-2d L %5d Cout %3d Cin %3d: %7.1f us  %5.0f TF  %5.2f TB/s of the operands read once   %s" % (K, L, cout, cin, us, fl / us / 1e6, gb / us * 1e3, r[0][0][12:48]), flush=True)
