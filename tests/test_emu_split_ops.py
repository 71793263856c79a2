"""Single-op parity of the fp16-split kernels (conv_h3_kernel, wgrad_h3d_kernel, wgrad_h3_kernel) through the C ABI's
wunet_op_*_split entry points, on the CPU fiber emulator at small geometries that reach every tile shape: un-segmented tiles
(L >= 256), 2 .. 16 items per tile (L = 128 .. 16), split-K, channel counts that are not multiples of 8 / 16 / 32, and operands
whose magnitudes sit far from 1 (the device-derived power-of-two scales).  tests/test_gpu_parity.py runs the same entry points at
all BASELINE layer geometries on the hardware.  Oracle: F.conv1d semantics restated in C (oracle/wunet_oracle.c, f64 build)."""
import numpy as np
import pytest

import emu_lib
from oracle import c_oracle


def rel_norm(a, r):
    return float(np.linalg.norm((a - r).ravel()) / max(np.linalg.norm(r.ravel()), 1e-30))


@pytest.mark.parametrize("B,Cin,Cout,L,K,xs,ws", [
    (1, 24, 40, 256, 15, 1.0, 1.0),        # one un-segmented tile
    (2, 20, 24, 512, 5, 1e-4, 3e3),        # channel counts off the 8 / 16 grid; small activations, large weights
    (3, 48, 24, 128, 5, 50.0, 1e-5),       # two-item tiles, odd batch (half-empty tile)
    (5, 16, 56, 64, 15, 1.0, 1.0),         # four-item tiles
    (9, 72, 40, 32, 5, 1e3, 1e-3),         # eight-item tiles: 2 items under one wave
    (17, 40, 72, 16, 15, 1.0, 1.0),        # sixteen-item tiles: 4 items under one wave
    # K tails of conv_h3d_kernel (channel groups left over after the last chunk of 32: one tail stage per group, taps spread over
    # the K quarters of the MFMA) - the cases above already hold 1 / 2 / 3 left-over groups with and without a full chunk; here
    # behind two full chunks, split K with splits that start inside the tail, and a 5-tap tail behind three chunks
    (2, 72, 80, 512, 15, 1.0, 1.0),        # 9 groups forward (2 chunks + 1), 10 backward (2 chunks + 2)
    (3, 104, 24, 256, 5, 1.0, 1.0),        # 13 groups forward: 3 chunks + a 5-tap tail stage of 2 steps
])
def test_split_ops_vs_oracle(B, Cin, Cout, L, K, xs, ws):
    rng = np.random.default_rng(B * 1000 + Cin)
    x = (xs * rng.standard_normal((B, Cin, L))).astype(np.float32)
    w = (ws * rng.standard_normal((Cout, Cin, K)) / np.sqrt(Cin * K)).astype(np.float32)
    b = (xs * ws * rng.standard_normal((Cout,))).astype(np.float32)
    gz = (rng.standard_normal((B, Cout, L)) / xs).astype(np.float32)
    zr = c_oracle.conv1d_fwd(x, w, b)
    dxr, dwr, _ = c_oracle.conv1d_bwd(gz, x, w)
    z = emu_lib.op_conv1d_split(x, w, b)
    assert rel_norm(z, zr) < 2e-6 and np.abs(z - zr).max() < 2e-6 * np.abs(zr).max()
    if Cin >= 16:
        dx = emu_lib.op_dgrad_split(gz, w, Cin)
        dw = emu_lib.op_wgrad_split(gz, x, K)
        assert rel_norm(dx, dxr) < 2e-6 and np.abs(dx - dxr).max() < 2e-6 * np.abs(dxr).max()
        assert rel_norm(dw, dwr) < 2e-6 and np.abs(dw - dwr).max() < 4e-6 * np.abs(dwr).max()


def test_split_ops_reject_shapes_the_kernels_do_not_cover():
    x = np.zeros((1, 16, 8), np.float32)
    w = np.zeros((16, 16, 5), np.float32)
    with pytest.raises(RuntimeError, match="L >= 16"):
        emu_lib.op_conv1d_split(x, w, None)
    x = np.zeros((2, 8, 256), np.float32)
    w = np.zeros((16, 8, 5), np.float32)
    with pytest.raises(RuntimeError, match="Cin >= 16"):
        emu_lib.op_dgrad_split(np.zeros((2, 16, 256), np.float32), w, 8)
