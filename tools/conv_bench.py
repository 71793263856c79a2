#!/usr/bin/env python
"""usage (GPU box): python tools/conv_bench.py [--trace] [--layers enc3,dec8,...] "<ENV=VAL ...>" "<ENV=VAL ...>" ...
Per-layer kernel times of the fp16-split conv / data-gradient kernels at the BASELINE geometries (12 levels, batch 64) for each
environment setting, through the single-op C-ABI entry points (wunet_op_conv1d_split / _dgrad_split: the planner's tiling) and the
library's HIP-event profiler.  --trace additionally prints the phase stamps of conv_h3d_kernel (wunet_debug_set_conv_trace)."""
import argparse
import ctypes
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "wave-u-net-for-speech-enhancement_amd"


def layers():
    out = []
    n, ci = 12, 24
    for i in range(1, n):
        out.append((f"enc{i}", i * ci, (i + 1) * ci, 15, 16384 >> i))
    for j in range(n):
        cout = (n - j) * ci
        L = 16384 >> (n - 1 - j)
        cin = (n * ci if j == 0 else (n - j + 1) * ci) + (n - j) * ci
        out.append((f"dec{j}", cin, cout, 5, L))
    return out


def profiled(lib, fn, reps):
    fn()
    torch.cuda.synchronize()
    lib.wunet_profile_enable(1)
    for _ in range(reps):
        fn()
    buf = ctypes.create_string_buffer(1 << 14)
    lib.wunet_profile_collect(buf, len(buf))
    lib.wunet_profile_enable(0)
    rows = {}
    for ln in buf.value.decode().strip().splitlines():
        name, n, ms, fl, by = ln.split("\t")
        rows[name] = (float(ms) / int(n) * 1e3, float(fl) / int(n))
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trace", action="store_true")
    ap.add_argument("--layers", default="")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--min-l", type=int, default=256)
    ap.add_argument("--what", choices=["conv", "wgrad"], default="conv", help="conv: forward + data gradient kernels; wgrad: weight gradient kernels")
    ap.add_argument("--batch", type=int, default=64, help="frames (the 256-MiB Infinity Cache holds a layer's operands at smaller batches)")
    ap.add_argument("cfgs", nargs="*", default=[""])
    a = ap.parse_args()
    lib = importlib.import_module(PKG + "._lib").load_hip()
    dev = torch.device("cuda:0")
    B = a.batch
    want = set(a.layers.split(",")) if a.layers else None
    print("%-6s %5s %4s %4s | " % ("layer", "L", "cin", "cout") + " | ".join("%-34s" % c[:34] for c in a.cfgs))
    tot = [[0.0, 0.0] for _ in a.cfgs]
    for name, cin, cout, K, L in layers():
        if L < a.min_l or (want and name not in want):
            continue
        g = torch.Generator(device=dev).manual_seed(L + cin)
        x = torch.randn(B, cin, L, device=dev, generator=g)
        w = torch.randn(cout, cin, K, device=dev, generator=g) / float(np.sqrt(cin * K))
        b = torch.randn(cout, device=dev, generator=g)
        gz = torch.randn(B, cout, L, device=dev, generator=g)
        z = torch.empty(B, cout, L, device=dev)
        dx = torch.empty(B, cin, L, device=dev)
        dw = torch.empty(cout, cin, K, device=dev)
        cells, ref = [], None
        for ci_, cfg in enumerate(a.cfgs):
            saved = {}
            for kv in cfg.split():
                k, v = kv.split("=")
                saved[k] = os.environ.get(k)
                os.environ[k] = v

            def run():
                if a.what == "wgrad":
                    assert lib.wunet_op_conv1d_wgrad_split(gz.data_ptr(), x.data_ptr(), dw.data_ptr(), B, cin, cout, L, K, None) == 0, lib.wunet_last_error()
                    return
                assert lib.wunet_op_conv1d_split(x.data_ptr(), w.data_ptr(), b.data_ptr(), z.data_ptr(), B, cin, cout, L, K, None) == 0, lib.wunet_last_error()
                assert lib.wunet_op_conv1d_dgrad_split(gz.data_ptr(), w.data_ptr(), dx.data_ptr(), B, cin, cout, L, K, None) == 0, lib.wunet_last_error()
            rows = profiled(lib, run, a.reps)
            torch.cuda.synchronize()
            res = (dw.clone(), dw.clone()) if a.what == "wgrad" else (z.clone(), dx.clone())
            if ref is None:
                ref = res
                same = ""
            else:
                same = "=" if (torch.equal(ref[0], res[0]) and torch.equal(ref[1], res[1])) else "DIFF"
            if a.what == "wgrad":
                ks = [(k, v) for k, v in rows.items() if k.startswith("wgrad_h3") and "reduce" not in k]
                us = sum(v[0] for _, v in ks)
                fl = 2.0 * B * L * cin * cout * K
            else:
                ks = [(k, v) for k, v in rows.items() if k.startswith("conv_h3")]
                # forward and data gradient may share one instantiation: the profile merges them, so report the sum of both launches
                us = sum(v[0] * (2 if len(ks) == 1 else 1) for _, v in ks)
                fl = 2.0 * 2.0 * B * L * cin * cout * K
            tot[ci_][0] += us
            tot[ci_][1] += fl
            cells.append("%7.1f us %4.0f TF %-4s %-12s" % (us, fl / us / 1e6, same, ",".join(k.split("kernel")[1] for k, _ in ks)[:12]))
            for k, v in saved.items():
                if v is None:
                    del os.environ[k]
                else:
                    os.environ[k] = v
        print("%-6s %5d %4d %4d | " % (name, L, cin, cout) + " | ".join(cells), flush=True)
    print("%-22s | " % "total fwd+dgrad" + " | ".join("%7.1f us %4.0f TF %17s" % (t[0], t[1] / max(t[0], 1e-9) / 1e6, "") for t in tot))

    if a.trace:
        nblk = 8192
        tr = torch.zeros(nblk * 64, dtype=torch.int64, device=dev)
        for name, cin, cout, K, L in layers():
            if L < a.min_l or (want and name not in want):
                continue
            x = torch.randn(B, cin, L, device=dev)
            w = torch.randn(cout, cin, K, device=dev) / float(np.sqrt(cin * K))
            b = torch.randn(cout, device=dev)
            z = torch.empty(B, cout, L, device=dev)
            for cfg in a.cfgs:
                if "XDMA=0" in cfg:
                    continue
                saved = {}
                for kv in cfg.split():
                    k, v = kv.split("=")
                    saved[k] = os.environ.get(k)
                    os.environ[k] = v
                lib.wunet_op_conv1d_split(x.data_ptr(), w.data_ptr(), b.data_ptr(), z.data_ptr(), B, cin, cout, L, K, None)
                tr.zero_()
                lib.wunet_debug_set_conv_trace(tr.data_ptr())
                lib.wunet_op_conv1d_split(x.data_ptr(), w.data_ptr(), b.data_ptr(), z.data_ptr(), B, cin, cout, L, K, None)
                lib.wunet_debug_set_conv_trace(None)
                torch.cuda.synchronize()
                for k, v in saved.items():
                    if v is None:
                        del os.environ[k]
                    else:
                        os.environ[k] = v
                t = tr.cpu().numpy().reshape(nblk, 64)
                t = t[t[:, 0] > 0]
                if not len(t):
                    continue
                t0 = t[:, 0].min()
                nst = (t > 0).sum(axis=1)
                ns = int(np.median(nst))
                t = t[nst == ns][:, :ns].astype(np.float64)
                d = np.diff(t, axis=1)
                print(f"trace {name} [{cfg}] blocks {len(t)} stamps/block {ns}: block span median {np.median(t[:, -1] - t[:, 0]):.0f} ticks; "
                      f"median deltas between stamps (ticks): " + " ".join("%d" % v for v in np.median(d, axis=0)), flush=True)
                if os.environ.get("WUNET_TRACE_REALTIME"):
                    # library built with -DWUNET_TRACE_REALTIME (tools/_lib_rt.so): stamps are the 100 MHz counter every XCD shares
                    full = tr.cpu().numpy().reshape(nblk, 64)
                    rows_ = full[(full > 0).sum(axis=1) >= 3]
                    cnt = (rows_ > 0).sum(axis=1)
                    first = rows_[:, 0].astype(np.float64)
                    lastv = np.array([r[c - 1] for r, c in zip(rows_, cnt)], dtype=np.float64)
                    t00 = first.min()
                    so = np.sort(first - t00) / 100.0
                    print(f"      realtime: {len(rows_)} blocks; first start -> last end {(lastv.max() - t00) / 100.0:.2f} us; block starts (us after the first) "
                          f"median {np.median(so):.2f} p90 {so[int(0.9 * (len(so) - 1))]:.2f} max {so[-1]:.2f}; block spans (us) median {np.median(lastv - first) / 100.0:.2f} "
                          f"max {np.max(lastv - first) / 100.0:.2f}; ends (us after the first start) median {np.median(lastv - t00) / 100.0:.2f}", flush=True)
                    ends = (lastv - t00) / 100.0
                    bx = np.nonzero((full > 0).sum(axis=1) >= 3)[0] % 8          # (grid.y == 1: row = blockIdx.x, XCD = blockIdx.x % 8)
                    print("      ends by XCD (median / max us): " + "  ".join("%.1f/%.1f" % (np.median(ends[bx == k]), ends[bx == k].max()) for k in range(8) if (bx == k).any())
                          + "; deciles of all ends: " + " ".join("%.1f" % v for v in np.quantile(ends, np.linspace(0, 1, 11))), flush=True)


if __name__ == "__main__":
    main()
