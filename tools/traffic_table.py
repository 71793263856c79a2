#!/usr/bin/env python
"""usage: python tools/traffic_table.py <bench.json (WUNET_BENCH_ALL=1)> [profiles/pmc_traffic.json]
Per-kernel-family HBM traffic of one training step: counter bytes (rocprofv3 --pmc FETCH_SIZE x 2 / WRITE_SIZE, read and written)
next to the bytes that family must move (the library's own annotation: operands read once + results written once), the time the
family takes (HIP events, serial pass of bench.py) and the time the SAME bytes take at the chip's measured streaming rates
(profiles/r3_stream_rate_microbench.txt: 6.4 TB/s for reads, 4.5 TB/s for writes of tensors beyond the Infinity Cache)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
READ_TBS, WRITE_TBS = 6.4, 4.5
b = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
pmc = json.load(open(sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "pmc_traffic.json")))
r = b["roofline"]
fam = {}
for k, v in pmc["kernels"].items():
    f = fam.setdefault(k.split("<")[0], {"n": 0.0, "rd": 0.0, "wr": 0.0, "ms": None, "alg": None})
    f["n"] += v["launches_per_step"]
    f["rd"] += 2.0 * v["FETCH_SIZE_KiB"] * 1024 * v["launches_per_step"]
    f["wr"] += v["WRITE_SIZE_KiB"] * 1024 * v["launches_per_step"]
for t in r["top5"]:
    f = fam.setdefault(t["kernel"].split("<")[0], {"n": 0.0, "rd": 0.0, "wr": 0.0, "ms": None, "alg": None})
    f["ms"] = (f["ms"] or 0.0) + t["ms_per_step"]
    if "algorithmic_bytes_per_step" in t:
        f["alg"] = (f["alg"] or 0.0) + t["algorithmic_bytes_per_step"]
for m in r["memory_bound_kernels"]:
    f = fam.setdefault(m["kernel"], {"n": 0.0, "rd": 0.0, "wr": 0.0, "ms": None, "alg": None})
    f["ms"] = (f["ms"] or 0.0) + m["ms_per_step"]
    f["alg"] = (f["alg"] or 0.0) + m["algorithmic_GBps"] * 1e9 * m["ms_per_step"] * 1e-3
tot = sum(f["rd"] + f["wr"] for f in fam.values())
print("HBM traffic of one training step by kernel family (12 levels, batch 64 x 16384; %s)" % b["config"]["workload"][:60])
print("counter = rocprofv3 PMC (2 x FETCH_SIZE, WRITE_SIZE); must-move = operands read once + results written once (the library's annotation);")
print("t_stream = counter bytes at the measured streaming rates (reads %.1f TB/s, writes %.1f TB/s); ms = HIP events, one stream" % (READ_TBS, WRITE_TBS))
print("%-32s %6s %9s %9s %9s %7s %8s %9s %8s" % ("family", "n/step", "read GB", "write GB", "must GB", "ratio", "ms", "t_stream", "ms/t_str"))
for k, f in sorted(fam.items(), key=lambda kv: -(kv[1]["rd"] + kv[1]["wr"])):
    cnt = f["rd"] + f["wr"]
    if cnt < 1e6:
        continue
    ts = (f["rd"] / READ_TBS + f["wr"] / WRITE_TBS) / 1e9
    print("%-32s %6.0f %9.3f %9.3f %9s %7s %8s %9.3f %8s" % (
        k, f["n"], f["rd"] / 1e9, f["wr"] / 1e9, "%.3f" % (f["alg"] / 1e9) if f["alg"] else "-", "%.2f" % (cnt / f["alg"]) if f["alg"] else "-",
        "%.3f" % f["ms"] if f["ms"] else "-", ts, "%.2f" % (f["ms"] / ts) if f["ms"] else "-"))
print("%-32s %6s %9.3f %9.3f   (whole step %.3f GB; algorithmic, every elementwise pass fused away: %.3f GB; ratio %.2f)" % (
    "all kernels", "", sum(f["rd"] for f in fam.values()) / 1e9, sum(f["wr"] for f in fam.values()) / 1e9, tot / 1e9,
    r["traffic_whole_step"]["algorithmic_bytes_per_step"] / 1e9 if r.get("traffic_whole_step") else float("nan"),
    tot / r["traffic_whole_step"]["algorithmic_bytes_per_step"] if r.get("traffic_whole_step") else float("nan")))
