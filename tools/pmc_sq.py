#!/usr/bin/env python
"""usage: tools/pmc_sq.py <dir of the SQ pass> <dir of the GRBM pass>   (rocprofv3 --kernel-trace --pmc ... --output-format csv, run by
tools/measure_round.sh with WUNET_NO_SIDE_STREAM=1 so no two kernels overlap).  Prints, per kernel: launches, average duration, the SQ
wave-cycle split (issuing / parked at s_waitcnt or a barrier / issue-stalled), LDS bank conflicts, the matrix-pipe busy fraction
(SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES and per-SIMD share of the kernel's cycles) and the EFFECTIVE CLOCK the kernel ran at
(GRBM_GUI_ACTIVE / duration - the guide's DVFS note: the chip clocks to its power budget)."""
import collections
import csv
import glob
import sys


def load(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    dur = collections.defaultdict(float)
    seen = set()
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"].replace("void ", "").split("(")[0]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (r.get("Dispatch_Id"), k)
            if key not in seen:
                seen.add(key)
                cnt[k] += 1
                if r.get("Start_Timestamp") and r.get("End_Timestamp"):
                    dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    if not any(dur.values()):       # older layouts: durations only in the kernel trace
        for path in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
            for r in csv.DictReader(open(path)):
                k = r["Kernel_Name"].replace("void ", "").split("(")[0]
                dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return acc, cnt, dur


sq, n_sq, d_sq = load(sys.argv[1])
gr, n_gr, d_gr = load(sys.argv[2])
print("# pass 1: rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES "
      "SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F16")
print("# pass 2: rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU")
print("# both of: WUNET_NO_SIDE_STREAM=1 python bench.py --steps 2 --warmup 1 (one stream: kernels do not overlap); counters summed over all launches of a")
print("# kernel name.  active / parked / stalled = SQ_ACTIVE_INST_ANY / SQ_WAIT_ANY / SQ_WAIT_INST_ANY over SQ_WAVE_CYCLES (quad-cycles of resident waves:")
print("# issuing / at s_waitcnt or a barrier / issue-stalled on the matrix pipe or a dependency).  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES")
print("# (the guide: MFMA_BUSY counts cycles per SIMD-instance summed, BUSY_CYCLES per SE/XCD instance: read it as a RELATIVE measure across kernels);")
print("# mfma/ideal = 16 cycles x SQ_INSTS_VALU_MFMA_MOPS_F16-derived MFMA count / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE/xcds): the share of the chip's")
print("# matrix-pipe cycles that issued an f16 MFMA.  clock = GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / duration.")
print("# lds_busy = SQ_LDS_IDX_ACTIVE / (256 CUs x GRBM_GUI_ACTIVE / 8): share of the chip's LDS-array cycles in use; valu/mfma = SQ_INSTS_VALU (the MFMAs included)")
print("# per f16 MFMA; lds/mfma = SQ_INSTS_LDS per f16 MFMA (wave instructions, pass 2 scaled to pass 1's launch count)")
hdr = "%-44s %5s %9s %7s %7s %7s %8s %9s %10s %9s %8s %9s %8s" % ("kernel", "n", "avg_us", "active", "parked", "stalled", "lds_conf", "mfma_busy", "mfma/ideal", "clock_GHz", "lds_busy", "valu/mfma", "lds/mfma")
print(hdr)
rows = sorted(sq, key=lambda k: -d_sq.get(k, 0.0))
for k in rows:
    c = sq[k]
    wc = c.get("SQ_WAVE_CYCLES", 0.0)
    if wc <= 0:
        continue
    g = gr.get(k, {})
    gui = g.get("GRBM_GUI_ACTIVE", 0.0)
    dur_g = d_gr.get(k, 0.0)
    clock = gui / 8.0 / dur_g if dur_g > 0 else float("nan")           # cycles per ns = GHz
    mops = c.get("SQ_INSTS_VALU_MFMA_MOPS_F16", 0.0)
    # MOPS counts 512 flops per unit (gfx94x convention): a 16x16x32 f16 MFMA = 16384 flop = 32 units, 16 pipe cycles each
    mfma_cycles = mops / 32.0 * 16.0
    gui_sq = gui * (d_sq.get(k, 0.0) / dur_g) if dur_g > 0 else 0.0      # the SQ pass's own duration scales the cycle budget
    ideal = 4.0 * 256.0 * gui_sq / 8.0
    n_mfma = mops / 32.0 * (n_gr.get(k, 0) / max(n_sq[k], 1))          # wave-level MFMA instructions (32 MOPS units each), for pass 2's launches
    lds_busy = g.get("SQ_LDS_IDX_ACTIVE", 0.0) / (256.0 * gui / 8.0) if gui > 0 else float("nan")
    print("%-44s %5d %9.1f %7.3f %7.3f %7.3f %8.4f %9.3f %10.3f %9.2f %8.3f %9s %8s" % (
        k[:44], n_sq[k], d_sq[k] / max(n_sq[k], 1) / 1e3, c.get("SQ_ACTIVE_INST_ANY", 0) / wc, c.get("SQ_WAIT_ANY", 0) / wc, c.get("SQ_WAIT_INST_ANY", 0) / wc,
        c.get("SQ_LDS_BANK_CONFLICT", 0) / wc, c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(c.get("SQ_BUSY_CYCLES", 0), 1.0),
        mfma_cycles / ideal if ideal > 0 else float("nan"), clock, lds_busy,
        "%.2f" % (g.get("SQ_INSTS_VALU", 0) / n_mfma) if n_mfma > 0 else "-", "%.2f" % (g.get("SQ_INSTS_LDS", 0) / n_mfma) if n_mfma > 0 else "-"))
print("# raw sums of the GRBM pass (per kernel): GRBM_GUI_ACTIVE, GRBM_COUNT, duration_ns")
for k in rows[:12]:
    g = gr.get(k, {})
    print("#   %-44s %.4e %.4e %.4e" % (k[:44], g.get("GRBM_GUI_ACTIVE", 0), g.get("GRBM_COUNT", 0), d_gr.get(k, 0)))
