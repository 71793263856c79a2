#!/usr/bin/env python
"""usage: tools/timeline.py <kernel_trace.csv> [out.txt]   - the kernels of the last complete training step of a rocprofv3 --kernel-trace
run of bench.py in start order: start offset, duration, idle gap on the chip before it (no kernel of any queue running), queue, name;
then the totals per kernel family, the busy time, the gaps and the step span."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("adam_kernel")]
# two adam launches per step: the step is what lies between the end of the previous step's second launch and this step's second launch
ends = adam[1::2]
a, b = ends[-2] + 1, ends[-1] + 1
step = rows[a:b]
t0 = int(step[0]["Start_Timestamp"])
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
busy_end = t0
gaps = 0.0
fam = {}
qs = {}
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = max(0, s - busy_end) / 1e3
    gaps += gap
    busy_end = max(busy_end, e)
    q = qs.setdefault(r.get("Queue_Id", "?"), len(qs))
    nm = r["Kernel_Name"].replace("void ", "").split("(")[0]
    fam[nm.split("<")[0]] = fam.get(nm.split("<")[0], 0.0) + (e - s) / 1e3
    print("%9.1f us  %7.1f us  gap %5.1f  q%d  grid %7s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, q, r.get("Grid_Size_X", "?"), nm[:70]), file=out)
span = (busy_end - t0) / 1e3
print("step span %.1f us, chip idle between kernels %.1f us, %d kernels" % (span, gaps, len(step)), file=out)
for k, v in sorted(fam.items(), key=lambda x: -x[1]):
    print("  %-40s %9.1f us" % (k, v), file=out)
