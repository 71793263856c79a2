#!/bin/bash
# usage (on the GPU box, via gpurun): tools/prof.sh <tag> <bench.py arguments ...>
# rocprofv3 --kernel-trace --stats of one bench.py run; prints the top kernels and leaves
# gpurun_out/<tag>_kernel_stats.csv (copy what should be judged into profiles/).
R=${GRAFT_REPO_ROOT:-/root/repo}; tag=$1; shift
O=$R/gpurun_out; mkdir -p $O; rm -rf /tmp/prof_$tag
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o p -- python $R/bench.py --no-cpu-baseline "$@" > $O/${tag}_bench.json 2> $O/${tag}_err.log
f=$(find /tmp/prof_$tag -name 'p_kernel_stats.csv' | head -1)
cp $f $O/${tag}_kernel_stats.csv
python - <<PY
import csv
rows = list(csv.DictReader(open("$O/${tag}_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms: %.3f" % (tot / 1e6))
for r in rows[:${TOPN:-30}]:
    print("%-86s %6s %10.1f us avg %8.3f ms tot %5.1f%%" % (r["Name"][:86], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, float(r["Percentage"])))
PY
