#!/bin/bash
# usage (GPU box): BASE=tools/_lib_base.so tools/serial_ab.sh [filter]  - per-kernel times of the serial step (one stream, rocprofv3 --kernel-trace --stats) of a saved
# library and of the current one on ONE box: which kernels a change moved.  Prints us per step per kernel name (calls / steps x average).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
PKG=wave-u-net-for-speech-enhancement_amd
for lib in base new; do
  if [ $lib = base ]; then export WUNET_LIB_PATH=$R/${BASE:-tools/_lib_base.so}; else export WUNET_LIB_PATH=$R/$PKG/csrc/libwunet_hip.so; fi
  rm -rf /tmp/sab_$lib
  WUNET_NO_SIDE_STREAM=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sab_$lib -o s -- python $R/bench.py --no-cpu-baseline --no-extras --no-roofline --steps 40 --warmup 10 > /dev/null 2>&1
done
python - "$1" <<'PY'
import csv, glob, sys
flt = sys.argv[1] if len(sys.argv) > 1 else ""
def load(lib):
    f = glob.glob("/tmp/sab_%s/**/s_kernel_stats.csv" % lib, recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    steps = [int(r["Calls"]) for r in rows if "conv_first" in r["Name"]][0]
    return {r["Name"].split("(")[0]: (int(r["Calls"]) / steps, float(r["TotalDurationNs"]) / steps / 1e3) for r in rows}, steps
a, sa = load("base"); b, sb = load("new")
ta = sum(v[1] for v in a.values()); tb = sum(v[1] for v in b.values())
print("kernel time per step: base %.1f us, new %.1f us (%d / %d steps under the profiler)" % (ta, tb, sa, sb))
for k in sorted(set(a) | set(b), key=lambda k: -max(a.get(k, (0, 0))[1], b.get(k, (0, 0))[1])):
    if flt and flt not in k: continue
    x, y = a.get(k, (0, 0)), b.get(k, (0, 0))
    if max(x[1], y[1]) < 3: continue
    print("%-70s base %5.1f x %7.1f us   new %5.1f x %7.1f us   %+7.1f" % (k[:70], x[0], x[1], y[0], y[1], y[1] - x[1]))
PY
