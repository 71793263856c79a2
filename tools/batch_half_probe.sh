#!/bin/bash
# usage (GPU box): tools/batch_half_probe.sh  - does a layer's working set fitting the 256 MiB Infinity Cache pay?  The serial step (one stream) under
# rocprofv3 --kernel-trace --stats at batch 64 and at batch 32: per kernel name, us per step at batch 64 against 2 x us per step at batch 32.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
for b in 64 32; do
  rm -rf /tmp/bh_$b
  WUNET_NO_SIDE_STREAM=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bh_$b -o s -- python $R/bench.py --batch $b --no-cpu-baseline --no-extras --no-roofline --steps 40 --warmup 10 > /tmp/bh_$b.json 2>/dev/null
done
python - <<'PY'
import csv, glob, json
def load(b):
    f = glob.glob("/tmp/bh_%d/**/s_kernel_stats.csv" % b, recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    steps = [int(r["Calls"]) for r in rows if "conv_first" in r["Name"]][0]
    return {r["Name"].split("(")[0]: (int(r["Calls"]) / steps, float(r["TotalDurationNs"]) / steps / 1e3) for r in rows}
a, b = load(64), load(32)
for n in (64, 32):
    j = json.loads(open("/tmp/bh_%d.json" % n).read().strip().splitlines()[-1]); print("batch %d: %.3f ms per step = %.0f frames/s (serial, under the profiler)" % (n, j["ms_per_step"], j["value"]))
print("kernel time per step: batch 64 %.1f us, 2 x batch 32 %.1f us" % (sum(v[1] for v in a.values()), 2 * sum(v[1] for v in b.values())))
for k in sorted(set(a) | set(b), key=lambda k: -a.get(k, (0, 0))[1]):
    x, y = a.get(k, (0, 0)), b.get(k, (0, 0))
    if x[1] < 20: continue
    print("%-64s b64 %4.1f x %7.1f us   2 x b32 %4.1f x %7.1f us   %+7.1f" % (k[:64], x[0], x[1], y[0], 2 * y[1], 2 * y[1] - x[1]))
PY
