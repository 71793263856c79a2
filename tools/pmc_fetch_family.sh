#!/bin/bash
# usage (GPU box): tools/pmc_fetch_family.sh <out.txt> [ENV=VALUE ...]  - HBM-side bytes (FETCH_SIZE x 2 on gfx950, WRITE_SIZE; separate passes) of one
# training step per kernel family, under the given environment: a quick per-family traffic check of ONE change (the round's full table is
# tools/measure_round.sh's)
OUT=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}; T=$(mktemp -d /tmp/pmcf.XXXX)
cd /tmp; export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  env "$@" WUNET_BENCH_NO_MEDIAN=1 timeout 400 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $T/$ctr -o p -- \
      python $R/bench.py --steps 2 --warmup 1 --graph off --no-cpu-baseline --no-roofline --no-extras > /dev/null 2>&1
done
python - <<PY > $OUT
import collections, csv, glob
def agg(pattern, ctr):
    a = collections.defaultdict(float); n = collections.defaultdict(int)
    for path in glob.glob(pattern, recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] != ctr: continue
            k = r["Kernel_Name"].replace("void ", "").split("<")[0].split("(")[0]
            a[k] += float(r["Counter_Value"]); n[k] += 1
    return a, n
f, nf = agg("$T/FETCH_SIZE/**/*counter_collection.csv", "FETCH_SIZE")
w, nw = agg("$T/WRITE_SIZE/**/*counter_collection.csv", "WRITE_SIZE")
steps = 3.0        # 1 warm-up + 2: launches / 3 = launches per step
print("env: $*   (GB per step; FETCH_SIZE / WRITE_SIZE are KiB, FETCH_SIZE doubled per the gfx950 note - as tools/collect_round.py)")
tr = tw = 0.0
for k in sorted(set(f) | set(w), key=lambda k: -(2 * f.get(k, 0) + w.get(k, 0))):
    rd, wr = 2.0 * f.get(k, 0.0) * 1024.0 / steps / 1e9, w.get(k, 0.0) * 1024.0 / steps / 1e9
    tr += rd; tw += wr
    if rd + wr > 0.005:
        print(f"{k:42s} launches/step {nf.get(k,0)/steps:6.1f}   read {rd:7.3f} GB   write {wr:7.3f} GB")
print(f"{'all kernels':42s}                        read {tr:7.3f} GB   write {tw:7.3f} GB   total {tr + tw:7.3f} GB")
PY
rm -rf $T
