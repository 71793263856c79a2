#!/bin/bash
# usage (GPU box, through gpurun): tools/measure_round.sh            -> gpurun_out/final/*
# Every measurement DESIGN.md / profiles/ quote for the round: default bench line, rocprofv3 stats (concurrent and one-stream),
# PMC passes for HBM traffic (FETCH_SIZE and WRITE_SIZE in separate runs, --kernel-trace only).  tools/collect_round.py <tag> then
# copies the summaries into profiles/ and rebuilds the same profiles/pmc_traffic.json (stamped with the source hash) here.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/final; rm -rf $O; mkdir -p $O
cd $R
# PMC passes first (collect_round.py --pmc-only stamps profiles/pmc_traffic.json on the box): the bench lines below then carry
# roofline.traffic measured on THIS box and THIS build
cd /tmp; export TMPDIR=/tmp
WUNET_BENCH_NO_MEDIAN=1 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-extras > /dev/null 2>&1
WUNET_BENCH_NO_MEDIAN=1 timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-extras > /dev/null 2>&1
python - <<PY
import collections, csv, glob, json
def agg(pattern, ctr):
    a = collections.defaultdict(lambda: [0, 0.0])
    for path in glob.glob(pattern, recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] != ctr: continue
            k = r["Kernel_Name"].replace("void ", "").split("(")[0]
            a[k][0] += 1; a[k][1] += float(r["Counter_Value"])
    return {k: v for k, v in a.items()}
json.dump({"fetch": agg("$O/pmc_fetch/**/*counter_collection.csv", "FETCH_SIZE"), "write": agg("$O/pmc_write/**/*counter_collection.csv", "WRITE_SIZE")},
          open("$O/pmc_raw.json", "w"))
PY
rm -rf $O/pmc_fetch $O/pmc_write
cd $R; python tools/collect_round.py ${TAG:-r2} --pmc-only
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.json; echo
timeout 200 python bench.py --gemm bf16 --no-cpu-baseline --no-extras > $O/bench_bf16.json 2>/dev/null
timeout 300 python bench.py --gemm bf16 --layers 16 --frame 65536 --batch 32 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_deep16_bf16.json 2>/dev/null
timeout 300 python bench.py --layers 16 --frame 65536 --batch 32 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_deep16_split.json 2>/dev/null
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/conc -o conc -- python $R/bench.py --no-cpu-baseline --no-extras > $O/bench_under_rocprof.json 2>/dev/null
WUNET_NO_SIDE_STREAM=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial -o serial -- python $R/bench.py --no-cpu-baseline --no-extras > $O/serial_bench.json 2>/dev/null
WUNET_NO_SIDE_STREAM=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fwd -o fwd -- python $R/bench.py --mode forward --no-cpu-baseline --no-roofline > $O/forward_bench.json 2>/dev/null
# keep what collect_round.py needs, drop the bulky traces
for d in conc serial fwd; do f=$(find $O/$d -name "${d}_kernel_stats.csv" | head -1); cp $f $O/${d}_kernel_stats.csv; g=$(find $O/$d -name "${d}_kernel_trace.csv" | head -1); [ "$d" = serial ] && cp $g $O/serial_kernel_trace.csv; rm -rf $O/$d; done
ls -la $O
