#!/bin/bash
# usage (GPU box): tools/two_stream_timeline.sh <out.txt>  - rocprofv3 --kernel-trace over the default bench (graph replays, both streams), the last complete
# step written out by tools/timeline.py: which kernels run side by side, where the main chain waits
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/tst
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tst -o t -- python $R/bench.py --no-cpu-baseline --no-extras --no-roofline --steps 20 --warmup 5 $BENCH_ARGS > /dev/null 2>&1
f=$(find /tmp/tst -name 't_kernel_trace.csv' | head -1)
python $R/tools/timeline.py $f $1
