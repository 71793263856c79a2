#!/bin/bash
# usage (GPU box): tools/ab.sh "<ENV=VAL ...>" "<ENV=VAL ...>" ...   - bench.py ms/step (3 runs each) per environment setting
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for cfg in "$@"; do
  for i in 1 2 3; do
    env $cfg python bench.py --no-cpu-baseline --no-roofline --steps ${STEPS:-40} --warmup 10 ${BENCH_ARGS} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', round(d['ms_per_step'],3), round(d['ms_per_step_median'],3), round(d['final_loss'],6))"
  done
done
