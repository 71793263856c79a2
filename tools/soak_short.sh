#!/bin/bash
# usage (GPU box): tools/soak_short.sh  - the loss after 5 .. 80 steps on one synthetic batch (bench.py --seed 1 --warmup 0): default split GEMMs, exact fp32,
# the split arithmetic with another order of the K sums (WUNET_H3_KTAIL=0), and the previous round's library - where do the trajectories part?
cd ${GRAFT_REPO_ROOT:-/root/repo}
export WUNET_BENCH_NO_MEDIAN=1
run() { python bench.py --seed 1 --steps $1 --warmup 0 $2 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | python -c "import sys,json; print('%.7f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['final_loss'])"; }
for steps in 5 10 20 40 80; do
  a=$(run $steps ""); b=$(run $steps "--gemm fp32"); c=$(WUNET_H3_KTAIL=0 run $steps ""); d=$(WUNET_LIB_PATH=$PWD/tools/_lib_round5.so run $steps "")
  python -c "print('steps %3d: split %s  fp32 %s (%+.3f %%)  split, other sum order %s (%+.3f %%)  round 5 split %s (%+.3f %%)' % ($steps, '$a', '$b', ($b/$a-1)*100, '$c', ($c/$a-1)*100, '$d', ($d/$a-1)*100))"
done
