cd ${GRAFT_REPO_ROOT:-/root/repo}
export WUNET_BENCH_NO_MEDIAN=1
run() { python bench.py --seed $1 --steps $2 --warmup 0 $3 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | python -c "import sys,json; print('%.7f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['final_loss'])"; }
for seed in 1 2 3; do
  steps=100
  a=$(run $seed $steps ""); b=$(run $seed $steps "--gemm fp32"); c=$(WUNET_H3_KTAIL=0 run $seed $steps "")
  d=$(WUNET_LIB_PATH=$PWD/tools/_lib_round5.so run $seed $steps ""); e=$(WUNET_LIB_PATH=$PWD/tools/_lib_round5.so run $seed $steps "--gemm fp32")
  echo "seed $seed steps $steps: round6 split $a fp32 $b split-other-sum-order $c | round5 split $d fp32 $e"
done
