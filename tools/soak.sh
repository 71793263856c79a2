cd ${GRAFT_REPO_ROOT:-/root/repo}
export WUNET_BENCH_NO_MEDIAN=1
for seed in 1 2 3; do for steps in 100 300 600; do
  a=$(python bench.py --seed $seed --steps $steps --warmup 0 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | python -c "import sys,json; print('%.7f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['final_loss'])")
  b=$(python bench.py --seed $seed --steps $steps --warmup 0 --gemm fp32 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | python -c "import sys,json; print('%.7f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['final_loss'])")
  python -c "print('  %4d %5d   %s        %s     %+.2f %%' % ($seed, $steps, '$a', '$b', ($a/$b-1)*100))"
done; done
