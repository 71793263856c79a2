timeout 900 python -m pytest tests -q -x -m gpu -s 2>&1 | grep -E "passed|failed|error|gemm path" | tail -8
timeout 200 python bench.py --steps 30 --warmup 5 2>&1 | tail -1 > gpurun_out/bench_split.json; cut -c1-400 gpurun_out/bench_split.json
timeout 200 python bench.py --steps 30 --warmup 5 --gemm fp32 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_fp32.json; cut -c90-220 gpurun_out/bench_fp32.json
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
