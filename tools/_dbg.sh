timeout 900 python -m pytest tests -q -x -m gpu 2>&1 | grep -E "passed|failed|error|Error" | tail -3
for i in 1 2; do timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c90-200; done
timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --mode forward 2>&1 | tail -1 | cut -c150-260
