export WUNET_H3=1
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "fp16_split or full_size" 2>&1 | tail -2
timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c90-200
timeout 200 python bench.py --mode forward --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c90-200
