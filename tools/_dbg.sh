export WUNET_H3=1
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "fp16_split or full_size" 2>&1 | tail -2
for o in 23; do WUNET_H3_ORDER=$o timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c90-200; done
WUNET_H3_ORDER=23 WUNET_NO_SIDE_STREAM=1 timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
j=json.loads(sys.stdin.read()); r=j['roofline']
print(j['ms_per_step'], [(t['kernel'], round(t['ms_per_step'],3), round(t['tflops'])) for t in r['top5']])"
