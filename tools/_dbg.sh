for tp in 0 256 128 0; do echo "TP $tp"; WUNET_H3W_TP=$tp timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c90-200; done
WUNET_NO_SIDE_STREAM=1 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
j=json.loads(sys.stdin.read()); r=j['roofline']
print(j['ms_per_step'], r['mfma_kernels_ms_per_step'])"
