#!/bin/bash
# conv_h3u_kernel on / off in one box: eval forward (WUNET_H3U=<eval min L>,<train min L>) and the training step with the fused loader from
# different levels; first and last arm of each group the same
cd ${GRAFT_REPO_ROOT:-/root/repo}
for u in ${EVAL_ARMS:-0,0 2048,0 1024,0 512,0 256,0 0,0}; do
  WUNET_H3U=$u python bench.py --mode forward --no-cpu-baseline --no-extras --steps 50 --warmup 10 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('eval  H3U=$u', 'ms %.4f median %.4f' % (j['ms_per_step'], j['ms_per_step_median']), 'fps %.0f' % j['value'])
for t in r['top5'][:6]:
    if 'h3u' in t['kernel']: print('    %-34s %.4f ms/step x%g  %.0f TF' % (t['kernel'], t['ms_per_step'], t['launches_per_step'], t['tflops']))
"
done
for u in ${TRAIN_ARMS:-2048,0 2048,2048 2048,4096 2048,8192 2048,0}; do
  WUNET_H3U=$u python bench.py --no-cpu-baseline --no-extras --no-roofline --steps 100 --warmup 20 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('train H3U=$u', 'ms %.4f' % j['ms_per_step'], 'fps %.0f' % j['value'])
"
done
