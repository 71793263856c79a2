#!/bin/bash
# eval forward (BASELINE configs[1], batch 64) with conv_h3u_kernel from different minimum levels: WUNET_H3U=<eval min L>,<train min L>; 0,0 = prep_h3_kernel + conv_h3d_kernel everywhere (same box, first and last arm the same)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for u in 0,0 8192,0 4096,0 2048,0 1024,0 256,0 0,0; do
  WUNET_H3U=$u python bench.py --mode forward --no-cpu-baseline --no-extras --steps 50 --warmup 10 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('H3U=$u', 'eval ms %.4f median %.4f' % (j['ms_per_step'], j['ms_per_step_median']), 'fps %.0f' % j['value'])
for t in r['top5'][:6]: print('    %-34s %.4f ms/step x%g  %.0f TF' % (t['kernel'], t['ms_per_step'], t['launches_per_step'], t['tflops']))
for m in r['memory_bound_kernels'][:3]: print('    %-34s %.4f ms/step x%g' % (m['kernel'], m['ms_per_step'], m['launches_per_step']))
"
done
