#!/bin/bash
# usage (GPU box): ARMS="0 2048 1024 512 256" tools/bsum_ab.sh   - the training step with the BatchNorm-backward sums in the data-gradient epilogue
# (conv_h3d_kernel<.., BSUM>) from different minimum consumer lengths (WUNET_BSUM, read when a context is planned; 0 = pass_a_kernel everywhere),
# interleaved on one box
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2 ${REPS:-}; do for v in ${ARMS:-0 1024}; do
  export WUNET_BSUM=$v
  python bench.py --no-cpu-baseline --no-extras --no-roofline --steps 100 --warmup 10 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('WUNET_BSUM=$v', 'train ms %.4f median %.4f loss %.7f' % (j['ms_per_step'], j['ms_per_step_median'], j['final_loss']))
"
done; done
unset WUNET_BSUM
