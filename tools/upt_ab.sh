#!/bin/bash
# usage (GPU box): tools/upt_ab.sh  - the training step with / without the transposed upsample in the data-gradient epilogue (WUNET_UPT, read when
# a context is planned), interleaved on one box
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2 3; do for v in 1 0; do
  export WUNET_UPT=$v
  python bench.py --no-cpu-baseline --no-extras --no-roofline --steps 100 --warmup 10 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('WUNET_UPT=$v', 'train ms %.4f median %.4f loss %.7f' % (j['ms_per_step'], j['ms_per_step_median'], j['final_loss']))
"
done; done
unset WUNET_UPT
