#!/bin/bash
# usage: tools/env_ab.sh VAR [VALUE]: the training step (bench.py, 100 steps) with VAR unset and VAR=VALUE (default 1), interleaved three times on one box
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2 3; do for v in "" ${2:-1}; do
  if [ -z "$v" ]; then unset $1; else export $1=$v; fi      # (the baseline arm must UNSET the variable: the library tests getenv() != nullptr,
                                                            #  and `env VAR= cmd` sets an empty one - that mistake hid a 1.7 ms regression in round 5)
  python bench.py --no-cpu-baseline --no-extras --no-roofline --steps 100 --warmup 10 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1=$v', 'train ms %.4f median %.4f loss %.7f' % (j['ms_per_step'], j['ms_per_step_median'], j['final_loss']))
"
done; done
