#!/bin/bash
# usage (GPU box): tools/round_vs_round.sh        - the previous round's library (tools/_lib_round<N-1>.so: built from its final commit by
# `git worktree add /tmp/wt 0a4c048 && make -C /tmp/wt/<pkg>/csrc`) against this round's on ONE box, alternating: training step (100 graph
# replays) and eval forward (50 forwards), both libraries through the ctypes binding.  The boxes of the pool differ by a few per cent:
# only this comparison says what the CODE changed.
cd ${GRAFT_REPO_ROOT:-/root/repo}
PKG=wave-u-net-for-speech-enhancement_amd
PREV=${PREV:-round5}; CUR=${CUR:-round6}
for rep in 1 2 3; do for lib in $PREV $CUR; do
  if [ $lib = $PREV ]; then export WUNET_LIB_PATH=$PWD/tools/_lib_$PREV.so; else export WUNET_LIB_PATH=$PWD/$PKG/csrc/libwunet_hip.so; fi
  python bench.py --no-cpu-baseline --no-extras --no-roofline --steps 100 --warmup 10 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib training step ms %.4f (median %.4f) = %.0f frames/s  loss %.7f' % (j['ms_per_step'], j['ms_per_step_median'], j['value'], j['final_loss']))
"
  python bench.py --mode forward --no-cpu-baseline --no-extras --no-roofline --steps 50 --warmup 10 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib eval forward  ms %.4f (median %.4f) = %.0f frames/s' % (j['ms_per_step'], j['ms_per_step_median'], j['value']))
"
done; done
