#!/bin/bash
# usage (GPU box): BASE=tools/_lib_base.so tools/lib_ab.sh   - a saved build of the library against the current one on ONE box, alternating:
# training step (100 graph replays) and eval forward (50 forwards), both through the ctypes binding
cd ${GRAFT_REPO_ROOT:-/root/repo}
PKG=wave-u-net-for-speech-enhancement_amd
for rep in 1 2 3; do for lib in base new; do
  if [ $lib = base ]; then export WUNET_LIB_PATH=$PWD/${BASE:-tools/_lib_base.so}; else export WUNET_LIB_PATH=$PWD/$PKG/csrc/libwunet_hip.so; fi
  python bench.py --no-cpu-baseline --no-extras --no-roofline --steps 100 --warmup 10 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib training step ms %.4f (median %.4f) = %.0f frames/s  loss %.7f' % (j['ms_per_step'], j['ms_per_step_median'], j['value'], j['final_loss']))
"
  python bench.py --mode forward --no-cpu-baseline --no-extras --no-roofline --steps 50 --warmup 10 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib eval forward  ms %.4f (median %.4f) = %.0f frames/s' % (j['ms_per_step'], j['ms_per_step_median'], j['value']))
"
done; done
