export WUNET_NO_H3A=1
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_conc -o conc -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > /dev/null 2>&1
ls $GRAFT_REPO_ROOT/gpurun_out/prof_conc | head -3
