R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; rm -rf $O; mkdir -p $O
cd $R
timeout 300 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json
timeout 200 python bench.py --gemm fp32 --no-cpu-baseline > $O/bench_fp32.json 2>/dev/null
timeout 200 python bench.py --mode forward --no-cpu-baseline > $O/bench_forward.json 2>/dev/null
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/conc -o conc -- python $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2>/dev/null
WUNET_NO_SIDE_STREAM=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial -o serial -- python $R/bench.py --no-cpu-baseline > $O/serial_bench.json 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2>&1
ls $O | head
