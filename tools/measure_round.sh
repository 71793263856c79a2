#!/bin/bash
# usage (GPU box, through gpurun): TAG=r4 tools/measure_round.sh            -> gpurun_out/final/*
# Every measurement DESIGN.md / profiles/ quote for the round: PMC passes for HBM traffic (FETCH_SIZE and WRITE_SIZE in separate runs,
# --kernel-trace only) of the headline workload and of the two extras that carry their own roofline (exact fp32, configs[4] bf16),
# the SQ / GRBM counter passes (matrix-pipe utilisation, effective clock), the default bench line, rocprofv3 stats (concurrent and
# one-stream).  tools/collect_round.py <tag> then copies the summaries into profiles/ and rebuilds profiles/pmc_traffic.json
# (stamped with the source hash) here.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/final; rm -rf $O; mkdir -p $O
# only from a clean tree: tools/run_measure_round.sh (build container) refuses a dirty csrc/ and leaves the commit + source hash here
if [ ! -f $R/tools/_git_state ] || [ "$(cut -d' ' -f3 $R/tools/_git_state)" != "$(cd $R && python -c 'import bench; print(bench.source_hash())')" ]; then
    echo "refused: tools/_git_state missing or not these sources - start the measurement with tools/run_measure_round.sh" >&2; exit 2
fi
cp $R/tools/_git_state $O/git_state.txt
if [ -z "$PMC_ONLY" ]; then      # the -m gpu suite and the smoke hook on the sources the set is taken from (the summary lines go to profiles/)
    ( cd $R && timeout 900 python -m pytest tests -m gpu -q > $O/gpu_tests_full.log 2>&1; grep -E "^[0-9]+ (passed|failed)|passed|failed|error" $O/gpu_tests_full.log | tail -5 > $O/gpu_tests.txt
      python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke >> $O/gpu_tests.txt; cat $O/gpu_tests.txt )
fi
cd /tmp; export TMPDIR=/tmp
pmc_pass() {   # <name> <bench args...>: FETCH_SIZE and WRITE_SIZE passes of 3 steps (1 warm-up + 2) -> $O/pmc_raw_<name>.json
    local name=$1; shift
    for ctr in FETCH_SIZE WRITE_SIZE; do
        WUNET_BENCH_NO_MEDIAN=1 timeout 400 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $O/pmc_${name}_$ctr -o p -- \
            python $R/bench.py --steps 2 --warmup 1 --graph off --no-cpu-baseline --no-roofline --no-extras "$@" > /dev/null 2>&1
    done
    python - <<PY
import collections, csv, glob, json
def agg(pattern, ctr):
    a = collections.defaultdict(lambda: [0, 0.0])
    for path in glob.glob(pattern, recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] != ctr: continue
            k = r["Kernel_Name"].replace("void ", "").split("(")[0]
            a[k][0] += 1; a[k][1] += float(r["Counter_Value"])
    return {k: v for k, v in a.items()}
json.dump({"fetch": agg("$O/pmc_${name}_FETCH_SIZE/**/*counter_collection.csv", "FETCH_SIZE"),
           "write": agg("$O/pmc_${name}_WRITE_SIZE/**/*counter_collection.csv", "WRITE_SIZE")}, open("$O/pmc_raw_${name}.json", "w"))
PY
    rm -rf $O/pmc_${name}_FETCH_SIZE $O/pmc_${name}_WRITE_SIZE
}
# PMC passes first (collect_round.py --pmc-only stamps profiles/pmc_traffic.json on the box): the bench line below then carries
# roofline.traffic measured on THIS box and THIS build
pmc_pass train
pmc_pass gemm_fp32 --gemm fp32
pmc_pass deep16_bf16 --gemm bf16 --layers 16 --frame 65536 --batch 32
pmc_pass eval_forward --mode forward
cd $R; python tools/collect_round.py $TAG --pmc-only
[ -n "$PMC_ONLY" ] && exit 0
# SQ / GRBM counters of the serial step (one stream: kernels do not overlap, so counters and durations belong to one kernel)
cd /tmp
WUNET_NO_SIDE_STREAM=1 WUNET_BENCH_NO_MEDIAN=1 timeout 400 rocprofv3 --kernel-trace \
    --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F16 \
    --output-format csv -d $O/pmc_sq -o s -- python $R/bench.py --steps 2 --warmup 1 --graph off --no-cpu-baseline --no-roofline --no-extras > /dev/null 2>&1
WUNET_NO_SIDE_STREAM=1 WUNET_BENCH_NO_MEDIAN=1 timeout 400 rocprofv3 --kernel-trace \
    --pmc GRBM_GUI_ACTIVE GRBM_COUNT SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU \
    --output-format csv -d $O/pmc_grbm -o g -- python $R/bench.py --steps 2 --warmup 1 --graph off --no-cpu-baseline --no-roofline --no-extras > /dev/null 2>&1
python $R/tools/pmc_sq.py $O/pmc_sq $O/pmc_grbm > $O/pmc_sq.txt 2> $O/pmc_sq.err
rm -rf $O/pmc_sq $O/pmc_grbm
cd $R
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json; echo
WUNET_BENCH_ALL=1 timeout 200 python bench.py --no-cpu-baseline --no-extras > $O/bench_all_kernels.json 2>/dev/null
python tools/traffic_table.py $O/bench_all_kernels.json > $O/traffic_by_family.txt 2>&1
timeout 200 python bench.py --gemm bf16 --no-cpu-baseline --no-extras > $O/bench_bf16.json 2>/dev/null
# the collectives' cost inside the step on one GPU (world size 1 through the library's RCCL entry), eager and as one captured graph
timeout 200 python bench.py --native-rccl --graph off --no-cpu-baseline --no-extras --no-roofline > $O/bench_native_rccl_eager.json 2>/dev/null
timeout 200 python bench.py --native-rccl --graph on --no-cpu-baseline --no-extras --no-roofline > $O/bench_native_rccl_graph.json 2>/dev/null
timeout 200 python bench.py --graph on --no-cpu-baseline --no-extras --no-roofline --steps 100 > $O/bench_graph.json 2>/dev/null
timeout 200 python bench.py --graph off --no-cpu-baseline --no-extras --no-roofline --steps 100 > $O/bench_eager.json 2>/dev/null
# (the training soak - three seeds, 100 / 300 / 600 steps, split against exact fp32 and the summation-order control - is profiles/r3_training_soak.txt,
#  measured by its own calls: bench.py --seed S --steps N --warmup 0 [--gemm fp32] with WUNET_BENCH_NO_MEDIAN=1)
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/conc -o conc -- python $R/bench.py --no-cpu-baseline --no-extras > $O/bench_under_rocprof.json 2>/dev/null
WUNET_NO_SIDE_STREAM=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial -o serial -- python $R/bench.py --no-cpu-baseline --no-extras > $O/serial_bench.json 2>/dev/null
WUNET_NO_SIDE_STREAM=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fwd -o fwd -- python $R/bench.py --mode forward --no-cpu-baseline --no-roofline > $O/forward_bench.json 2>/dev/null
# keep what collect_round.py needs, drop the bulky traces
for d in conc serial fwd; do f=$(find $O/$d -name "${d}_kernel_stats.csv" | head -1); cp $f $O/${d}_kernel_stats.csv; g=$(find $O/$d -name "${d}_kernel_trace.csv" | head -1); [ "$d" = serial ] && cp $g $O/serial_kernel_trace.csv; rm -rf $O/$d; done
python $R/tools/timeline.py $O/serial_kernel_trace.csv > $O/step_timeline.txt 2>/dev/null
# host side of a step (back to back / synchronised every step), the conv kernel with parts compiled out (tools/conv_ablation.sh build first, in the
# container), the LDS-DMA issue microbench (hipcc --offload-arch=gfx950 -O3 -o tools/microbench/_dma_issue tools/microbench/dma_issue.hip)
cd $R
{ echo "tools/host_phases.py on one box, with the C++ marshalling extension (torch_ext/wunet_torch.cpp, the default when built) and with WUNET_NO_TORCH_EXT=1 (ctypes)"; timeout 120 python tools/host_phases.py 2>/dev/null | grep -v amdgpu; WUNET_NO_TORCH_EXT=1 timeout 120 python tools/host_phases.py 2>/dev/null | grep -v amdgpu | sed 's/^/ctypes: /'; } > $O/host_phases.txt
timeout 300 python tools/next_rows_bench.py 2>/dev/null | grep -v amdgpu > $O/next_rows_bench.txt       # f1 / f3 / f4 of SURVEY.md section 8(f)
# round 5: conv_h3u_kernel (eval decoder levels) - threshold sweep, ablation (tools/h3u_ablation.sh build first, in the container), stage timeline
# (tools/build_h3u_trace.sh first); the previous round's library against this one on this box (tools/_lib_round4.so)
if [ -n "$H3U_SET" ]; then       # (round 5's own studies of conv_h3u_kernel / EVOP: H3U_SET=1 repeats them)
{ echo "conv_h3u_kernel from different minimum levels, WUNET_H3U=<eval min L>,<train min L> (0: prep_h3_kernel + conv_h3d_kernel); eval forward and training step, batch 64, one box, first and last arm of each group the same (tools/h3u_ab.sh)"; EVAL_ARMS="0,0 2048,0 1024,0 512,0 256,0 0,0" TRAIN_ARMS="512,0 512,4096 512,2048 512,1024 512,512 512,0" timeout 500 bash tools/h3u_ab.sh; } > $O/h3u_sweep.txt 2>&1
ls tools/_lib_u64.so > /dev/null 2>&1 && H3U=8192,0 timeout 300 bash tools/h3u_ablation.sh run > /dev/null 2>&1 && cp gpurun_out/h3u_ablation.txt $O/h3u_ablation.txt
[ -f tools/_lib_trace.so ] && timeout 120 python tools/h3u_trace.py 2>/dev/null | grep -v amdgpu > $O/h3u_stage_timeline.txt
fi
# the previous round's library (tools/_lib_round5.so) against this round's on this box; this round's switchable changes, each against its off arm
[ -f tools/_lib_${PREV:-round5}.so ] && timeout 500 bash tools/round_vs_round.sh > $O/prev_vs_cur_same_box.txt 2>&1
timeout 300 bash tools/upt_ab.sh > $O/upt_ab.txt 2>&1
timeout 300 bash tools/env_ab.sh WUNET_WGRAD_XCD 0 > $O/wgrad_xcd_ab.txt 2>&1
[ -n "$H3U_SET" ] && { for rep in 1 2; do for v in "" 1; do if [ -z "$v" ]; then unset WUNET_NO_EVOP; else export WUNET_NO_EVOP=1; fi
    python bench.py --mode forward --no-cpu-baseline --no-extras --no-roofline --steps 50 --warmup 10 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('WUNET_NO_EVOP=$v eval forward ms %.4f median %.4f' % (j['ms_per_step'], j['ms_per_step_median']))"; done; done; unset WUNET_NO_EVOP; } > $O/evop_ab.txt 2>&1
ls tools/_lib_abl2.so > /dev/null 2>&1 && timeout 300 bash tools/conv_ablation.sh run > /dev/null 2>&1 && cp gpurun_out/conv_ablation.txt $O/conv_ablation.txt
[ -x tools/microbench/_dma_issue ] && timeout 60 tools/microbench/_dma_issue > $O/dma_issue_microbench.txt
ls -la $O
