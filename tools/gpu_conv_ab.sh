#!/bin/bash
# usage (GPU box, through gpurun): tools/gpu_conv_ab.sh   - parity of the DMA-staged conv kernel, per-layer A/B, whole-step A/B
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "split_ops_at_baseline or golden or gemm_paths or backward_range or full_size" 2>&1 | tail -15 > $O/ab_pytest.txt; tail -5 $O/ab_pytest.txt
timeout 400 python tools/conv_bench.py --trace --layers enc1,enc3,dec8,dec11 "WUNET_H3_XDMA=0" "WUNET_H3_XDMA=1" > $O/convbench_trace.txt 2>&1
timeout 400 python tools/conv_bench.py "WUNET_H3_XDMA=0" "WUNET_H3_XDMA=1" "WUNET_H3_XDMA=1 WUNET_H3_ORDER=432 WUNET_H3D_ORDER=432" "WUNET_H3_XDMA=1 WUNET_H3_ORDER=32 WUNET_H3D_ORDER=32" > $O/convbench.txt 2>&1; tail -40 $O/convbench.txt | cut -c1-200
STEPS=30 timeout 400 tools/ab.sh "WUNET_H3_XDMA=0" "WUNET_H3_XDMA=1" > $O/ab_step.txt 2>&1; cat $O/ab_step.txt
