#!/bin/bash
# Shader clock and socket power sampled by rocm-smi while long launches of tools/microbench/dma_issue.hip (MFMAs only / LDS-DMA from L2 / from HBM / both) and
# bench.py's training and eval-forward steps run: the evidence behind "the matrix pipe's ceiling is a power limit" (HISTORY.md section 9).
# GPU box; writes gpurun_out/power_probe.txt
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; OUT=gpurun_out/power_probe.txt; : > $OUT
sample() {   # <label> <command...>: start the command, sample rocm-smi 8 times 0.3 s apart after <delay> s, wait for it
    local label=$1 delay=$2; shift; shift
    "$@" > /tmp/pp_cmd.txt 2>&1 &
    local pid=$!
    sleep $delay
    local clk="" pw=""
    for i in 1 2 3 4 5 6 7 8; do
        s=$(rocm-smi --showclocks --showpower 2>/dev/null)
        c=$(echo "$s" | grep "sclk clock" | head -1 | sed 's/.*(\([0-9]*\)Mhz).*/\1/')
        p=$(echo "$s" | grep "Power (W)" | head -1 | sed 's/.*: *//')
        clk="$clk $c"; pw="$pw $p"
        sleep 0.3
    done
    wait $pid
    printf "%-46s | sclk MHz:%s | socket W:%s\n" "$label" "$clk" "$pw" >> $OUT
    grep -E "ms " /tmp/pp_cmd.txt | head -2 | sed 's/^/      /' >> $OUT
}
echo "# rocm-smi samples (8 per workload, 0.3 s apart) while ONE long launch / a long loop of steps runs" >> $OUT
sample "idle" 0.2 sleep 3
sample "MFMAs only (2 waves / SIMD, 180 per stage)" 1.5 tools/microbench/_dma_issue 1500000 0
sample "LDS-DMA only, source in L2" 1.5 tools/microbench/_dma_issue 4000000 1
sample "LDS-DMA only, source in HBM" 1.5 tools/microbench/_dma_issue 800000 2
sample "DMA burst from L2 + MFMAs" 1.5 tools/microbench/_dma_issue 1200000 3
sample "DMA burst from HBM + MFMAs" 1.5 tools/microbench/_dma_issue 800000 4
sample "bench.py training steps (split GEMMs)" 6 python bench.py --steps 1500 --warmup 5 --no-cpu-baseline --no-extras --no-roofline
sample "bench.py --gemm fp32 training steps" 6 python bench.py --gemm fp32 --steps 700 --warmup 5 --no-cpu-baseline --no-extras --no-roofline
sample "bench.py --mode forward (eval)" 6 python bench.py --mode forward --steps 4000 --warmup 5 --no-cpu-baseline --no-extras --no-roofline
cat $OUT
