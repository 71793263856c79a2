#!/bin/bash
# What bounds conv_h3d_kernel: the kernel rebuilt with parts of it compiled out (-DWUNET_ABL=<bits>, wunet_h3d.h; the product build has
# WUNET_ABL = 0 and none of this code), every variant timed on the BASELINE geometries by tools/conv_bench.py.  Results of the ablated
# builds are garbage by construction - only their times mean anything.
#   bits: 1 no MFMAs   2 no DMA at all   4 no output stores   8 no x-tile DMA   16 no W DMA   32 no vmcnt wait at the top of a stage
#         128 every x piece from the 16-byte zero pad (cache hit)   256 every W run from the first run of the pack (cache hit)
#         512 two MFMA passes instead of three (the W_lo x_hi products dropped: what a pass costs)
#         1024 the x tile's DMA without the de-interleave (every instruction 1 KiB contiguous, wrong columns, same bytes): what the gather costs
#   tools/conv_ablation.sh build      (container: hipcc)  ->  tools/_lib_abl<bits>.so  (git-ignored, shipped by gpurun)
#   tools/conv_ablation.sh run        (GPU box)           ->  gpurun_out/conv_ablation.txt
set -e
cd "$(dirname "$0")/.."
CS=wave-u-net-for-speech-enhancement_amd/csrc
VARIANTS="1 2 4 6 8 16 32 128 384 512"
if [ "$1" = build ]; then
    make -C $CS -j8 > /dev/null
    for a in $VARIANTS; do
        ( cd $CS && /opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I. -I../../include -Wno-unused-result -Wno-unused-value \
              -DWUNET_ABL=$a -c h3d_inst.cpp -o /tmp/h3d_abl$a.o ) &
    done
    wait
    for a in $VARIANTS; do
        ( cd $CS && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_lib_abl$a.so wunet_plan.o wunet_launchers.o wunet_forward.o \
              wunet_backward.o wunet_ops.o wunet_comm.o h3_inst.o /tmp/h3d_abl$a.o conv_15.o conv_5.o wgrad_15.o wgrad_5.o -ldl )
    done
    ls -la tools/_lib_abl*.so
    exit 0
fi
mkdir -p gpurun_out
OUT=gpurun_out/conv_ablation.txt
{
    echo "conv_h3d_kernel with parts compiled out (tools/conv_ablation.sh; forward + data gradient of every level of >= 256 samples, batch 64; us per layer, one box)"
    printf "%-6s" layer
    for a in 0 $VARIANTS; do printf " %7s" "abl$a"; done
    echo
} > $OUT
for a in 0 $VARIANTS; do
    if [ $a = 0 ]; then unset WUNET_LIB_PATH; else export WUNET_LIB_PATH=$PWD/tools/_lib_abl$a.so; fi
    python tools/conv_bench.py --min-l 256 2>/dev/null | grep -E "^(enc|dec|total)" | awk '{ if ($1 == "total") print "total", $4; else print $1, $6 }' > /tmp/abl_$a.txt
done
python - >> $OUT <<'P'
cols = [0, 1, 2, 4, 6, 8, 16, 32, 128, 384, 512]
rows = {}
order = []
for a in cols:
    for ln in open("/tmp/abl_%d.txt" % a):
        k, v = ln.split()
        if k not in rows: rows[k] = {}; order.append(k)
        rows[k][a] = float(v)
for k in order:
    print("%-6s" % k + "".join(" %7.1f" % rows[k].get(a, float("nan")) for a in cols))
print("0 the product kernel; 1 no MFMAs (staging + stores only); 2 no DMA; 4 no stores; 6 neither (MFMAs, LDS reads, barriers only); 8 no x DMA;")
print("16 no W DMA; 32 no vmcnt wait at the stage top; 128 x pieces from a cached page (W real); 384 x and W from cached pages; 512 two MFMA passes instead of three")
P
cat $OUT
