#!/bin/bash
# What bounds conv_h3u_kernel: the kernel rebuilt with parts compiled out (-DWUNET_H3U_ABL=<bits>, wunet_h3u.h), the eval forward timed per variant.
#   bits: 1 no prefetch loads   4 no MFMAs   8 no W DMA   16 no conversion at all (no LDS writes)   32 no fragment reads, no MFMAs
#         64 the loads stay alive (their registers are summed) but nothing is converted   128 conversion without its LDS writes
#   e.g. 100 = 64 + 32 + 4: the memory pipeline alone;  36: the loader waves alone;  64: loads in flight but nothing converted (the MFMA waves + the memory pipeline)
#   (16 WITHOUT 1 is not a valid build since the prefetch loads come from inline asm: with nothing reading their destination registers hipcc hands
#    those registers to address arithmetic while the loads are in flight - a wild address sooner or later.  Use 64: it keeps the registers alive.)
#   tools/h3u_ablation.sh build (container) -> tools/_lib_u<bits>.so ;  tools/h3u_ablation.sh run (GPU box) -> gpurun_out/h3u_ablation.txt
set -e
cd "$(dirname "$0")/.."
CS=wave-u-net-for-speech-enhancement_amd/csrc
VARIANTS=${VARIANTS:-"1 4 8 36 44 64 68 100"}
if [ "$1" = build ]; then
    make -C $CS -j8 > /dev/null
    for a in $VARIANTS; do
        ( cd $CS && /opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I. -I../../include -Wno-unused-result -Wno-unused-value \
              -DWUNET_H3U_ABL=$a -c h3u_inst.cpp -o /tmp/h3u_abl$a.o ) &
    done
    wait
    for a in $VARIANTS; do
        ( cd $CS && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_lib_u$a.so wunet_plan.o wunet_launchers.o wunet_forward.o \
              wunet_backward.o wunet_ops.o wunet_comm.o h3_inst.o h3d_inst.o /tmp/h3u_abl$a.o conv_15.o conv_5.o wgrad_15.o wgrad_5.o -ldl )
    done
    ls tools/_lib_u*.so
    exit 0
fi
mkdir -p gpurun_out
for a in 0 $VARIANTS; do
    if [ $a = 0 ]; then unset WUNET_LIB_PATH; else export WUNET_LIB_PATH=$PWD/tools/_lib_u$a.so; fi
    WUNET_H3U=${H3U:-4096,0} python bench.py --mode forward --no-cpu-baseline --no-extras --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('abl %3d' % $a, ' '.join('%s %.1f us x%g' % (t['kernel'][5:], 1e3*t['ms_per_step']/t['launches_per_step'], t['launches_per_step']) for t in r['top5'] if 'h3u' in t['kernel']))
"
done | tee gpurun_out/h3u_ablation.txt
