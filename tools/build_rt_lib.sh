#!/bin/bash
# tools/_lib_rt.so: the library with conv_h3d_kernel's phase stamps on the 100 MHz real-time counter (-DWUNET_TRACE_REALTIME), for
# WUNET_LIB_PATH=$PWD/tools/_lib_rt.so WUNET_TRACE_REALTIME=1 python tools/conv_bench.py --trace ...   (block start / end times of one launch)
set -e
cd "$(dirname "$0")/.."
CS=wave-u-net-for-speech-enhancement_amd/csrc
make -C $CS -j8 > /dev/null
( cd $CS && /opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I. -I../../include -Wno-unused-result -Wno-unused-value \
      -DWUNET_TRACE_REALTIME -c h3d_inst.cpp -o /tmp/h3d_rt.o )
( cd $CS && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_lib_rt.so wunet_plan.o wunet_launchers.o wunet_forward.o \
      wunet_backward.o wunet_ops.o wunet_comm.o h3_inst.o /tmp/h3d_rt.o conv_15.o conv_5.o wgrad_15.o wgrad_5.o -ldl )
ls -la tools/_lib_rt.so
