#!/bin/bash
# tools/_lib_trace.so: the product library with conv_h3u_kernel's stage stamps compiled in (-DWUNET_H3U_TRACE); tools/h3u_trace.py (GPU box) reads them
set -e
cd "$(dirname "$0")/../wave-u-net-for-speech-enhancement_amd/csrc"
make -j8 > /dev/null
/opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I. -I../../include -Wno-unused-result -Wno-unused-value -DWUNET_H3U_TRACE -c h3u_inst.cpp -o /tmp/h3u_tr.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_lib_trace.so wunet_plan.o wunet_launchers.o wunet_forward.o wunet_backward.o wunet_ops.o wunet_comm.o \
    h3_inst.o h3d_inst.o /tmp/h3u_tr.o conv_15.o conv_5.o wgrad_15.o wgrad_5.o -ldl
ls -la ../../tools/_lib_trace.so
