#!/bin/bash
# usage: tools/isa.sh <unit.cpp> [out.s]   - gfx950 ISA of one instantiation unit + registers / scratch per kernel
C=/root/repo/wave-u-net-for-speech-enhancement_amd/csrc; O=${2:-/tmp/isa/$(basename $1 .cpp).s}; mkdir -p $(dirname $O)
/opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -I$C -I/root/repo/include -S --cuda-device-only -o $O $C/$1 2>&1 | grep -v "hip-link"
grep "^; NumVgprs\|; ScratchSize\|^_Z.*:" $O | paste - - - | awk '{print $1, $6, $7, $9,$10}'
