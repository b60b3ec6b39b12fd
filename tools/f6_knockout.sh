#!/bin/bash
# Builds (here) / runs (on the GPU box) the knock-out variants of the fp6 forward kernel.
# usage: [KOS="0 57 ..."] f6_knockout.sh build | run [batch] [0 = zero operands]
cd "$(dirname "$0")/.."
KOS=${KOS:-"0 1 9 25 57"}
if [ "$1" = build ]; then
  for k in $KOS; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -DHOIG_F6_KO=$k -DHOIG_F6_KO_VALUE=$k -DHOIG_F6_DMA=${DMA:-1} -Iinclude -Ihoig_amd/csrc -Wno-unused-result \
      tools/f6_knockout.cpp hoig_amd/csrc/conv_f6.hip -o tools/_build/f6_ko_${k}${SUF:-} &
  done
  wait
else
  for k in $KOS; do tools/_build/f6_ko_${k}${SUF:-} ${2:-16} ${3:-1}; done
fi
