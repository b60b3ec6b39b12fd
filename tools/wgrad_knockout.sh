#!/bin/bash
# usage: [KOS="0 1 ..."] wgrad_knockout.sh build | run [batch] [0 = zero operands] [HOIG_PREC_* value]
cd "$(dirname "$0")/.."
KOS=${KOS:-"0 1 2 4 6 8 16 48"}
if [ "$1" = build ]; then
  for k in $KOS; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -DHOIG_WG_KO=$k -DHOIG_WG_KO_VALUE=$k -Iinclude -Ihoig_amd/csrc -Wno-unused-result \
      tools/wgrad_knockout.cpp hoig_amd/csrc/wgrad_igemm_bf16.hip -o tools/_build/wg_ko_$k &
  done
  wait
else
  for k in $KOS; do tools/_build/wg_ko_$k ${2:-16} ${3:-1} ${4:-3} ${5:-512} ${6:-}; done
fi
