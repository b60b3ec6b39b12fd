#!/bin/bash
# Diagnostic build of libhoig_hip.so with in-kernel cycle stamps in the 3x3 halo conv (tools/stamp_halo.py): tools/_build/
set -e
cd "$(dirname "$0")/../hoig_amd/csrc"
OUT=../../tools/_build
mkdir -p $OUT
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -I../../include -I. -Wno-unused-result -DHOIG_STAMP=${HOIG_STAMP:-1}"
for f in conv_igemm conv_igemm_bf16 conv_small norm attn sample pointwise input_prep raster; do
  EXTRA=""; { [ $f = input_prep ] || [ $f = raster ]; } && EXTRA="-ffp-contract=off"
  /opt/rocm/bin/hipcc $FLAGS $EXTRA -c $f.hip -o $OUT/$f.stamp.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libhoig_hip_stamp${HOIG_STAMP:-1}.so $OUT/*.stamp.o
echo built $OUT/libhoig_hip_stamp${HOIG_STAMP:-1}.so
