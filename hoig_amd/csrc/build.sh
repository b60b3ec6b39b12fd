#!/bin/bash
# Builds libhoig_hip.so (gfx950 code objects) in-tree: hoig_amd/csrc/_build/libhoig_hip.so
set -e
cd "$(dirname "$0")"
mkdir -p _build
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -I../../include -I. -Wno-unused-result"
pids=()
for f in conv_igemm conv_igemm_bf16 wgrad_igemm_bf16 dgrad_k128 conv_halo16 conv_wino conv_s2_16 wgrad_dma conv_igemm16 conv_flat16 wgrad_flat conv_head16 conv_f6 conv_small conv_thin norm attn sample pointwise input_prep raster mano data_prep tuning; do
  if [ ! -f _build/$f.o ] || [ $f.hip -nt _build/$f.o ] || [ common.h -nt _build/$f.o ] || [ conv_bf16_common.h -nt _build/$f.o ] || [ conv_m16_common.h -nt _build/$f.o ] || [ tuning.h -nt _build/$f.o ] || [ ../../include/hoig_kernels.h -nt _build/$f.o ]; then
    # input_prep.hip reproduces float->int truncations of the reference: no FMA contraction there (see its header)
    EXTRA=""; { [ $f = input_prep ] || [ $f = raster ] || [ $f = data_prep ]; } && EXTRA="-ffp-contract=off"
    $HIPCC $FLAGS $EXTRA -c $f.hip -o _build/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o _build/libhoig_hip.so _build/*.o
echo built _build/libhoig_hip.so
