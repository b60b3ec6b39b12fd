#!/bin/bash
# rocprofv3 PMC passes over the dominant weight-gradient launch on the LDS-DMA kernel (run through gpurun):
#   tools/pmc_wgrad5.sh [images] -> gpurun_out/r05_pmc_dominant_wgrad[_<images>].json  (+ the register kernel beside it, same passes)
set -u
O=$GRAFT_REPO_ROOT/gpurun_out
B=${1:-16}
export TMPDIR=/tmp
cd /tmp
for mode in split plain; do
  for c in "fetch FETCH_SIZE" "write WRITE_SIZE" \
           "sq SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" \
           "lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM"; do
    set -- $c; n=$1; shift
    rm -rf /tmp/p_wg_$n
    rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/p_wg_$n -- python3 $GRAFT_REPO_ROOT/tools/dominant_wgrad.py $B $mode > /dev/null 2>&1
    cp $(find /tmp/p_wg_$n -name "*counter_collection.csv" | head -1) /tmp/wg_$n.csv
  done
  cd $GRAFT_REPO_ROOT
  k=wgrad_dma_kernel; f=r05_pmc_dominant_wgrad; [ $mode = plain ] && { k=wgrad_halo_bf16_kernel; f=r05_pmc_dominant_wgrad_register_kernel; }
  [ $B != 16 ] && f=${f}_$B
  python tools/pmc_summary.py $k $O/$f.json fetch=/tmp/wg_fetch.csv write=/tmp/wg_write.csv sq=/tmp/wg_sq.csv lds=/tmp/wg_lds.csv > /dev/null
  cat $O/$f.json
  cd /tmp
done
