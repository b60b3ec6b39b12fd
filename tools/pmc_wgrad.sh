#!/bin/bash
# rocprofv3 PMC passes over the dominant weight-gradient launch (run through gpurun): -> gpurun_out/r03_pmc_dominant_wgrad.json
set -u
O=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
cd /tmp
for c in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"; do
  set -- $c; n=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/p_wg_$n -- python3 $GRAFT_REPO_ROOT/tools/dominant_wgrad.py > /dev/null 2>&1
  cp $(find /tmp/p_wg_$n -name "*counter_collection.csv" | head -1) /tmp/wg_$n.csv
done
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py wgrad_halo_bf16_kernel $O/r03_pmc_dominant_wgrad.json fetch=/tmp/wg_fetch.csv write=/tmp/wg_write.csv sq=/tmp/wg_sq.csv
cat $O/r03_pmc_dominant_wgrad.json
