#!/bin/bash
# Round-2 evidence run on the MI355X box (through gpurun): tests, bench, rocprofv3 stats and PMC passes -> gpurun_out/
set -u
cd /root/repo
O=/root/repo/gpurun_out
TAG=${1:-r02}
python -c "from hoig_amd import _lib; _lib.lib" || { echo "library does not load: stale snapshot?"; exit 9; }
python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/${TAG}_pytest.log
python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
export TMPDIR=/tmp
cd /tmp
prof() {  # name, then the rocprofv3 arguments
  local name=$1; shift
  rm -rf /tmp/p_$name
  rocprofv3 "$@" > /dev/null 2>&1
}
HOIG_WGRAD_STREAM=0 HOIG_G_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_serial -- python3 /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-gen-fwd > /dev/null 2>&1
cp $(find /tmp/p_serial -name "*kernel_stats.csv" | head -1) $O/${TAG}_serial_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_bench -- python3 /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-gen-fwd > /dev/null 2>&1
cp $(find /tmp/p_bench -name "*kernel_stats.csv" | head -1) $O/${TAG}_bench_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_dom -- python3 /root/repo/tools/dominant_conv.py f16f6 > /dev/null 2>&1
cp $(find /tmp/p_dom -name "*kernel_stats.csv" | head -1) $O/${TAG}_dominant_conv_f6_kernel_stats.csv
for c in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"; do
  set -- $c; n=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/p_pmc_$n -- python3 /root/repo/tools/dominant_conv.py f16f6 > /dev/null 2>&1
  cp $(find /tmp/p_pmc_$n -name "*counter_collection.csv" | head -1) $O/${TAG}_pmc_${n}_dominant_conv_f6.csv
  B=32 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/p_pmcf_$n -- python3 /root/repo/tools/fwd_only.py > /dev/null 2>&1
  cp $(find /tmp/p_pmcf_$n -name "*counter_collection.csv" | head -1) /tmp/pmcf_$n.csv
done
cd /root/repo
python tools/pmc_summary.py conv_halo3_f6_kernel $O/${TAG}_pmc_dominant_conv_f6.json fetch=$O/${TAG}_pmc_fetch_dominant_conv_f6.csv write=$O/${TAG}_pmc_write_dominant_conv_f6.csv sq=$O/${TAG}_pmc_sq_dominant_conv_f6.csv > /dev/null
python tools/pmc_summary.py ALL $O/${TAG}_pmc_genfwd_b32.json fetch=/tmp/pmcf_fetch.csv write=/tmp/pmcf_write.csv sq=/tmp/pmcf_sq.csv > /dev/null
cd /tmp
HOIG_G_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_fwd -- python3 /root/repo/tools/fwd_only.py > /dev/null 2>&1
cp $(find /tmp/p_fwd -name "*kernel_stats.csv" | head -1) $O/${TAG}_genfwd_b32_kernel_stats.csv
cd /root/repo
HOIG_WGRAD_STREAM=0 HOIG_G_STREAMS=0 python tools/conv_table.py f16f6 2>/dev/null | grep -v "created\|amdgpu" > $O/${TAG}_conv_table.txt
tail -15 $O/${TAG}_pytest.log; cat $O/${TAG}_bench.json; cat $O/${TAG}_pmc_dominant_conv_f6.json | head -30; cat $O/${TAG}_pmc_genfwd_b32.json | head -40
