#!/bin/bash
# Round-6 evidence run on the MI355X box (through gpurun): tests, bench, rocprofv3 stats and PMC passes -> gpurun_out/<tag>_*
set -u
O=$GRAFT_REPO_ROOT/gpurun_out
TAG=${1:-r06}
mkdir -p $O
cd $GRAFT_REPO_ROOT
python -c "from hoig_amd import _lib; _lib.lib" || { echo "library does not load: stale snapshot?"; exit 9; }
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  ( time timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -15 ) > $O/${TAG}_pytest_gpu.log 2>&1
  # the opt-in duplicates too (the exact-fp32 gradient comparison at 128x128 lives there), once per round
  ( time timeout 2400 python -m pytest tests -m gpu_slow -q 2>&1 | tail -8 ) > $O/${TAG}_pytest_gpu_slow.log 2>&1
fi
python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
export TMPDIR=/tmp
cd /tmp
HOIG_WGRAD_STREAM=0 HOIG_STREAMS=0 HOIG_BENCH_NO_ROOF=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_serial -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-gen-fwd --graph-steps 0 --host-samples 0 > /dev/null 2>&1
cp $(find /tmp/p_serial -name "*kernel_stats.csv" | head -1) $O/${TAG}_serial_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-gen-fwd --graph-steps 0 --host-samples 0 > /dev/null 2>&1
cp $(find /tmp/p_bench -name "*kernel_stats.csv" | head -1) $O/${TAG}_bench_kernel_stats.csv
cp $(find /tmp/p_bench -name "*kernel_trace.csv" | head -1) /tmp/${TAG}_bench_trace.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_dom -- python3 $GRAFT_REPO_ROOT/tools/dominant_conv.py bf16x3 > /dev/null 2>&1
cp $(find /tmp/p_dom -name "*kernel_stats.csv" | head -1) $O/${TAG}_dominant_conv_kernel_stats.csv
for c in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"; do
  set -- $c; n=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/p_pmc_$n -- python3 $GRAFT_REPO_ROOT/tools/dominant_conv.py bf16x3 > /dev/null 2>&1
  cp $(find /tmp/p_pmc_$n -name "*counter_collection.csv" | head -1) $O/${TAG}_pmc_${n}_dominant_conv.csv
  B=32 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/p_pmcf_$n -- python3 $GRAFT_REPO_ROOT/tools/fwd_only.py > /dev/null 2>&1
  cp $(find /tmp/p_pmcf_$n -name "*counter_collection.csv" | head -1) /tmp/pmcf_$n.csv
done
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py conv_halo3_m16_kernel $O/${TAG}_pmc_dominant_conv.json fetch=$O/${TAG}_pmc_fetch_dominant_conv.csv write=$O/${TAG}_pmc_write_dominant_conv.csv sq=$O/${TAG}_pmc_sq_dominant_conv.csv > /dev/null
python tools/pmc_summary.py ALL $O/${TAG}_pmc_genfwd_b32.json fetch=/tmp/pmcf_fetch.csv write=/tmp/pmcf_write.csv sq=/tmp/pmcf_sq.csv > /dev/null
python tools/timeline.py /tmp/${TAG}_bench_trace.csv 2000 > $O/${TAG}_timeline_eager.txt 2>&1
# the same step as a captured hipGraph: where its 2.7 ms go
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_graph -- python3 $GRAFT_REPO_ROOT/bench.py --graph --steps 3 --warmup 1 --no-cpu-baseline --no-gen-fwd --graph-steps 0 --host-samples 0 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/timeline.py $(find /tmp/p_graph -name "*kernel_trace.csv" | head -1) 2000 > $O/${TAG}_timeline_graph.txt 2>&1
# the dominant SHAPE inside the multi-stream step and inside the one-stream step (the template serves several shapes)
python tools/dominant_in_step.py /tmp/${TAG}_bench_trace.csv conv_halo3_m16_kernel 131072 > $O/${TAG}_dominant_in_step.txt 2>&1
python tools/dominant_in_step.py $(find /tmp/p_serial -name "*kernel_trace.csv" | head -1) conv_halo3_m16_kernel 131072 >> $O/${TAG}_dominant_in_step.txt 2>&1
python tools/kstats_top.py $O/${TAG}_serial_kernel_stats.csv 5 70 > $O/${TAG}_serial_top.txt
HOIG_WGRAD_STREAM=0 HOIG_STREAMS=0 ROWS=150 python tools/conv_table.py bf16x3:f16x2 2>/dev/null | grep -v "created\|amdgpu" > $O/${TAG}_conv_table.txt
# the same per-SHAPE table inside the multi-stream step (events on each launch's own stream: durations include what runs beside it)
ROWS=40 python tools/conv_table.py bf16x3:f16x2 2>/dev/null | grep -v "created\|amdgpu" > $O/${TAG}_conv_table_streams.txt
cd /tmp
HOIG_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_fwd -- python3 $GRAFT_REPO_ROOT/tools/fwd_only.py > /dev/null 2>&1
cp $(find /tmp/p_fwd -name "*kernel_stats.csv" | head -1) $O/${TAG}_genfwd_b32_kernel_stats.csv
cd $GRAFT_REPO_ROOT
python tools/ab_graph.py 4 30 2>/dev/null | grep -v "created\|amdgpu\|WARNING" > $O/${TAG}_graph_ab.txt
python tools/ddp_overhead.py 20 2>&1 | grep -v "created\|amdgpu\|WARN\|socket" > $O/${TAG}_ddp_overhead.txt
python tools/knockout_step.py 2>/dev/null | grep -v "created\|amdgpu\|WARNING" > $O/${TAG}_knockout_step.txt
cat $O/${TAG}_dominant_in_step.txt $O/${TAG}_knockout_step.txt $O/${TAG}_graph_ab.txt $O/${TAG}_ddp_overhead.txt
tail -5 $O/${TAG}_pytest_gpu.log 2>/dev/null; cat $O/${TAG}_bench.json; head -20 $O/${TAG}_pmc_dominant_conv.json; head -12 $O/${TAG}_serial_top.txt
