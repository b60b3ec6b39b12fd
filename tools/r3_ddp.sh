out=${1:-gpurun_out/r3l}; mkdir -p $out
timeout 1500 python -m pytest tests/test_ddp_rccl_gpu.py tests/test_graph_gpu.py -x -q 2>&1 | grep -v "^frame\|^$" | tail -30 > $out/pytest_ddp.log
tail -15 $out/pytest_ddp.log
python tools/ddp_overhead.py 20 2>&1 | grep "ms/step" > $out/ddp_overhead.txt
taskset -c 0 python tools/ddp_overhead.py 20 2>&1 | grep "ms/step" > $out/ddp_overhead_1core.txt
cat $out/ddp_overhead.txt $out/ddp_overhead_1core.txt
python tools/bench_thin.py 2>&1 | grep -v "Warn\|amdgpu" > $out/bench_thin.txt; cat $out/bench_thin.txt
