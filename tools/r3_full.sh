out=${1:-gpurun_out/r3k}; mkdir -p $out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $out/pytest_gpu.log
tail -5 $out/pytest_gpu.log
python tools/ddp_overhead.py 20 2>&1 | grep -v "Warn\|amdgpu.ids" > $out/ddp_overhead.txt
taskset -c 0 python tools/ddp_overhead.py 20 2>&1 | grep -v "Warn\|amdgpu.ids" > $out/ddp_overhead_1core.txt
cat $out/ddp_overhead.txt $out/ddp_overhead_1core.txt
