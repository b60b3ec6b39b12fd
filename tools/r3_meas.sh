out=${1:-gpurun_out/r3h}; mkdir -p $out
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "thin or pointwise" 2>&1 | tail -3 > $out/pytest.log
HOIG_STREAMS=0 HOIG_WGRAD_STREAM=0 ROWS=140 python tools/conv_table.py bf16x3:f16x2 > $out/conv_table.txt 2>/dev/null
bash tools/r3_prof_serial.sh $out
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-gen-fwd --graph-steps 0"
$B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['ms_per_step'], d['value'])" >> $out/ab.txt 2>&1
cat $out/pytest.log $out/ab.txt; head -50 $out/serial_top.txt
