out=${1:-gpurun_out/r3j}; mkdir -p $out
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "thin or band or local_attention or pointwise" 2>&1 | tail -5 > $out/pytest.log
cat $out/pytest.log
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-gen-fwd --graph-steps 0"
run() { name=$1; shift; env "$@" timeout 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', d['ms_per_step'], d['value'])" >> $out/ab.txt 2>&1; }
run band X=1
run noband HOIG_NO_BAND=1
run band X=1
run noband HOIG_NO_BAND=1
cat $out/ab.txt
HOIG_STREAMS=0 HOIG_WGRAD_STREAM=0 ROWS=140 python tools/conv_table.py bf16x3:f16x2 > $out/conv_table.txt 2>/dev/null
grep "5x5\|7x7" $out/conv_table.txt | cut -c1-110
bash tools/r3_prof_serial.sh $out
grep "thin\|band" $out/serial_top.txt
