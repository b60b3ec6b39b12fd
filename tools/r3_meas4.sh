out=${1:-gpurun_out/r3o}; mkdir -p $out
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "valid_5x5 or local_attention" -s 2>&1 | grep "valid 5x5\|passed\|failed\|Error" | tail -12 > $out/pytest.log
cat $out/pytest.log
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-gen-fwd --graph-steps 0"
run() { name=$1; shift; env "$@" timeout 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', d['ms_per_step'], d['value'])" >> $out/ab.txt 2>&1; }
run flat X=1
run noflat HOIG_NO_FLAT=1
run flat X=1
run noflat HOIG_NO_FLAT=1
cat $out/ab.txt
HOIG_STREAMS=0 HOIG_WGRAD_STREAM=0 ROWS=140 python tools/conv_table.py bf16x3:f16x2 > $out/conv_table.txt 2>/dev/null
grep "5x5" $out/conv_table.txt | cut -c1-110
