out=${1:-gpurun_out/r3m}; mkdir -p $out
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "thin or conv2d_fwd_bwd or epilogue" 2>&1 | tail -4 > $out/pytest.log
timeout 900 python -m pytest tests/test_trainer_gpu.py -x -q -k "golden and bf16x3 and not visuals" 2>&1 | tail -4 >> $out/pytest.log
cat $out/pytest.log
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-gen-fwd --graph-steps 0"
run() { name=$1; shift; env "$@" timeout 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', d['ms_per_step'], d['value'])" >> $out/ab.txt 2>&1; }
run fused X=1
run unfused HOIG_FUSE_HEADS=0
run fused X=1
run unfused HOIG_FUSE_HEADS=0
cat $out/ab.txt
python tools/bench_thin.py 2>&1 | grep -v "Warn\|amdgpu" > $out/bench_thin.txt; cat $out/bench_thin.txt
