mkdir -p gpurun_out/r3f
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "thin or conv2d_fwd_bwd" -s 2>&1 | grep -v Warn | tail -60 > gpurun_out/r3f/pytest_thin.log
tail -40 gpurun_out/r3f/pytest_thin.log
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-gen-fwd --graph-steps 0"
run() { name=$1; shift; env "$@" timeout 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', d['ms_per_step'], d['value'])" >> gpurun_out/r3f/ab.txt 2>&1; }
run thin X=1
run nothin HOIG_NO_THIN=1
run thin X=1
run nothin HOIG_NO_THIN=1
cat gpurun_out/r3f/ab.txt
