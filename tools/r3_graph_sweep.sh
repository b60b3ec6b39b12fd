mkdir -p gpurun_out/r3b
timeout 900 python -m pytest tests/test_graph_gpu.py -x -q 2>&1 | tail -30 > gpurun_out/r3b/pytest_graph.log
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-gen-fwd --graph-steps 0"
run() { name=$1; shift; env "$@" timeout 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', d['ms_per_step'], d['value'])" >> gpurun_out/r3b/sweep.txt 2>&1; }
run base X=1
run q2 DEBUG_HIP_FORCE_GRAPH_QUEUES=2
run q8 DEBUG_HIP_FORCE_GRAPH_QUEUES=8
run q16 DEBUG_HIP_FORCE_GRAPH_QUEUES=16
run hw8 GPU_MAX_HW_QUEUES=8
run hw8q8 GPU_MAX_HW_QUEUES=8 DEBUG_HIP_FORCE_GRAPH_QUEUES=8
run nopkt DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run eager HOIG_GRAPH=0
run eager_hw8 HOIG_GRAPH=0 GPU_MAX_HW_QUEUES=8
cat gpurun_out/r3b/pytest_graph.log gpurun_out/r3b/sweep.txt
