mkdir -p gpurun_out/r3d
timeout 900 python -m pytest tests/test_graph_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/r3d/pytest_graph.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_graph -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-gen-fwd --graph-steps 0 > /tmp/prof_graph.log 2>&1
f=$(find /tmp/prof_graph -name "*kernel_trace.csv" | head -1)
cd $GRAFT_REPO_ROOT
python tools/timeline.py $f 1000 > gpurun_out/r3d/timeline_graph.txt 2>&1
python tools/gpu_idle.py $f > gpurun_out/r3d/idle_graph.txt 2>&1
cd /tmp
HOIG_GRAPH=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_eager -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-gen-fwd --graph-steps 0 > /tmp/prof_eager.log 2>&1
f=$(find /tmp/prof_eager -name "*kernel_trace.csv" | head -1)
cd $GRAFT_REPO_ROOT
python tools/timeline.py $f 1000 > gpurun_out/r3d/timeline_eager.txt 2>&1
tail -3 /tmp/prof_graph.log /tmp/prof_eager.log
cat gpurun_out/r3d/pytest_graph.log; head -30 gpurun_out/r3d/timeline_graph.txt
