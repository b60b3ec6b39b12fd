# round-3 evidence run: f16f6 rows of the precision frontier, trajectory, gradient spread at 256x256, timelines
mkdir -p gpurun_out/r3e
python tools/grad_parity.py 256 1 4 f16f6@1 f16f6@192 bf16x3:f16x2 > gpurun_out/r3e/grad_parity_256.txt 2>&1
python tools/precision_frontier.py f16f6 bf16x3:f16x2 > gpurun_out/r3e/precision_frontier_f16f6.txt 2>&1
F6_MIN_TILES=192 python tools/precision_frontier.py f16f6 > gpurun_out/r3e/precision_frontier_f16f6_prod.txt 2>&1
python tools/trajectory.py 100 64 4 f32 f32#again f16f6 f16f6#again bf16x3:f16x2 > gpurun_out/r3e/trajectory_64.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_graph -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-gen-fwd --graph-steps 0 > /tmp/prof_graph.log 2>&1
f=$(find /tmp/prof_graph -name "*kernel_trace.csv" | head -1)
cd $GRAFT_REPO_ROOT
python tools/timeline.py $f 2000 > gpurun_out/r3e/timeline_graph.txt 2>&1
tail -n 12 gpurun_out/r3e/grad_parity_256.txt; head -40 gpurun_out/r3e/timeline_graph.txt
