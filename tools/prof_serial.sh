# one-stream profile of the training step (kernel times add up): rocprofv3 --kernel-trace --stats over 1 + 2 + 3 steps, no roofline loop
out=${1:-gpurun_out/r3prof}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_serial
HOIG_STREAMS=0 HOIG_WGRAD_STREAM=0 HOIG_GRAPH=0 HOIG_BENCH_NO_ROOF=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_serial -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-gen-fwd --graph-steps 0 > /tmp/prof_serial.log 2>&1
f=$(find /tmp/prof_serial -name "*kernel_stats.csv" | head -1)
cd $GRAFT_REPO_ROOT
cp $f $out/serial_kernel_stats.csv
python tools/kstats_top.py $f 5 60 > $out/serial_top.txt
tail -n 2 /tmp/prof_serial.log
