#!/bin/bash
# Step time against the hardware-queue class of the step's side streams (HOIG_STREAM_MAP; hoig_amd/ops.py new_stream): one
# bench.py run of 50 timed steps per assignment.  Roles in order: optimiser side, bg, obj, src, loss x2, D, weight-gradient.
out=${1:-gpurun_out/stream_map_sweep.txt}
B="python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-gen-fwd --graph-steps 0"
for m in "$@"; do :; done
maps=${MAPS:-"3,1,3,1,2,2,2,2 2,1,3,1,2,2,2,2 0,1,3,1,2,2,2,2 1,1,3,1,2,2,2,2 3,3,1,1,2,2,2,2 3,1,1,3,2,2,2,2 3,1,3,1,3,3,2,2 3,1,3,1,2,2,0,2 3,1,3,1,2,2,2,0 3,1,3,1,2,3,2,2 3,1,3,1,2,2,3,2 3,1,3,1,2,2,1,2 3,1,3,1,0,0,2,2 3,1,3,1,0,0,0,0 3,1,2,1,3,3,3,3 3,2,3,1,2,2,2,2 3,1,3,1,2,2,2,2"}
for m in $maps; do
  HOIG_STREAM_MAP=$m timeout 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m', d['ms_per_step'])" >> $out
done
cat $out
