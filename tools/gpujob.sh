#!/bin/bash
# usage: tools/gpujob.sh <tag> [timeout]   -- runs tools/_ab.sh on a GPU box, retrying while no slot is free
tag=$1; to=${2:-3200}
for i in 1 2 3 4 5 6 7 8; do
  /usr/local/graft/bin/gpurun --timeout $to -- 'bash tools/_ab.sh' > /root/repo/gpurun_out/${tag}_stdout.txt 2>&1
  rc=$?
  if grep -q "status=transient" /root/repo/gpurun_out/${tag}_stdout.txt; then sleep 60; continue; fi
  break
done
echo "gpujob $tag finished rc=$rc" >> /root/repo/gpurun_out/${tag}_stdout.txt
