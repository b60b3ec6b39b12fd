#!/bin/bash
# usage: tools/ab_bench.sh "ENV_A" "ENV_B" [rounds]   -- alternating bench runs in one box
A="$1"; B="$2"; R=${3:-3}
for i in $(seq $R); do
  for cfg in "$A" "$B"; do
    echo -n "[$cfg] "
    env $cfg HOIG_BENCH_NO_ROOF=1 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --graph-steps 0 --no-gen-fwd 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
  done
done
