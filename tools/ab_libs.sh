#!/bin/bash
# alternating bench runs with two builds of the library (old = HEAD, new = working tree)
B=hoig_amd/csrc/_build
for i in 1 2 3; do
  for v in old new; do
    cp $B/lib_$v.so.keep $B/libhoig_hip.so
    echo -n "[$v] "
    HOIG_BENCH_NO_ROOF=1 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --graph-steps 0 --no-gen-fwd 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
  done
done
