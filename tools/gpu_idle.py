"""GPU busy / idle time of the training step from a rocprofv3 --kernel-trace CSV (is the step host-bound?).
    python tools/gpu_idle.py <kernel_trace.csv> [n_last_steps_kernel_name]
Prints the union of kernel intervals (any stream), the idle time between them, and the largest gaps with the kernels around
them, over the window between the 2nd and the last adam_kernel launch (whole optimiser steps)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows), key=lambda t: t[0])
adam = [i for i, k in enumerate(ks) if 'adam_kernel' in k[2]]
# two adam launches per step (G, D): take whole steps between D-adam launches
lo, hi = adam[1], adam[-1]
win = ks[lo + 1:hi + 1]
steps = (len(adam) - 2) // 2
t0, t1 = win[0][0], max(k[1] for k in win)
busy, cur_s, cur_e = 0, win[0][0], win[0][1]
gaps = []
prev = win[0]
for k in win[1:]:
    if k[0] > cur_e:
        busy += cur_e - cur_s
        gaps.append((k[0] - cur_e, prev[2][:60], k[2][:60]))
        cur_s, cur_e = k[0], k[1]
    else:
        cur_e = max(cur_e, k[1])
    prev = k if k[1] >= prev[1] else prev
busy += cur_e - cur_s
wall = t1 - t0
print('steps %d  wall %.2f ms/step  busy %.2f ms/step  idle %.2f ms/step (%.1f%%)  kernels/step %d'
      % (steps, wall / steps / 1e6, busy / steps / 1e6, (wall - busy) / steps / 1e6, 100.0 * (wall - busy) / wall, len(win) // steps))
small = sum(g[0] for g in gaps if g[0] < 20000)
print('gaps < 20 us: %d, total %.2f ms/step; gaps >= 20 us: %d, total %.2f ms/step'
      % (sum(1 for g in gaps if g[0] < 20000) // steps, small / steps / 1e6, sum(1 for g in gaps if g[0] >= 20000) // steps,
         (wall - busy - small) / steps / 1e6))
for g in sorted(gaps, reverse=True)[:12]:
    print('  gap %.1f us  after %-60s before %s' % (g[0] / 1e3, g[1], g[2]))
