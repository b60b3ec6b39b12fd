"""Concurrency timeline of ONE training step from a rocprofv3 --kernel-trace CSV: how many kernels are in flight over the step,
where the chip runs a single kernel (or nothing), and which kernels those are.
    python tools/timeline.py <kernel_trace.csv> [bucket_us=1000]
The step = the window between the last two launches of D's Adam (adam_pack_kernel / adam_dev_kernel / adam_kernel: two per step, G then D)."""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
bucket = int(sys.argv[2]) * 1000 if len(sys.argv) > 2 else 1000000
ks = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', ''), int(r.get('Grid_Size', 0) or 0),
              int(r.get('Workgroup_Size', 1) or 1)) for r in rows), key=lambda t: t[0])
adam = [i for i, k in enumerate(ks) if re.search(r'adam_(dev_|pack_)?kernel', k[2])]
lo, hi = adam[-3], adam[-1]            # D-adam of step n-1 .. D-adam of step n
win = ks[lo + 1:hi + 1]
t0, t1 = win[0][0], max(k[1] for k in win)
def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('at::native::', '')
    n = re.sub(r'^void ', '', n)
    n = re.sub(r'\(.*', '', n)
    return n[:44]
# sweep
ev = []
for s, e, n, q, g, w in win:
    ev.append((s, 1, n)); ev.append((e, -1, n))
ev.sort()
hist = collections.Counter(); cur = 0; last = t0
alone = collections.Counter()
live = collections.Counter()
for t, d, n in ev:
    hist[min(cur, 6)] += t - last
    if cur == 1:
        alone[next(iter(k for k, v in live.items() if v > 0))] += t - last
    last = t; cur += d; live[n] += d
wall = t1 - t0
print('step wall %.2f ms, %d kernels, sum of kernel durations %.2f ms, mean in flight %.2f' % (wall / 1e6, len(win), sum(e - s for s, e, *_ in win) / 1e6,
      sum(e - s for s, e, *_ in win) / wall))
print('time with k kernels in flight: ' + '  '.join('%s:%.1fms' % (str(k) if k < 6 else '6+', v / 1e6) for k, v in sorted(hist.items())))
print('kernels that run ALONE (top 15 by time alone):')
for n, v in alone.most_common(15):
    print('   %7.2f ms  %s' % (v / 1e6, short(n)))
nb = (wall + bucket - 1) // bucket
print('per %.1f-ms bucket: mean kernels in flight | small-grid share | top kernels by time' % (bucket / 1e6))
for b in range(nb):
    a, z = t0 + b * bucket, t0 + (b + 1) * bucket
    tot = 0; by = collections.Counter(); smallt = 0
    for s, e, n, q, g, w in win:
        o = min(e, z) - max(s, a)
        if o > 0:
            tot += o; by[short(n)] += o
            if g // max(w, 1) < 256: smallt += o
    top = ', '.join('%s %.0f%%' % (n, 100.0 * v / max(tot, 1)) for n, v in by.most_common(3))
    print('  %3d  %.2f | %3.0f%% | %s' % (b, tot / bucket, 100.0 * smallt / max(tot, 1), top))
