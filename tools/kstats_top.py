"""Top kernels of a rocprofv3 --stats kernel_stats.csv, per training step.  python tools/kstats_top.py <csv> <steps> [n]"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]); n = int(sys.argv[3]) if len(sys.argv) > 3 else 45
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total %.2f ms per step over %s steps' % (tot / 1e6 / steps, sys.argv[2]))
acc = 0
for r in rows[:n]:
    t = float(r['TotalDurationNs']); acc += t
    name = r['Name'].replace('(anonymous namespace)::', '').replace('at::native::', '')
    name = re.sub(r'^void ', '', name); name = re.sub(r'\(.*', '', name)
    print('%6.2f%% %6.2f%%  %6.3f ms/step  calls/step %6.1f avg %8.1f us  %s' % (100 * t / tot, 100 * acc / tot, t / 1e6 / steps,
          float(r['Calls']) / steps, float(r['AverageNs']) / 1e3, name[:90]))
