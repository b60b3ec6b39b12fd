"""Print a rocprofv3 kernel_stats.csv compactly: python tools/kstats.py <csv> [substring ...]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pats = sys.argv[2:]
for r in rows:
    n = re.sub(r'\(anonymous namespace\)::', '', r['Name'])
    if pats and not any(p in n for p in pats):
        continue
    print('%6.2f%% %6d calls %9.1f us avg  %s' % (float(r['Percentage']), int(r['Calls']), float(r['AverageNs']) / 1e3, n[:80]))
