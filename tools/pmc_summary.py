"""Summarise rocprofv3 --pmc counter_collection CSVs for one kernel (TEST / PROFILING INFRASTRUCTURE).
    python tools/pmc_summary.py <kernel substring> <out.json> name=path_to_counter_collection.csv ...
Each CSV comes from its own `rocprofv3 --kernel-trace --pmc <COUNTERS> --output-format csv` pass (FETCH_SIZE and WRITE_SIZE do
not fit one pass on gfx950).  Traffic = (2 * FETCH_SIZE + WRITE_SIZE) KB: on gfx950 FETCH_SIZE tallies 128-B read requests at
64 B (MI355X_MICROARCH.md, HBM section); the counters sit on the L2's memory side, Infinity-Cache hits included."""
import csv, json, sys
from collections import defaultdict

kern, out = sys.argv[1], sys.argv[2]
res = {'kernel_filter': kern}
for arg in sys.argv[3:]:
    name, path = arg.split('=', 1)
    sums, n, dur = defaultdict(float), defaultdict(int), []
    seen = set()
    for r in csv.DictReader(open(path)):
        if kern != 'ALL' and kern not in r['Kernel_Name']:
            continue
        sums[r['Counter_Name']] += float(r['Counter_Value'])
        n[r['Counter_Name']] += 1
        if r['Dispatch_Id'] not in seen:
            seen.add(r['Dispatch_Id'])
            dur.append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    # one kernel: per-launch averages; ALL: totals over every dispatch of the run
    res[name] = {k: (sums[k] if kern == 'ALL' else sums[k] / max(n[k], 1)) for k in sums}
    res[name]['launches'] = len(seen)
    res[name]['avg_kernel_us'] = (sum(dur) if kern == 'ALL' else sum(dur) / max(len(dur), 1)) / 1e3
f = res.get('fetch', {}).get('FETCH_SIZE')
w = res.get('write', {}).get('WRITE_SIZE')
if f is not None and w is not None:
    res['traffic_bytes_total' if kern == 'ALL' else 'traffic_bytes_per_launch'] = (2.0 * f + w) * 1024.0
    res['correction'] = 'traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE counts 64 B per 128-B request)'
sq = res.get('sq')
if sq and 'SQ_VALU_MFMA_BUSY_CYCLES' in sq and 'GRBM_GUI_ACTIVE' in sq:
    cyc = sq['GRBM_GUI_ACTIVE']
    res['mfma_pipe_busy_fraction'] = sq['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * cyc / 8.0) if cyc else None
    res['in_kernel_clock_GHz'] = cyc / 8.0 / (sq['avg_kernel_us'] * 1e3) if sq['avg_kernel_us'] else None
json.dump(res, open(out, 'w'), indent=1)
print(json.dumps(res, indent=1))
