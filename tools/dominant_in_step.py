"""In-step time of one launch SHAPE (not of a kernel template, which serves several layer shapes) from a rocprofv3 kernel trace
(VERDICT r3 item 5).    python tools/dominant_in_step.py <kernel_trace.csv> <kernel-name substring> <grid size x> [GFLOP per launch]
The step's dominant launch: 3x3 stride-1 512->512 at 32x32 over 16 stacked images = 256 workgroups of 512 threads
(Grid_Size_X 131072) of conv_halo3_m16_kernel<2,4,2,128,true>; 77.31 GFLOP algorithmic.
CAVEAT (round 4): the grid does not identify that shape uniquely -- the 8-image 128 -> 1024 SPADE convolutions launch the same
template on the same 256 workgroups -- so the figure this prints averages both; the per-SHAPE in-step time that DESIGN.md quotes
comes from tools/conv_table.py, which brackets every launch with events on its own stream and keys them by the convolution
descriptor (profiles/r04_conv_table.txt: one stream; r04_conv_table_streams.txt: the multi-stream step)."""
import csv
import sys

path, name, grid = sys.argv[1], sys.argv[2], int(sys.argv[3])
gflop = float(sys.argv[4]) if len(sys.argv) > 4 else 77.309
rows = list(csv.DictReader(open(path)))
gx = 'Grid_Size_X' if 'Grid_Size_X' in rows[0] else ('Grid_Size' if 'Grid_Size' in rows[0] else None)
if gx is None:
    sys.exit('no grid-size column in %s: %s' % (path, list(rows[0])))
dur = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows if name in r['Kernel_Name'] and int(r[gx]) == grid]
if not dur:
    sys.exit('no launch of %r with grid %d' % (name, grid))
dur.sort()
avg = sum(dur) / len(dur)
print('%s, grid %d: %d launches, avg %.1f us (min %.1f, median %.1f, max %.1f) = %.1f TFLOP/s algorithmic at %.2f GFLOP per launch'
      % (name, grid, len(dur), avg, dur[0], dur[len(dur) // 2], dur[-1], gflop / avg * 1e3, gflop))
