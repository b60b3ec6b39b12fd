"""Times hoig_amd.input_prep.prepare_inputs (HIP: 3 launches per sample + 1 per batch) for one batch of rasteriser outputs,
beside the oracle restatement of the reference's per-sample torch code on the host cores.
Usage: python tools/bench_input_prep.py [batch]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hoig_amd import synthetic, input_prep as IP          # noqa: E402
from oracle import input_prep_oracle as P                 # noqa: E402  (the CPU baseline leg)

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
r = synthetic.make_raster(B, 8)
dev = torch.device('cuda', 0)
tabs = {k: IP.ObjectTables(tb, dev) for k, tb in r['tables'].items()}
args = [r[k].to(dev) for k in ('src_img', 'ref_img', 'src_faces', 'src_fim', 'src_wim', 'ref_fim', 'ref_wim')]
tl = [tabs[k] for k in r['obj_ids']]
for _ in range(3):
    IP.prepare_inputs(*args, tl)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter()
e0.record()
for _ in range(20):
    IP.prepare_inputs(*args, tl)
e1.record()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 20 * 1e3
gpu = e0.elapsed_time(e1) / 20
# algorithmic bytes per sample: inputs (2 images, 2 fim, 2 wim) + outputs (52 planar channels + T + 4 masks), fp32
byts = B * 65536 * 4 * (6 + 2 + 6 + 4 + 15 + 15 + 6 + 6 + 2 + 4)
print('HIP  batch %d: %.3f ms per batch on the stream (%.3f ms wall incl. launches), %.1f GB/s of algorithmic bytes'
      % (B, gpu, wall, byts / gpu / 1e6))
ct = [r['tables'][k] for k in r['obj_ids']]
cargs = [r[k] for k in ('src_img', 'ref_img', 'src_faces', 'src_fim', 'src_wim', 'ref_fim', 'ref_wim')]
P.prepare_inputs(*cargs, ct)
t0 = time.perf_counter()
for _ in range(3):
    P.prepare_inputs(*cargs, ct)
cpu = (time.perf_counter() - t0) / 3 * 1e3
print('CPU  oracle (torch, %d threads): %.1f ms per batch  -> x%.0f' % (torch.get_num_threads(), cpu, cpu / wall))
