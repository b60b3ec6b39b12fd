"""Times the tile-binned rasteriser at the reference's scale (1538 hand faces + a YCB object mesh per sample, two views per
sample: here `views` images of `faces` faces at 256 x 256) and reports face tests avoided vs the reference's brute force.
Usage: python tools/bench_raster.py [views] [faces]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from hoig_amd import raster                         # noqa: E402
from common import synthetic_mesh_faces             # noqa: E402  (mesh generator only)

views = int(sys.argv[1]) if len(sys.argv) > 1 else 16
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 15000
f = synthetic_mesh_faces(views, n_random=max(0, nf - 2208), seed=4).cuda()
F = f.shape[1]
for _ in range(3):
    fim, wim = raster.rasterize_fim_wim(f, 256)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    fim, wim = raster.rasterize_fim_wim(f, 256)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
brute = views * 65536.0 * F
print('%d views x %d faces at 256x256: %.3f ms per batch (%.1f us per view); coverage %.1f %%; the reference\'s brute force '
      'is %.2f G face tests per batch' % (views, F, ms, ms / views * 1e3, 100 * float((fim >= 0).float().mean()), brute / 1e9))
