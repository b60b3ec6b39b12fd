"""Runs only the step's dominant kernel (conv3x3 s1 512->512 at 32x32, 16 images = src+tsf stacked) a few times, for the
rocprofv3 --pmc passes that measure its HBM traffic (FETCH_SIZE / WRITE_SIZE need separate passes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hoig_amd import ops

prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16x3'
ops.set_precision(prec)
x = torch.randn(16, 32, 32, 512, device='cuda')
w = ops.pack_weight(torch.randn(512, 512, 3, 3, device='cuda') * 0.02)
for _ in range(10):
    y = ops.conv2d(x, w, None, 1, 1)
torch.cuda.synchronize()
print('ok', float(y.abs().mean()))
