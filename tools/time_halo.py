"""Times the 3x3 stride-1 512->512 halo convolution at 32x32 (forward launches, split mode) for a few image counts."""
import os
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hoig_amd import ops

ops.set_precision(sys.argv[1] if len(sys.argv) > 1 else 'bf16x3')
for B, C, H in ((16, 512, 32), (8, 512, 32), (32, 512, 32), (8, 256, 64), (8, 128, 128)):
    x = torch.randn(B, H, H, C, device='cuda')
    w = ops.pack_weight(torch.randn(C, C, 3, 3, device='cuda') * 0.02)
    w._hoig_owner = types.SimpleNamespace(version=0, packed_planes=lambda w_, for_dgrad: None)
    for _ in range(10):
        y = ops.conv2d(x, w, None, 1, 1)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            y = ops.conv2d(x, w, None, 1, 1)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 50)
    fl = 2.0 * B * H * H * C * C * 9
    print('B=%d C=%d H=%d: %.1f us  %.1f TF/s' % (B, C, H, best * 1e3, fl / best / 1e9))
