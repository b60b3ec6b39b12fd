"""Interleaved A/B of the 7x7 heads' forward: tuning key 'head16' 0 (fp32 VALU kernel) against 1 (conv_head16.hip), on random data,
at the shapes of the training step (8 / 16 images) and of the batch-32 generator forward.
usage: python tools/ab_heads.py [rounds]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hoig_amd import _lib as L, ops            # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
ops.set_precision('bf16x3')
SHAPES = [(8, 5, 256), (16, 5, 256), (16, 3, 256), (32, 5, 256), (64, 3, 256), (4, 5, 512)]
for B, Co, S in SHAPES:
    x = torch.randn(B, S, S, 64, device='cuda').abs()
    w = ops.pack_weight(torch.randn(Co, 64, 7, 7, device='cuda') * 0.02)
    splits, acts = ((3, Co - 3), (L.ACT_TANH, L.ACT_SIGMOID)) if Co > 3 else ((3,), (L.ACT_TANH,))
    t = {0: [], 1: []}
    for r in range(rounds + 1):
        for v in (0, 1):
            L.set_tuning('head16', v)
            with torch.no_grad():
                ops.conv_heads(x, w, splits, acts)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    ops.conv_heads(x, w, splits, acts)
                e1.record()
            torch.cuda.synchronize()
            if r:
                t[v].append(e0.elapsed_time(e1) * 100)          # us per call (includes the per-head channel copies)
    gf = 2.0 * B * S * S * 49 * 64 * Co / 1e9
    a, b = sorted(t[0])[len(t[0]) // 2], sorted(t[1])[len(t[1]) // 2]
    print('heads fwd %2d img %dx%d 64->%d   valu %7.1f us (%5.1f TF)   mfma %7.1f us (%5.1f TF)   x%.2f' % (B, S, S, Co, a, gf / a * 1e3, b, gf / b * 1e3, a / b),
          flush=True)
    del x
L.set_tuning('head16', 1)
