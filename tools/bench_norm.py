"""Instance-norm kernels at the shapes of the training step: forward and backward time, bytes moved per launch and the rate (one
stream, HIP events).  Bytes: forward reads x (+ gamma|beta for SPADE) and writes y; backward reads x, dy (+ y or the SPADE parameters)
and writes dx (+ dgamma|dbeta).    python tools/bench_norm.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hoig_amd import ops, _lib as L          # noqa: E402

ops.set_precision('bf16x3:f16x2')
SHAPES = [(8, 32, 32, 512), (16, 32, 32, 512), (8, 64, 64, 256), (16, 64, 64, 256), (16, 128, 128, 128), (8, 128, 128, 128),
          (16, 256, 256, 64), (8, 256, 256, 64)]


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


print('%-22s %-7s %9s %9s %9s %9s' % ('shape', 'kind', 'fwd us', 'TB/s', 'bwd us', 'TB/s'))
for B, H, W, C in SHAPES:
    n = B * H * W * C * 4
    for kind in ('in_relu', 'spade'):
        x = torch.randn(B, H, W, C, device='cuda', requires_grad=True)
        g = torch.rand(C, device='cuda', requires_grad=True)
        b = torch.rand(C, device='cuda', requires_grad=True)
        gb = torch.randn(B, H, W, 2 * C, device='cuda', requires_grad=True) * 0.1 if kind == 'spade' else None
        if kind == 'spade':
            gb = gb.detach().requires_grad_(True)
            fwd = lambda: ops.spade_norm_fused(x, gb, act=L.ACT_RELU)
            fb, bb = n * 4, n * 7            # x + 2 gb + y ; x, dy, y, gb(gamma) + dx + 2 dgb
        else:
            fwd = lambda: ops.instance_norm(x, g, b, act=L.ACT_RELU)
            fb, bb = n * 2, n * 3            # x + y ; x, dy + dx (the ReLU mask is recomputed)
        with torch.no_grad():
            tf = timeit(fwd)
        y = fwd()
        dy = torch.randn_like(y)
        tb = timeit(lambda: torch.autograd.grad(y, [x] + ([gb] if gb is not None else []), dy, retain_graph=True))
        print('%2d x %3dx%3d x %4d    %-7s %9.1f %9.2f %9.1f %9.2f' % (B, H, W, C, kind, tf, fb / tf / 1e6, tb, bb / tb / 1e6))
