"""Micro-benchmark of the thin-channel convolutions (conv_thin.hip) on the shapes of the training step, through the generic C-ABI
entry points (the real dispatch): forward / data gradient / weight gradient, HIP events, algorithmic TFLOP/s and the HBM floor.
    python tools/bench_thin.py        (HOIG_NO_THIN=1 for the previous kernels)"""
import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hoig_amd import ops, _lib as L
SHAPES = [  # B, Ci, Co, H, k       (stride 1, 'same')
    (8, 64, 1, 256, 7), (8, 64, 3, 256, 7), (16, 64, 3, 256, 7),
    (8, 3, 64, 256, 7), (16, 3, 64, 256, 7), (16, 8, 64, 256, 7),
    (8, 3, 128, 32, 3), (16, 12, 128, 32, 3), (8, 3, 64, 256, 3),
]

def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

st = torch.cuda.current_stream().cuda_stream
p = lambda t: t.data_ptr()
print('%-28s %10s %10s %10s   (us; HBM floor of the big tensor at 5 TB/s)' % ('shape', 'fwd', 'dgrad', 'wgrad'))
for (B, Ci, Co, H, k) in SHAPES:
    x = torch.randn(B, H, H, Ci, device='cuda'); dy = torch.randn(B, H, H, Co, device='cuda')
    w = ops.pack_weight(torch.randn(Co, Ci, k, k, device='cuda') * 0.02)
    y = torch.empty_like(dy); dx = torch.empty_like(x); dw = torch.zeros_like(w)
    mk = lambda prec: L.ConvDesc(B, H, H, Ci, H, H, Co, k, k, 1, k // 2, 0, 0, 0.0, prec)
    df, dd = mk(L.PREC_BF16X3), mk(L.PREC_F16X2)
    tf = timeit(lambda: L.call('hoig_conv2d_fwd', ctypes.byref(df), p(x), p(w), None, p(y), st))
    td = timeit(lambda: L.call('hoig_conv2d_bwd_data', ctypes.byref(dd), p(dy), p(w), p(dx), st))
    tw = timeit(lambda: L.call('hoig_conv2d_bwd_weight', ctypes.byref(dd), p(x), p(dy), p(dw), None, st))
    big = B * H * H * max(Ci, Co) * 4
    print('%2d %3d->%3d @%3d k%d          %10.1f %10.1f %10.1f   floor %.1f' % (B, Ci, Co, H, k, tf, td, tw, big / 5e12 * 1e6))
