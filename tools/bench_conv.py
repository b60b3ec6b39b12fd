"""Micro-benchmark of the convolution kernels on the shapes of one HOGAN forward, B=8, 256x256.
Times fwd / dgrad / wgrad separately with HIP events on the launch stream; prints algorithmic TFLOP/s."""
import sys
import os
import ctypes
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hoig_amd import ops, _lib as L

SHAPES = [  # Ci, Co, H, k, stride, pad, transposed
    (512, 512, 32, 3, 1, 1, False),
    (128, 512, 32, 3, 1, 1, False),
    (128, 128, 128, 3, 1, 1, False),
    (64, 128, 256, 3, 2, 1, False),
    (256, 512, 64, 3, 2, 1, False),
    (512, 256, 32, 3, 2, 1, True),
    (128, 64, 128, 3, 2, 1, True),
    (512, 256, 64, 3, 1, 1, False),
    (128, 64, 256, 3, 1, 1, False),
    (64, 64, 256, 3, 1, 1, False),
]


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    B = int(os.environ.get('B', 8))
    precs = sys.argv[1:] or ['f32', 'bf16x3']
    only = os.environ.get('ONLY')
    for (Ci, Co, H, k, st, pad, tr) in SHAPES[:int(only)] if only else SHAPES:
        x = torch.randn(B, H, H, Ci, device='cuda')
        wl = torch.randn((Ci, Co, k, k) if tr else (Co, Ci, k, k), device='cuda') * 0.02
        w = ops.pack_weight(wl, transposed=tr)
        Ho = H * 2 if tr else (H + 2 * pad - k) // st + 1
        flops = 2.0 * B * (H * H if tr else Ho * Ho) * Co * Ci * k * k
        dy = torch.randn(B, Ho, Ho, Co, device='cuda')
        dw = torch.zeros_like(w)
        dx = torch.empty_like(x)
        y = torch.empty_like(dy)
        line = '%4d->%4d @%3d k%d s%d %s  GF %7.2f |' % (Ci, Co, H, k, st, 'T' if tr else ' ', flops / 1e9)
        for pn in precs:
            prec = ops._PREC[pn]
            d = L.ConvDesc(B, H, H, Ci, Ho, Ho, Co, k, k, st, pad, 1 if tr else 0, 0, 0.0, prec)
            p = lambda t: t.data_ptr()
            stream = torch.cuda.current_stream().cuda_stream
            if prec == L.PREC_F32:
                f = lambda: L.call('hoig_conv2d_fwd', ctypes.byref(d), p(x), p(w), None, p(y), stream)
                b = lambda: L.call('hoig_conv2d_bwd_data', ctypes.byref(d), p(dy), p(w), p(dx), stream)
            else:
                hi, lo = ops._packed_planes(w, tr, False)
                thi, tlo = ops._packed_planes(w, tr, True)
                f = lambda: L.call('hoig_conv2d_fwd_packed', ctypes.byref(d), p(x), p(hi), p(lo), None, p(y), stream)
                b = lambda: L.call('hoig_conv2d_bwd_data_packed', ctypes.byref(d), p(dy), p(thi), p(tlo), p(dx), stream)
            g = lambda: L.call('hoig_conv2d_bwd_weight', ctypes.byref(d), p(x), p(dy), p(dw), None, stream)
            tf, tb, tg = timeit(f), timeit(b), timeit(g)
            line += ' %s fwd %6.1f dgrad %6.1f wgrad %6.1f TF |' % (pn, flops / tf / 1e9, flops / tb / 1e9, flops / tg / 1e9)
        print(line, flush=True)


if __name__ == '__main__':
    main()
