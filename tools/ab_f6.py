"""The f16f6 forward kernel (hoig_conv2d_fwd_f6_ex, conv_f6.hip) against the three-term kernel (hoig_conv2d_fwd_packed) on single launches:
interleaved rounds in one process, random data, HIP events.   python tools/ab_f6.py [rounds] [iters]"""
import ctypes
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hoig_amd import ops, _lib as L          # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
SHAPES = [(16, 512, 512, 32, 32), (32, 512, 512, 32, 32), (64, 512, 512, 32, 32), (32, 256, 256, 64, 64), (32, 128, 128, 128, 128),
          (32, 256, 128, 128, 128), (32, 64, 64, 256, 256)]
ops.set_precision('bf16x3:f16x2')
st = torch.cuda.current_stream().cuda_stream
p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
for B, Ci, Co, H, W in SHAPES:
    x = torch.randn(B, H, W, Ci, device='cuda')
    w = ops.pack_weight(torch.randn(Co, Ci, 3, 3, device='cuda') * 0.02)
    y = torch.empty(B, H, W, Co, device='cuda')
    sc, sh = torch.rand(B, Ci, device='cuda') + 0.5, torch.randn(B, Ci, device='cuda')
    sums = torch.zeros(B, 2, Co, device='cuda')
    d3 = L.ConvDesc(B, H, W, Ci, H, W, Co, 3, 3, 1, 1, 0, 0, 0.0, L.PREC_BF16X3)
    d6 = L.ConvDesc(B, H, W, Ci, H, W, Co, 3, 3, 1, 1, 0, 0, 0.0, L.PREC_F16F6)
    hi, lo = ops._packed_planes(w, False, False)
    qh, ql = ops._f6_planes(w)
    fns = {'x3': lambda: L.call('hoig_conv2d_fwd_packed', ctypes.byref(d3), p(x), p(hi), p(lo), None, p(y), st),
           'f6': lambda: L.call('hoig_conv2d_fwd_f6_ex', ctypes.byref(d6), p(x), 0, None, p(hi), p(qh), p(ql), None, None, None, 0, p(y), None, st),
           'f6+normin+sums': lambda: L.call('hoig_conv2d_fwd_f6_ex', ctypes.byref(d6), p(x), 0, None, p(hi), p(qh), p(ql), None, p(sc), p(sh), 0,
                                            p(y), p(sums), st)}
    times = {k: [] for k in fns}
    for r in range(rounds + 1):
        for k, fn in fns.items():
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            if r:
                times[k].append(e0.elapsed_time(e1) / iters * 1e3)
    gf = 2.0 * B * H * W * Co * Ci * 9 / 1e9
    m = {k: statistics.median(v) for k, v in times.items()}
    print('fwd %2d %3dx%-3d %4d->%-4d | three terms %7.1f us %6.1f TF | f16f6 %7.1f us %6.1f TF (%+5.1f %%) | + norm-in + sums %7.1f us' % (
        B, H, W, Ci, Co, m['x3'], gf / m['x3'] * 1e3, m['f6'], gf / m['f6'] * 1e3, (m['f6'] / m['x3'] - 1) * 100, m['f6+normin+sums']), flush=True)
