"""The Winograd forward (hoig_conv2d_fwd_wino) against the direct three-term kernel (hoig_conv2d_fwd_packed) on single launches:
interleaved rounds in one process, random data, HIP events.   python tools/ab_wino.py [rounds] [iters]"""
import ctypes
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hoig_amd import ops, _lib as L          # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 7
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
SHAPES = [(16, 512, 512, 32, 32), (8, 512, 512, 32, 32)] if os.environ.get('WINO_SHORT') else [(16, 512, 512, 32, 32), (8, 512, 512, 32, 32), (32, 512, 512, 32, 32), (16, 128, 1024, 32, 32), (8, 256, 256, 64, 64),
          (16, 128, 256, 128, 128), (8, 128, 128, 128, 128), (8, 64, 64, 256, 256)]
ops.set_precision('bf16x3:f16x2')
st = torch.cuda.current_stream().cuda_stream
p = lambda t: ctypes.c_void_p(t.data_ptr())
for B, Ci, Co, H, W in SHAPES:
    x = torch.randn(B, H, W, Ci, device='cuda')
    w = ops.pack_weight(torch.randn(Co, Ci, 3, 3, device='cuda') * 0.02)
    y = torch.empty(B, H, W, Co, device='cuda')
    d = L.ConvDesc(B, H, W, Ci, H, W, Co, 3, 3, 1, 1, 0, 0, 0.0, L.PREC_BF16X3)
    hi, lo = ops._packed_planes(w, False, False)
    n = L.lib.hoig_wino_plane_halfs(Co, Ci)
    uh, ul = torch.empty(n, dtype=torch.int16, device='cuda'), torch.empty(n, dtype=torch.int16, device='cuda')
    L.call('hoig_pack_conv_weight_wino', p(w), Co, Ci, p(uh), p(ul), st)
    fns = {'direct': lambda: L.call('hoig_conv2d_fwd_packed', ctypes.byref(d), p(x), p(hi), p(lo), None, p(y), st),
           'wino': lambda: L.call('hoig_conv2d_fwd_wino', ctypes.byref(d), p(x), p(uh), p(ul), None, p(y), st),
           'pack': lambda: L.call('hoig_pack_conv_weight_wino', p(w), Co, Ci, p(uh), p(ul), st)}
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
    flop = 2.0 * B * H * W * Ci * Co * 9
    md = {k: statistics.median(v) for k, v in times.items()}
    print('fwd %2d %3dx%-3d %4d->%-4d | direct %7.1f us %6.1f TF | wino %7.1f us %6.1f TF (%+5.1f %%) | weight pack %6.1f us'
          % (B, H, W, Ci, Co, md['direct'], flop / md['direct'] / 1e6, md['wino'], flop / md['wino'] / 1e6,
             (md['wino'] / md['direct'] - 1) * 100, md['pack']), flush=True)
