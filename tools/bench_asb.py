"""Times hoig_attn_sample_bwd alone (C=512, 32x32, B=8 and C=128, 128x128)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hoig_amd import ops
from hoig_amd.ops import call, _p, _st
for (C, h) in [(512, 32), (128, 128)]:
    B = 8
    M = B * h * h
    flow = torch.rand(B, 2, h, h, device='cuda') * 4 - 3
    dS = torch.randn(M, 25 * C, device='cuda')
    attn = torch.softmax(torch.randn(M, 25, device='cuda'), -1)
    dout = torch.randn(M, C, device='cuda')
    dsrc = torch.zeros(B, h, h, C, device='cuda')
    def run():
        call('hoig_attn_sample_bwd', _p(flow), _p(dS), _p(attn), _p(dout), _p(dsrc), B, h, h, C, _st())
    for _ in range(3):
        run()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(10):
        run()
    e.record(); torch.cuda.synchronize()
    t = s.elapsed_time(e) / 10
    print('C=%d h=%d  %.1f us   dS %.0f MB -> %.2f TB/s' % (C, h, t * 1e3, dS.numel() * 4 / 1e6, dS.numel() * 4 / t / 1e9), flush=True)
