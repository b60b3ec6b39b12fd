"""Times the local-attention op (forward / backward) at the three generator resolutions, B=8."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hoig_amd import ops
from hoig_amd.nn import split_attn_weight
ops.set_precision(sys.argv[1] if len(sys.argv) > 1 else 'bf16x3')
for (C, h) in [(512, 32), (256, 64), (128, 128)]:
    B = 8
    src = torch.randn(B, h, h, C, device='cuda', requires_grad=True)
    tgt = torch.randn(B, h, h, C, device='cuda', requires_grad=True)
    flow = torch.rand(B, 2, h, h, device='cuda') * 4 - 3
    w1 = torch.randn(128, 2 * C, 5, 5, device='cuda') * 0.02
    wt, ws = [ops.pack_weight(t.contiguous()).requires_grad_(True) for t in split_attn_weight(w1)]
    b1 = torch.zeros(128, device='cuda', requires_grad=True)
    w2 = ops.pack_weight(torch.randn(25, 128, 1, 1, device='cuda') * 0.1).requires_grad_(True)
    b2 = torch.zeros(25, device='cuda', requires_grad=True)
    gy = torch.randn(B, h, h, C, device='cuda')
    def fwd():
        return ops.local_attention(src, tgt, flow, wt, ws, b1, w2, b2)
    for _ in range(2):
        fwd().backward(gy)
    s, e, e2 = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    torch.cuda.synchronize(); tf = tb = 0.0
    for _ in range(5):
        s.record(); y = fwd(); e.record(); y.backward(gy); e2.record(); torch.cuda.synchronize()
        tf += s.elapsed_time(e); tb += e.elapsed_time(e2)
    gf = 2.0 * B * h * h * 128 * 50 * C / 1e9
    print('C=%d h=%d  fwd %.3f ms  bwd %.3f ms  (fc1 %.1f GF)' % (C, h, tf / 5, tb / 5, gf), flush=True)
