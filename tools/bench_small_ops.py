"""Back-to-back launch time of the step's small HBM / latency-bound operators at the shapes the training step runs them with
(batch 8 at 256^2): effective bandwidth = algorithmic bytes / time.  usage: python tools/bench_small_ops.py"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hoig_amd import ops, _lib as L

dev = torch.device('cuda')
st = lambda: torch.cuda.current_stream().cuda_stream
p = lambda t: None if t is None else t.data_ptr()


def timed(name, nbytes, fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    print('%-58s %8.1f us  %6.2f TB/s' % (name, us, nbytes / us / 1e6), flush=True)


r = lambda *s: torch.randn(*s, device=dev)
for B, H, C in ((8, 32, 128), (16, 32, 128), (8, 64, 128), (8, 128, 128), (8, 32, 1024), (16, 32, 512)):
    y, dy, g, db = r(B, H, H, C), r(B, H, H, C), r(B, H, H, C), torch.zeros(C, device=dev)
    timed('act_bwd_colsum [%d,%d,%d,%d]' % (B, H, H, C), 3 * y.numel() * 4,
          lambda: L.call('hoig_act_bwd_colsum', p(y), p(dy), p(g), p(db), L.ACT_RELU, 0.0, y.numel() // C, C, st()))
    timed('colsum_accum   [%d,%d,%d,%d]' % (B, H, H, C), y.numel() * 4,
          lambda: L.call('hoig_colsum_accum', p(dy), p(db), y.numel() // C, C, st()))
for B, H, C, pad in ((8, 32, 512, 2), (8, 32, 512, 4), (8, 64, 256, 2), (8, 128, 128, 2)):
    dyp, dx, add = r(B, H + 2 * pad, H + 2 * pad, C), r(B, H, H, C), r(B, H, H, C)
    x, yp = r(B, H, H, C), r(B, H + 2 * pad, H + 2 * pad, C)
    timed('replicate_pad_bwd_add [%d,%d,%d,%d] pad %d' % (B, H, H, C, pad), (dyp.numel() + 2 * dx.numel()) * 4,
          lambda: L.call('hoig_replicate_pad_bwd_add', p(dyp), p(add), p(dx), B, H, H, C, pad, st()))
    timed('replicate_pad_fwd     [%d,%d,%d,%d] pad %d' % (B, H, H, C, pad), (yp.numel() + x.numel()) * 4,
          lambda: L.call('hoig_replicate_pad_fwd', p(x), p(yp), B, H, H, C, pad, st()))
for B in (8, 16):
    x = r(B, 32, 32, 512).requires_grad_(True)
    w, b = r(512).requires_grad_(True), r(512).requires_grad_(True)
    res = r(B, 32, 32, 512)
    timed('instance_norm fwd (affine, relu) [%d,32,32,512]' % B, 2 * x.numel() * 4, lambda: ops.instance_norm(x.detach(), w.detach(), b.detach(), act=L.ACT_RELU))
    timed('instance_norm fwd (affine, +res) [%d,32,32,512]' % B, 3 * x.numel() * 4, lambda: ops.instance_norm(x.detach(), w.detach(), b.detach(), residual=res))
    yv = ops.instance_norm(x, w, b, act=L.ACT_RELU)
    gy = r(B, 32, 32, 512)
    timed('instance_norm bwd (affine, relu) [%d,32,32,512]' % B, 3 * x.numel() * 4, lambda: torch.autograd.grad(yv, x, gy, retain_graph=True))
for B, H, C in ((8, 128, 128), (8, 256, 64), (16, 256, 64)):
    x = r(B, H, H, C).requires_grad_(True)
    w, b = r(C).requires_grad_(True), r(C).requires_grad_(True)
    timed('instance_norm fwd (affine, relu) [%d,%d,%d,%d]' % (B, H, H, C), 3 * x.numel() * 4, lambda: ops.instance_norm(x.detach(), w.detach(), b.detach(), act=L.ACT_RELU))
    yv = ops.instance_norm(x, w, b, act=L.ACT_RELU)
    gy = r(B, H, H, C)
    timed('instance_norm bwd (affine, relu) [%d,%d,%d,%d]' % (B, H, H, C), 5 * x.numel() * 4, lambda: torch.autograd.grad(yv, x, gy, retain_graph=True))
a, b2 = r(8, 32, 32, 512), r(8, 32, 32, 512)
timed('add [8,32,32,512]', 3 * a.numel() * 4, lambda: ops.add(a, b2))
