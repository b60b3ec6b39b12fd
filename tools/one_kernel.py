"""Runs one conv kernel class a few times (for rocprofv3 --pmc passes): python tools/one_kernel.py {fwd|dgrad|wgrad} [prec]"""
import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hoig_amd import ops, _lib as L
which = sys.argv[1]
prec = ops._PREC[sys.argv[2] if len(sys.argv) > 2 else 'bf16x3']
B, H, Ci, Co = 8, 32, 512, 512
x = torch.randn(B, H, H, Ci, device='cuda'); dy = torch.randn(B, H, H, Co, device='cuda')
w = ops.pack_weight(torch.randn(Co, Ci, 3, 3, device='cuda') * 0.02)
d = L.ConvDesc(B, H, H, Ci, H, H, Co, 3, 3, 1, 1, 0, 0, 0.0, prec)
y = torch.empty_like(dy); dx = torch.empty_like(x); dw = torch.zeros_like(w)
st = torch.cuda.current_stream().cuda_stream
p = lambda t: t.data_ptr()
hi, lo = ops._packed_planes(w, False, False); thi, tlo = ops._packed_planes(w, False, True)
for _ in range(6):
    if which == 'fwd': L.call('hoig_conv2d_fwd_packed', ctypes.byref(d), p(x), p(hi), p(lo), None, p(y), st)
    elif which == 'dgrad': L.call('hoig_conv2d_bwd_data_packed', ctypes.byref(d), p(dy), p(thi), p(tlo), p(dx), st)
    else: L.call('hoig_conv2d_bwd_weight', ctypes.byref(d), p(x), p(dy), p(dw), None, st)
torch.cuda.synchronize(); print('done')
