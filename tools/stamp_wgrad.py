"""Diagnostic: per-wave cycle sums of the phases of the 3x3 weight-gradient halo kernel (HOIG_STAMP build, tools/build_stamp.sh):
issue (global loads of the next pixel tile) / compute (transpose reads + MFMA) / barrier / publish (split + ds_write) / barrier,
the loop's share of the kernel and the clock.  Usage: python tools/stamp_wgrad.py [images] [stamp level]"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hoig_amd._lib import ConvDesc, ACT_NONE, PREC_BF16X3, PREC_F16X2      # noqa: E402 (structure layout only)

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
level = sys.argv[2] if len(sys.argv) > 2 else '1'
lib = ctypes.CDLL(os.path.join(ROOT, 'tools', '_build', 'libhoig_hip_stamp%s.so' % level))
C, H = 512, 32
x = torch.randn(B, H, H, C, device='cuda')
dy = torch.randn(B, H, H, C, device='cuda')
dw = torch.zeros(C, 3, 3, C, device='cuda')
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
vp = ctypes.c_void_p
prec = PREC_BF16X3 if len(sys.argv) > 3 and sys.argv[3] == 'x3' else PREC_F16X2          # the shipped weight-gradient arithmetic: two terms
d = ConvDesc(B, H, H, C, H, H, C, 3, 3, 1, 1, 0, ACT_NONE, 0.0, prec)
nwg, waves = 4096, 16                                   # (an upper bound: unstamped rows stay zero and are dropped)
dbg = torch.zeros(nwg * waves * 8, dtype=torch.int64, device='cuda')


def run():
    rc = lib.hoig_conv2d_bwd_weight(ctypes.byref(d), vp(x.data_ptr()), vp(dy.data_ptr()), vp(dw.data_ptr()), None, st)
    assert rc == 0, rc


lib.hoig_debug_set_stamp_buffer(None)
for _ in range(300):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    run()
e1.record()
lib.hoig_debug_set_stamp_buffer(vp(dbg.data_ptr()))
run()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 50 * 1e3
t = dbg.cpu().numpy().reshape(nwg * waves, 8).astype(np.float64)
t = t[t[:, 6] > 0]
fl = 2.0 * B * H * H * C * C * 9
print('B=%d: %.1f us per launch (%.0f TF/s); %d stamped waves; kernel cycles per wave median %.0f, loop %.0f (%.0f %%); clock %.2f GHz'
      % (B, us, fl / us / 1e6, len(t), np.median(t[:, 6]), np.median(t[:, 5]), 100 * np.median(t[:, 5] / t[:, 6]),
         np.median(t[:, 6] / t[:, 7]) * 0.1))
for i, n in enumerate(['issue', 'compute', 'barrier1', 'publish', 'barrier2']):
    print('  %-9s %5.1f %% of the loop' % (n, 100 * np.median(t[:, i] / t[:, 5])))
th = int(os.environ.get('HOIG_WGRAD_HALO_TH', '4'))
mt = B * (H // th) // 4                                 # pixel tiles per workgroup (four pixel splits)
nm = (2 if prec == PREC_F16X2 else 3) * 3 * 2 * th     # MFMAs per wave and pixel tile
print('  per pixel tile: %.0f cycles per wave; %d MFMAs per wave and tile = %d cycles per SIMD with 12 waves per CU'
      % (np.median(t[:, 5]) / mt, nm, nm * 32 * 3))
