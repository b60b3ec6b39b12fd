"""Diagnostic: per-wave cycle sums of the phases of the 3x3 halo conv's step loop (tools/build_stamp.sh build).
phases: issue (global loads of the next step) / compute (LDS reads + MFMA) / barrier 1 / publish (ds_write) / barrier 2."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hoig_amd._lib import ConvDesc, ACT_NONE, PREC_BF16X3      # noqa: E402 (structure layout only)

lib = ctypes.CDLL(os.path.join(ROOT, 'tools', '_build', 'libhoig_hip_stamp.so'))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
C, H = 512, 32
x = torch.randn(B, H, H, C, device='cuda')
w = (torch.randn(C, 3, 3, C, device='cuda') * 0.02).contiguous()      # packed [Co][R][S][Ci]
hi = torch.empty(w.numel(), dtype=torch.int16, device='cuda')
lo = torch.empty_like(hi)
y = torch.empty(B, H, H, C, device='cuda')
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
vp = ctypes.c_void_p
assert lib.hoig_pack_conv_weight_bf16(vp(w.data_ptr()), C, 9, C, 0, vp(hi.data_ptr()), vp(lo.data_ptr()), st) == 0
d = ConvDesc(B, H, H, C, H, H, C, 3, 3, 1, 1, 0, ACT_NONE, 0.0, PREC_BF16X3)
nblk = B * (H // 4) * (H // 32) * (C // 128)
waves = 4 if nblk >= 384 else 8
tall = nblk // 2 >= 256                    # launch_halo3: 8 x 32-pixel tiles on eight waves, weights double-buffered
if tall:
    nblk, waves = nblk // 2, 8
dbg = torch.zeros(nblk * waves * 8, dtype=torch.int64, device='cuda')


def run():
    rc = lib.hoig_conv2d_fwd_packed(ctypes.byref(d), vp(x.data_ptr()), vp(hi.data_ptr()), vp(lo.data_ptr()), None,
                                    vp(y.data_ptr()), st)
    assert rc == 0, rc


for _ in range(20):
    run()
torch.cuda.synchronize()
lib.hoig_debug_set_stamp_buffer(vp(dbg.data_ptr()))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record()
torch.cuda.synchronize()
t = dbg.cpu().numpy().reshape(nblk * waves, 8)
names = ['issue', 'compute', 'barrier1', 'publish', 'barrier2']
tot = t[:, 5].astype(np.float64)
print('B=%d  %d workgroups x %d waves; kernel %.1f us; loop cycles per wave: median %.0f (min %.0f max %.0f)'
      % (B, nblk, waves, e0.elapsed_time(e1) * 1e3, np.median(tot), tot.min(), tot.max()))
for i, n in enumerate(names):
    v = t[:, i].astype(np.float64)
    print('  %-9s median %8.0f cycles  %5.1f %% of the loop' % (n, np.median(v), 100 * np.median(v / tot)))
steps = (C // 32) * 3
print('  per step: %.0f cycles total, %.0f compute (MFMA-only floor %d)' % (np.median(tot) / steps, np.median(t[:, 1]) / steps,
                                                                         3 * 2 * (12 if (waves == 4 or tall) else 6) * 32))
