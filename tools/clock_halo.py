"""In-kernel clock of the step's dominant launch (conv3x3 s1 512->512 at 32x32, 16 images, split mode): the diagnostic build
HOIG_STAMP=2 (tools/build_stamp.sh) executes two stamps around the step loop, s_memtime (shader cycles) and s_memrealtime
(100 MHz); clock = their quotient, taken after >= 2 s of back-to-back launches (MI355X_MICROARCH.md 'DVFS give-back' item 6).
Prints the clock, the loop's share of MFMA-pipe cycles at that clock, and the same on zero-filled operands."""
import ctypes
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hoig_amd._lib import ConvDesc, ACT_NONE, PREC_BF16X3      # noqa: E402 (structure layout only)

lib = ctypes.CDLL(os.path.join(ROOT, 'tools', '_build', 'libhoig_hip_stamp2.so'))
B, C, H = 16, 512, 32
vp = ctypes.c_void_p
st = vp(torch.cuda.current_stream().cuda_stream)
nblk, waves = B * (H // 8) * (H // 32) * (C // 128), 8
steps = (C // 32) * 3
mfma_cycles = steps * 6 * 12 * 32 * 2          # per SIMD: two waves x 72 MFMAs of 32 cycles per step


def measure(zero):
    x = torch.zeros(B, H, H, C, device='cuda') if zero else torch.randn(B, H, H, C, device='cuda')
    w = (torch.zeros(C, 3, 3, C, device='cuda') if zero else torch.randn(C, 3, 3, C, device='cuda') * 0.02).contiguous()
    hi = torch.empty(w.numel(), dtype=torch.int16, device='cuda')
    lo = torch.empty_like(hi)
    y = torch.empty(B, H, H, C, device='cuda')
    assert lib.hoig_pack_conv_weight_bf16(vp(w.data_ptr()), C, 9, C, 0, vp(hi.data_ptr()), vp(lo.data_ptr()), st) == 0
    d = ConvDesc(B, H, H, C, H, H, C, 3, 3, 1, 1, 0, ACT_NONE, 0.0, PREC_BF16X3)
    dbg = torch.zeros(nblk * waves * 8, dtype=torch.int64, device='cuda')

    def run():
        assert lib.hoig_conv2d_fwd_packed(ctypes.byref(d), vp(x.data_ptr()), vp(hi.data_ptr()), vp(lo.data_ptr()), None,
                                          vp(y.data_ptr()), st) == 0
    lib.hoig_debug_set_stamp_buffer(None)
    t0 = time.time()
    while time.time() - t0 < 2.5:
        for _ in range(200):
            run()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        run()
    e1.record()
    lib.hoig_debug_set_stamp_buffer(vp(dbg.data_ptr()))
    run()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 200 * 1e3
    t = dbg.cpu().numpy().reshape(nblk * waves, 8).astype(np.float64)
    cyc, rt = t[:, 5], t[:, 6]
    clk = np.median(cyc / rt) * 0.1                               # GHz
    fl = 2.0 * B * H * H * C * C * 9
    print('%-6s launch %.1f us (%.0f TF/s effective, %.0f TF/s of split MFMAs); loop %.0f cycles in %.1f us: clock %.2f GHz; '
          'MFMA pipe busy %.0f %% of the loop at that clock; peak at that clock %.0f TF/s effective'
          % ('zeros' if zero else 'random', us, fl / us / 1e6, 3 * fl / us / 1e6, np.median(cyc), np.median(rt) / 100.0, clk,
             100.0 * mfma_cycles / np.median(cyc), 2500.0 / 3 * clk / 2.4))


measure(False)
measure(True)
measure(False)
