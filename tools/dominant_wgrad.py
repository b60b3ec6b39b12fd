"""Runs only the step's dominant WEIGHT-GRADIENT launch (conv3x3 s1 512->512 at 32x32, 16 images; wgrad_halo_bf16_kernel<3,3,2,false,4>
in the default bf16x3:f16x2 arithmetic) a few times, for rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE / SQ counters in separate passes)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hoig_amd import ops, _lib as L

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
SPLIT = len(sys.argv) > 2 and sys.argv[2] == 'split'        # pre-split dy on the LDS-DMA kernel (wgrad_dma.hip)
x = torch.randn(B, 32, 32, 512, device='cuda')
dy = torch.randn(B, 32, 32, 512, device='cuda')
w = ops.pack_weight(torch.randn(512, 512, 3, 3, device='cuda') * 0.02)
dw = torch.zeros_like(w)
d = L.ConvDesc(B, 32, 32, 512, 32, 32, 512, 3, 3, 1, 1, 0, 0, 0.0, L.PREC_F16X2)
st = torch.cuda.current_stream().cuda_stream
dys = torch.empty(B, 32, 32, 2, 512, dtype=torch.bfloat16, device='cuda')
L.call('hoig_split_planes_bf16', dy.data_ptr(), dys.data_ptr(), B * 32 * 32, 512, st)
for _ in range(10):
    if SPLIT:
        L.call('hoig_conv2d_bwd_weight_split', ctypes.byref(d), x.data_ptr(), dys.data_ptr(), dw.data_ptr(), st)
    else:
        L.call('hoig_conv2d_bwd_weight', ctypes.byref(d), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), None, st)
torch.cuda.synchronize()
print('ok', float(dw.abs().mean()))
