"""micro-benchmark: cat_channels of the discriminator input (TEST INFRASTRUCTURE)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hoig_amd import ops
for shp in ((8, 256, 256, 3, 16), (8, 256, 256, 64, 64), (16, 256, 256, 8, 3)):
    B, H, W, c1, c2 = shp
    a, b = torch.randn(B, H, W, c1, device='cuda'), torch.randn(B, H, W, c2, device='cuda')
    for _ in range(3):
        ops.cat_channels([a, b])
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(20):
        ops.cat_channels([a, b])
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 20
    print(shp, '%.1f us  %.2f TB/s' % (ms * 1e3, 2 * B * H * W * (c1 + c2) * 4 / ms / 1e9))
