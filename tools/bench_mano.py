"""Launch time of the MANO layer (hoig_mano_lbs) beside the CPU restatement (TEST INFRASTRUCTURE: imports the oracle).
usage: python tools/bench_mano.py [B=16]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import mano_oracle as M
from hoig_amd.mano import ManoModel, mano_vertices
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
md = M.synthetic_model(0)
model = ManoModel.from_dict(md)
g = np.random.Generator(np.random.Philox(key=[1, B]))
root, hand, betas, t = (g.standard_normal(s).astype(np.float32) for s in ((B, 3), (B, 45), (B, 10), (B, 3)))
c = lambda a: torch.from_numpy(a).cuda()
args = (c(root), c(hand), c(betas), c(t))
out = torch.empty(B, 778 + 3000, 3, device='cuda')
for _ in range(20):
    mano_vertices(model, *args, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 200
e0.record()
for _ in range(n):
    mano_vertices(model, *args, out=out)
e1.record(); torch.cuda.synchronize()
gpu_us = e0.elapsed_time(e1) * 1e3 / n
t0 = time.perf_counter()
for _ in range(5):
    M.smplx_mano_forward(md, root, hand, betas, t)
cpu_us = (time.perf_counter() - t0) * 1e6 / 5
print('MANO layer, B=%d: %.1f us per call on the device (launch-to-launch, incl. host issue), %.0f us numpy float64 on one host core'
      % (B, gpu_us, cpu_us))
