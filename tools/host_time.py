"""Host-side (launch) time vs GPU time of the training step: is the step CPU-bound?"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from hoig_amd import ops, synthetic
from hoig_amd.models import ModelsFactory
from common import opt_namespace
ops.set_precision('bf16x3')
opt = opt_namespace(gen_name='generator_spade_attn', local_rank=0, image_size=256)
torch.manual_seed(8)
model = ModelsFactory.get_by_name('trainer', opt, use_ddp=False)
model.set_train()
model.set_input(synthetic.make_inputs(8, 256, seed=8))
for _ in range(3):
    model.optimize_parameters()
torch.cuda.synchronize()
n = 10
t0 = time.perf_counter()
for _ in range(n):
    model.optimize_parameters()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('host launch time per step %.1f ms; wall per step %.1f ms; GPU tail after last launch %.1f ms'
      % ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3, (t2 - t1) * 1e3))
