"""Lists the aten operators (torch-side work: autograd accumulation, fills, copies) of one training step."""
import os, sys, torch
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
for _ in range(2):
    model.optimize_parameters()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    model.optimize_parameters()
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages() if e.key.startswith('aten::')]
rows.sort(key=lambda e: -e.count)
for e in rows[:40]:
    print('%-40s count %5d  cpu %8.1f us  cuda %8.1f us' % (e.key, e.count, e.cpu_time_total, getattr(e, 'device_time_total', 0)))
