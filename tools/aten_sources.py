"""Which host call sites launch the ATen / runtime kernels of a training step (autograd accumulation adds, copies, fills): a
torch.profiler run of three steps with Python stacks, grouped by (ATen op, innermost hoig_amd frame)."""
import collections
import os
import sys

import torch
from torch.profiler import profile, ProfilerActivity

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('HOIG_WGRAD_STREAM', '0')
os.environ.setdefault('HOIG_STREAMS', '0')
from hoig_amd import ops, synthetic                       # noqa: E402
from hoig_amd.options import opt_namespace                # noqa: E402
from hoig_amd.models import ModelsFactory                 # noqa: E402

ops.set_precision(os.environ.get('HOIG_PRECISION', 'bf16x3:f16x2'))
opt = opt_namespace(gen_name='generator_spade_attn')
m = ModelsFactory.get_by_name('trainer', opt, use_ddp=False)
m.set_input(synthetic.make_inputs(8, 256, seed=1))
for _ in range(2):
    m.optimize_parameters()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    m.optimize_parameters()
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    kern = getattr(ev, 'kernels', None) or []
    if not ev.name.startswith('aten::') or not kern:
        continue
    dev_us = sum(k.duration for k in kern)
    site = 'autograd engine / no python frame'
    for fr in ev.stack or []:
        if 'hoig_amd' in fr:
            site = fr.split('hoig_amd/')[-1]
            break
    shp = str([list(x) for x in (ev.input_shapes or []) if x][:2])
    k = (ev.name, site + ' ' + shp)
    agg[k][0] += 1
    agg[k][1] += dev_us
tot = sum(v[1] for v in agg.values())
print('ATen device time per step: %.2f ms' % (tot / 1e3))
for (name, site), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    print('%7.1f us %5d  %-28s %s' % (t, n, name, site))
