"""Per-entry-point / per-shape table of EVERY C-ABI launch of one training step (name, integer arguments, calls, time).
Usage: python tools/op_table.py [precision] [batch] [side] [name filter]"""
import collections
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from hoig_amd import ops, nn as hnn, synthetic, _lib as L   # noqa: E402
from hoig_amd.models import ModelsFactory               # noqa: E402
from common import opt_namespace                        # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16x3'
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
side = int(sys.argv[3]) if len(sys.argv) > 3 else 256
filt = sys.argv[4] if len(sys.argv) > 4 else ''
ops.set_precision(prec)
records = []
enabled = [False]
_call = L.call


def sig(a):
    out = []
    for v in a:
        if isinstance(v, bool):
            continue
        if isinstance(v, int) and abs(v) < (1 << 40) and not (v > (1 << 32)):
            out.append(v)
        elif hasattr(v, '_obj') and isinstance(v._obj, L.ConvDesc):
            d = v._obj
            out.append('conv[%d %dx%d %d->%dx%d %d k%d s%d%s]' % (d.B, d.Hi, d.Wi, d.Ci, d.Ho, d.Wo, d.Co, d.R, d.stride,
                                                                  'T' if d.transposed else ''))
    return tuple(out)


def call(name, *a):
    if not enabled[0]:
        return _call(name, *a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = _call(name, *a)
    e1.record()
    records.append((name, sig(a), e0, e1))
    return r


for mod in (ops, hnn, L):
    if getattr(mod, 'call', None) is _call:
        mod.call = call

opt = opt_namespace(gen_name='generator_spade_attn', local_rank=0, image_size=side)
torch.manual_seed(8)
model = ModelsFactory.get_by_name('trainer', opt, use_ddp=False)
model.set_train()
model.set_input(synthetic.make_inputs(batch, side, seed=8))
for _ in range(2):
    model.optimize_parameters()
torch.cuda.synchronize()
enabled[0] = True
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record()
model.optimize_parameters()
t1.record()
torch.cuda.synchronize()
enabled[0] = False
agg = collections.OrderedDict()
for name, s, e0, e1 in records:
    a = agg.setdefault((name, s), [0, 0.0])
    a[0] += 1
    a[1] += e0.elapsed_time(e1)
by_name = collections.defaultdict(lambda: [0, 0.0])
for (name, s), (n, ms) in agg.items():
    by_name[name][0] += n
    by_name[name][1] += ms
tot = sum(v[1] for v in by_name.values())
print('step %.2f ms; C-ABI launches %d, %.2f ms inside them' % (t0.elapsed_time(t1), len(records), tot))
for name, (n, ms) in sorted(by_name.items(), key=lambda kv: -kv[1][1]):
    print('%-34s %5d calls %9.3f ms' % (name, n, ms))
print()
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
for (name, s), (n, ms) in rows:
    if filt and filt not in name:
        continue
    print('%-30s %-64s %4d %8.3f ms  %7.1f us/call' % (name, ' '.join(str(v) for v in s)[:64], n, ms, 1e3 * ms / n))
