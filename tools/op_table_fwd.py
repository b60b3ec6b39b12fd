"""Per-entry-point / per-shape table of every C-ABI launch of ONE generator forward in eval mode (eval.py's path: Trainer.forward under
no_grad), serial on one stream.   HOIG_STREAMS=0 python tools/op_table_fwd.py [batch] [side] [name filter]"""
import collections
import os
import sys

import torch

os.environ.setdefault('HOIG_STREAMS', '0')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from hoig_amd import ops, nn as hnn, synthetic, _lib as L   # noqa: E402
import hoig_amd.ops_norm, hoig_amd.ops_small, hoig_amd.ops_attn, hoig_amd.ops_loss   # noqa: E402,E401
from hoig_amd.models import ModelsFactory               # noqa: E402
from common import opt_namespace                        # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
side = int(sys.argv[2]) if len(sys.argv) > 2 else 256
filt = sys.argv[3] if len(sys.argv) > 3 else ''
ops.set_precision(os.environ.get('HOIG_PRECISION', 'bf16x3:f16x2'))
records = []
enabled = [False]


def sig(a):
    out = []
    for v in a:
        if isinstance(v, bool):
            continue
        if isinstance(v, int) and abs(v) < (1 << 24):
            out.append(v)
        elif hasattr(v, '_obj') and isinstance(v._obj, L.ConvDesc):
            d = v._obj
            out.append('conv[%d %dx%d %d->%dx%d %d k%d s%d%s p%d]' % (d.B, d.Hi, d.Wi, d.Ci, d.Ho, d.Wo, d.Co, d.R, d.stride,
                                                                      'T' if d.transposed else '', d.precision))
    return tuple(out)


class LibProxy(object):
    def __init__(self, lib):
        object.__setattr__(self, '_lib', lib)

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if not enabled[0] or not name.startswith('hoig_') or 'tuning' in name or 'bytes' in name:
            return fn

        def timed(*a):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(*a)
            e1.record()
            if r != L.EUNSUPPORTED:
                records.append((name, sig(a), e0, e1))
            return r
        return timed


L.lib = LibProxy(L.lib)
opt = opt_namespace(gen_name='generator_spade_attn', local_rank=0, image_size=side)
torch.manual_seed(8)
model = ModelsFactory.get_by_name('trainer', opt, use_ddp=False)
model.set_eval()
model.set_input(synthetic.make_inputs(batch, side, seed=8))
with torch.no_grad():
    for _ in range(2):
        model.forward()
    torch.cuda.synchronize()
    enabled[0] = True
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    model.forward()
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
print('forward of %d images: %.2f ms (with the timing events); C-ABI launches %d, %.2f ms inside them' % (batch, t0.elapsed_time(t1), len(records), tot))
for name, (n, ms) in sorted(by_name.items(), key=lambda kv: -kv[1][1]):
    print('%-36s %5d calls %9.3f ms  %5.1f %%' % (name, n, ms, 100 * ms / tot))
print()
for (name, s), (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if filt and filt not in name:
        continue
    print('%-32s %-72s %4d %8.3f ms  %7.1f us/call' % (name, ' '.join(str(v) for v in s)[:72], n, ms, 1e3 * ms / n))
