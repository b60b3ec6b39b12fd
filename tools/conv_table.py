"""Per-shape table of every convolution launch of one training step (forward / data gradient / weight gradient):
calls, time, TFLOP/s.  Each launch is bracketed by events (this serialises nothing: one stream), so the sum matches the
kernel-trace view.  Usage: python tools/conv_table.py [precision] [batch] [side]"""
import collections
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from hoig_amd import ops, synthetic, _lib as L          # noqa: E402
from hoig_amd.models import ModelsFactory               # noqa: E402
from common import opt_namespace                        # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16x3'
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
side = int(sys.argv[3]) if len(sys.argv) > 3 else 256
ops.set_precision(prec)

records = []
enabled = [False]


def key_of(d):
    return (d.B, d.Hi, d.Wi, d.Ci, d.Ho, d.Wo, d.Co, d.R, d.stride, d.pad, d.transposed)


def timed(kind, d, fn):
    if not enabled[0]:
        return fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = fn()
    e1.record()
    records.append((kind, key_of(d), e0, e1))
    return r


_fwd, _dgrad, _call = ops._conv_fwd_raw, ops._conv_dgrad_raw, ops.call


def fwd(d, *a, **k):
    return timed('fwd', d, lambda: _fwd(d, *a, **k))


def dgrad(d, *a, **k):
    return timed('dgrad', d, lambda: _dgrad(d, *a, **k))


def call(name, *a):
    if name == 'hoig_conv2d_bwd_weight':
        d = a[0]._obj
        return timed('wgrad', d, lambda: _call(name, *a))
    return _call(name, *a)


ops._conv_fwd_raw, ops._conv_dgrad_raw, ops.call = fwd, dgrad, call


def wrap_lib(name, kind):
    """the pre-split entry points are called through L.lib directly (ops._Conv._backward_split): time them too"""
    orig = getattr(L.lib, name)

    def f(dref, *a):
        return timed(kind, dref._obj, lambda: orig(dref, *a))
    setattr(L.lib, name, f)


from hoig_amd import _lib as L          # noqa: E402
wrap_lib('hoig_conv2d_bwd_weight_split', 'wgrad')
wrap_lib('hoig_conv2d_bwd_data_packed_split', 'dgrad')

opt = opt_namespace(gen_name='generator_spade_attn', local_rank=0, image_size=side, hip_graph=False)
torch.manual_seed(8)
model = ModelsFactory.get_by_name('trainer', opt, use_ddp=False)
model.set_train()
model.set_input(synthetic.make_inputs(batch, side, seed=8))
for _ in range(2):
    model.optimize_parameters()
torch.cuda.synchronize()
enabled[0] = True
model.optimize_parameters()
torch.cuda.synchronize()
enabled[0] = False

agg = collections.OrderedDict()
for kind, key, e0, e1 in records:
    a = agg.setdefault((kind, key), [0, 0.0])
    a[0] += 1
    a[1] += e0.elapsed_time(e1)
rows = []
for (kind, key), (n, ms) in agg.items():
    B, Hi, Wi, Ci, Ho, Wo, Co, R, stride, pad, tr = key
    if tr:
        flop = 2.0 * B * Hi * Wi * Ci * Co * R * R
    else:
        flop = 2.0 * B * Ho * Wo * Ci * Co * R * R
    rows.append((ms, kind, key, n, flop * n))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print('total conv time %.2f ms over %d launches (%s, batch %d, %d^2)' % (tot, len(records), prec, batch, side))
print('%-6s %-52s %5s %9s %8s %7s' % ('kind', 'B Hi Wi Ci -> Ho Wo Co  RxR s p T', 'calls', 'ms', 'TF/s', 'cum%'))
cum = 0.0
for ms, kind, key, n, flop in rows[:int(os.environ.get('ROWS', '70'))]:
    cum += ms
    B, Hi, Wi, Ci, Ho, Wo, Co, R, stride, pad, tr = key
    desc = '%d %dx%d %d -> %dx%d %d  %dx%d s%d p%d %s' % (B, Hi, Wi, Ci, Ho, Wo, Co, R, R, stride, pad, 'T' if tr else '')
    print('%-6s %-52s %5d %9.3f %8.1f %6.1f%%' % (kind, desc, n, ms, flop / ms / 1e9, 100 * cum / tot))
by_kind = collections.defaultdict(lambda: [0.0, 0.0])
for ms, kind, key, n, flop in rows:
    by_kind[kind][0] += ms
    by_kind[kind][1] += flop
for k, (ms, flop) in by_kind.items():
    print('%-6s %8.2f ms  %7.2f TFLOP  %6.1f TF/s' % (k, ms, flop / 1e12, flop / ms / 1e9))
