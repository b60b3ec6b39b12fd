"""Step time with stream roles moved to other hardware-queue classes (hoig_amd.ops._QUEUE_OF_ROLE: see the comment there).
usage: python tools/queue_sweep.py [role=class,role=class ...]   -> one line: the map's changes and ms per step
The table in ops.py is the shipped default; this tool only patches it in its own process (one process per candidate: the streams are
created when the Trainer is built)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hoig_amd import ops, synthetic            # noqa: E402
from hoig_amd.options import opt_namespace    # noqa: E402
from hoig_amd.models import ModelsFactory     # noqa: E402

change = sys.argv[1] if len(sys.argv) > 1 and '=' in sys.argv[1] else ''
for kv in filter(None, change.split(',')):
    k, v = kv.split('=')
    assert k in ops._QUEUE_OF_ROLE, k
    ops._QUEUE_OF_ROLE[k] = int(v)
ops.set_precision(os.environ.get('HOIG_PRECISION', 'bf16x3:f16x2'))
m = ModelsFactory.get_by_name('trainer', opt_namespace(gen_name='generator_spade_attn'), use_ddp=False)
m.set_input(synthetic.make_inputs(8, 256, seed=1))
for _ in range(6):
    m.optimize_parameters()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 25
for _ in range(n):
    m.optimize_parameters()
torch.cuda.synchronize()
print('%-28s %.2f ms/step   %s' % (change or 'default', (time.perf_counter() - t0) * 1e3 / n,
                                  ' '.join('%s=%d' % kv for kv in sorted(ops._QUEUE_OF_ROLE.items()))), flush=True)
