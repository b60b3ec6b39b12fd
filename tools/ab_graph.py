"""Eager step against the captured (hipGraph-replayed) step in ONE process, alternating: VERDICT r5 item 7.  bench.py times its other
form once, after the main run; boxes and even legs of one process differ by 1-2 ms, so the gap is measured here as the median of
interleaved legs.     python tools/ab_graph.py [rounds] [steps per leg]"""
import contextlib
import io
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hoig_amd import ops, synthetic                                  # noqa: E402
from hoig_amd.models import ModelsFactory, trainer as T              # noqa: E402
from hoig_amd.options import opt_namespace                           # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
ops.set_precision('bf16x3:f16x2')
torch.manual_seed(8)
with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
    m = ModelsFactory.get_by_name('trainer', opt_namespace(hip_graph=True), use_ddp=False)
m.set_train()
inputs = synthetic.make_inputs(8, 256, seed=8)
res = {False: [], True: []}
host = {False: [], True: []}
for r in range(rounds):
    for graph in (False, True):
        m._use_graph = graph
        m.set_input(inputs)
        for _ in range(T._GRAPH_WARMUP + 3):
            m.optimize_parameters()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        h = 0.0
        for _ in range(steps):
            h0 = time.perf_counter()
            m.optimize_parameters()
            h += time.perf_counter() - h0
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        res[graph].append(ms)
        host[graph].append(h / steps * 1e3)
        print('round %d  %-8s %7.2f ms/step   host %6.2f ms/step' % (r, 'captured' if graph else 'eager', ms, h / steps * 1e3), flush=True)
e, g = statistics.median(res[False]), statistics.median(res[True])
print('median: eager %.2f ms, captured %.2f ms (%+.2f ms); host per step %.1f / %.1f ms' % (e, g, g - e, statistics.median(host[False]),
                                                                                           statistics.median(host[True])))
