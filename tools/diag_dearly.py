"""Why does issuing the D step before G's backward (tuning key d_early) hide the negative control of tests/test_stream_order_gpu.py
(the optimiser side stream without its wait for the caller's stream)?  Times (ms after the step's start) of the end of the bg stream's
work, of the caller's stream after the backward, and of G's Adam on the side stream, with the bg branch delayed by ~30 ms."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from common import product_trainer          # noqa: E402
from hoig_amd import _lib as L, ops         # noqa: E402
from hoig_amd.models import trainer as T    # noqa: E402

ops.set_precision('bf16x3:f16x2')
for early in (0, 1):
    L.set_tuning('d_early', early)
    ops._TEST_DELAYS.clear()
    ops._TEST_DELAYS['g_bg'] = 240_000_000
    m = product_trainer('generator_spade_attn', 2, 128, hip_graph=False)
    m._side.wait_stream = lambda stream: None
    if os.environ.get('WARM'):
        m.optimize_parameters()
    torch.cuda.synchronize()
    T._TEST_TRACE = trace = []
    start = torch.cuda.Event(enable_timing=True)
    start.record()
    orig, marks = m._phase_g, {}

    def phase_g(*a, **k):
        r = orig(*a, **k)
        for name, st in (('bg_stream_done', m._net(m._G)._streams[0]), ('main_after_backward', torch.cuda.current_stream())):
            e = torch.cuda.Event(enable_timing=True)
            e.record(st)
            marks[name] = e
        return r
    m._phase_g = phase_g
    import time
    t0 = time.perf_counter()
    m.optimize_parameters()
    host_ms = (time.perf_counter() - t0) * 1e3
    torch.cuda.synchronize()
    T._TEST_TRACE = None
    print('d_early = %d   (host returned after %.1f ms; %s step of the trainer)' % (early, host_ms, 'second' if os.environ.get('WARM') else 'FIRST'))
    bgw = m._G.export_dict(m._optimizer_G.exp_avg)
    print('   |exp_avg| of tsf_model.resnets.4.main.4.weight: %.3e' % float(bgw['tsf_model.resnets.4.main.4.weight'].abs().sum()))
    for k, e in marks.items():
        print('   %-22s %8.2f ms' % (k, start.elapsed_time(e)))
    for tag, e in trace:
        print('   %-22s %8.2f ms' % (tag, start.elapsed_time(e)))
    ops._TEST_DELAYS.clear()
L.set_tuning('d_early', 1)
