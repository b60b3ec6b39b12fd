"""Diagnostic: with Trainer._join_backward_streams removed and the bg branch delayed, does G's optimiser step start before the bg
branch's backward has finished?  Prints the times (ms after the step's start) of: the end of the work queued on the bg stream,
the start / end of G's Adam on the side stream."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from common import product_trainer
from hoig_amd import ops
from hoig_amd.models import trainer as T

ops.set_precision('bf16x3:f16x2')
delay = int(sys.argv[1]) if len(sys.argv) > 1 else 60_000_000
for join in (True, False):
    keep = T.Trainer._join_backward_streams
    if not join:
        T.Trainer._join_backward_streams = lambda self: None
    ops._TEST_DELAYS.clear(); ops._TEST_DELAYS['g_bg'] = delay
    m = product_trainer('generator_spade_attn', 2, 128, hip_graph=False)
    m.optimize_parameters(); torch.cuda.synchronize()
    T._TEST_TRACE = trace = []
    start = torch.cuda.Event(enable_timing=True); start.record()
    orig = m._phase_g
    marks = {}
    def phase_g(*a, **k):
        r = orig(*a, **k)
        s_bg = m._net(m._G)._streams[0]
        e = torch.cuda.Event(enable_timing=True); e.record(s_bg); marks['bg_stream_done'] = e
        e2 = torch.cuda.Event(enable_timing=True); e2.record(torch.cuda.current_stream()); marks['main_after_backward'] = e2
        return r
    m._phase_g = phase_g
    m.optimize_parameters(); torch.cuda.synchronize()
    T._TEST_TRACE = None
    print('join =', join)
    for k, e in marks.items():
        print('   %-22s %8.2f ms' % (k, start.elapsed_time(e)))
    for tag, e in trace:
        print('   %-22s %8.2f ms' % (tag, start.elapsed_time(e)))
    bg = m._G.export_dict(m._optimizer_G.exp_avg)
    print('   |exp_avg| of bg_model.model.0.weight: %.3e' % float(bg['bg_model.model.0.weight'].abs().sum()))
    T.Trainer._join_backward_streams = keep
    ops._TEST_DELAYS.clear()
