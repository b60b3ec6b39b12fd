"""Host cost of replaying the captured step: (a) host time of ONE replay issued right after a device synchronise (nothing to wait
for: pure launch cost), (b) steady-state step time, captured vs eager, in one process.  HOIG_STREAMS=0 for the one-stream graph."""
import os, sys, time, contextlib, io
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hoig_amd import ops, synthetic
from hoig_amd.models import ModelsFactory
from hoig_amd.options import opt_namespace
ops.set_precision(os.environ.get('HOIG_PRECISION', 'bf16x3:f16x2'))
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for graph in (True, False):
    opt = opt_namespace(hip_graph=graph)
    torch.manual_seed(8)
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        m = ModelsFactory.get_by_name('trainer', opt, use_ddp=False)
    m.set_train()
    m.set_input(synthetic.make_inputs(8, 256, seed=8))
    for _ in range(5):
        m.optimize_parameters()
    torch.cuda.synchronize()
    one = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m.optimize_parameters()
        one.append((time.perf_counter() - t0) * 1e3)
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        m.optimize_parameters()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e3
    print('%-8s steady %.2f ms/step   host time of one step issued on an idle device: %s ms' % (
        'captured' if graph else 'eager', dt, ' '.join('%.2f' % v for v in one)), flush=True)
    del m
    torch.cuda.empty_cache()
