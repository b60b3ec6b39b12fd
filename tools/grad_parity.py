"""Gradient parity of one optimiser step against the CPU oracle at a given size, REPEATED, so that the run-to-run spread is seen
next to the arithmetic's own error (TEST INFRASTRUCTURE: imports tests/common.py and the oracle).
    python tools/grad_parity.py side batch repeats mode[@min_tiles][+subnet=mode,...] ...
    e.g.  256 1 4 f16f6@1 bf16x3:f16x2 bf16x3:f16x2+vgg=f16x2,d=f16x2      (the '+' part: ops.set_subnet_precision)"""
import os, sys, io, contextlib, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from common import oracle_trainer, product_trainer
from gpu_util import rel_err, rel_l2
from hoig_amd import ops
from precision_frontier import oracle_side

side, batch, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
ofwd, oerr, ograd = oracle_side(side, batch)
print('%-22s | %-9s | %-9s | %s' % ('mode@f6_min_tiles run', 'fwd max', 'loss max', 'gradient rel-L2 median / p95 / worst (tensor)'))
for spec in sys.argv[4:]:
    base, _, submap = spec.partition('+')
    mode, _, tiles = base.partition('@')
    ops.set_precision(mode)
    ops.set_subnet_precision(submap)
    old = ops.set_f6_min_tiles(int(tiles) if tiles else 192)
    for r in range(reps):
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            m = product_trainer('generator_spade_attn', batch, side, hip_graph=False)
        with torch.no_grad():
            fwd = m.forward()
        ferr = max(rel_err(a, b) for a, b in zip(fwd, ofwd))
        m.optimize_parameters()
        e = m.get_current_errors()
        lerr = max(abs(e[k] - oerr[k]) / max(abs(oerr[k]), 1e-2) for k in oerr)
        gr = []
        for tag, net in (('G', m._G), ('D', m._D)):
            for k, v in net.export_dict(net.flat_grad).items():
                if (tag, k) in ograd:
                    gr.append((rel_l2(v, ograd[(tag, k)]), tag + '.' + k))
        gr.sort()
        print('%-18s %3d | %.3e | %.3e | %.2e / %.2e / %.2e (%s)' % (spec, r, ferr, lerr, gr[len(gr) // 2][0], gr[int(0.95 * len(gr))][0],
                                                                  gr[-1][0], gr[-1][1]), flush=True)
        del m
        torch.cuda.empty_cache()
    ops.set_f6_min_tiles(old)
    ops.set_subnet_precision(None)
