"""Precision / throughput frontier of the conv arithmetic: for every HOIG_PRECISION mode, the error of
the six forward outputs (max-norm relative, the reading of north_star's "1e-3 relative fp32" used by the tests), of the seven
loss terms of one optimiser step, and of all gradient tensors (relative L2: median / p95 / worst) against the CPU oracle on the
same seeded inputs and weights, at 64x64 (batch 2), 128x128 (batch 1) and 256x256 (batch 1).
    python tools/precision_frontier.py [modes...] > profiles/r02_precision_frontier.txt
TEST INFRASTRUCTURE (imports tests/common.py and the oracle)."""
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from common import oracle_trainer, product_trainer      # noqa: E402
from gpu_util import rel_err, rel_l2                    # noqa: E402
from hoig_amd import ops                                # noqa: E402

SKIP = ('.conv_0.bias', )
SKIP_D = ('model.2.bias', 'model.5.bias', 'model.8.bias', 'model.11.bias')


def oracle_side(side, batch):
    ot = oracle_trainer('generator_spade_attn', batch, side)
    with torch.no_grad():
        fwd = [o.clone() for o in ot.forward()]
    ot.optimize_parameters()
    grads = {}
    for tag, net in (('G', ot.G), ('D', ot.D)):
        for name, p in net.items():
            if p.grad is not None and not name.endswith(SKIP) and name not in SKIP_D:
                grads[(tag, name)] = p.grad.clone()
    return fwd, ot.get_current_errors(), grads


def main():
    modes = sys.argv[1:] or ['f32', 'bf16x3', 'bf16']
    ops.set_f6_min_tiles(int(os.environ.get('F6_MIN_TILES', '1')))          # 'f16f6': the fp6 forward kernel on every eligible layer, also at these small sizes
    print('%-8s %5s | %-10s | %-10s | %-28s | %s' % ('mode', 'side', 'fwd max', 'loss max', 'grad rel-L2 med/p95/worst', 'worst loss term'))
    for side, batch in ((64, 2), (128, 1), (256, 1)):
        ofwd, oerr, ograd = oracle_side(side, batch)
        for mode in modes:
            ops.set_precision(mode)
            m = product_trainer('generator_spade_attn', batch, side)
            with torch.no_grad():
                fwd = m.forward()
            ferr = max(rel_err(a, b) for a, b in zip(fwd, ofwd))
            m.optimize_parameters()
            e = m.get_current_errors()
            lerr, lkey = max((abs(e[k] - oerr[k]) / max(abs(oerr[k]), 1e-2), k) for k in oerr)
            gr = {}
            for tag, net in (('G', m._G), ('D', m._D)):
                for k, v in net.export_dict(net.flat_grad).items():
                    if (tag, k) in ograd:
                        gr[(tag, k)] = rel_l2(v, ograd[(tag, k)])
            vals = sorted(gr.values())
            print('%-8s %5d | %.3e  | %.3e  | %.2e / %.2e / %.2e | %s' % (mode, side, ferr, lerr, vals[len(vals) // 2],
                                                                         vals[int(0.95 * len(vals))], vals[-1], lkey), flush=True)
            del m
            torch.cuda.empty_cache()
    ops.set_precision('f32')


if __name__ == '__main__':
    main()
