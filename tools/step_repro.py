"""Run-to-run reproducibility of the first optimiser steps (TEST INFRASTRUCTURE: imports tests/common and the oracle).
usage: python tools/step_repro.py [precision] [repeats] [steps] [side] [batch]; env HOIG_STREAMS / HOIG_GRAPH as usual."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import contextlib, io
from common import product_trainer, oracle_trainer
from hoig_amd import ops
prec = sys.argv[1] if len(sys.argv) > 1 else 'f16f6'
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
side = int(sys.argv[4]) if len(sys.argv) > 4 else 64
batch = int(sys.argv[5]) if len(sys.argv) > 5 else 2
ops.set_precision(prec)
ops.set_f6_min_tiles(1)
keys = None
def show(tag, hist):
    for s, e in enumerate(hist):
        print('%-10s step %d  %s' % (tag, s, '  '.join('%s=%.6f' % (k, e[k]) for k in keys)), flush=True)
if os.environ.get('ORACLE', '1') == '1':
    ot = oracle_trainer('generator_spade_attn', batch, side)
    hist = []
    for s in range(steps):
        ot.optimize_parameters(); hist.append(ot.get_current_errors())
    keys = list(hist[0].keys())
    show('oracle', hist)
for r in range(reps):
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        m = product_trainer('generator_spade_attn', batch, side, hip_graph=os.environ.get('HOIG_GRAPH', '0') == '1')
    hist = []
    for s in range(steps):
        m.optimize_parameters(); hist.append(m.get_current_errors())
    keys = keys or list(hist[0].keys())
    show('run%d' % r, hist)
    del m
