"""Loss trajectories of the training step under different arithmetic modes (TEST INFRASTRUCTURE): same seeded weights and a
fixed synthetic batch, N optimiser steps, the seven loss terms every few steps.  GAN training amplifies any difference
(Adam's first steps are sign-like), so the trajectories separate slowly; what is checked by eye is that they stay together
to within run-to-run scatter and that nothing drifts or blows up under the two-term backward.
    python tools/trajectory.py [steps] [side] [batch] mode [mode ...]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from common import product_trainer
from hoig_amd import ops

steps, side, batch = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
modes = sys.argv[4:]
marks = sorted(set([1, 2, 5, 10, 20, 35, 50, 75, 100, steps]))
res = {}
ops.set_f6_min_tiles(int(os.environ.get('F6_MIN_TILES', '1')))          # 'f16f6': the fp6 forward kernel on every eligible layer, also at this small size
for mode in modes:
    ops.set_precision(mode.split('#')[0])
    m = product_trainer('generator_spade_attn', batch, side)
    traj = []
    for s in range(1, steps + 1):
        m.optimize_parameters()
        if s in marks:
            traj.append((s, m.get_current_errors()))
    res[mode] = traj
    del m
    torch.cuda.empty_cache()
keys = list(res[modes[0]][0][1].keys())
print('%-22s %5s ' % ('mode', 'step') + ' '.join('%13s' % k for k in keys))
for i in range(len(res[modes[0]])):
    for mode in modes:
        s, e = res[mode][i]
        print('%-22s %5d ' % (mode, s) + ' '.join('%13.6f' % e[k] for k in keys))
    print()
