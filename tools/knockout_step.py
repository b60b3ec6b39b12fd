"""How much WALL time of the training step each kernel family is worth: the step is timed with the family's C-ABI entry points
replaced by no-ops (results are garbage -- only the clock is read).  With several concurrent chains per step the sum of kernel
durations says little about the critical path; this measures it.
usage: python tools/knockout_step.py [family ...]   (default: every family, one after the other, each in this process)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hoig_amd import _lib as L, ops, synthetic            # noqa: E402
from hoig_amd.options import opt_namespace                # noqa: E402
from hoig_amd.models import ModelsFactory                 # noqa: E402

import re          # noqa: E402

# Families are built from the library's own symbol table (hoig_amd._lib._SIGS) by name pattern, so an entry point added later
# lands in its family by itself (round 3's hand-written lists missed the *_stats, *_f6 and cat wgrad entry points, and
# its forward / weight-gradient figures were understated by the launches that kept running).
PATTERNS = [
    ('norm_fwd', r'hoig_inorm_(fwd_fused|stats|stats_from_sums|apply|apply_ld)$'),
    ('norm_bwd', r'hoig_inorm_bwd'),
    ('conv_wgrad', r'hoig_conv2d_(cat_)?bwd_weight(_split)?(_pair)?$'),
    ('conv_dgrad', r'hoig_conv2d_(cat_)?bwd_data'),
    ('conv_fwd', r'hoig_conv2d_(cat_)?fwd'),
    ('attention', r'hoig_(attn_(pixel|src_gather|gs_gather)|replicate_pad)'),
    ('optimiser', r'hoig_(adam_step_dev|adam_tick|adam_step|adam_pack_step|pack_conv_weights)'),
    ('pointwise', r'hoig_(add|add_act|act_bwd|act_bwd_colsum|colsum_accum|copy_channels|cat2_channels|compose_fwd|compose_bwd|'
                  r'maxpool2_fwd|maxpool2_bwd|loss_accumulate|loss_fwd_bwd|tv_accumulate|tv_fwd_bwd|sum|sum_scaled)$'),
]
FAMILIES = {name: [] for name, _ in PATTERNS}
for sym in sorted(L._SIGS):
    for name, pat in PATTERNS:
        if re.match(pat, sym):
            FAMILIES[name].append(sym)
            break


def step_ms(m, steps=12, warmup=3):
    for _ in range(warmup):
        m.optimize_parameters()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        m.optimize_parameters()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / steps


ops.set_precision(os.environ.get('HOIG_PRECISION', 'bf16x3:f16x2'))
opt = opt_namespace(gen_name='generator_spade_attn')
m = ModelsFactory.get_by_name('trainer', opt, use_ddp=False)
m.set_input(synthetic.make_inputs(8, 256, seed=1))
full = step_ms(m)
print('full step                                   %.2f ms' % full, flush=True)
for fam, names in FAMILIES.items():
    print('# %-10s %s' % (fam, ' '.join(n[5:] for n in names)))
stub = lambda *a: 0
# The optimiser stays knocked out in every other run: the weights then never change, so a family whose outputs are garbage (stale
# finite memory, not NaN) cannot poison them -- a step on NaN data draws less power and runs 4 ms FASTER, which would be booked
# as that family's time.  Every family is compared with the step without the optimiser.
for n in FAMILIES['optimiser']:
    setattr(L.lib, n, stub)
base = step_ms(m)
print('without optimiser (reference of the rest)   %.2f ms  (%+.2f)' % (base, base - full), flush=True)
for fam in (sys.argv[1:] or [f for f in FAMILIES if f != 'optimiser']):
    names = [n for n in FAMILIES[fam] if hasattr(L.lib, n)]
    real = {n: getattr(L.lib, n) for n in names}
    for n in names:
        setattr(L.lib, n, stub)
    t = step_ms(m)
    for n, f in real.items():
        setattr(L.lib, n, f)
    print('without optimiser and %-20s  %.2f ms  (%+.2f)' % (fam, t, t - base), flush=True)
print('without optimiser, again                    %.2f ms' % step_ms(m))
