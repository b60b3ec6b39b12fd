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

FAMILIES = {
    'norm_fwd': ['hoig_inorm_fwd_fused', 'hoig_inorm_stats', 'hoig_inorm_apply', 'hoig_inorm_apply_ld'],
    'norm_bwd': ['hoig_inorm_bwd_fused_add', 'hoig_inorm_bwd_fused', 'hoig_inorm_bwd', 'hoig_inorm_bwd_ld', 'hoig_inorm_bwd_add_ld'],
    'conv_fwd': ['hoig_conv2d_fwd_packed', 'hoig_conv2d_fwd', 'hoig_conv2d_cat_fwd_packed', 'hoig_conv2d_fwd_heads'],
    'conv_dgrad': ['hoig_conv2d_bwd_data_packed', 'hoig_conv2d_bwd_data_packed_add', 'hoig_conv2d_bwd_data',
                   'hoig_conv2d_cat_bwd_data_packed'],
    'conv_wgrad': ['hoig_conv2d_bwd_weight'],
    'attention': ['hoig_attn_pixel_fwd', 'hoig_attn_pixel_bwd', 'hoig_attn_src_gather', 'hoig_attn_gs_gather',
                  'hoig_replicate_pad_fwd', 'hoig_replicate_pad_bwd_add', 'hoig_replicate_pad_bwd'],
    'optimiser': ['hoig_adam_step_dev', 'hoig_adam_tick', 'hoig_pack_conv_weights_bf16_all'],
    'pointwise': ['hoig_add', 'hoig_add_act', 'hoig_act_bwd', 'hoig_act_bwd_colsum', 'hoig_colsum_accum', 'hoig_copy_channels',
                  'hoig_compose_fwd', 'hoig_compose_bwd', 'hoig_maxpool2_fwd', 'hoig_maxpool2_bwd',
                  'hoig_loss_accumulate', 'hoig_tv_accumulate', 'hoig_sum', 'hoig_sum_scaled'],
}


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
