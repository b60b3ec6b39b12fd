"""Optimiser step + operand-plane refresh of the generator's flat buffer, timed alone: FusedAdam.fuse_planes False (two launches) / True."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hoig_amd import _lib as L, ops, synthetic            # noqa: E402
from hoig_amd.options import opt_namespace                # noqa: E402
from hoig_amd.models import ModelsFactory                 # noqa: E402

ops.set_precision('bf16x3:f16x2')
m = ModelsFactory.get_by_name('trainer', opt_namespace(gen_name='generator_spade_attn'), use_ddp=False)
m.set_input(synthetic.make_inputs(8, 256, seed=1))
m.optimize_parameters()
opt = m._optimizer_G
tree = opt.tree
print('parameters %.1f M, plane tiles %d, plain chunks %d' % (tree.flat.numel() / 1e6, tree._plane_tiles, tree._plain_chunks[1]))
for r in range(3):
    for v in (0, 1):
        opt.fuse_planes = bool(v)
        opt.step(); tree._refresh_planes()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            opt.step()
            tree._refresh_planes()
        e1.record()
        torch.cuda.synchronize()
        print('fuse_planes=%d  %.1f us per step + refresh' % (v, e0.elapsed_time(e1) * 100), flush=True)
