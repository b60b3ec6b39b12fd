"""End to end: the training step fed by the loader + the raw-batch `Trainer.set_input` (MANO layer, rasteriser, input preparation) against
the same step on prepared synthetic inputs (the loader stage had only been timed on an idle device; the raw path's index
tensors used to be blocking pageable copies).  Batch 8 at 256 x 256, synthetic HO3D-v3-shaped tree and synthetic assets
(tests/data_fixture.py, tests/test_hand_recovery_gpu.py), `workers` DataLoader workers.
usage: python tools/bench_loader_step.py [steps] [workers]"""
import os
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import data_fixture as FX                                  # noqa: E402
from test_hand_recovery_gpu import _assets                 # noqa: E402
from common import opt_namespace                           # noqa: E402
from hoig_amd import ops, synthetic                        # noqa: E402
from hoig_amd.data import CustomDatasetDataLoader          # noqa: E402
from hoig_amd.mano import ManoModel                        # noqa: E402
from hoig_amd.models import ModelsFactory                  # noqa: E402
from oracle import mano_oracle as M                        # noqa: E402  (a synthetic MANO model: test infrastructure, as in the tests)

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 24
workers = int(sys.argv[2]) if len(sys.argv) > 2 else 4
B = 8
ops.set_precision('bf16x3:f16x2')
assets, nv = _assets([2, 5], 21)
with tempfile.TemporaryDirectory() as root:
    opt_d = FX.build(root, seed=8, frames=B, n_obj_verts=nv)
    opt_d.batch_size, opt_d.n_threads_train = B, workers
    pairs = [('%s/%04d.png' % (('ABF1_0', 'MC2_0')[k % 2], k % B), '%s/%04d.png' % (('ABF1_0', 'MC2_0')[k % 2], (k + 3) % B)) for k in range(B * steps)]
    FX.write_pairs(opt_d, pairs)
    opt = opt_namespace(gen_name='generator_spade_attn', local_rank=0, image_size=256)
    opt.mano_model = ManoModel.from_dict(M.synthetic_model(4))
    opt.object_assets = assets
    torch.manual_seed(3)
    model = ModelsFactory.get_by_name('trainer', opt, use_ddp=False)
    model.set_train()
    # (a) prepared inputs, staged once: what bench.py times
    model.set_input(synthetic.make_inputs(B, 256, seed=8))
    for _ in range(6):
        model.optimize_parameters()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        model.optimize_parameters()
    torch.cuda.synchronize()
    ms_a = (time.perf_counter() - t0) / steps * 1e3
    # (b) train_ddp.py:88-92: for batch in loader: set_input(batch); optimize_parameters()
    torch.cuda.empty_cache()
    loader = CustomDatasetDataLoader(opt_d, is_for_train=True)
    n, t0 = 0, None
    for batch in loader.load_data():
        if n == 4:                                          # (the first batches pay the worker start-up and the first raw set_input)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        model.set_input(batch)
        model.optimize_parameters()
        n += 1
    torch.cuda.synchronize()
    ms_b = (time.perf_counter() - t0) / (n - 4) * 1e3
    print('step on prepared synthetic inputs (staged once)          : %.2f ms' % ms_a)
    print('loader (%d workers) + raw set_input + step, %2d timed steps : %.2f ms  (%+.2f ms)' % (workers, n - 4, ms_b, ms_b - ms_a))
