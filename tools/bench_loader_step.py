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
    def timed(fn, n=steps):
        for _ in range(4):
            fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n * 1e3
    prepared = synthetic.make_inputs(B, 256, seed=8)

    def with_prepared():
        model.set_input(prepared)
        model.optimize_parameters()
    ms_p = timed(with_prepared)
    prepared_dev = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in prepared.items()}

    def with_prepared_dev():
        model.set_input(prepared_dev)
        model.optimize_parameters()
    ms_pd = timed(with_prepared_dev)
    ms_pd_alone = timed(lambda: model.set_input(prepared_dev))
    # (b) train_ddp.py:88-92: for batch in loader: set_input(batch); optimize_parameters()
    torch.cuda.empty_cache()
    opt_d.loader_prepares = False                           # first: the round-5 form (the raw stage inside Trainer.set_input)
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
    fixed = batch                                            # the loop's last batch: the raw path without the loader

    def loader_leg(use_batch):
        k, t = 0, None
        for b in CustomDatasetDataLoader(opt_d, is_for_train=True).load_data():
            if k == 4:
                torch.cuda.synchronize()
                t = time.perf_counter()
            if use_batch:
                model.set_input(b)
            model.optimize_parameters()
            k += 1
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / (k - 4) * 1e3
    ms_idle_loader = loader_leg(False)                       # the loader runs, its batches are dropped: the loader's own interference
    ms_b2 = loader_leg(True)                                 # (the timed loop above once more: run-to-run spread)
    # round 6: the raw stage as the loader's last device step, one batch ahead (the loader's opt carries the MANO model and the assets,
    # as train_ddp.py's single opt does)
    opt_d.loader_prepares = True
    opt_d.mano_model, opt_d.object_assets, opt_d.image_size = opt.mano_model, opt.object_assets, 256
    ms_ahead = [loader_leg(True), loader_leg(True)]
    opt_d.loader_prepares = False

    def with_raw():
        model.set_input(fixed)
        model.optimize_parameters()
    ms_r = timed(with_raw)
    # where the raw path's extra time goes: host time of the two calls (no synchronisation inside the loop: the host runs ahead of the
    # device by whatever slack the step leaves) and the stage alone on an idle device
    host_in = host_st = 0.0
    torch.cuda.synchronize()
    for _ in range(steps):
        t = time.perf_counter()
        model.set_input(fixed)
        t1 = time.perf_counter()
        model.optimize_parameters()
        host_in, host_st = host_in + (t1 - t), host_st + (time.perf_counter() - t1)
    torch.cuda.synchronize()
    host_in, host_st = host_in / steps * 1e3, host_st / steps * 1e3

    def stage_only():
        model.set_input(fixed)
    ms_stage = timed(stage_only)
    print('step on prepared synthetic inputs (staged once)          : %.2f ms' % ms_a)
    print('set_input(prepared tensors) + step, every step           : %.2f ms  (%+.2f ms)' % (ms_p, ms_p - ms_a))
    print('set_input(prepared tensors ON THE DEVICE) + step            : %.2f ms  (%+.2f ms; the call alone on an idle device %.2f ms)'
          % (ms_pd, ms_pd - ms_a, ms_pd_alone))
    print('set_input(one raw device batch, no loader) + step        : %.2f ms  (%+.2f ms)' % (ms_r, ms_r - ms_a))
    print('  host time of the two calls in that loop                  : set_input %.2f ms, optimize_parameters %.2f ms' % (host_in, host_st))
    print('  raw set_input alone, back to back on an idle device    : %.2f ms' % ms_stage)
    print('loader running, batches dropped (step on the staged inputs): %.2f ms  (%+.2f ms)' % (ms_idle_loader, ms_idle_loader - ms_a))
    print('loader (%d workers) + raw set_input + step, %2d timed steps : %.2f ms  (%+.2f ms); again %.2f ms' % (workers, n - 4, ms_b, ms_b - ms_a, ms_b2))
    print('loader runs the raw stage ONE BATCH AHEAD + set_input(prepared entries of the batch) + step : %.2f / %.2f ms  (%+.2f ms)'
          % (ms_ahead[0], ms_ahead[1], min(ms_ahead) - ms_a))
