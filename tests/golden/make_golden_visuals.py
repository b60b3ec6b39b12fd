"""Generates tests/golden/hov3_spade_attn_64_visuals.npz: the 18 uint8 visuals of the REFERENCE's own
``Trainer.forward(keep_data_for_visuals=True)`` / ``get_current_visuals()`` (models/trainer.py:405-415,497-551), i.e. the
reference's ``util.tensor2im`` / ``tensor2maskim`` / ``Colorize`` (utils/util.py:22-74,249-272) applied to the reference's
own forward outputs, on the seeded 64x64 batch-2 inputs and weights of hov3_spade_attn_64.npz.  Build container only:

    python tests/golden/make_golden_visuals.py

Two notes on how the reference code is driven on CPU:
  * ``tensor2im`` unnormalises IN PLACE after ``img.cpu().float()`` (util.py:255-258).  On the GPU, where the reference runs,
    ``.cpu()`` copies; on CPU tensors it returns the tensor itself and the caller's data would be overwritten (e.g.
    ``_real_src`` before ``_vis_batch_src`` is taken from it, trainer.py:528,549).  The harness therefore hands every call a
    clone -- the GPU behaviour.
  * ``torchvision.utils.make_grid(padding=0)`` is the harness's stand-in (oracle/ref_harness.py), torchvision is not
    installed: the three batch grids are 'composition-pinned' in their tiling, value-pinned in their arithmetic.
Also stored: ``fake_tsf_imgs`` etc. as float32, so that the uint8 conversion kernel can be value-tested in isolation."""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_harness as RH, hogan_oracle as O      # noqa: E402
from hoig_amd import synthetic                               # noqa: E402

SEEDS = dict(G=8, D=9, VGG=10, inputs=8)


def main(gen_name='generator_spade_attn', side=64, batch=2):
    t = RH.build_reference_trainer(RH.namespace(gen_name=gen_name))
    util = importlib.import_module('utils.util')
    real_t2i = util.tensor2im
    util.tensor2im = lambda img, *a, **k: real_t2i(img.clone(), *a, **k)
    t.colorize = util.Colorize(n=16)                                    # trainer.py:212
    cfg = O.make_cfg(gen_name)
    t._G.load_state_dict(O.make_weights(O.gen_param_shapes(cfg), seed=SEEDS['G'], mode='random'))
    inp = synthetic.make_inputs(batch, side, seed=SEEDS['inputs'])
    for k, v in inp.items():
        setattr(t, '_' + k, v.clone())
    with torch.no_grad():
        outs = t.forward(keep_data_for_visuals=True)
    vis = t.get_current_visuals()
    out = dict(gen_name=gen_name, side=side, batch=batch, keys=np.array(list(vis.keys())))
    for k, v in vis.items():
        assert v.dtype == np.uint8, (k, v.dtype)
        out['vis_' + k] = v
    out['fake_tsf_imgs'] = outs[3].numpy()
    out['fake_masks_bg'] = outs[4].numpy()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'hov3_spade_attn_64_visuals.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB', {k: v.shape for k, v in vis.items()})


if __name__ == '__main__':
    main()
