"""Generates tests/golden/*.npz by running the REFERENCE's own Python (imported from /root/reference on CPU through
oracle/ref_harness.py) on seeded synthetic inputs and platform-independent seeded weights.  Run in the build
container only:   python tests/golden/make_golden.py
The vectors are data (inputs are re-derived from seeds; expected outputs are stored); nothing of the reference's
source travels.

Provenance per file:
  hov3_spade_64.npz       generator_spade      (grid_sample warping): every op is the reference's own -> fully pinned.
  hov3_spade_attn_64.npz  generator_spade_attn (default): reference composition code (extract_attn.py,
                          generator.py:480-491) over the ORACLE's K1-K4 (the CUDA kernels cannot run on CPU)
                          -> 'composition-pinned'.
  hov3_spade_attn_tiny_64.npz  generator_spade_attn_tiny (attention on layers 4-9 only): composition-pinned likewise.
  hov3_base_64.npz        generator_base (no SPADE, grid_sample warping): fully pinned.
  dexycb_spade_attn_64.npz  the HOIG_DexYCB copy of the trainer (HOIG_DexYCB/models/trainer.py: bg 13, hand cond 9,
                          D input 24 channels, no arm mask), generator_spade_attn: composition-pinned.
Usage:  python tests/golden/make_golden.py <gen_name> [hov3|dexycb]      (one reference copy per process)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_harness as RH, hogan_oracle as O      # noqa: E402
from hoig_amd import synthetic                               # noqa: E402

SEEDS = dict(G=8, D=9, VGG=10, inputs=8)
OUT_NAMES = ['fake_src_bg', 'fake_tsf_bg', 'fake_src_imgs', 'fake_tsf_imgs', 'fake_masks_bg', 'fake_masks_hand']


def run(gen_name, fname, side=64, batch=2, steps=2, copy='hov3'):
    opt = RH.namespace(gen_name=gen_name)
    t = RH.build_reference_trainer(opt, copy=copy)
    cfg = O.make_cfg(gen_name, copy)
    sdG = O.make_weights(O.gen_param_shapes(cfg), seed=SEEDS['G'], mode='random')
    sdD = O.make_weights(O.disc_param_shapes(cfg), seed=SEEDS['D'], mode='random')
    sdV = O.make_weights(O.vgg_param_shapes(), seed=SEEDS['VGG'], kind='vgg')
    t._G.load_state_dict(sdG)
    t._D.load_state_dict(sdD)
    t._crt_tsf.vgg.load_state_dict(sdV)
    inp = synthetic.make_inputs(batch, side, seed=SEEDS['inputs'], dataset=copy)
    for k, v in inp.items():
        setattr(t, '_' + k, v.clone())
    out = dict(gen_name=gen_name, side=side, batch=batch, steps=steps, **{'seed_' + k: v for k, v in SEEDS.items()})
    if copy != 'hov3':
        out['copy'] = copy
    with torch.no_grad():
        for name, v in zip(OUT_NAMES, t.forward()):
            out['fwd_' + name] = v.numpy().astype(np.float32)
        tsf_cond = torch.cat([inp['input_G_tsf_obj'][:, 3:], inp['input_G_tsf_hand'][:, 3:]] +
                             ([inp['armask_tsf']] if copy == 'hov3' else []), 1)
        out['d_real_out'] = t._D.forward(torch.cat([inp['real_tsf'], tsf_cond], 1)).numpy()
    errs = []
    for s in range(steps):
        t.optimize_parameters()
        e = t.get_current_errors()
        errs.append([e[k] for k in e])
        if s == 0:
            # gradients of the D step and a few G gradient tensors of step 0 (G grads are overwritten per step)
            out['grad_D_model.0.weight'] = t._D.model[0].weight.grad.numpy().copy()
            out['grad_D_model.14.weight'] = t._D.model[14].weight.grad.numpy().copy()
            out['grad_G_bg_model.model.0.weight'] = t._G.bg_model.model[0].weight.grad.numpy().copy()
            out['grad_G_tsf_model.img_reg.0.weight'] = t._G.tsf_model.img_reg[0].weight.grad.numpy().copy()
            # (a conv bias feeding an instance norm has an identically-zero gradient: pick the SPADE gamma bias instead)
            if hasattr(t._G.src_model.resnets[0], 'norm_0'):
                out['grad_G_src_model.resnets.0.norm_0.mlp_gamma.bias'] = \
                    t._G.src_model.resnets[0].norm_0.mlp_gamma.bias.grad.numpy().copy()
            out['grad_G_obj_model.skippers.2.0.weight'] = t._G.obj_model.skippers[2][0].weight.grad.numpy().copy()
            if hasattr(t._G, 'attn_9'):
                out['grad_G_attn_9.fully_connect_layer.2.weight'] = \
                    t._G.attn_9.fully_connect_layer[2].weight.grad.numpy().copy()
            if hasattr(t._G, 'attn_2'):
                out['grad_G_attn_2.fully_connect_layer.0.bias'] = \
                    t._G.attn_2.fully_connect_layer[0].bias.grad.numpy().copy()
    out['error_keys'] = np.array(list(e.keys()))
    out['errors'] = np.array(errs, dtype=np.float64)
    gs, ds = t._G.state_dict(), t._D.state_dict()
    out['post_G_l2'] = np.array([float(v.double().norm()) for v in gs.values()])
    out['post_D_l2'] = np.array([float(v.double().norm()) for v in ds.values()])
    out['post_D_model.14.weight'] = ds['model.14.weight'].numpy()
    out['post_G_obj_model.img_reg.0.weight'] = gs['obj_model.img_reg.0.weight'].numpy()
    out['param_names_G'] = np.array(list(gs.keys()))
    out['param_names_D'] = np.array(list(ds.keys()))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), fname)
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB')


if __name__ == '__main__':
    which = sys.argv[1] if len(sys.argv) > 1 else 'generator_spade_attn'
    copy = sys.argv[2] if len(sys.argv) > 2 else 'hov3'
    short = {'generator_spade_attn': 'spade_attn', 'generator_spade': 'spade', 'generator_base': 'base',
             'generator_spade_attn_tiny': 'spade_attn_tiny'}[which]
    run(which, '%s_%s_64.npz' % (copy, short), copy=copy)
