"""Shared builders for the parity tests (test infrastructure)."""
import os
import types

import numpy as np
import torch

from oracle import hogan_oracle as O
from hoig_amd import synthetic

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
SEEDS = dict(G=8, D=9, VGG=10, inputs=8)
OUT_NAMES = ['fake_src_bg', 'fake_tsf_bg', 'fake_src_imgs', 'fake_tsf_imgs', 'fake_masks_bg', 'fake_masks_hand']


def opt_namespace(**over):
    """The option fields Trainer reads, with the values of scripts/train_hov3_ddp.sh."""
    d = dict(gpu_ids='0', is_train=True, checkpoints_dir='/tmp/hoig_ckpt', name='t', map_name='uv_seg', cond_nc=2,
             local_rank=0, gen_name='generator_spade_attn', use_spade=True, repeat_num=6, norm_type='instance',
             image_size=256, tex_size=3, bg_both=False, use_vgg=True, mask_bce=True, lr_G=2e-4, lr_D=2e-4,
             G_adam_b1=0.5, G_adam_b2=0.999, D_adam_b1=0.5, D_adam_b2=0.999, lambda_D_prob=1.0, lambda_rec=10.0,
             lambda_tsf=10.0, lambda_mask=1.0, lambda_mask_smooth=1.0, final_lr=2e-6, nepochs_decay=15,
             load_path='None', load_epoch=-1, dataset_mode='hov3')
    d.update(over)
    return types.SimpleNamespace(**d)


def seeded_state(gen_name, dataset='hov3'):
    cfg = O.make_cfg(gen_name, dataset)
    sdG = O.make_weights(O.gen_param_shapes(cfg), seed=SEEDS['G'], mode='random')
    sdD = O.make_weights(O.disc_param_shapes(cfg), seed=SEEDS['D'], mode='random')
    sdV = O.make_weights(O.vgg_param_shapes(), seed=SEEDS['VGG'], kind='vgg')
    return cfg, sdG, sdD, sdV


def oracle_trainer(gen_name, batch, side, dataset='hov3'):
    cfg, sdG, sdD, sdV = seeded_state(gen_name, dataset)
    ot = O.OracleTrainer(cfg, sdG, sdD, sdV)
    ot.set_prepared_input(synthetic.make_inputs(batch, side, seed=SEEDS['inputs'], dataset=dataset))
    return ot


def product_trainer(gen_name, batch, side, dataset='hov3', **over):
    from hoig_amd.models import ModelsFactory
    cfg, sdG, sdD, sdV = seeded_state(gen_name, dataset)
    opt = opt_namespace(gen_name=gen_name, dataset_mode=dataset, **over)
    m = ModelsFactory.get_by_name('trainer', opt)
    m._G.load_state_dict(sdG)
    m._D.load_state_dict(sdD)
    m._crt_tsf.vgg.load_state_dict(sdV)
    m.set_input(synthetic.make_inputs(batch, side, seed=SEEDS['inputs'], dataset=dataset))
    return m


def load_golden(name):
    return np.load(os.path.join(GOLDEN_DIR, name), allow_pickle=False)
