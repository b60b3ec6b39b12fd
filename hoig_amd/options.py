"""The option fields ``Trainer`` reads (SURVEY.md §8b "Construction"), with the values of the reference's training
recipe (scripts/train_hov3_ddp.sh:18-31): what bench.py, the smoke test and the parity tests build their ``opt`` from
when no reference ``options/`` parser is around.  A real run passes the reference's own argparse Namespace instead."""
import types


def is_dexycb(opt):
    """The HOIG_DexYCB copy's layouts (13 / 9 / 24 input channels, its MANO and camera conventions, its resume rules) are selected by the
    copy's own option value -- ``--dataset_mode ycb`` (scripts/train_ycb_ddp.sh:7) -- or by 'dexycb'."""
    m = str(getattr(opt, 'dataset_mode', 'hov3')).lower()
    return 'dex' in m or 'ycb' in m


def opt_namespace(**over):
    d = dict(gpu_ids='0', is_train=True, checkpoints_dir='/tmp/hoig_ckpt', name='t', map_name='uv_seg', cond_nc=2,
             local_rank=0, gen_name='generator_spade_attn', use_spade=True, repeat_num=6, norm_type='instance',
             image_size=256, tex_size=3, bg_both=False, use_vgg=True, mask_bce=True, lr_G=2e-4, lr_D=2e-4,
             G_adam_b1=0.5, G_adam_b2=0.999, D_adam_b1=0.5, D_adam_b2=0.999, lambda_D_prob=1.0, lambda_rec=10.0,
             lambda_tsf=10.0, lambda_mask=1.0, lambda_mask_smooth=1.0, final_lr=2e-6, nepochs_no_decay=15, nepochs_decay=15,
             load_path='None', load_epoch=-1, dataset_mode='hov3',
             # not a reference option: ImageNet VGG19 weights are not vendored and there is no network, so benchmarks and
             # parity tests opt in to deterministic surrogate weights (hoig_amd/models/trainer.py:_init_losses)
             vgg_surrogate=True)
    d.update(over)
    return types.SimpleNamespace(**d)
