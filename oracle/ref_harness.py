"""Reference-import harness (TEST INFRASTRUCTURE, this container only).

Imports the *reference's own* Python modules from /root/reference/HOIG_HOv3 on
CPU so that (a) the oracle restatement in ``oracle/hogan_oracle.py`` can be
pinned against them and (b) golden vectors can be generated
(``tests/golden/make_golden.py``).  Nothing here may run on the GPU box: the
reference tree does not exist there.

Recipe: stub the modules that are absent from the image
(h5py, cv2, smplx, neural_renderer, torchvision, the two CUDA extensions),
make ``.cuda()`` the identity, and build ``Trainer`` with ``__new__`` so the
renderer assets are never touched.

The two CUDA-only ops of the reference (``block_extractor_cuda``,
``local_attn_reshape_cuda``; thirdparty/block_extractor/block_extractor_kernel.cu:20-170,
thirdparty/local_attn_reshape/local_attn_reshape_kernel.cu:20-108) cannot run
here.  With ``attn_ops='oracle'`` the harness substitutes the oracle's
restatement of K1-K4 for those two modules ONLY, so that the reference's own
composition code (extract_attn.py:23-29, generator.py:480-491) executes for
real on top of it.  Vectors produced that way are labelled
``composition-pinned`` (the kernels themselves stay pinned only by the three
checks the reference's manual scripts define, see tests/test_oracle_attn.py).
"""
import os
import sys
import types
import importlib

REF_ROOT = os.environ.get('HOIG_REFERENCE_ROOT', '/root/reference')
REF_HOV3 = os.path.join(REF_ROOT, 'HOIG_HOv3')
REF_DEXYCB = os.path.join(REF_ROOT, 'HOIG_DexYCB')


def available():
    return os.path.isdir(os.path.join(REF_HOV3, 'models'))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


_installed = {}


def install(copy='hov3', attn_ops='oracle'):
    """Make ``import models.trainer`` resolve to the reference copy `copy`."""
    import torch
    import torch.nn as nn

    root = REF_HOV3 if copy == 'hov3' else REF_DEXYCB
    if _installed.get('root') == root:
        return
    if _installed:
        raise RuntimeError('ref_harness: one reference copy per process')
    sys.dont_write_bytecode = True

    for name in ['h5py', 'cv2', 'neural_renderer', 'smplx']:
        _stub(name)
    _stub('smplx.lbs', transform_mat=lambda *a, **k: None)
    sys.modules['smplx'].lbs = sys.modules['smplx.lbs']

    # torchvision: only vgg19().features and make_grid are touched on the path
    def _vgg19(pretrained=False, **kw):
        cfg = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M',
               512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M']
        layers, cin = [], 3
        for v in cfg:
            if v == 'M':
                layers.append(nn.MaxPool2d(2, 2))
            else:
                layers += [nn.Conv2d(cin, v, 3, padding=1), nn.ReLU(inplace=True)]
                cin = v
        return types.SimpleNamespace(features=nn.Sequential(*layers))

    def _make_grid(t, nrow=8, padding=0, **kw):
        # torchvision.utils.make_grid(padding=0) for a (B,C,H,W) batch
        b, c, h, w = t.shape
        if c == 1:
            t = t.repeat(1, 3, 1, 1)
            c = 3
        ncol = min(nrow, b)
        nrw = (b + ncol - 1) // ncol
        grid = t.new_zeros(c, h * nrw, w * ncol)
        for i in range(b):
            r, q = i // ncol, i % ncol
            grid[:, r * h:(r + 1) * h, q * w:(q + 1) * w] = t[i]
        return grid

    tv = _stub('torchvision')
    tv.utils = _stub('torchvision.utils', make_grid=_make_grid)
    tv.models = _stub('torchvision.models', vgg19=_vgg19)
    tv.transforms = _stub('torchvision.transforms')
    tv.transforms.functional = _stub('torchvision.transforms.functional')

    _stub('block_extractor_cuda')
    _stub('local_attn_reshape_cuda')
    if attn_ops == 'oracle':
        here = os.path.dirname(os.path.abspath(__file__))
        if os.path.dirname(here) not in sys.path:
            sys.path.insert(0, os.path.dirname(here))
        from oracle import hogan_oracle as O

        class BlockExtractor(nn.Module):          # API of block_extractor.py:45-54
            def __init__(self, kernel_size=3):
                super().__init__()
                self.kernel_size = kernel_size

            def forward(self, source, flow_field):
                return O.block_extract(source, flow_field, self.kernel_size)

        class LocalAttnReshape(nn.Module):        # API of local_attn_reshape.py:40-46
            def forward(self, inputs, kernel_size=3):
                return O.local_attn_reshape(inputs, kernel_size)

        for pkg in ['thirdparty', 'thirdparty.block_extractor', 'thirdparty.local_attn_reshape']:
            _stub(pkg).__path__ = []
        _stub('thirdparty.block_extractor.block_extractor', BlockExtractor=BlockExtractor)
        _stub('thirdparty.local_attn_reshape.local_attn_reshape', LocalAttnReshape=LocalAttnReshape)

    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self

    sys.path.insert(0, root)
    _installed['root'] = root


def namespace(**over):
    """The argparse fields Trainer reads (options/*.py defaults + the flags of
    scripts/train_hov3_ddp.sh:23-52)."""
    d = dict(gpu_ids='', is_train=True, checkpoints_dir='/tmp/hoig_ref_ckpt', name='ref',
             map_name='uv_seg', cond_nc=2, local_rank=0, gen_name='generator_spade_attn',
             use_spade=True, repeat_num=6, norm_type='instance', image_size=256, tex_size=3,
             bg_both=False, use_vgg=True, mask_bce=True, lr_G=2e-4, lr_D=2e-4,
             G_adam_b1=0.5, G_adam_b2=0.999, D_adam_b1=0.5, D_adam_b2=0.999,
             lambda_D_prob=1.0, lambda_rec=10.0, lambda_tsf=10.0, lambda_mask=1.0,
             lambda_mask_smooth=1.0, final_lr=2e-6, nepochs_decay=15, load_path='None',
             load_epoch=-1)
    d.update(over)
    return types.SimpleNamespace(**d)


def build_reference_trainer(opt, copy='hov3', attn_ops='oracle'):
    """A reference ``Trainer`` whose G/D/optimisers/losses were built by the
    reference's own code (trainer.py:258-322) without the renderer."""
    import torch
    install(copy, attn_ops)
    trainer_mod = importlib.import_module('models.trainer')
    vgg_mod = importlib.import_module('models.networks.vgg19')
    T = trainer_mod.Trainer
    t = T.__new__(T)
    t._name = 'Trainer'
    t._opt = opt
    t._gpu_ids = opt.gpu_ids
    t._is_train = opt.is_train
    t._use_ddp = False
    t._Tensor = torch.Tensor
    t._save_dir = os.path.join(opt.checkpoints_dir, opt.name)
    t._G = t._create_generator()
    t._D = t._create_discriminator()
    if opt.is_train:
        t._init_train_vars()
        t._crt_l1 = torch.nn.L1Loss()
        t._crt_mask = torch.nn.BCELoss() if opt.mask_bce else torch.nn.MSELoss()
        t._crt_tsf = vgg_mod.VGGLoss(vgg=vgg_mod.Vgg19())
        for n in ['_loss_g_rec', '_loss_g_tsf', '_loss_g_adv', '_loss_g_smooth', '_loss_g_mask',
                  '_loss_g_mask_smooth', '_d_real', '_d_fake']:
            setattr(t, n, torch.zeros(1))
    return t


def reference_input_prep(raster, bg_both=False, copy='hov3'):
    """Runs the reference's OWN ``HandRecoveryFlow.forward`` (models/trainer.py:46-145) and the MANORenderer methods it
    calls (utils/nmr.py:567-595,874-968,973-1100) on CPU, with ONLY the rasteriser call ``render_fim_wim`` (neural_renderer
    CUDA, needs MANO + YCB assets) replaced by the seeded synthetic rasteriser outputs of
    ``hoig_amd.synthetic.make_raster`` and the per-object buffers by its synthetic tables.  Returns the reference's return
    tuple (12 entries, the last one None)."""
    import torch
    install(copy)
    trainer_mod = importlib.import_module('models.trainer')
    nmr_mod = importlib.import_module('utils.nmr')
    names = trainer_mod.OBJNAMES

    render = nmr_mod.MANORenderer.__new__(nmr_mod.MANORenderer)
    torch.nn.Module.__init__(render)
    render.image_size = 256
    for k, tb in raster['tables'].items():
        n = names[k]
        render.register_buffer('faces_' + n, torch.arange(tb['n_faces'] * 3, dtype=torch.int32).reshape(-1, 3) % 64)
        for key in ('map_fn', 'sem_full', 'fim_uv', 'wim_uv', 'faces_uv_coord', 'obj_tex_img'):
            render.register_buffer('%s_%s' % (key, n), tb[key].clone())
    calls = []

    def render_fim_wim(cam, vertices, obj_name, faces=None):           # src view, then ref view, per sample (trainer.py:66,74)
        i, which = len(calls) // 2, 'src' if len(calls) % 2 == 0 else 'ref'
        calls.append(which)
        nf = raster['tables'][raster['obj_ids'][i]]['n_faces']
        return (raster['src_faces'][i:i + 1, :nf].clone(), raster[which + '_fim'][i:i + 1].clone(),
                raster[which + '_wim'][i:i + 1].clone())
    render.render_fim_wim = render_fim_wim

    bs = raster['src_img'].shape[0]
    info = dict(objName=torch.tensor(raster['obj_ids'])[:, None], cam=torch.zeros(bs, 3), verts=torch.zeros(bs, 64, 3))
    hdr = trainer_mod.HandRecoveryFlow.__new__(trainer_mod.HandRecoveryFlow)
    torch.nn.Module.__init__(hdr)
    hdr._opt = types.SimpleNamespace(bg_both=bg_both)
    hdr._hmr = types.SimpleNamespace(get_details=lambda mano: info)
    hdr._render = render
    with torch.no_grad():
        return hdr.forward(raster['src_img'].clone(), raster['ref_img'].clone(), None, None)
