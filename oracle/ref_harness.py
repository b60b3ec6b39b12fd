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


def reference_vertex_stage(cam, vertices, faces_idx, viewing_angle=30.0, copy='hov3'):
    """The reference's own vertex stage of ``render_fim_wim`` (utils/nmr.py:503-511) on CPU: ``orthographic_proj_withz_idrot``
    (nmr.py:109-140), the y flip, and neural_renderer's pure-Python ``look_at`` / ``vertices_to_faces`` loaded from their files
    (the package itself imports its CUDA extensions)."""
    import importlib.util
    import numpy as np
    install(copy)
    nmr_mod = importlib.import_module('utils.nmr')
    root = REF_HOV3 if copy == 'hov3' else REF_DEXYCB
    mods = {}
    for name in ('look_at', 'vertices_to_faces'):
        spec = importlib.util.spec_from_file_location(
            'hoig_ref_nr_' + name, os.path.join(root, 'thirdparty', 'neural_renderer', 'neural_renderer', name + '.py'))
        mods[name] = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mods[name])
    eye = [0, 0, -(1. / np.tan(np.radians(viewing_angle)) + 1)]                      # nmr.py:357
    proj = nmr_mod.orthographic_proj_withz_idrot(vertices, cam)
    proj[:, :, 1] *= -1
    v = mods['look_at'].look_at(proj, eye)
    faces = faces_idx if faces_idx.dim() == 3 else faces_idx[None].repeat(cam.shape[0], 1, 1)
    return mods['vertices_to_faces'].vertices_to_faces(v, faces)


def reference_rasterize_wrapper(faces, image_size, copy='hov3'):
    """The reference's own Python wrapper ``rasterize_face_index_map_and_weight_map(faces, image_size, False)``
    (thirdparty/neural_renderer/neural_renderer/rasterize.py: output allocation and fill :50-52, the call sequence of
    ``Rasterize.forward``, the vertical flips :334-338) executed on CPU over the ORACLE's restatement of the two CUDA kernels
    it calls (oracle/raster.c::oracle_rasterize_kernels) -- 'composition-pinned': pins the fill / flip / return conventions,
    not the kernels."""
    import ctypes
    import importlib.util
    import numpy as np
    import torch
    install(copy)
    root = REF_HOV3 if copy == 'hov3' else REF_DEXYCB
    here = os.path.dirname(os.path.abspath(__file__))
    clib = ctypes.CDLL(os.path.join(here, '_build', 'libhoig_oracle_c.so'))
    vp = ctypes.c_void_p
    clib.oracle_rasterize_kernels.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float,
                                              vp, vp, vp]

    def forward_face_index_map(faces, face_index_map, weight_map, depth_map, face_inv_map, faces_inv, image_size, near, far,
                               return_rgb, return_alpha, return_depth):
        f = np.ascontiguousarray(faces.numpy(), dtype=np.float32)
        fim, wim, inv = face_index_map.numpy(), weight_map.numpy(), faces_inv.numpy()
        clib.oracle_rasterize_kernels(f.ctypes.data, f.shape[0], f.shape[1], image_size, near, far, fim.ctypes.data,
                                      wim.ctypes.data, inv.ctypes.data)
        return face_index_map, weight_map, depth_map, face_inv_map

    for pkg in ('neural_renderer', 'neural_renderer.cuda'):
        if pkg not in sys.modules or not hasattr(sys.modules[pkg], '__path__'):
            _stub(pkg).__path__ = []
    _stub('neural_renderer.cuda.rasterize', forward_face_index_map=forward_face_index_map)
    sys.modules['neural_renderer.cuda'].rasterize = sys.modules['neural_renderer.cuda.rasterize']
    saved = (torch.cuda.FloatTensor, torch.cuda.IntTensor)
    torch.cuda.FloatTensor, torch.cuda.IntTensor = torch.FloatTensor, torch.IntTensor
    try:
        spec = importlib.util.spec_from_file_location(
            'hoig_ref_nr_rasterize', os.path.join(root, 'thirdparty', 'neural_renderer', 'neural_renderer', 'rasterize.py'))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        with torch.no_grad():
            return mod.rasterize_face_index_map_and_weight_map(faces.clone(), image_size, False)
    finally:
        torch.cuda.FloatTensor, torch.cuda.IntTensor = saved
