"""Shared builders for the parity tests (test infrastructure)."""
import os
import numpy as np
import torch

from oracle import hogan_oracle as O
from hoig_amd import synthetic
from hoig_amd.options import opt_namespace  # noqa: F401  (re-exported for the tests)

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
SEEDS = dict(G=8, D=9, VGG=10, inputs=8)
OUT_NAMES = ['fake_src_bg', 'fake_tsf_bg', 'fake_src_imgs', 'fake_tsf_imgs', 'fake_masks_bg', 'fake_masks_hand']


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


def product_trainer(gen_name, batch, side, dataset='hov3', use_ddp=False, inputs=None, **over):
    from hoig_amd.models import ModelsFactory
    cfg, sdG, sdD, sdV = seeded_state(gen_name, dataset)
    opt = opt_namespace(gen_name=gen_name, dataset_mode=dataset, **over)
    m = ModelsFactory.get_by_name('trainer', opt, use_ddp=use_ddp)
    m._G.load_state_dict(sdG)
    m._D.load_state_dict(sdD)
    m._crt_tsf.vgg.load_state_dict(sdV)
    m.set_input(inputs if inputs is not None else synthetic.make_inputs(batch, side, seed=SEEDS['inputs'], dataset=dataset))
    return m


def load_golden(name):
    return np.load(os.path.join(GOLDEN_DIR, name), allow_pickle=False)


# ---- rasteriser test helpers (oracle/raster.c through ctypes; test infrastructure) ----
def oracle_rasterize(faces, image_size=256, near=0.1, far=100.0):
    """faces (B,F,3,3) CPU tensor -> (fim int32 (B,S,S), wim (B,S,S,3)) by the plain-C oracle."""
    import ctypes
    import numpy as np
    import torch
    lib = ctypes.CDLL(os.path.join(os.path.dirname(GOLDEN_DIR), os.pardir, 'oracle', '_build', 'libhoig_oracle_c.so'))
    f = np.ascontiguousarray(faces.numpy(), dtype=np.float32)
    B, F = f.shape[0], f.shape[1]
    S = image_size
    fim = np.empty((B, S, S), np.int32)
    wim = np.empty((B, S, S, 3), np.float32)
    inv = np.empty((B, F, 9), np.float32)
    vp = ctypes.c_void_p
    lib.oracle_rasterize_fim_wim.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float, vp,
                                             vp, vp]
    lib.oracle_rasterize_fim_wim(f.ctypes.data, B, F, S, near, far, fim.ctypes.data, wim.ctypes.data, inv.ctypes.data)
    return torch.from_numpy(fim), torch.from_numpy(wim)


def synthetic_mesh_faces(batch, n_random=1500, seed=5):
    """(B,F,3,3) face tensors in the rasteriser's input convention (x, y in [-1,1], y up, z = depth): a rotated, lumpy UV
    sphere (closed surface: back faces, depth order, shared edges) plus random triangles of all sizes, some partly or fully
    outside the image, some beyond near / far, some degenerate."""
    import numpy as np
    import torch
    out = []
    for b in range(batch):
        g = np.random.default_rng([seed, b])
        nu, nv = 48, 24
        u = np.linspace(0, 2 * np.pi, nu, endpoint=False)
        v = np.linspace(0.05, np.pi - 0.05, nv)
        uu, vv = np.meshgrid(u, v)
        r = 0.55 + 0.08 * np.sin(3 * uu) * np.sin(2 * vv)
        P = np.stack([r * np.sin(vv) * np.cos(uu), r * np.cos(vv), r * np.sin(vv) * np.sin(uu)], -1).reshape(-1, 3)
        a = g.uniform(0, 2 * np.pi, 3)
        Rx = np.array([[1, 0, 0], [0, np.cos(a[0]), -np.sin(a[0])], [0, np.sin(a[0]), np.cos(a[0])]])
        Ry = np.array([[np.cos(a[1]), 0, np.sin(a[1])], [0, 1, 0], [-np.sin(a[1]), 0, np.cos(a[1])]])
        P = P @ Rx.T @ Ry.T + np.array([g.uniform(-0.3, 0.3), g.uniform(-0.3, 0.3), 2.7])
        idx = []
        for j in range(nv - 1):
            for i in range(nu):
                p00, p01 = j * nu + i, j * nu + (i + 1) % nu
                p10, p11 = (j + 1) * nu + i, (j + 1) * nu + (i + 1) % nu
                idx += [(p00, p10, p01), (p01, p10, p11)]
        sphere = P[np.array(idx)]
        c = g.uniform(-1.3, 1.3, size=(n_random, 1, 2))
        size = g.choice([0.01, 0.05, 0.2, 0.8], size=(n_random, 1, 1))
        xy = c + g.uniform(-1, 1, size=(n_random, 3, 2)) * size
        z = g.uniform(0.05, 5.0, size=(n_random, 3, 1)) * g.choice([1.0, 1.0, 1.0, 40.0], size=(n_random, 1, 1))
        rnd = np.concatenate([xy, z], -1)
        if n_random >= 20:
            rnd[:10, 1] = rnd[:10, 0]                               # degenerate (two equal vertices)
            rnd[10:20, :, 2] = 1.5                                  # coplanar depth ties among overlapping faces
            rnd[10:20, :, :2] = g.uniform(-0.2, 0.2, size=(1, 3, 2)) + g.uniform(-0.01, 0.01, size=(10, 3, 2))
        out.append(np.concatenate([sphere, rnd], 0).astype(np.float32))
    return torch.from_numpy(np.stack(out))
