"""The MANO hand layer on the device: what ``HandModelRecovery.get_details`` obtains from its third-party hand
model -- ``smplx.create(.., 'mano', use_pca=False, is_rhand=True, flat_hand_mean=True)`` in HOIG_HOv3 (models/networks/hmr.py:55,
84-85) and ``manopth.manolayer.ManoLayer(flat_hand_mean=False, ncomps=45, side='right', use_pca=True)`` in HOIG_DexYCB
(models/networks/hmr.py:55-60,85-86) -- as one HIP launch (hoig_amd/csrc/mano.hip, ``hoig_mano_lbs``) that writes the hand
vertices straight into the [hand | object] vertex buffer the rasteriser reads (hmr.py:87-93).

``HandModelRecovery`` mirrors the reference class (constructor argument, ``get_details(theta)`` and the keys of its result) for
both dataset copies (``variant='hov3' | 'dexycb'``).  The model itself comes from the licensed MANO_RIGHT.pkl, which is not
shipped with the reference (.gitignore:3): ``ManoModel.from_pickle`` reads it where it exists, tests use a synthetic model.
"""
import pickle

import numpy as np
import torch

from . import _lib as L

NUM_JOINTS, NUM_HAND_POSE, NUM_BETAS = 16, 45, 10


class ManoModel(object):
    """The tensors of a MANO model on the device, in the layout hoig_mano_lbs reads (include/hoig_kernels.h)."""

    def __init__(self, v_template, shapedirs, posedirs, J_regressor, parents, lbs_weights, hands_mean=None, hands_components=None,
                 device=None):
        if device is None:
            device = torch.device('cuda', torch.cuda.current_device())
        f64 = lambda a: np.asarray(a, np.float64)
        V = int(np.asarray(v_template).shape[0])
        posedirs = np.asarray(posedirs)
        if posedirs.shape == (V, 3, 9 * (NUM_JOINTS - 1)):               # the model file's (V,3,135) -> smplx's buffer (135, V*3)
            posedirs = posedirs.reshape(V * 3, -1).T
        if tuple(posedirs.shape) != (9 * (NUM_JOINTS - 1), V * 3):
            raise ValueError('posedirs must be (V,3,135) or (135,V*3), got %r' % (posedirs.shape,))
        if tuple(np.asarray(shapedirs).shape) != (V, 3, NUM_BETAS) or tuple(np.asarray(lbs_weights).shape) != (V, NUM_JOINTS) \
                or tuple(np.asarray(J_regressor).shape) != (NUM_JOINTS, V):
            raise ValueError('shapedirs (V,3,10), lbs_weights (V,16), J_regressor (16,V) expected')
        par = np.asarray(parents).astype(np.int64).copy()
        par[0] = 0
        if any(par[i] >= i for i in range(1, NUM_JOINTS)):
            raise ValueError('parents must precede their children')
        dev = lambda a, dt=torch.float32: torch.as_tensor(np.ascontiguousarray(a)).to(device=device, dtype=dt).contiguous()
        self.V, self.device = V, device
        self.v_template, self.shapedirs, self.posedirs = dev(v_template), dev(shapedirs), dev(posedirs)
        self.lbs_weights, self.parents = dev(lbs_weights), dev(par, torch.int32)
        # J = J_regressor . (v_template + shapedirs . betas): linear in betas, folded once in double precision
        self.j_template = dev(f64(J_regressor) @ f64(v_template))
        self.j_shapedirs = dev(np.einsum('jv,vkl->jkl', f64(J_regressor), f64(shapedirs)))
        self.hands_mean = dev(hands_mean) if hands_mean is not None else None
        self.hands_components = dev(hands_components) if hands_components is not None else None

    @classmethod
    def from_pickle(cls, path, device=None):
        """MANO_RIGHT.pkl as distributed (a latin-1 pickle of numpy / chumpy / scipy-sparse objects; chumpy must be importable for
        files that hold chumpy arrays)."""
        with open(path, 'rb') as f:
            d = pickle.load(f, encoding='latin1')
        arr = lambda x: np.asarray(x.todense() if hasattr(x, 'todense') else (x.r if hasattr(x, 'r') else x))
        kin = arr(d['kintree_table'])[0].astype(np.int64)
        return cls(arr(d['v_template']), arr(d['shapedirs'])[:, :, :NUM_BETAS], arr(d['posedirs']), arr(d['J_regressor']), kin,
                   arr(d['weights']), arr(d['hands_mean']), arr(d['hands_components']), device=device)

    @classmethod
    def from_dict(cls, d, device=None):
        return cls(d['v_template'], d['shapedirs'], d['posedirs'], d['J_regressor'], d['parents'], d['lbs_weights'],
                   d.get('hands_mean'), d.get('hands_components'), device=device)


def mano_vertices(model, root, hand, betas, transl=None, use_pca=False, flat_hand_mean=True, ncomps=NUM_HAND_POSE, out=None,
                  return_joints=False):
    """root (B,3), hand (B,45) axis-angle or (B,ncomps) PCA coefficients, betas (B,10), transl (B,3) | None -> vertices (B,V,3)
    [, posed joints (B,16,3)].  `out`: a (B, >=V, 3) tensor whose first V rows receive the vertices (returned as that view)."""
    if not root.is_cuda:
        raise NotImplementedError('hoig_amd.mano runs on the HIP device only (no CPU path)')
    B = int(root.shape[0])
    f = lambda t: None if t is None else t.to(device=model.device, dtype=torch.float32).contiguous()
    root, hand, betas, transl = f(root), f(hand), f(betas), f(transl)
    nh = ncomps if use_pca else NUM_HAND_POSE
    if tuple(root.shape) != (B, 3) or tuple(hand.shape) != (B, nh) or tuple(betas.shape) != (B, NUM_BETAS) or \
            (transl is not None and tuple(transl.shape) != (B, 3)):
        raise ValueError('root (B,3), hand (B,%d), betas (B,10), transl (B,3) expected' % nh)
    comps = None
    if use_pca:
        if model.hands_components is None:
            raise ValueError('use_pca=True needs a model with hands_components')
        comps = model.hands_components[:ncomps].contiguous()
    mean = None if flat_hand_mean else model.hands_mean
    if not flat_hand_mean and mean is None:
        raise ValueError('flat_hand_mean=False needs a model with hands_mean')
    if out is None:
        out = torch.empty(B, model.V, 3, dtype=torch.float32, device=model.device)
    if out.dim() != 3 or out.shape[0] != B or out.shape[1] < model.V or out.shape[2] != 3 or not out.is_contiguous() \
            or out.dtype != torch.float32:
        raise ValueError('out must be a contiguous fp32 (B, >= V, 3) tensor')
    joints = torch.empty(B, NUM_JOINTS, 3, dtype=torch.float32, device=model.device) if return_joints else None
    p = lambda t: None if t is None else t.data_ptr()
    L.call('hoig_mano_lbs', p(model.v_template), p(model.shapedirs), p(model.posedirs), p(model.j_template), p(model.j_shapedirs),
           p(model.lbs_weights), p(model.parents), p(mean), p(comps), int(ncomps if use_pca else 0), model.V, p(root), p(hand),
           p(betas), p(transl), p(out), int(out.shape[1]), p(joints), B, torch.cuda.current_stream().cuda_stream)
    verts = out[:, :model.V]
    return (verts, joints) if return_joints else verts


class HandModelRecovery(object):
    """models/networks/hmr.py:45-95 of both dataset copies: ``get_details(theta)`` -> {'cam', 'verts', 'objName'}.
    `mano` is a ManoModel or the path of a MANO_RIGHT.pkl (the reference passes the model directory to smplx / manopth)."""

    def __init__(self, mano, feature_dim=2048, theta_dim=31, variant='hov3', device=None):
        if variant not in ('hov3', 'dexycb'):
            raise ValueError("variant must be 'hov3' or 'dexycb'")
        self.model = mano if isinstance(mano, ManoModel) else ManoModel.from_pickle(mano, device=device)
        self.variant, self.feature_dim, self.theta_dim = variant, feature_dim, theta_dim

    def get_details(self, theta):
        bs = theta['cam'].shape[0]
        dev = self.model.device
        pose, shape = theta['pose'].to(dev), theta['shape'].to(dev)
        vobj = theta['vertices_obj'].to(device=dev, dtype=torch.float32)
        verts = torch.empty(bs, self.model.V + vobj.shape[1], 3, dtype=torch.float32, device=dev)
        if self.variant == 'hov3':          # hmr.py:77-85: full axis-angle pose, flat hand mean, translation `handtrans`
            mano_vertices(self.model, pose[:, :3], pose[:, 3:], shape, theta['handtrans'].to(dev), out=verts)
        else:                               # HOIG_DexYCB hmr.py:83-86: pose[:, :48] = root + 45 PCA coefficients, pose[:, 48:51] = trans
            mano_vertices(self.model, pose[:, :3], pose[:, 3:48], shape, pose[:, 48:51], use_pca=True, flat_hand_mean=False, out=verts)
        verts[:, self.model.V:] = vobj      # torch.cat([vertices_hand, vertices_obj], dim=1) without the copy of the hand half
        cam, trans = theta['cam'].to(dev), theta['trans'].to(dev)
        return {'cam': torch.cat([cam.reshape(bs, -1), trans.reshape(bs, -1)], dim=1), 'verts': verts, 'objName': theta['objName']}
