"""Trainer.set_input on the RAW dataloader batch (imageA/B, maskA/B, manoA/B: what train_ddp.py:92 passes; trainer.py:324-362)
through hoig_amd.hand_recovery.HandRecoveryFlow -- MANO layer -> projection -> rasteriser -> tensor stage, all on the device --
against the chain of the three stages' CPU oracles (oracle/mano_oracle.py, oracle/raster.c, oracle/input_prep_oracle.py) on a
synthetic MANO model, synthetic object meshes and synthetic renderer tables standing in for the licensed / unshipped assets.
The MANO and rasteriser oracles are parity-unpinned, so this test pins the WIRING: stage order, vertex
counts per object, the source view's projected faces handed to the tensor stage, padding faces that no pixel hits, batch staging.

The device's vertices differ from the float64 oracle's by ~2e-6, i.e. ~2e-3 pixel after projection: a triangle edge then falls on
the other side of a pixel centre for a few hundred of the ~100 000 covered pixels of this scene of 1 900 random triangles, and
those pixels (and their 3x3 / 15x15 erosion neighbourhoods) legitimately differ: the bounds below are fractions of differing
values, which a wiring mistake (another object's faces, the wrong view's vertices, a missing y flip) exceeds by an order of
magnitude."""
import numpy as np
import pytest
import torch

from common import opt_namespace, oracle_rasterize
from hoig_amd import synthetic
from oracle import input_prep_oracle as P
from oracle import mano_oracle as M

pytestmark = pytest.mark.gpu
S = 256


def _assets(obj_ids, seed):
    """object id -> {'faces': (F,3) over the [778 hand | n object] vertex buffer, + the renderer tables}, and n object vertices."""
    out, nv = {}, {}
    for k in sorted(set(obj_ids)):
        tb = synthetic.make_object_tables(k, seed)
        g = np.random.Generator(np.random.Philox(key=[seed, 77 + k]))
        n_hand, n_obj = synthetic.N_HAND_FACES, tb['n_faces'] - synthetic.N_HAND_FACES
        vo = 120 + 10 * k
        hand = g.integers(0, 778, size=(n_hand, 3))
        obj = 778 + g.integers(0, vo, size=(n_obj, 3))
        obj[0] = (778, 778 + vo - 1, 778 + 1)                     # the last object vertex is used: length = 778 + vo
        out[k] = dict(tb, faces=torch.from_numpy(np.concatenate([hand, obj]).astype(np.int64)))
        nv[k] = vo
    return out, nv


def _raw_batch(B, seed, obj_ids, nv):
    g = np.random.Generator(np.random.Philox(key=[seed, B]))
    f = lambda a: torch.from_numpy(np.asarray(a, np.float32))
    vmax = max(nv.values())
    cam = np.tile(np.array([[600., 0., 128.], [0., 600., 128.], [0., 0., 1.]], np.float32), (B, 1, 1))
    trans = np.tile(np.array([[1., 0., 0.], [0., 1., 0.]], np.float32), (B, 1, 1))

    def mano(view):
        gg = np.random.Generator(np.random.Philox(key=[seed + 1000 * view, B]))
        t = np.concatenate([gg.uniform(-0.05, 0.05, (B, 2)), gg.uniform(-0.55, -0.45, (B, 1))], axis=1)      # in front of the camera
        vobj = np.zeros((B, vmax, 3), np.float32)
        for i, k in enumerate(obj_ids):
            vobj[i, :nv[k]] = t[i] + gg.uniform(-0.08, 0.08, (nv[k], 3)) + (0.06, 0.0, 0.0)
        return {'pose': f(np.concatenate([gg.standard_normal((B, 3)) * 0.5, gg.standard_normal((B, 45)) * 0.3], axis=1)),
                'shape': f(gg.standard_normal((B, 10))), 'handtrans': f(t), 'cam': f(cam), 'trans': f(trans),
                'vertices_obj': f(vobj), 'objName': torch.tensor(obj_ids)}
    arm = f((g.uniform(size=(B, 1, S, S)) < 0.1) * 1.99)
    return dict(imageA=f(g.uniform(-1, 1, (B, 3, S, S))), imageB=f(g.uniform(-1, 1, (B, 3, S, S))), maskA=arm, maskB=arm.flip(0),
                manoA=mano(0), manoB=mano(1))


def _oracle_chain(batch, md, assets, obj_ids):
    """The same three stages on the CPU: float64 MANO, the reference's projection arithmetic in torch fp32 (hoig_amd.raster's
    batched torch ops run on CPU tensors too: plumbing, not a kernel), the plain-C rasteriser, the tensor-stage oracle."""
    from hoig_amd import raster
    views = []
    fmax = max(assets[k]['n_faces'] for k in obj_ids)
    for mano in (batch['manoA'], batch['manoB']):
        pose = mano['pose'].numpy()
        v, _ = M.smplx_mano_forward(md, pose[:, :3], pose[:, 3:], mano['shape'].numpy(), mano['handtrans'].numpy())
        verts = torch.cat([torch.from_numpy(v.astype(np.float32)), mano['vertices_obj']], dim=1)
        cam = torch.cat([mano['cam'].reshape(len(obj_ids), -1), mano['trans'].reshape(len(obj_ids), -1)], dim=1)
        faces = torch.full((len(obj_ids), fmax, 3, 3), -1.0e6)
        for i, k in enumerate(obj_ids):
            length = int(assets[k]['faces'].max()) + 1
            fi = raster.project_to_faces(cam[i:i + 1], verts[i:i + 1, :length], assets[k]['faces'])
            faces[i, :fi.shape[1]] = fi[0]
        fim, wim = oracle_rasterize(faces, S)
        views.append((faces, fim, wim))
    (sf, sfim, swim), (_, rfim, rwim) = views
    out = P.prepare_inputs(batch['imageA'], batch['imageB'], sf, sfim, swim, rfim, rwim, [assets[k] for k in obj_ids], False, False)
    return P.to_prepared(out, batch['imageA'], batch['imageB'], batch['maskA'], batch['maskB']), sfim, rfim


def test_trainer_stages_a_raw_batch_like_the_oracle_chain():
    from hoig_amd import ops
    from hoig_amd.mano import ManoModel
    from hoig_amd.models import ModelsFactory
    ops.set_precision('bf16x3')
    try:
        B, seed = 3, 21
        obj_ids = [2, 5, 2]
        assets, nv = _assets(obj_ids, seed)
        md = M.synthetic_model(4)
        batch = _raw_batch(B, seed, obj_ids, nv)
        opt = opt_namespace(gen_name='generator_spade_attn', local_rank=0, image_size=S)
        opt.mano_model = ManoModel.from_dict(md)
        opt.object_assets = assets
        torch.manual_seed(3)
        model = ModelsFactory.get_by_name('trainer', opt, use_ddp=False)
        model.set_train()
        model.set_input(batch)
        want, sfim, rfim = _oracle_chain(batch, md, assets, obj_ids)
        # the rasterised scene is not trivial: hand and object faces are both visible in both views
        for fim in (sfim, rfim):
            assert int(((fim >= 0) & (fim < synthetic.N_HAND_FACES)).sum()) > 2000 and int((fim >= synthetic.N_HAND_FACES).sum()) > 2000

        def frac_differing(a, b, tol=1e-4):
            return float(((a.cpu() - b).abs() > tol).float().mean())
        # masks and inputs: equal except around the few pixels where a triangle edge crossed a pixel centre (see the module docstring)
        assert frac_differing(model._hand_mask, want['hand_mask']) < 1e-2
        assert frac_differing(model._bg_mask, want['bg_mask']) < 1e-2
        assert frac_differing(model._input_G_src_obj, want['input_G_src_obj']) < 2e-2
        assert frac_differing(model._input_G_tsf_obj, want['input_G_tsf_obj']) < 2e-2
        assert frac_differing(model._input_G_src_hand, want['input_G_src_hand']) < 2e-2
        assert frac_differing(model._input_G_tsf_hand, want['input_G_tsf_hand']) < 2e-2
        assert frac_differing(model._input_G_bg, want['input_G_bg']) < 5e-2          # (15x15 erosion spreads a flipped pixel)
        assert frac_differing(model._T, want['T']) < 2e-2
        assert torch.equal(model._real_src.cpu(), batch['imageA']) and torch.equal(model._armask_tsf.cpu(), batch['maskB'])
        # and a step runs on it
        model.optimize_parameters()
        assert all(np.isfinite(v) for v in model.get_current_errors().values())
    finally:
        ops.set_precision('f32')


def test_raw_batch_without_assets_is_refused_with_a_reason():
    from hoig_amd.models import ModelsFactory
    opt = opt_namespace(gen_name='generator_spade_attn', local_rank=0, image_size=S)
    model = ModelsFactory.get_by_name('trainer', opt, use_ddp=False)
    batch = _raw_batch(1, 3, [0], {0: 100})
    with pytest.raises(NotImplementedError, match='mano_model'):
        model.set_input(batch)
    with pytest.raises(KeyError):
        model.set_input({'imageA': batch['imageA']})
    del model
