"""hoig_mano_lbs (hoig_amd/mano.py, csrc/mano.hip) against the CPU restatement of smplx.lbs / manopth's ManoLayer
(oracle/mano_oracle.py), through the mirror of the reference's HandModelRecovery (models/networks/hmr.py:45-95, both dataset
copies).  fp32 against float64: 2e-6 absolute on coordinates of ~0.1 (metres)."""
import numpy as np
import pytest
import torch

from oracle import mano_oracle as M

pytestmark = pytest.mark.gpu
ATOL = 2e-6


def _params(B, seed=3):
    g = np.random.Generator(np.random.Philox(key=[seed, B]))
    f = lambda a: a.astype(np.float32)
    return (f(g.standard_normal((B, 3))), f(g.standard_normal((B, 45)) * 0.4), f(g.standard_normal((B, 10))),
            f(g.standard_normal((B, 3)) * 0.3))


@pytest.mark.parametrize('B', [1, 7, 32])
def test_axis_angle_form_vs_oracle(B):
    from hoig_amd.mano import ManoModel, mano_vertices
    md = M.synthetic_model(B)
    model = ManoModel.from_dict(md)
    root, hand, betas, t = _params(B)
    if B > 1:
        root[0] = 0                                             # the zero rotation (angle = |1e-8|)
        hand[1] = 0
    want_v, want_j = M.smplx_mano_forward(md, root, hand, betas, t)
    c = lambda a: torch.from_numpy(a).cuda()
    v, j = mano_vertices(model, c(root), c(hand), c(betas), c(t), return_joints=True)
    assert tuple(v.shape) == (B, 778, 3) and tuple(j.shape) == (B, 16, 3)
    assert np.abs(v.cpu().numpy() - want_v).max() < ATOL
    assert np.abs(j.cpu().numpy() - want_j).max() < ATOL
    v2 = mano_vertices(model, c(root), c(hand), c(betas))       # no translation
    assert np.abs(v2.cpu().numpy() - (want_v - t[:, None])).max() < ATOL


@pytest.mark.parametrize('ncomps', [45, 12])
def test_pca_form_vs_oracle(ncomps):
    from hoig_amd.mano import ManoModel, mano_vertices
    md = M.synthetic_model(5)
    model = ManoModel.from_dict(md)
    root, coeffs, betas, t = _params(6, seed=9)
    coeffs = coeffs[:, :ncomps].copy()
    sub = dict(md, hands_components=md['hands_components'][:ncomps])
    want_v, _ = M.manopth_forward(sub, np.concatenate([root, coeffs], axis=1), betas, t)
    c = lambda a: torch.from_numpy(a).cuda()
    v = mano_vertices(model, c(root), c(coeffs), c(betas), c(t), use_pca=True, flat_hand_mean=False, ncomps=ncomps)
    assert np.abs(v.cpu().numpy() - want_v).max() < ATOL


@pytest.mark.parametrize('variant', ['hov3', 'dexycb'])
def test_hand_model_recovery_get_details(variant):
    """hmr.py:69-95: verts = [hand | object] vertices, cam = [cam | trans] flattened, objName passed through."""
    from hoig_amd.mano import ManoModel, HandModelRecovery
    md = M.synthetic_model(11)
    hmr = HandModelRecovery(ManoModel.from_dict(md), variant=variant)
    B, VO = 3, 50
    root, hand, betas, t = _params(B, seed=4)
    g = torch.Generator().manual_seed(2)
    theta = {'cam': torch.randn(B, 3, 3, generator=g), 'trans': torch.randn(B, 2, 3, generator=g),
             'shape': torch.from_numpy(betas), 'vertices_obj': torch.randn(B, VO, 3, generator=g), 'objName': torch.arange(B)}
    if variant == 'hov3':
        theta['pose'] = torch.from_numpy(np.concatenate([root, hand], axis=1))
        theta['handtrans'] = torch.from_numpy(t)
        want_v, _ = M.smplx_mano_forward(md, root, hand, betas, t)
    else:
        theta['pose'] = torch.from_numpy(np.concatenate([root, hand, t], axis=1))          # 3 + 45 PCA coefficients + 3
        want_v, _ = M.manopth_forward(md, np.concatenate([root, hand], axis=1), betas, t)
    out = hmr.get_details(theta)
    assert tuple(out['verts'].shape) == (B, 778 + VO, 3) and tuple(out['cam'].shape) == (B, 15)
    assert np.abs(out['verts'][:, :778].cpu().numpy() - want_v).max() < ATOL
    assert torch.equal(out['verts'][:, 778:].cpu(), theta['vertices_obj'])
    assert torch.equal(out['cam'].cpu(), torch.cat([theta['cam'].reshape(B, -1), theta['trans'].reshape(B, -1)], dim=1))
    assert out['objName'] is theta['objName']


def test_rejects_malformed_arguments():
    from hoig_amd.mano import ManoModel, mano_vertices
    md = M.synthetic_model(1)
    model = ManoModel.from_dict(md)
    z = lambda *s: torch.zeros(*s, device='cuda')
    with pytest.raises(ValueError):
        mano_vertices(model, z(2, 3), z(2, 44), z(2, 10))
    with pytest.raises(ValueError):
        mano_vertices(model, z(2, 3), z(2, 45), z(2, 10), out=z(2, 700, 3))
    with pytest.raises(NotImplementedError):
        mano_vertices(model, torch.zeros(2, 3), z(2, 45), z(2, 10))
    with pytest.raises(ValueError):
        ManoModel.from_dict(dict(md, parents=np.array([-1, 2, 1] + list(range(2, 15)))))
