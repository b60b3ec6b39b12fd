"""The MANO-layer restatement (oracle/mano_oracle.py; parity unpinned: smplx / manopth and the model file are not in the build
container) against the algebraic properties of the published algorithm and its two call forms against each other."""
import numpy as np
from scipy.spatial.transform import Rotation

from oracle import mano_oracle as M


def _params(B, seed=3):
    g = np.random.Generator(np.random.Philox(key=[seed, B]))
    return (g.standard_normal((B, 3)), g.standard_normal((B, 45)) * 0.4, g.standard_normal((B, 10)), g.standard_normal((B, 3)) * 0.3)


def test_rodrigues_forms():
    g = np.random.Generator(np.random.Philox(key=[1, 1]))
    r = np.concatenate([g.standard_normal((50, 3)) * 2.0, np.zeros((1, 3)), [[1e-6, 0, 0]]])
    R = M.batch_rodrigues(r)
    assert np.abs(R - Rotation.from_rotvec(r).as_matrix()).max() < 1e-7           # (the 1e-8 inside the norm)
    assert np.abs(M.batch_rodrigues_manopth(r) - R).max() < 1e-7
    assert np.abs(np.matmul(R, R.transpose(0, 2, 1)) - np.eye(3)).max() < 1e-7


def test_rest_pose_is_the_shaped_template():
    m = M.synthetic_model(0)
    _, _, betas, t = _params(3)
    v, j = M.smplx_mano_forward(m, np.zeros((3, 3)), np.zeros((3, 45)), betas, t)
    v_shaped = m['v_template'][None] + np.einsum('bl,mkl->bmk', betas, m['shapedirs'].astype(np.float64))
    assert np.abs(v - (v_shaped + t[:, None])).max() < 1e-6                        # (R(0) differs from I by ~1e-8)
    assert np.abs(j - (np.einsum('jv,bvk->bjk', m['J_regressor'].astype(np.float64), v_shaped) + t[:, None])).max() < 1e-6


def test_root_rotation_is_rigid_about_the_root_joint():
    m = M.synthetic_model(1)
    root, _, betas, t = _params(4)
    v0, j0 = M.smplx_mano_forward(m, np.zeros((4, 3)), np.zeros((4, 45)), betas, np.zeros((4, 3)))
    v, j = M.smplx_mano_forward(m, root, np.zeros((4, 45)), betas, t)
    R = Rotation.from_rotvec(root).as_matrix()
    want = np.einsum('bij,bvj->bvi', R, v0 - j0[:, :1]) + j0[:, :1] + t[:, None]
    assert np.abs(v - want).max() < 1e-6
    assert np.abs(j - (np.einsum('bij,bvj->bvi', R, j0 - j0[:, :1]) + j0[:, :1] + t[:, None])).max() < 1e-6


def test_a_finger_joint_moves_only_what_hangs_on_it():
    m = M.synthetic_model(2)
    m['posedirs'] = np.zeros_like(m['posedirs'])                                   # (skinning alone)
    w = np.zeros_like(m['lbs_weights'])
    w[np.arange(778), np.arange(778) % 16] = 1.0                                   # every vertex rigidly on one joint
    m['lbs_weights'] = w
    hand = np.zeros((1, 45))
    hand[0, 3 * (5 - 1):3 * 5] = [0.3, -0.5, 0.2]                                  # joint 5 (children: 6)
    v0, _ = M.smplx_mano_forward(m, np.zeros((1, 3)), np.zeros((1, 45)), np.zeros((1, 10)), np.zeros((1, 3)))
    v, _ = M.smplx_mano_forward(m, np.zeros((1, 3)), hand, np.zeros((1, 10)), np.zeros((1, 3)))
    moved = np.abs(v - v0).max(axis=2)[0] > 1e-6
    on = np.isin(np.arange(778) % 16, [5, 6])
    assert moved[on].all() and not moved[~on].any()


def test_manopth_form_equals_smplx_form_on_the_expanded_pose():
    m = M.synthetic_model(3)
    root, coeffs, betas, t = _params(5)
    hand = m['hands_mean'][None].astype(np.float64) + coeffs @ m['hands_components'].astype(np.float64)
    v1, j1 = M.smplx_mano_forward(m, root, hand, betas, t)
    v2, j2 = M.manopth_forward(m, np.concatenate([root, coeffs], axis=1), betas, t)
    assert np.abs(v1 - v2).max() < 1e-7 and np.abs(j1 - j2).max() < 1e-7
