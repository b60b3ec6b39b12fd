"""TEST INFRASTRUCTURE -- CPU restatement (numpy, float64) of the MANO hand layer that turns pose / shape parameters into the 778
hand vertices in front of the HOGAN input preparation.  Only tests/ may import this file.

The reference does not contain the algorithm; it calls two third-party packages:
  * HOIG_HOv3/models/networks/hmr.py:55,84-85 -- ``smplx.create(path, 'mano', use_pca=False, is_rhand=True, flat_hand_mean=True)``
    and ``layer(global_orient, hand_pose, betas, transl).vertices``; requirements.txt:153 pins smplx==0.1.28.  Restated here:
    ``smplx.body_models.MANO.forward`` and ``smplx.lbs.{lbs, batch_rodrigues, batch_rigid_transform, blend_shapes,
    vertices2joints}`` of that release (the published SMPL/MANO linear blend skinning).
  * HOIG_DexYCB/models/networks/hmr.py:55-60,85-86 -- ``manopth.manolayer.ManoLayer(flat_hand_mean=False, ncomps=45,
    side='right', use_pca=True)``, ``layer(pose[:, :48], shape, pose[:, 48:51])`` and ``vertices /= 1000`` (manopth is installed
    from its git repository, no pinned version; restated from ``ManoLayer.forward``: PCA coefficients -> axis-angle pose, the same
    skinning, translation, millimetre scale).

PARITY UNPINNED: neither package nor a MANO model file (MANO_RIGHT.pkl, licensed, not shipped: .gitignore:3) exists in the build
container and the reference holds no golden vectors for this stage, so this restatement is checked only against the algebraic
properties of the published algorithm (tests/test_mano_oracle.py) and the two variants against each other.
"""
import numpy as np

NUM_JOINTS = 16
# kintree_table[0] of the MANO model files (smplx stores it as `parents`; manopth hard-codes the same tree as three levels per
# finger, manolayer.py `lev1_idxs = [1, 4, 7, 10, 13]` ...)
MANO_PARENTS = np.array([-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13, 14])


def batch_rodrigues(rot_vecs, epsilon=1e-8):
    """smplx.lbs.batch_rodrigues: (N,3) axis-angle -> (N,3,3).  The angle is |r + 1e-8| (sic: epsilon is added to every
    component before the norm), the axis r / angle."""
    rot_vecs = np.asarray(rot_vecs, np.float64)
    angle = np.linalg.norm(rot_vecs + epsilon, axis=1, keepdims=True)
    rot_dir = rot_vecs / angle
    cos, sin = np.cos(angle)[:, None, :], np.sin(angle)[:, None, :]
    rx, ry, rz = rot_dir[:, 0], rot_dir[:, 1], rot_dir[:, 2]
    zeros = np.zeros_like(rx)
    K = np.stack([zeros, -rz, ry, rz, zeros, -rx, -ry, rx, zeros], axis=1).reshape(-1, 3, 3)
    ident = np.eye(3)[None]
    return ident + sin * K + (1 - cos) * np.matmul(K, K)


def batch_rodrigues_manopth(axisang):
    """manopth.rodrigues_layer.batch_rodrigues: through a unit quaternion (|r + 1e-8| / 2 half angle).  Same rotation."""
    axisang = np.asarray(axisang, np.float64)
    angle = np.linalg.norm(axisang + 1e-8, axis=1, keepdims=True)
    normalized = axisang / angle
    half = angle * 0.5
    quat = np.concatenate([np.cos(half), np.sin(half) * normalized], axis=1)
    quat = quat / np.linalg.norm(quat, axis=1, keepdims=True)
    w, x, y, z = quat[:, 0], quat[:, 1], quat[:, 2], quat[:, 3]
    w2, x2, y2, z2 = w * w, x * x, y * y, z * z
    wx, wy, wz, xy, xz, yz = w * x, w * y, w * z, x * y, x * z, y * z
    return np.stack([w2 + x2 - y2 - z2, 2 * xy - 2 * wz, 2 * wy + 2 * xz,
                     2 * wz + 2 * xy, w2 - x2 + y2 - z2, 2 * yz - 2 * wx,
                     2 * xz - 2 * wy, 2 * wx + 2 * yz, w2 - x2 - y2 + z2], axis=1).reshape(-1, 3, 3)


def batch_rigid_transform(rot_mats, joints, parents):
    """smplx.lbs.batch_rigid_transform: (B,J,3,3), (B,J,3) -> posed joints (B,J,3), relative transforms (B,J,4,4)."""
    B, J = joints.shape[:2]
    rel = joints.copy()
    rel[:, 1:] -= joints[:, parents[1:]]
    tm = np.zeros((B, J, 4, 4))
    tm[:, :, :3, :3] = rot_mats
    tm[:, :, :3, 3] = rel
    tm[:, :, 3, 3] = 1.0
    chain = [tm[:, 0]]
    for i in range(1, J):
        chain.append(np.matmul(chain[parents[i]], tm[:, i]))
    transforms = np.stack(chain, axis=1)
    posed = transforms[:, :, :3, 3].copy()
    jh = np.concatenate([joints, np.zeros((B, J, 1))], axis=2)[..., None]
    rel_t = transforms.copy()
    rel_t[:, :, :, 3] -= np.matmul(transforms, jh)[..., 0]
    return posed, rel_t


def lbs(betas, pose, model, rodrigues=batch_rodrigues):
    """smplx.lbs.lbs with pose2rot=True.  model: dict of v_template (V,3), shapedirs (V,3,10), posedirs (135, V*3),
    J_regressor (16,V), parents (16,), lbs_weights (V,16).  -> vertices (B,V,3), posed joints (B,16,3)."""
    betas, pose = np.asarray(betas, np.float64), np.asarray(pose, np.float64)
    B = betas.shape[0]
    vt, sd = np.asarray(model['v_template'], np.float64), np.asarray(model['shapedirs'], np.float64)
    v_shaped = vt[None] + np.einsum('bl,mkl->bmk', betas, sd)
    Jn = np.einsum('bik,ji->bjk', v_shaped, np.asarray(model['J_regressor'], np.float64))
    rot = rodrigues(pose.reshape(-1, 3)).reshape(B, -1, 3, 3)
    pose_feature = (rot[:, 1:] - np.eye(3)).reshape(B, -1)
    v_posed = v_shaped + np.matmul(pose_feature, np.asarray(model['posedirs'], np.float64)).reshape(B, -1, 3)
    posed_j, A = batch_rigid_transform(rot, Jn, np.asarray(model['parents']))
    T = np.matmul(np.asarray(model['lbs_weights'], np.float64)[None], A.reshape(B, NUM_JOINTS, 16)).reshape(B, -1, 4, 4)
    vh = np.concatenate([v_posed, np.ones((B, v_posed.shape[1], 1))], axis=2)
    verts = np.matmul(T, vh[..., None])[:, :, :3, 0]
    return verts, posed_j


def smplx_mano_forward(model, global_orient, hand_pose, betas, transl):
    """smplx.body_models.MANO.forward with use_pca=False, flat_hand_mean=True (hmr.py:55,84): full_pose = [global_orient,
    hand_pose] (+ pose_mean, all zeros for a flat hand mean), lbs, then vertices += transl."""
    full = np.concatenate([np.asarray(global_orient, np.float64), np.asarray(hand_pose, np.float64)], axis=1)
    verts, joints = lbs(betas, full, model)
    t = np.asarray(transl, np.float64)[:, None]
    return verts + t, joints + t


def manopth_forward(model, pose_coeffs, betas, trans):
    """manopth ManoLayer.forward(th_pose_coeffs (B, 3 + ncomps), th_betas, th_trans) with use_pca=True, flat_hand_mean=False,
    ncomps=45, followed by the reference's `/ 1000` (HOIG_DexYCB hmr.py:85-86): hand pose = hands_mean + coeffs @ components,
    the same skinning (levels of the same tree), + trans, x 1000 (mm), / 1000."""
    pose_coeffs = np.asarray(pose_coeffs, np.float64)
    hand = np.asarray(model['hands_mean'], np.float64)[None] + pose_coeffs[:, 3:] @ np.asarray(model['hands_components'], np.float64)
    full = np.concatenate([pose_coeffs[:, :3], hand], axis=1)
    verts, joints = lbs(betas, full, model, rodrigues=batch_rodrigues_manopth)
    t = np.asarray(trans, np.float64)[:, None]
    return (verts + t) * 1000 / 1000, (joints + t) * 1000 / 1000


def synthetic_model(seed=0, V=778):
    """A random model with MANO's dimensions and structure (blend weights are convex combinations, the joint regressor's rows
    sum to one) for tests: the real MANO_RIGHT.pkl is licensed and not shipped."""
    g = np.random.Generator(np.random.Philox(key=[seed, 778]))
    w = g.random((V, NUM_JOINTS)) ** 8
    w /= w.sum(1, keepdims=True)
    jr = g.random((NUM_JOINTS, V)) ** 16
    jr /= jr.sum(1, keepdims=True)
    comps = np.linalg.qr(g.standard_normal((45, 45)))[0]
    return {'v_template': (g.standard_normal((V, 3)) * 0.04).astype(np.float32),
            'shapedirs': (g.standard_normal((V, 3, 10)) * 0.004).astype(np.float32),
            'posedirs': (g.standard_normal((135, V * 3)) * 0.002).astype(np.float32),
            'J_regressor': jr.astype(np.float32), 'parents': MANO_PARENTS.copy(), 'lbs_weights': w.astype(np.float32),
            'hands_mean': (g.standard_normal(45) * 0.2).astype(np.float32), 'hands_components': comps.astype(np.float32)}
