"""A synthetic asset tree in the reference renderer's layout (TEST INFRASTRUCTURE): assets/MANO_UV_right.obj (778 vertices, 1538
triangular faces with texture indices), assets/obj/<name>/<name>.obj + texture_map.png for two objects, assets/semantics_hand.pkl.
Seeded; the meshes are random soups with a UV layout of small well-separated triangles (so that the atlas rasterisation has a unique
front face per texel and the checks do not hinge on depth ties)."""
import os
import pickle

import numpy as np

OBJ_NAMES = ['004_sugar_box', '011_banana']            # two directory names; MANORenderer ranks them by sorted()


def _uv_triangles(g, nf, cols):
    """nf small triangles on a grid inside (0,1)^2 -> (3 nf, 2) vt coordinates and (nf,3) indices"""
    rows = (nf + cols - 1) // cols
    vt, idx = [], []
    for f in range(nf):
        cy, cx = divmod(f, cols)
        ox, oy = (cx + 0.1) / cols, (cy + 0.1) / rows
        sx, sy = 0.8 / cols, 0.8 / rows
        tri = np.array([[0.05, 0.05], [0.95, 0.1], [0.3, 0.9]]) + g.uniform(-0.04, 0.04, (3, 2))
        vt.extend((ox + tri[:, 0] * sx, oy + tri[:, 1] * sy) for _ in [0])
        idx.append([3 * f, 3 * f + 1, 3 * f + 2])
    vt = np.concatenate([np.stack(p, axis=1) for p in vt], axis=0)
    return vt, np.asarray(idx)


def write_obj(path, verts, faces, vt, faces_vt):
    with open(path, 'w') as fp:
        for v in verts:
            fp.write('v %.6f %.6f %.6f\n' % tuple(v))
        for t in vt:
            fp.write('vt %.6f %.6f\n' % tuple(t))
        fp.write('vn 0.000000 0.000000 1.000000\n')
        for f, ft in zip(faces, faces_vt):
            fp.write('f %d/%d/1 %d/%d/1 %d/%d/1\n' % (f[0] + 1, ft[0] + 1, f[1] + 1, ft[1] + 1, f[2] + 1, ft[2] + 1))


def build(root, seed=31):
    g = np.random.Generator(np.random.Philox(key=[seed, 5]))
    a = os.path.join(root, 'assets')
    os.makedirs(os.path.join(a, 'obj'), exist_ok=True)
    vt, fvt = _uv_triangles(g, 1538, 40)
    write_obj(os.path.join(a, 'MANO_UV_right.obj'), g.uniform(-0.1, 0.1, (778, 3)), g.integers(0, 778, (1538, 3)), vt, fvt)
    perm = g.permutation(1538)
    parts = ['palm', 'thumb', 'index_finger', 'middle_finger', 'ring_finger', 'little_finger']
    cuts = [0, 500, 700, 900, 1100, 1300, 1500]               # (the last 38 faces carry no label)
    sem = {'right': {k: perm[cuts[i]:cuts[i + 1]].tolist() for i, k in enumerate(parts)}}
    with open(os.path.join(a, 'semantics_hand.pkl'), 'wb') as f:
        pickle.dump(sem, f)
    from PIL import Image
    for j, name in enumerate(OBJ_NAMES):
        d = os.path.join(a, 'obj', name)
        os.makedirs(d, exist_ok=True)
        nv, nf = 90 + 30 * j, 150 + 40 * j
        vt, fvt = _uv_triangles(g, nf, 14)
        write_obj(os.path.join(d, name + '.obj'), g.uniform(-0.08, 0.08, (nv, 3)), g.integers(0, nv, (nf, 3)), vt, fvt)
        Image.fromarray(g.integers(0, 256, (300 + 20 * j, 280, 3), dtype=np.uint8)).save(os.path.join(d, 'texture_map.png'))
    return root
