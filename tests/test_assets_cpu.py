"""hoig_amd.assets, host half: the renderer's per-object tables built from OBJ files (utils/nmr.py:283-391) against a fixture made by
the REFERENCE's own mesh.py / look_at.py / vertices_to_faces.py on the same synthetic asset tree (tests/golden/make_golden_assets.py):
bit for bit."""
import os
import pickle

import numpy as np
import pytest

import assets_fixture as AF
from common import load_golden


def _tree(tmp_path):
    root = AF.build(str(tmp_path))
    a = os.path.join(root, 'assets')
    names = sorted(os.listdir(os.path.join(a, 'obj')))
    objects = {j: (os.path.join(a, 'obj', n, n + '.obj'), os.path.join(a, 'obj', n, 'texture_map.png')) for j, n in enumerate(names)}
    with open(os.path.join(a, 'semantics_hand.pkl'), 'rb') as f:
        sem = pickle.load(f)
    return os.path.join(a, 'MANO_UV_right.obj'), objects, sem


def test_host_tables_match_the_references_own_functions(tmp_path):
    from hoig_amd import assets
    hand, objects, sem = _tree(tmp_path)
    got = assets.host_tables(hand, objects, sem)
    want = load_golden('assets_tables.npz')
    assert sorted(got) == [0, 1]
    for j in got:
        for key in ('faces', 'map_fn', 'sem_full', 'faces_uv_coord', 'raster_hand', 'raster_obj'):
            w = want['%d/%s' % (j, key)]
            g = got[j][key]
            assert g.shape == w.shape and g.dtype == w.dtype, (j, key, g.shape, w.shape, g.dtype, w.dtype)
            assert np.array_equal(g, w), (j, key, np.abs(g.astype(np.float64) - w).max())


def test_obj_reader_refuses_what_the_two_reference_readers_read_differently(tmp_path):
    from hoig_amd import assets
    p = tmp_path / 'quad.obj'
    p.write_text('v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\nf 1/1/1 2/2/1 3/3/1 4/4/1\n')
    with pytest.raises(ValueError, match='triangle'):
        assets.load_obj(str(p))
    q = tmp_path / 'tri.obj'
    q.write_text('# comment\nv 0 0 0\nv 1 0 0\nv 1 1 0\nvt 0 0\nvt 1 0\nvt 1 1\nf 1/1 2/2 3/3\n')
    ob = assets.load_obj(str(q))
    assert ob['faces'].tolist() == [[0, 1, 2]] and ob['faces_vts'].tolist() == [[0, 1, 2]]
    m = assets.uv_seg_mapping(ob)
    assert m.shape == (2, 3) and m[1].tolist() == [0.0, 0.0, 1.0]
