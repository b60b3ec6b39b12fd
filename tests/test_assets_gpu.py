"""hoig_amd.assets, device half and end to end: the texture atlas's face-index / weight maps (the product's rasteriser over the UV
layouts: against oracle/raster.c on the REFERENCE-made face tensors of tests/golden/assets_tables.npz), the object texture image
(against oracle/data_oracle.py's cv2.resize restatement) -- both oracles parity-unpinned -- and the tables as
`opt.object_assets`: ObjectTables accepts them and Trainer.set_input(raw batch) runs on them."""
import os
import pickle

import numpy as np
import pytest
import torch

import assets_fixture as AF
from common import load_golden, oracle_rasterize

pytestmark = pytest.mark.gpu


def _tree(tmp_path):
    root = AF.build(str(tmp_path))
    a = os.path.join(root, 'assets')
    names = sorted(os.listdir(os.path.join(a, 'obj')))
    objects = {j: (os.path.join(a, 'obj', n, n + '.obj'), os.path.join(a, 'obj', n, 'texture_map.png')) for j, n in enumerate(names)}
    with open(os.path.join(a, 'semantics_hand.pkl'), 'rb') as f:
        sem = pickle.load(f)
    return root, os.path.join(a, 'MANO_UV_right.obj'), objects, sem


def test_atlas_maps_and_texture_against_the_oracles(tmp_path):
    from hoig_amd import assets, input_prep as IP
    from hoig_amd.data.hov3_dataset import imread_bgr
    from oracle import data_oracle as DO
    root, hand, objects, sem = _tree(tmp_path)
    got = assets.build_object_assets(hand, objects, sem)
    same = assets.object_assets_from_tree(root)                       # the reference's directory layout, ids = sorted positions
    want = load_golden('assets_tables.npz')
    hf, hw = oracle_rasterize(torch.from_numpy(want['0/raster_hand']), 256)
    for j in (0, 1):
        t = got[j]
        of, ow = oracle_rasterize(torch.from_numpy(want['%d/raster_obj' % j]), 256)
        fim = torch.cat([hf, torch.full((1, 256, 128), -1, dtype=torch.int32), of + (of != -1).int() * 1538], dim=2)
        wim = torch.cat([hw, torch.zeros(1, 256, 128, 3), ow], dim=2)
        assert torch.equal(t['fim_uv'].cpu(), fim) and torch.equal(t['wim_uv'].cpu(), wim)
        assert (t['fim_uv'][:, :, :256] >= 0).float().mean() > 0.2 and (t['fim_uv'][:, :, 384:] >= 1538).float().mean() > 0.2
        tex = DO.resize_linear_u8(np.ascontiguousarray(imread_bgr(objects[j][1])[:, :, ::-1]), (256, 256)).astype(np.float32) / 255.0 * 2.0 - 1
        assert np.array_equal(t['obj_tex_img'].cpu().numpy(), tex)
        for k in t:
            assert torch.equal(torch.as_tensor(t[k]).cpu(), torch.as_tensor(same[j][k]).cpu()), k
        tb = IP.ObjectTables(t, torch.device('cuda'))                 # (checks every shape and the index range of fim_uv)
        assert tb.n_faces == t['faces'].shape[0] == 1538 + (150 + 40 * j)


def test_trainer_takes_a_raw_batch_on_tables_built_from_the_asset_files(tmp_path):
    """The last "caller must supply" of SURVEY 8f row 3: opt.object_assets made by hoig_amd.assets from OBJ / PNG / pickle files."""
    from test_hand_recovery_gpu import _raw_batch
    from common import opt_namespace
    from hoig_amd import assets, ops
    from hoig_amd.mano import ManoModel
    from hoig_amd.models import ModelsFactory
    from oracle import mano_oracle as M
    root, hand, objects, sem = _tree(tmp_path)
    opt = opt_namespace(gen_name='generator_spade_attn', local_rank=0, image_size=256)
    opt.mano_model = ManoModel.from_dict(M.synthetic_model(4))
    opt.object_assets = assets.object_assets_from_tree(root)
    nv = {0: 90, 1: 120}                                              # the fixture's object vertex counts
    ops.set_precision('bf16x3')
    try:
        torch.manual_seed(3)
        model = ModelsFactory.get_by_name('trainer', opt, use_ddp=False)
        model.set_train()
        model.set_input(_raw_batch(2, 5, [1, 0], nv))
        model.optimize_parameters()
        assert all(np.isfinite(v) for v in model.get_current_errors().values())
        assert model._input_G_src_obj.shape == (2, 15, 256, 256)
        # both objects' own label planes are populated: faces of object rank j carry label j + 7 (nmr.py:317) = channel 6 + j
        seg = model._input_G_src_obj[:, 6:]
        assert float(seg[0, 1].sum()) > 0 and float(seg[1, 0].sum()) > 0
    finally:
        ops.set_precision('f32')
