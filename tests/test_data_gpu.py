"""SURVEY 8f row 4 on the device: hoig_warp_affine_u8 / hoig_resize_linear_u8 (hoig_amd/csrc/data_prep.hip) against the CPU oracle
(oracle/data_oracle.py) bit for bit, and the loader end to end -- CustomDatasetDataLoader.load_data() on a synthetic HO3D-v3-shaped tree
against the oracle's restatement of HOv3Dataset._get_sample / __getitem__ (HOIG_HOv3/data/hov3_dataset.py:198-267)."""
import ctypes

import numpy as np
import pytest
import torch

import data_fixture as FX
from oracle import data_oracle as O

pytestmark = pytest.mark.gpu
_p = lambda t: ctypes.c_void_p(t.data_ptr())


def _st():
    return torch.cuda.current_stream().cuda_stream


def _imgs(B, h, w, c, seed):
    return np.random.Generator(np.random.Philox(key=[seed, B * h * w])).integers(0, 256, (B, h, w, c)).astype(np.uint8)


@pytest.mark.parametrize('C', [1, 3, 4])
def test_warp_affine_matches_the_oracle_bit_for_bit(C):
    from hoig_amd import _lib as L
    g = np.random.Generator(np.random.Philox(key=[9, C]))
    B, Hs, Ws, Hd, Wd = 5, 97, 131, 64, 80
    src = _imgs(B, Hs, Ws, C, 1)
    Ms = []
    for b in range(B):                                   # rotation + anisotropic scale + shear + translation; parts fall outside the source
        th, sx, sy = g.uniform(-0.6, 0.6), g.uniform(0.4, 2.5), g.uniform(0.4, 2.5)
        A = np.array([[np.cos(th) * sx, -np.sin(th) * sy + g.uniform(-0.2, 0.2)], [np.sin(th) * sx, np.cos(th) * sy]])
        Ms.append(np.concatenate([A, g.uniform(-30, 30, (2, 1))], axis=1).astype(np.float32))
    Ms[0] = np.array([[1, 0, 0], [0, 1, 0]], np.float32)                          # identity: the 32767 / 1 weights
    Ms[1] = np.array([[0.5, 0, 3.25], [0, 0.5, -7.5]], np.float32)
    m_dev = torch.from_numpy(np.stack(Ms).astype(np.float64).reshape(B, 6)).cuda()
    s_dev = torch.from_numpy(src).cuda()
    out = torch.empty((B, Hd, Wd, C), dtype=torch.uint8, device='cuda')
    L.call('hoig_warp_affine_u8', _p(s_dev), B, Hs, Ws, C, _p(m_dev), Hd, Wd, 0, _p(out), _st())
    want = np.stack([O.warp_affine_linear_u8(src[b], Ms[b], (Wd, Hd)) for b in range(B)])
    assert np.array_equal(out.cpu().numpy(), want)
    assert 0.2 < (want > 0).mean() < 1.0                                            # the cases are not all inside / all border
    mask = torch.empty((B, 1, Hd, Wd), dtype=torch.float32, device='cuda')
    L.call('hoig_warp_affine_u8', _p(s_dev), B, Hs, Ws, C, _p(m_dev), Hd, Wd, 2, _p(mask), _st())
    assert np.array_equal(mask.cpu().numpy()[:, 0], (want.astype(np.float32) / 128.0)[..., -1])
    if C == 3:
        img = torch.empty((B, 3, Hd, Wd), dtype=torch.float32, device='cuda')
        L.call('hoig_warp_affine_u8', _p(s_dev), B, Hs, Ws, C, _p(m_dev), Hd, Wd, 1, _p(img), _st())
        t = (want.astype(np.float32) / 255.0)[..., ::-1].transpose(0, 3, 1, 2)
        assert np.array_equal(img.cpu().numpy(), (t - np.float32(0.5)) / np.float32(0.5))


@pytest.mark.parametrize('Hs,Ws,Hd,Wd', [(240, 320, 480, 640), (480, 640, 480, 640), (37, 53, 111, 71), (100, 120, 33, 47), (5, 7, 64, 3)])
def test_resize_matches_the_oracle_bit_for_bit(Hs, Ws, Hd, Wd):
    from hoig_amd import _lib as L
    B, C = 3, 3
    src = _imgs(B, Hs, Ws, C, 2)
    s_dev = torch.from_numpy(src).cuda()
    out = torch.empty((B, Hd, Wd, C), dtype=torch.uint8, device='cuda')
    L.call('hoig_resize_linear_u8', _p(s_dev), B, Hs, Ws, C, _p(out), Hd, Wd, _st())
    want = np.stack([O.resize_linear_u8(src[b], (Wd, Hd)) for b in range(B)])
    assert np.array_equal(out.cpu().numpy(), want)


def test_loader_batches_match_the_oracle(tmp_path):
    from hoig_amd.data import CustomDatasetDataLoader
    opt = FX.build(str(tmp_path), seed=5)
    pairs = [('ABF1_0/0001.png', 'MC2_0/0003.png'), ('MC2_0/0000.png', 'ABF1_0/0002.png'), ('ABF1_0/0000.png', 'ABF1_0/0003.png')]
    FX.write_pairs(opt, pairs)
    oracle = {}
    # worker processes are FORKED: after the large-batch tests of a session the caching allocator holds tens of GB of mapped device
    # memory, and forking that address space took ~20 s per worker (50 s of this test in the full suite, 3 s alone)
    import gc
    gc.collect()                 # (Trainers of earlier tests sit in reference cycles until the collector runs; their tensors pin the cache)
    torch.cuda.empty_cache()
    for workers in (0, 1):                      # (one forked worker: every fork of this process costs ~20 s late in a session)
        opt.n_threads_train = workers
        loader = CustomDatasetDataLoader(opt, is_for_train=True)
        assert len(loader) == 3
        batches = list(loader.load_data())
        assert [len(b['nameA']) for b in batches] == [2, 1]                          # drop_last=False (data/__init__.py:28)
        got_a = [n for b in batches for n in b['nameA']]
        assert got_a == [p[0] for p in pairs]
        at = 0
        for b in batches:
            n = len(b['nameA'])
            if at not in oracle:
                oracle[at] = FX.oracle_batch(opt, [p[0] for p in pairs[at:at + n]], [p[1] for p in pairs[at:at + n]])
            va, vb = oracle[at]
            at += n
            for side, want in (('A', va), ('B', vb)):
                assert b['image' + side].is_cuda and b['image' + side].dtype == torch.float32
                assert np.array_equal(b['image' + side].cpu().numpy(), want['image'])
                assert np.array_equal(b['mask' + side].cpu().numpy(), want['mask'])
                mano = b['mano' + side]
                for k in ('cam', 'trans', 'pose', 'shape', 'handtrans'):
                    assert np.array_equal(mano[k].cpu().numpy(), want[k]), k
                assert mano['objName'].tolist() == want['objName'].tolist() and not mano['objName'].is_cuda
                v = mano['vertices_obj'].cpu().numpy()
                assert v.shape == (n, 7866, 3) and v.dtype == np.float32
                # float64 products summed in another order on the device: equal after the float32 rounding except for rare ties
                assert np.abs(v - want['vertices_obj']).max() <= 2e-8 and (v != want['vertices_obj']).mean() < 1e-3
                assert np.array_equal(v.any(axis=2), want['vertices_obj'].any(axis=2))


def test_ycb_loader_batches_match_the_oracle(tmp_path):
    from hoig_amd.data import CustomDatasetDataLoader
    opt = FX.build_ycb(str(tmp_path), seed=6)
    v0, v1 = '20200709-subject-01/20200709_141754/836212060125', '20200813-subject-02/20200813_145612/932122062010'
    pairs = [(v0 + '/1', v1 + '/2'), (v1 + '/0', v0 + '/2'), (v0 + '/0', v0 + '/1')]
    FX.write_pairs(opt, pairs)
    loader = CustomDatasetDataLoader(opt, is_for_train=True)
    at = 0
    for b in loader.load_data():
        n = len(b['nameA'])
        assert 'maskA' not in b and 'handtrans' not in b['manoA']
        for side, col in (('A', 0), ('B', 1)):
            want = FX.oracle_batch_ycb(opt, [p[col] for p in pairs[at:at + n]])
            assert b['name' + side] == [p[col] for p in pairs[at:at + n]]
            assert np.array_equal(b['image' + side].cpu().numpy(), want['image'])
            mano = b['mano' + side]
            for k in ('cam', 'trans', 'pose', 'shape'):
                assert np.array_equal(mano[k].cpu().numpy(), want[k]), k
            assert mano['objName'].tolist() == want['objName'].tolist()
            v = mano['vertices_obj'].cpu().numpy()
            assert v.shape == (n, 8000, 3) and np.abs(v - want['vertices_obj']).max() <= 1e-7 and (v != want['vertices_obj']).mean() < 1e-3
            assert np.array_equal(v.any(axis=2), want['vertices_obj'].any(axis=2))
        at += n
    assert at == 3


def test_trainer_takes_the_loaders_batch(tmp_path):
    """train_ddp.py:88-92: for batch in loader: model.set_input(batch); model.optimize_parameters() -- with the synthetic MANO model and
    renderer tables of tests/test_hand_recovery_gpu.py standing in for the unshipped assets."""
    from test_hand_recovery_gpu import _assets
    from common import opt_namespace
    from hoig_amd import ops
    from hoig_amd.data import CustomDatasetDataLoader
    from hoig_amd.mano import ManoModel
    from hoig_amd.models import ModelsFactory
    from oracle import mano_oracle as M
    obj_ids = [2, 5]
    assets, nv = _assets(obj_ids, 21)
    opt_d = FX.build(str(tmp_path), seed=8, n_obj_verts=nv)
    FX.write_pairs(opt_d, [('ABF1_0/0001.png', 'ABF1_0/0003.png'), ('MC2_0/0000.png', 'MC2_0/0002.png')])
    loader = CustomDatasetDataLoader(opt_d, is_for_train=True)
    opt = opt_namespace(gen_name='generator_spade_attn', local_rank=0, image_size=256)
    opt.mano_model = ManoModel.from_dict(M.synthetic_model(4))
    opt.object_assets = assets
    ops.set_precision('bf16x3')
    try:
        torch.manual_seed(3)
        model = ModelsFactory.get_by_name('trainer', opt, use_ddp=False)
        model.set_train()
        for batch in loader.load_data():
            model.set_input(batch)
            model.optimize_parameters()
        assert all(np.isfinite(v) for v in model.get_current_errors().values())
        assert torch.equal(model._real_src, batch['imageA']) and torch.equal(model._armask_tsf, batch['maskB'])
    finally:
        ops.set_precision('f32')


def test_loader_runs_the_raw_stage_one_batch_ahead_with_the_same_result(tmp_path):
    """Round 6: when the loader's opt carries the MANO model and the object assets (train_ddp.py passes one opt to both), the loader
    runs the raw-batch stage itself -- on its stream, behind the batch's pixel work, one batch ahead of the step -- and the batch
    carries the prepared tensors; Trainer.set_input then stages EXACTLY what it stages from the raw entries of the same batch."""
    from test_hand_recovery_gpu import _assets
    from common import opt_namespace
    from hoig_amd import ops
    from hoig_amd.data import CustomDatasetDataLoader
    from hoig_amd.mano import ManoModel
    from hoig_amd.models import ModelsFactory
    from hoig_amd.models.trainer import PREPARED_KEYS, RAW_KEYS
    from oracle import mano_oracle as M
    assets, nv = _assets([2, 5], 21)
    opt_d = FX.build(str(tmp_path), seed=8, n_obj_verts=nv)
    FX.write_pairs(opt_d, [('ABF1_0/0001.png', 'ABF1_0/0003.png'), ('MC2_0/0000.png', 'MC2_0/0002.png'),
                           ('MC2_0/0001.png', 'MC2_0/0003.png'), ('ABF1_0/0000.png', 'ABF1_0/0002.png')])
    opt_d.serial_batches = True
    opt = opt_namespace(gen_name='generator_spade_attn', local_rank=0, image_size=256)
    opt.mano_model = opt_d.mano_model = ManoModel.from_dict(M.synthetic_model(4))
    opt.object_assets = opt_d.object_assets = assets
    opt_d.image_size = 256
    ops.set_precision('bf16x3')
    try:
        torch.manual_seed(3)
        model = ModelsFactory.get_by_name('trainer', opt, use_ddp=False)
        model.set_train()
        staged = {}
        for ahead in (True, False):
            opt_d.loader_prepares = ahead
            got = []
            for batch in CustomDatasetDataLoader(opt_d, is_for_train=True).load_data():
                assert all(k in batch for k in RAW_KEYS)
                assert all(k in batch for k in PREPARED_KEYS) == ahead
                model.set_input(batch)
                got.append({k: v.clone() for k, v in model._n.items()})
                model.optimize_parameters()                      # (the loader's next batch is prepared beside this step)
            torch.cuda.synchronize()
            staged[ahead] = got
        assert len(staged[True]) == len(staged[False]) == 2
        for a, b in zip(staged[True], staged[False]):
            assert a.keys() == b.keys()
            for k in a:
                assert torch.equal(a[k], b[k]), k
        assert all(np.isfinite(v) for v in model.get_current_errors().values())
    finally:
        ops.set_precision('f32')


def test_dexycb_trainer_takes_the_ycb_loaders_batch(tmp_path):
    """The HOIG_DexYCB copy end to end: --dataset_mode ycb (its scripts/train_ycb_ddp.sh:7) selects its loader, its MANO / camera
    conventions and its channel layout; the loader's batch goes through Trainer.set_input and a step."""
    from test_hand_recovery_gpu import _assets
    from common import opt_namespace
    from hoig_amd import ops
    from hoig_amd.data import CustomDatasetDataLoader
    from hoig_amd.mano import ManoModel
    from hoig_amd.models import ModelsFactory
    from oracle import mano_oracle as M
    obj_ids = [10, 6]                                    # 019_pitcher_base, 008_pudding_box: the grasped objects of the fixture's two videos
    assets, nv = _assets(obj_ids, 23)
    opt_d = FX.build_ycb(str(tmp_path), seed=9, n_obj_verts=nv)
    v0, v1 = '20200709-subject-01/20200709_141754/836212060125', '20200813-subject-02/20200813_145612/932122062010'
    FX.write_pairs(opt_d, [(v0 + '/0', v0 + '/2'), (v1 + '/1', v1 + '/0')])
    loader = CustomDatasetDataLoader(opt_d, is_for_train=True)
    opt = opt_namespace(gen_name='generator_spade_attn', local_rank=0, image_size=256, dataset_mode='ycb')
    opt.mano_model = ManoModel.from_dict(M.synthetic_model(4))
    opt.object_assets = assets
    ops.set_precision('bf16x3')
    try:
        torch.manual_seed(5)
        model = ModelsFactory.get_by_name('trainer', opt, use_ddp=False)
        assert model._dexycb
        model.set_train()
        n = 0
        for batch in loader.load_data():
            assert batch['manoA']['cam'].shape == (2, 4) and batch['manoA']['pose'].shape == (2, 51)
            model.set_input(batch)
            model.optimize_parameters()
            n += 1
        assert n == 1 and all(np.isfinite(v) for v in model.get_current_errors().values())
        assert model._input_G_src_hand.shape[1] == 12 and torch.equal(model._real_tsf, batch['imageB'])
    finally:
        ops.set_precision('f32')
