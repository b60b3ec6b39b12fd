"""GPU: hoig_amd.input_prep (HIP, through the C ABI) against the oracle restatement of HandRecoveryFlow.forward
(trainer.py:46-145) and against the reference-made fixture.  Every discrete output -- the four crop masks, the eroded
background mask, one-hot segments, the -2 flow sentinel, which atlas texels are painted -- must match EXACTLY; the
interpolated values (two bilinear samplers, 3-tap barycentric sums) within 2e-6 absolute on data in [-2, 10]."""
import os

import numpy as np
import pytest
import torch

from hoig_amd import synthetic
from oracle import input_prep_oracle as P

pytestmark = pytest.mark.gpu
NAMES = ['input_G_src_bg', 'input_G_tsf_bg', 'input_G_src_obj', 'input_G_tsf_obj', 'input_G_src_hand', 'input_G_ref_hand',
         'T_hand', 'src_crop_mask_bg', 'ref_crop_mask_bg', 'src_crop_mask_hand', 'ref_crop_mask_hand']
ATOL = 2e-6


def run_hip(r, bg_both, dexycb=False):
    from hoig_amd import input_prep as IP
    dev = torch.device('cuda', 0)
    tabs = {k: IP.ObjectTables(tb, dev) for k, tb in r['tables'].items()}
    c = lambda k: r[k].to(dev)
    return IP.prepare_inputs(c('src_img'), c('ref_img'), c('src_faces'), c('src_fim'), c('src_wim'), c('ref_fim'),
                             c('ref_wim'), [tabs[k] for k in r['obj_ids']], bg_both, dexycb)


def run_oracle(r, bg_both, dexycb=False):
    tabs = [r['tables'][k] for k in r['obj_ids']]
    return P.prepare_inputs(r['src_img'], r['ref_img'], r['src_faces'], r['src_fim'], r['src_wim'], r['ref_fim'],
                            r['ref_wim'], tabs, bg_both, dexycb)


def compare(hip, ora):
    for name, a, b in zip(NAMES, hip, ora):
        assert (a is None) == (b is None), name
        if a is None:
            continue
        a = a.cpu()
        assert a.shape == b.shape, name
        if 'mask' in name:
            assert torch.equal(a, b), name
        else:
            bad = ((a - b).abs() > ATOL)
            assert not bad.any(), '%s: %d of %d values differ, max %g' % (name, int(bad.sum()), a.numel(),
                                                                        float((a - b).abs().max()))
    # discrete channels inside the float tensors: exact
    assert torch.equal(hip[0][:, 3].cpu(), ora[0][:, 3])                          # 15x15-eroded background mask
    for i in (2, 3):
        assert torch.equal(hip[i][:, 5:].cpu(), ora[i][:, 5:])                    # flag + one-hot object segments
    assert torch.equal((hip[6] == -2).cpu(), ora[6] == -2)                        # the flow sentinel


@pytest.mark.parametrize('seed,batch,bg_both', [(8, 2, False), (8, 2, True), (11, 3, False), (13, 8, False)])
def test_hip_matches_oracle(seed, batch, bg_both):
    r = synthetic.make_raster(batch, seed)
    compare(run_hip(r, bg_both), run_oracle(r, bg_both))


def test_hip_matches_oracle_dexycb_layout():
    r = synthetic.make_raster(2, 14)
    hip, ora = run_hip(r, True, dexycb=True), run_oracle(r, True, dexycb=True)
    assert hip[4].shape[1] == 12 and hip[5].shape[1] == 12
    compare(hip, ora)
    for i in (4, 5):
        assert torch.equal(hip[i][:, 5:].cpu(), ora[i][:, 5:])                    # flag + the six hand-part one-hots


def test_hip_matches_reference_fixture():
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'input_prep_256.npz'))
    r = synthetic.make_raster(int(g['batch']), int(g['seed']))
    for bg_both in (False, True):
        out = run_hip(r, bg_both)
        for name, v in zip(NAMES, out):
            if v is None:
                continue
            a = v.cpu().numpy()
            sub = a[:, 1::4, 2::4, :] if name == 'T_hand' else a[:, :, 1::4, 2::4]
            want = g['bg_both%d/%s/sub' % (bg_both, name)]
            if 'mask' in name:
                np.testing.assert_array_equal(sub, want, err_msg=name)
            else:
                np.testing.assert_allclose(sub, want, rtol=0, atol=ATOL, err_msg=name)
            np.testing.assert_allclose(a.astype(np.float64).sum(), float(g['bg_both%d/%s/sum' % (bg_both, name)]),
                                       rtol=1e-6, atol=1e-2)


def test_batched_launches_equal_the_per_sample_launches(monkeypatch):
    """Round 6: prepare_inputs issues hoig_prep_texture_batched / hoig_prep_lookup_batched (five launches per batch) where it used to
    loop over the samples (three launches each): the same arithmetic on the same tables, so every output is bit-identical; the
    per-sample entry points stay the C ABI of one sample and the path of batches beyond HOIG_PREP_MAX_BATCH."""
    from hoig_amd import input_prep as IP, synthetic
    r = synthetic.make_raster(3, 12)
    dev = torch.device('cuda', 0)
    args = [r[k].to(dev) for k in ('src_img', 'ref_img', 'src_faces', 'src_fim', 'src_wim', 'ref_fim', 'ref_wim')]
    tabs = [IP.ObjectTables(r['tables'][k], dev) for k in r['obj_ids']]
    assert len({id(t) for t in tabs}) > 1 or len(tabs) == 1
    a = IP.prepare_inputs(*args, tabs, bg_both=True)
    monkeypatch.setattr(IP, 'MAX_BATCH', 0)
    b = IP.prepare_inputs(*args, tabs, bg_both=True)
    torch.cuda.synchronize()
    for x, y in zip(a, b):
        assert (x is None) == (y is None)
        if x is not None:
            assert torch.equal(x, y)


def test_edge_cases_empty_and_full():
    r = synthetic.make_raster(1, 8)
    r['src_fim'] = -torch.ones_like(r['src_fim'])
    r['ref_fim'] = -torch.ones_like(r['ref_fim'])
    r['src_wim'] = torch.zeros_like(r['src_wim'])
    r['ref_wim'] = torch.zeros_like(r['ref_wim'])
    out = run_hip(r, True)
    compare(out, run_oracle(r, True))
    assert (out[6] == -2).all() and (out[7] == 1).all() and (out[9] == 1).all()
    r['src_fim'] = torch.zeros_like(r['src_fim'])
    r['ref_fim'] = torch.zeros_like(r['ref_fim'])
    r['src_wim'] = torch.full_like(r['src_wim'], 1.0 / 3)
    r['ref_wim'] = torch.full_like(r['ref_wim'], 1.0 / 3)
    out = run_hip(r, False)
    compare(out, run_oracle(r, False))
    assert (out[9] == 0).all() and (out[6] != -2).all()


def test_rejects_cpu_tensors_and_wrong_size():
    from hoig_amd import input_prep as IP
    r = synthetic.make_raster(1, 8)
    tabs = [IP.ObjectTables(r['tables'][r['obj_ids'][0]], torch.device('cuda', 0))]
    with pytest.raises(NotImplementedError):
        IP.prepare_inputs(r['src_img'], r['ref_img'], r['src_faces'], r['src_fim'], r['src_wim'], r['ref_fim'], r['ref_wim'],
                          tabs)
    bad = r['src_fim'].clone()
    bad[0, 0, 0] = 10 ** 6
    with pytest.raises(IndexError):
        IP.prepare_inputs(r['src_img'].cuda(), r['ref_img'].cuda(), r['src_faces'].cuda(), bad.cuda(), r['src_wim'].cuda(),
                          r['ref_fim'].cuda(), r['ref_wim'].cuda(), tabs, validate=True)
    # 'deferred' (the training loop's mode): no host wait; the bad index is clamped for the kernels and reported afterwards
    IP.prepare_inputs(r['src_img'].cuda(), r['ref_img'].cuda(), r['src_faces'].cuda(), bad.cuda(), r['src_wim'].cuda(),
                      r['ref_fim'].cuda(), r['ref_wim'].cuda(), tabs, validate='deferred')
    with pytest.raises(IndexError):
        IP.flush_range_checks()
    IP.flush_range_checks()                                  # reported once
    with pytest.raises(ValueError):
        IP.prepare_inputs(r['src_img'][:, :, :128, :128].cuda(), r['ref_img'].cuda(), r['src_faces'].cuda(),
                          r['src_fim'].cuda(), r['src_wim'].cuda(), r['ref_fim'].cuda(), r['ref_wim'].cuda(), tabs)


def test_trainer_accepts_rasterised_batches():
    """Trainer.set_input on a raw batch (images + rasteriser outputs + tables): stages what the oracle's tuple implies
    (trainer.py:346-362) and one optimisation step runs on it."""
    from hoig_amd import ops
    from hoig_amd.models import ModelsFactory
    from common import opt_namespace
    ops.set_precision('bf16x3')
    r = synthetic.make_raster(1, 9)
    opt = opt_namespace(gen_name='generator_spade_attn', local_rank=0, image_size=256)
    torch.manual_seed(3)
    model = ModelsFactory.get_by_name('trainer', opt, use_ddp=False)
    model.set_train()
    arm = torch.zeros(1, 1, 256, 256)
    batch = dict(src_img=r['src_img'], ref_img=r['ref_img'], src_faces=r['src_faces'], src_fim=r['src_fim'],
                 src_wim=r['src_wim'], ref_fim=r['ref_fim'], ref_wim=r['ref_wim'],
                 tables=[r['tables'][k] for k in r['obj_ids']], maskA=arm, maskB=arm)
    model.set_input(batch)
    want = P.to_prepared(run_oracle(r, False), r['src_img'], r['ref_img'], arm, arm)
    assert torch.equal(model._bg_mask.cpu(), want['bg_mask']) and torch.equal(model._hand_mask.cpu(), want['hand_mask'])
    assert (model._input_G_src_obj.cpu() - want['input_G_src_obj']).abs().max() <= ATOL
    assert (model._input_G_tsf_hand.cpu() - want['input_G_tsf_hand']).abs().max() <= ATOL
    assert (model._T.cpu() - want['T']).abs().max() <= ATOL
    model.optimize_parameters()
    errs = model.get_current_errors()
    assert all(np.isfinite(v) for v in errs.values()), errs


def test_trainer_reports_a_bad_face_index_of_the_last_batch():
    """a batch staged with the deferred range check is reported by the next host-side read (get_current_errors / save /
    get_current_visuals), not only when another batch follows; an eval-mode batch is checked before it runs."""
    from hoig_amd import ops
    from hoig_amd.models import ModelsFactory
    from common import opt_namespace
    ops.set_precision('bf16x3')
    r = synthetic.make_raster(1, 9)
    opt = opt_namespace(gen_name='generator_spade_attn', local_rank=0, image_size=256)
    torch.manual_seed(3)
    model = ModelsFactory.get_by_name('trainer', opt, use_ddp=False)
    arm = torch.zeros(1, 1, 256, 256)
    bad = r['ref_fim'].clone()
    bad[0, 5, 5] = 10 ** 6
    batch = dict(src_img=r['src_img'], ref_img=r['ref_img'], src_faces=r['src_faces'], src_fim=r['src_fim'],
                 src_wim=r['src_wim'], ref_fim=bad, ref_wim=r['ref_wim'],
                 tables=[r['tables'][k] for k in r['obj_ids']], maskA=arm, maskB=arm)
    model.set_train()
    model.set_input(batch)                       # no host wait, indices clamped for the kernels
    model.optimize_parameters()
    with pytest.raises(IndexError):
        model.get_current_errors()
    model.get_current_errors()                   # reported once
    model.set_eval()
    with pytest.raises(IndexError):
        model.set_input(batch)
    model.set_train()
