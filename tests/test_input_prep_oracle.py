"""CPU: the input-preparation oracle (oracle/input_prep_oracle.py: HandRecoveryFlow.forward after the rasteriser,
trainer.py:46-145) against (1) the committed fixture tests/golden/input_prep_256.npz, made by the reference's OWN code
(tests/golden/make_golden_input_prep.py), bit for bit, and (2) where /root/reference exists (build container), a live run of
that code on other seeds; plus the edge cases (no face anywhere, hand everywhere)."""
import os
import zlib

import numpy as np
import pytest
import torch

from hoig_amd import synthetic
from oracle import input_prep_oracle as P, ref_harness as RH

NAMES = ['input_G_src_bg', 'input_G_tsf_bg', 'input_G_src_obj', 'input_G_tsf_obj', 'input_G_src_hand', 'input_G_ref_hand',
         'T_hand', 'src_crop_mask_bg', 'ref_crop_mask_bg', 'src_crop_mask_hand', 'ref_crop_mask_hand']
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'input_prep_256.npz')


def run_oracle(r, bg_both, dexycb=False):
    tabs = [r['tables'][k] for k in r['obj_ids']]
    return P.prepare_inputs(r['src_img'], r['ref_img'], r['src_faces'], r['src_fim'], r['src_wim'], r['ref_fim'],
                            r['ref_wim'], tabs, bg_both, dexycb)


@pytest.mark.parametrize('bg_both,copy', [(False, 'hov3'), (True, 'hov3'), (False, 'dexycb'), (True, 'dexycb')])
def test_oracle_matches_reference_fixture_bitwise(bg_both, copy):
    g = np.load(GOLD if copy == 'hov3' else GOLD.replace('.npz', '_dexycb.npz'))
    assert str(g['copy']) == copy
    r = synthetic.make_raster(int(g['batch']), int(g['seed']))
    out = run_oracle(r, bg_both, dexycb=copy == 'dexycb')
    assert out[4].shape[1] == (12 if copy == 'dexycb' else 6)
    assert out[11] is None and (out[1] is None) == (not bg_both)
    for name, v in zip(NAMES, out):
        if v is None:
            continue
        key = 'bg_both%d/%s' % (bg_both, name)
        a = np.ascontiguousarray(v.numpy(), dtype=np.float32)
        sub = a[:, 1::4, 2::4, :] if name == 'T_hand' else a[:, :, 1::4, 2::4]
        np.testing.assert_array_equal(sub, g[key + '/sub'], err_msg=name)
        assert np.uint32(zlib.crc32(a.tobytes())) == g[key + '/crc'], name
        np.testing.assert_allclose(a.astype(np.float64).sum(), float(g[key + '/sum']), rtol=1e-12)


@pytest.mark.skipif(not RH.available(), reason='the reference tree only exists in the build container')
@pytest.mark.parametrize('seed,batch,bg_both', [(11, 3, False), (12, 1, True)])
def test_oracle_matches_reference_live(seed, batch, bg_both):
    r = synthetic.make_raster(batch, seed)
    ref = RH.reference_input_prep(r, bg_both=bg_both)
    out = run_oracle(r, bg_both)
    for name, a, b in zip(NAMES, ref, out):
        assert (a is None) == (b is None), name
        if a is not None:
            assert torch.equal(a, b), name


def test_fixture_scene_exercises_both_outcomes():
    """The synthetic scene must hit the branches that matter: visible and hidden atlas texels, hand / object / background
    pixels, the -2 sentinel and valid flow, eroded borders."""
    r = synthetic.make_raster(2, 8)
    tb = r['tables'][r['obj_ids'][0]]
    f2 = r['src_faces'][0, :tb['n_faces'], :, 0:2].clone()
    f2[:, :, 1] *= -1
    tex = P.texture_backward_warp(r['src_img'][0:1], f2, r['src_fim'][0], tb)
    painted = (tex[0, :, :, :384] == 1.0).all(0).float().mean().item()
    assert 0.02 < painted < 0.9                      # some charts are hidden in the source view, some are visible
    out = run_oracle(r, False)
    T = out[6]
    assert ((T == -2).all(-1)).float().mean() > 0.5 and ((T != -2).any(-1)).float().mean() > 0.01
    assert 0 < out[9].mean() < 1 and 0 < out[7].mean() < 1
    assert out[2][:, 6:].sum() > 0 and out[4][:, 3].abs().sum() > 0


def test_edge_cases_empty_and_full():
    r = synthetic.make_raster(1, 8)
    tabs = [r['tables'][k] for k in r['obj_ids']]
    empty = -torch.ones_like(r['src_fim'])
    zero_w = torch.zeros_like(r['src_wim'])
    out = P.prepare_inputs(r['src_img'], r['ref_img'], r['src_faces'], empty, zero_w, empty, zero_w, tabs, True)
    assert (out[6] == -2).all()                                      # no face: the flow is the sentinel everywhere
    assert (out[7] == 1).all() and (out[9] == 1).all()               # everything is background / not hand
    assert torch.equal(out[0][:, :3], r['src_img']) and (out[0][:, 3] == 1).all()
    assert (out[2][:, :3] == 0).all() and (out[4][:, :3] == 0).all()
    full = torch.zeros_like(r['src_fim'])                            # hand face 0 everywhere
    w = torch.full_like(r['src_wim'], 1.0 / 3)
    out = P.prepare_inputs(r['src_img'], r['ref_img'], r['src_faces'], full, w, full, w, tabs, False)
    assert (out[9] == 0).all() and (out[7] == 0).all()               # hand everywhere: both crop masks are empty
    assert torch.equal(out[4][:, :3], r['src_img'])                  # hand input = the whole source image
    assert (out[6] != -2).all()
