"""GPU end-to-end parity: hoig_amd Trainer (HIP kernels through the C ABI) against the CPU oracle and the committed
reference golden vectors, same seeded inputs and weights.  Bound written here = north_star's: 1e-3 relative (fp32)."""
import numpy as np
import pytest
import torch

from common import oracle_trainer, product_trainer, load_golden, OUT_NAMES
from gpu_util import rel_err, rel_l2

pytestmark = pytest.mark.gpu
TOL = 1e-3


@pytest.fixture(autouse=True)
def _restore_precision():
    from hoig_amd import ops
    yield
    ops.set_precision('f32')


@pytest.mark.parametrize('precision', ['f32', 'bf16x3'])
@pytest.mark.parametrize('gen_name,fname', [('generator_spade_attn', 'hov3_spade_attn_64.npz'),
                                            ('generator_spade', 'hov3_spade_64.npz')])
def test_trainer_matches_reference_golden(gen_name, fname, precision):
    """Both shipped arithmetic modes (exact-fp32 MFMA and split-bf16 MFMA) must meet the same 1e-3 bound."""
    from hoig_amd import ops
    ops.set_precision(precision)
    g = load_golden(fname)
    m = product_trainer(gen_name, int(g['batch']), int(g['side']))
    assert list(m._G.state_dict().keys()) == [str(s) for s in g['param_names_G']]
    with torch.no_grad():
        outs = m.forward()
    for name, v in zip(OUT_NAMES, outs):
        assert tuple(v.shape) == g['fwd_' + name].shape
        assert rel_err(v, torch.from_numpy(g['fwd_' + name])) < TOL, name
    keys = [str(k) for k in g['error_keys']]
    lr = 2e-4
    for s in range(int(g['steps'])):
        m.optimize_parameters()
        e = m.get_current_errors()
        got, want = np.array([e[k] for k in keys]), g['errors'][s]
        assert np.all(np.abs(got - want) <= 2e-3 * np.maximum(np.abs(want), 1e-2)), (s, got, want)
        if s == 0:
            for k in g.files:
                if k.startswith('grad_G_'):
                    assert rel_l2(m._G.P[k[7:]].grad, torch.from_numpy(g[k])) < 5e-3, k
                if k.startswith('grad_D_'):
                    assert rel_l2(m._D.P[k[7:]].grad, torch.from_numpy(g[k])) < 5e-3, k
    sd = m._D.state_dict()
    # Adam's first steps are ~ lr*sign(g): an element whose gradient is at rounding-noise level may flip; bound by 2 steps
    assert (sd['model.14.weight'].cpu() - torch.from_numpy(g['post_D_model.14.weight'])).abs().max() <= 2.2 * 2 * lr
    # per-tensor L2 norms of all 425 generator tensors after the two steps.  A parameter whose true gradient is
    # identically zero (a conv bias feeding an instance norm) random-walks by +-lr per step under Adam on BOTH sides,
    # so the bound is Adam's worst-case drift, |dw_i| <= lr per step; large weight tensors must also agree to 1e-3.
    sdg = m._G.state_dict()
    steps = int(g['steps'])
    for (name, v), want in zip(sdg.items(), g['post_G_l2']):
        got = float(v.double().norm())
        assert abs(got - want) <= 2.2 * steps * lr * np.sqrt(v.numel()), name
        if v.dim() == 4 and v.numel() >= 65536:
            assert abs(got - want) <= 1e-3 * want, name


def test_trainer_vs_oracle_dexycb_channels():
    """DexYCB channel configuration (bg 13, hand cond 9, D 24, no arm mask): forward + one step against the oracle."""
    ot = oracle_trainer('generator_spade_attn', 1, 64, dataset='dexycb')
    m = product_trainer('generator_spade_attn', 1, 64, dataset='dexycb')
    with torch.no_grad():
        ro, po = ot.forward(), m.forward()
    for a, b in zip(po, ro):
        assert rel_err(a, b) < TOL
    ot.optimize_parameters()
    m.optimize_parameters()
    eo, ep = ot.get_current_errors(), m.get_current_errors()
    for k in eo:
        assert abs(eo[k] - ep[k]) <= 2e-3 * max(abs(eo[k]), 1e-2), (k, eo[k], ep[k])


def test_api_surface_and_checkpoint_roundtrip(tmp_path):
    m = product_trainer('generator_spade_attn', 2, 64, checkpoints_dir=str(tmp_path))
    m.optimize_parameters(keep_data_for_visuals=True)
    vis = m.get_current_visuals()
    assert len(vis) == 18
    assert vis['15_batch_fake_img'].dtype == np.uint8 and vis['15_batch_fake_img'].shape == (3, 64, 128)
    assert vis['12_fake_mask_bg'].shape == (1, 64, 64)
    assert set(m.get_current_scalars()) == {'lr_G', 'lr_D'}
    m.save(3)
    e1 = None
    m2 = product_trainer('generator_spade_attn', 2, 64, checkpoints_dir=str(tmp_path), load_epoch=3)
    for k, v in m._G.state_dict().items():
        assert torch.equal(v, m2._G.state_dict()[k]), k
    assert m2._optimizer_G.step_count == 1
    m.optimize_parameters()
    m2.optimize_parameters()
    for k, v in m._G.state_dict().items():
        assert torch.equal(v, m2._G.state_dict()[k]), k       # resumed run continues bit-identically
    m.update_learning_rate()
    assert abs(m.get_current_scalars()['lr_G'] - (2e-4 - (2e-4 - 2e-6) / 15)) < 1e-12
    m.set_eval()
    m.optimize_parameters()                                     # no-op outside training (trainer.py:418)
    assert hasattr(m, 'backward_G') and hasattr(m, 'backward_D')
    with pytest.raises(NotImplementedError):
        m.set_input({'imageA': torch.zeros(1)})
