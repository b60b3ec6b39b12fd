"""CPU: the oracle restatement against the committed golden vectors (made by tests/golden/make_golden.py from the
reference's own Python).  fp32 CPU on both sides -> tight tolerance (thread-count dependent summation only)."""
import numpy as np
import pytest
import torch

from common import oracle_trainer, load_golden, OUT_NAMES


GOLDEN = [('generator_spade_attn', 'hov3_spade_attn_64.npz'), ('generator_spade', 'hov3_spade_64.npz'),
          ('generator_spade_attn_tiny', 'hov3_spade_attn_tiny_64.npz'), ('generator_base', 'hov3_base_64.npz'),
          ('generator_spade_attn', 'dexycb_spade_attn_64.npz')]


@pytest.mark.parametrize('gen_name,fname', GOLDEN)
def test_oracle_matches_reference_golden(gen_name, fname):
    g = load_golden(fname)
    assert str(g['gen_name']) == gen_name
    dataset = str(g['copy']) if 'copy' in g.files else 'hov3'
    ot = oracle_trainer(gen_name, int(g['batch']), int(g['side']), dataset=dataset)
    assert list(ot.G.keys()) == [str(s) for s in g['param_names_G']]
    assert list(ot.D.keys()) == [str(s) for s in g['param_names_D']]
    with torch.no_grad():
        outs = ot.forward()
    for name, v in zip(OUT_NAMES, outs):
        np.testing.assert_allclose(v.numpy(), g['fwd_' + name], rtol=0, atol=2e-5)
    keys = [str(k) for k in g['error_keys']]
    for s in range(int(g['steps'])):
        ot.optimize_parameters()
        e = ot.get_current_errors()
        np.testing.assert_allclose([e[k] for k in keys], g['errors'][s], rtol=2e-4)
        if s == 0:
            for k in g.files:
                if k.startswith('grad_G_'):
                    np.testing.assert_allclose(ot.G[k[7:]].grad.numpy(), g[k], rtol=0, atol=3e-4 * np.abs(g[k]).max())
                if k.startswith('grad_D_'):
                    np.testing.assert_allclose(ot.D[k[7:]].grad.numpy(), g[k], rtol=0, atol=3e-4 * np.abs(g[k]).max())
    l2 = np.array([float(v.detach().double().norm()) for v in ot.G.values()])
    np.testing.assert_allclose(l2, g['post_G_l2'], rtol=1e-4)
    np.testing.assert_allclose(ot.D['model.14.weight'].detach().numpy(), g['post_D_model.14.weight'], atol=5e-4)
