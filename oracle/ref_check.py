"""TEST INFRASTRUCTURE (build container only): runs the REFERENCE's own Trainer (imported from /root/reference through
oracle/ref_harness.py) and the oracle restatement (oracle/hogan_oracle.py) side by side on identical seeded weights and
synthetic inputs, and prints one JSON line of differences.  One reference copy per process, hence a script that
tests/test_oracle_vs_reference.py runs once per (gen_name, copy):

    python -m oracle.ref_check <gen_name> <hov3|dexycb> [side] [batch] [steps]

Compared: the six outputs of Trainer.forward (trainer.py:373-415), the seven loss terms of every step
(trainer.py:483-492), every parameter of G and D after the steps (i.e. both Adam updates, trainer.py:425-434), and the G / D
gradients left by the last step."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from oracle import ref_harness as RH, hogan_oracle as O      # noqa: E402
from hoig_amd import synthetic                               # noqa: E402

SEEDS = dict(G=8, D=9, VGG=10, inputs=8)


def main(gen_name, copy, side=64, batch=1, steps=2):
    torch.set_num_threads(int(os.environ.get('HOIG_REF_CHECK_THREADS', '4')))
    cfg = O.make_cfg(gen_name, copy)
    sdG = O.make_weights(O.gen_param_shapes(cfg), seed=SEEDS['G'], mode='random')
    sdD = O.make_weights(O.disc_param_shapes(cfg), seed=SEEDS['D'], mode='random')
    sdV = O.make_weights(O.vgg_param_shapes(), seed=SEEDS['VGG'], kind='vgg')
    inp = synthetic.make_inputs(batch, side, seed=SEEDS['inputs'], dataset=copy)

    t = RH.build_reference_trainer(RH.namespace(gen_name=gen_name), copy=copy)
    assert list(t._G.state_dict().keys()) == list(sdG.keys()), 'generator schema differs from the reference'
    assert list(t._D.state_dict().keys()) == list(sdD.keys()), 'discriminator schema differs from the reference'
    t._G.load_state_dict(sdG)
    t._D.load_state_dict(sdD)
    t._crt_tsf.vgg.load_state_dict(sdV)
    for k, v in inp.items():
        setattr(t, '_' + k, v.clone())
    ot = O.OracleTrainer(cfg, sdG, sdD, sdV)
    ot.set_prepared_input(inp)

    out = dict(gen_name=gen_name, copy=copy, side=side, batch=batch, steps=steps)
    with torch.no_grad():
        r, o = t.forward(), ot.forward()
    out['fwd_max_abs'] = max(float((a - b).abs().max()) for a, b in zip(r, o))
    out['fwd_bit_exact'] = all(torch.equal(a, b) for a, b in zip(r, o))
    loss_rel = 0.0
    for _ in range(steps):
        t.optimize_parameters()
        ot.optimize_parameters()
        er, eo = t.get_current_errors(), ot.get_current_errors()
        assert list(er.keys()) == list(eo.keys())
        loss_rel = max(loss_rel, max(abs(er[k] - eo[k]) / max(abs(er[k]), 1e-6) for k in er))
    out['loss_max_rel'] = loss_rel
    wdiff, gdiff, exact = 0.0, 0.0, True
    for net_r, net_o in ((t._G, ot.G), (t._D, ot.D)):
        for (name, p), (name_o, q) in zip(net_r.named_parameters(), net_o.items()):
            assert name == name_o
            wdiff = max(wdiff, float((p.detach() - q.detach()).abs().max()))
            exact = exact and torch.equal(p.detach(), q.detach())
            if p.grad is not None and q.grad is not None:
                gdiff = max(gdiff, float((p.grad - q.grad).abs().max() / p.grad.abs().max().clamp_min(1e-30)))
    out['post_weight_max_abs'] = wdiff
    out['post_weight_bit_exact'] = exact
    out['grad_max_rel'] = gdiff
    print(json.dumps(out))


if __name__ == '__main__':
    a = sys.argv[1:]
    main(a[0], a[1], *[int(x) for x in a[2:]])
