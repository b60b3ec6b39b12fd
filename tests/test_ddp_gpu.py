"""The N>1 training step ON THE GPU: two ranks share cuda:0 and exchange gradients over gloo (RCCL refuses two ranks on one
device; the collective is the only thing that differs from the 8-GPU run), through the real Trainer with use_ddp=True --
flat-buffer broadcast, G's exchange + Adam on the side stream beside the D step, weight gradients on their side stream.
Checks: both ranks end with identical weights, the averaged gradient equals the ORACLE's gradient of the combined batch,
and the averaged gradient (read from Adam's first moment after one step,
(1 - beta1) * g) equals that of a single-process step on the combined batch (mean-reduced losses + per-sample instance norm
make data parallelism exact up to summation order; the weights themselves move by ~lr*sign(g) in Adam's first step, so
rounding-level gradient elements may flip and are not compared element-wise)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
SIDE, STEPS = 64, 1
PROBE = ['bg_model.model.12.main.0.weight', 'src_model.resnets.1.conv_0.weight', 'obj_model.decoders.0.0.weight',
         'attn_6.fully_connect_layer.0.weight']


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _slice(inputs, lo, hi):
    """Samples lo..hi-1 of a synthetic batch (bg_mask / hand_mask hold src then tsf, each of the full batch)."""
    full = inputs['real_src'].shape[0]
    out = {}
    for k, v in inputs.items():
        if v.shape[0] == 2 * full:
            out[k] = torch.cat([v[lo:hi], v[full + lo:full + hi]], 0).contiguous()
        else:
            out[k] = v[lo:hi].contiguous()
    return out


def _run(rank, world, port, q, mode=None, steps=STEPS):
    import torch.distributed as dist
    from common import product_trainer, SEEDS
    from hoig_amd import ops, synthetic
    ddp = world > 1
    if ddp:
        os.environ['MASTER_ADDR'] = '127.0.0.1'
        os.environ['MASTER_PORT'] = str(port)
        dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    ops.set_precision('bf16x3')
    batch = synthetic.make_inputs(2, SIDE, seed=SEEDS['inputs'])
    # the seeded weights of the parity tests (the oracle below starts from the same ones)
    model = product_trainer('generator_spade_attn', 2, SIDE, use_ddp=ddp, inputs=_slice(batch, rank, rank + 1) if ddp else batch,
                            **(dict(ddp_mode=mode) if mode else {}))
    model.set_train()
    g = model._G.module if ddp else model._G
    before = {k: g.state_dict()[k].cpu().numpy().copy() for k in PROBE}
    early = []
    for _ in range(steps):
        model.optimize_parameters()
        if ddp:
            early.append(model._G.sync.early_launches)
    torch.cuda.synchronize()
    sd = g.state_dict()
    mom = g.export_dict(model._optimizer_G.exp_avg)
    q.put((rank, {k: sd[k].cpu().numpy().copy() for k in PROBE}, before,
           float(g.flat.double().sum().item()), {k: mom[k].cpu().numpy().copy() for k in PROBE}, early))
    if ddp:
        dist.barrier()
        dist.destroy_process_group()


def _spawn(world, mode=None, steps=STEPS):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_run, args=(r, world, port, q, mode, steps)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


def test_trainer_ddp_world2_matches_single_process():
    (_, w0, b0, s0, m0, _), (_, w1, b1, s1, m1, _) = _spawn(2)
    (_, ws, bs, _, ms, _), = _spawn(1)
    # the ORACLE's gradient of the combined batch of 2 (what DDP's average over the two 1-sample ranks must equal:
    # mean-reduced losses, per-sample instance norm): models/trainer.py:425-434 on the CPU restatement
    from common import oracle_trainer
    ot = oracle_trainer('generator_spade_attn', 2, SIDE)
    ot.optimize_parameters()
    for k in PROBE:
        want = ot.G[k].grad.numpy()
        got = m0[k] / (1.0 - 0.5)                      # Adam's first moment after one step = (1 - beta1) * g, beta1 = 0.5
        rel_o = np.linalg.norm(got - want) / np.linalg.norm(want)
        print('ddp vs ORACLE  %-44s gradient rel-L2 %.2e' % (k, rel_o))
        assert rel_o < 2e-2, (k, rel_o)
        assert np.array_equal(b0[k], b1[k]) and np.array_equal(b0[k], bs[k]), k      # same start everywhere
        assert np.array_equal(w0[k], w1[k]) and np.array_equal(m0[k], m1[k]), k      # ranks stay bit-identical
        assert np.linalg.norm(ms[k]) > 0
        rel = np.linalg.norm(m0[k] - ms[k]) / np.linalg.norm(ms[k])                  # averaged gradient == combined-batch gradient
        print('ddp vs single  %-44s gradient rel-L2 %.2e' % (k, rel))
        assert rel < 2e-2, (k, rel)
        # and the weights moved the same way for the bulk of the elements
        agree = np.mean(np.sign(w0[k] - b0[k]) == np.sign(ws[k] - bs[k]))
        print('                %-44s update sign agreement %.3f' % (k, agree))
        assert agree > 0.9, (k, agree)
    assert s0 == s1


def test_bucket_mode_two_ranks_on_the_gpu_path():
    """opt.ddp_mode = 'bucket' through the real Trainer with two ranks (gloo, one GPU): after the learning step the slices of G's
    gradient are exchanged WHILE the backward is being issued -- from a communication stream ordered behind the chains that wrote each
    slice -- and the two ranks must still end every step with bit-identical weights and Adam moments (they apply the same sums), equal
    to the default mode's to the run-to-run floor of the weight gradients' fp32 atomics."""
    (_, w0, _, s0, m0, e0), (_, w1, _, s1, m1, e1) = _spawn(2, 'bucket', 3)
    (_, wa, _, _, ma, ea), _ = _spawn(2, 'after', 3)
    print('early launches per step: rank 0 %s, rank 1 %s; mode after %s' % (e0, e1, ea))
    assert e0 == e1 and e0[0] == 0 and e0[1] > 0 and e0[2] == e0[1] and ea == [0, 0, 0]
    assert s0 == s1
    for k in PROBE:
        assert np.array_equal(w0[k], w1[k]) and np.array_equal(m0[k], m1[k]), k
        assert np.isfinite(w0[k]).all() and np.linalg.norm(m0[k]) > 0
        # three Adam steps of ~lr each: no weight further than that from the default mode's (rounding-level gradient elements may flip sign)
        assert np.abs(w0[k] - wa[k]).max() <= 3 * 2.2 * 2e-4, k
