"""The training step as a captured hipGraph (Trainer._graph_step) against the same step issued kernel by
kernel and against the CPU oracle.  The graph must (a) compute what the eager step computes -- every loss term, the weights
after several optimiser steps, Adam's step count and bias corrections (they live in device memory and advance inside the
graph), (b) follow new inputs and a new learning rate without re-capture, (c) leave D untouched when `trainable=False`."""
import numpy as np
import pytest
import torch

from common import oracle_trainer, product_trainer, SEEDS

pytestmark = pytest.mark.gpu
LOSS_TOL = 1e-3            # north_star's bound on outputs; losses of later steps are held to it too (tests/test_trainer_gpu.py)
LR = 2e-4


@pytest.fixture(autouse=True)
def _precision():
    from hoig_amd import ops
    old = ops.set_f6_min_tiles(1)
    ops.set_precision('bf16x3:f16x2')       # the benchmarked arithmetic
    yield
    ops.set_precision('f32')
    ops.set_f6_min_tiles(old)


def _close(a, b, tol=LOSS_TOL):
    return abs(a - b) <= tol * max(abs(b), 1e-2)


def _captured(m):
    return [k for k, g in m._graphs.items() if g['graphs'] is not None]


def _copy_state(src, dst):
    """Make `dst` continue from exactly `src`'s weights and Adam state.  (This seeded GAN is chaotic after its first update --
    g_adv jumps to 61 on step 1 and two runs of the SAME eager code differ by 1 % in g_adv on step 2, tools/step_repro.py --
    so two trainers are only comparable step by step from a common state.)"""
    torch.cuda.synchronize()
    with torch.no_grad():
        for a, b in ((src._G, dst._G), (src._D, dst._D)):
            b.flat.copy_(a.flat)
            b.version += 1
        for a, b in ((src._optimizer_G, dst._optimizer_G), (src._optimizer_D, dst._optimizer_D)):
            b.exp_avg.copy_(a.exp_avg)
            b.exp_avg_sq.copy_(a.exp_avg_sq)
            b.step_count = a.step_count
    torch.cuda.synchronize()


@pytest.mark.parametrize('precision', ['bf16x3:f16x2', pytest.param('f16f6', marks=pytest.mark.gpu_slow)])
def test_captured_step_matches_eager_step_and_oracle(precision):
    from hoig_amd import ops
    from hoig_amd.models import trainer as T
    ops.set_precision(precision)
    mg = product_trainer('generator_spade_attn', 2, 64, hip_graph=True)
    me = product_trainer('generator_spade_attn', 2, 64, hip_graph=False)
    ot = oracle_trainer('generator_spade_attn', 2, 64)
    for s in range(T._GRAPH_WARMUP):                       # the eager iterations before the capture: against the oracle
        mg.optimize_parameters()
        ot.optimize_parameters()
        eg, eo = mg.get_current_errors(), ot.get_current_errors()
        assert not _captured(mg)
        later = 2e-3 if precision == 'f16f6' else LOSS_TOL          # (tests/test_trainer_gpu.py: LOSS_TOL_LATER_F6)
        for k in eo:
            assert _close(eg[k], eo[k], later if s else 1e-4), (s, k, eg[k], eo[k])
    for s in range(T._GRAPH_WARMUP):
        me.optimize_parameters()
    assert not me._graphs
    for rnd in range(3):                                   # captured on the first round, replayed on the others
        _copy_state(me, mg)
        mg.optimize_parameters()
        me.optimize_parameters()
        assert len(_captured(mg)) == 1
        eg, ee = mg.get_current_errors(), me.get_current_errors()
        for k in ee:
            # the same kernels on the same data from the same state: run-to-run differences only (fp32 atomics; in f16f6 a
            # last-bit difference can move an fp6 rounding, tests/test_configs_gpu.py)
            assert np.isfinite(eg[k]) and _close(eg[k], ee[k]), (rnd, k, eg[k], ee[k])
        steps = T._GRAPH_WARMUP + rnd + 1
        torch.cuda.synchronize()
        for opt_g, opt_e in ((mg._optimizer_G, me._optimizer_G), (mg._optimizer_D, me._optimizer_D)):
            assert opt_g.step_count == opt_e.step_count == steps
            assert float(opt_g._state[4]) == steps == float(opt_e._state[4])        # the device's own count
            assert torch.equal(opt_g._derived, opt_e._derived)                      # same bias corrections
        # weights after the step: an element whose gradient is at rounding-noise level moves by +-lr on either side
        # (tests/test_trainer_gpu.py), everything else must agree closely
        for net_g, net_e in ((mg._G, me._G), (mg._D, me._D)):
            assert float((net_g.flat - net_e.flat).abs().max()) <= 2.2 * LR
            assert float((net_g.flat - net_e.flat).norm() / net_e.flat.norm()) < 1e-4
        for opt_g, opt_e in ((mg._optimizer_G, me._optimizer_G), (mg._optimizer_D, me._optimizer_D)):
            assert float((opt_g.exp_avg - opt_e.exp_avg).norm() / opt_e.exp_avg.norm()) < 2e-2
    # Adam's moments in the reference's checkpoint layout (base_model.py:78-90)
    st = mg._optimizer_G.state_dict()
    assert int(st['state'][0]['step']) == T._GRAPH_WARMUP + 3


def test_captured_step_follows_new_inputs_and_learning_rate():
    from hoig_amd import synthetic
    from hoig_amd.models import trainer as T
    mg = product_trainer('generator_spade_attn', 2, 64, hip_graph=True)
    me = product_trainer('generator_spade_attn', 2, 64, hip_graph=False)
    for _ in range(T._GRAPH_WARMUP + 1):
        mg.optimize_parameters()
        me.optimize_parameters()
    assert len(_captured(mg)) == 1
    first = mg.get_current_errors()
    other = synthetic.make_inputs(2, 64, seed=SEEDS['inputs'] + 5)
    _copy_state(me, mg)
    for m in (mg, me):
        m.set_input(other)                  # copied into the staged buffers the graph reads
        m.update_learning_rate()            # trainer.py:574-591: the new rate reaches the device before the next replay
        m.optimize_parameters()
    assert len(_captured(mg)) == 1          # no re-capture
    eg, ee = mg.get_current_errors(), me.get_current_errors()
    assert abs(eg['g_rec'] - first['g_rec']) > 1e-2 * abs(first['g_rec'])          # another batch -> other losses
    for k in ee:
        assert _close(eg[k], ee[k]), (k, eg[k], ee[k])
    want_lr = LR - (LR - 2e-6) / 15
    torch.cuda.synchronize()
    assert abs(float(mg._optimizer_G._state[0]) - want_lr) < 1e-12 and abs(float(mg._optimizer_D._state[0]) - want_lr) < 1e-12
    assert float((mg._G.flat - me._G.flat).abs().max()) <= 2.2 * LR
    # a batch of another shape falls back to eager warm-up steps and gets its own graph; the first one is kept
    mg.set_input(synthetic.make_inputs(1, 64, seed=SEEDS['inputs']))
    for _ in range(T._GRAPH_WARMUP + 1):
        mg.optimize_parameters()
    assert len(_captured(mg)) == 2
    assert all(np.isfinite(v) for v in mg.get_current_errors().values())
    mg.set_input(other)
    mg.optimize_parameters()
    assert len(_captured(mg)) == 2


def test_captured_step_with_frozen_discriminator():
    """optimize_parameters(trainable=False) (trainer.py:417-428: the G half only) is a graph of its own; D's weights, moments and
    step count do not move."""
    from hoig_amd.models import trainer as T
    m = product_trainer('generator_spade_attn', 2, 64, hip_graph=True)
    d0 = m._D.flat.clone()
    g0 = m._G.flat.clone()
    for _ in range(T._GRAPH_WARMUP + 2):
        m.optimize_parameters(trainable=False)
    assert len(_captured(m)) == 1
    torch.cuda.synchronize()
    assert torch.equal(m._D.flat, d0) and m._optimizer_D.step_count == 0
    assert not torch.equal(m._G.flat, g0) and m._optimizer_G.step_count == T._GRAPH_WARMUP + 2
    m.optimize_parameters()                 # the full step after it: eager warm-up of another graph key
    assert m._optimizer_D.step_count == 1


def test_readers_wait_for_a_delayed_optimiser_side_stream():
    """The optimiser steps run on a side stream; EVERY stream that reads a network afterwards (the main stream, the D stream,
    the loss streams, the generator's branch streams) must be ordered behind them.  Delay the side stream by ~50 ms per step:
    a reader that is not ordered would use stale or half-updated weights / planes, or zero gradients that are still being
    read, and the losses would leave the undelayed run's."""
    from hoig_amd.models import trainer as T
    runs = []
    for delay in (0, 100_000_000):
        T._TEST_SIDE_DELAY = delay
        try:
            m = product_trainer('generator_spade_attn', 2, 64, hip_graph=False)
            hist = []
            for _ in range(2):             # (later steps of this seeded GAN are chaotic run to run: see _copy_state)
                m.optimize_parameters()
                hist.append(m.get_current_errors())
            torch.cuda.synchronize()
            runs.append((hist, m._G.flat.clone(), m._D.flat.clone()))
        finally:
            T._TEST_SIDE_DELAY = 0
    (h0, g0, d0), (h1, g1, d1) = runs
    for a, b in zip(h0, h1):
        for k in a:
            assert _close(a[k], b[k]), (k, a[k], b[k])
    assert float((g0 - g1).abs().max()) <= 2.2 * 2 * LR and float((d0 - d1).abs().max()) <= 2.2 * 2 * LR
