"""SURVEY 8c(iii) / VERDICT r5 row (g) on the GPU: the HIP path's images against the oracle's through the two stand-in metrics of
tests/quality_metrics.py (random-feature Frechet distance: fid_score.py:146-200's formula; LPIPS-shaped distance: lpips.py:41-56's form).
Default run: the stage that has an exact answer -- same weights, no training: the two image sets must coincide (and the eval forward's
cheaper arithmetic stays inside north_star's 1e-3).  Opt-in (-m "gpu or gpu_slow"): the protocol of tools/quality_surrogate.py with
training steps on both sides and the oracle-against-itself noise floor (profiles/r06_quality_surrogate.txt holds the N = 256 /
K = 5, 20, 50 run)."""
import os
import sys

import numpy as np
import pytest
import torch

from common import seeded_state

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _precision():
    yield
    from hoig_amd import ops
    ops.set_precision('f32')             # (quality_surrogate.hip_run selects the bench arithmetic process-wide)


def test_images_of_the_hip_path_and_of_the_oracle_coincide_at_the_same_weights():
    import quality_metrics as Q
    import quality_surrogate as S
    n, side, dims = 48, 64, 24
    hip = S.hip_run(n, side, 4, [0])                 # 'init' only (steps [0]: no optimiser step)
    cores = len(os.sched_getaffinity(0))
    ora = S.oracle_run(n, side, 4, [0], min(cores, 16))
    sd = seeded_state(S.GEN)[3]
    r = S.compare(hip['init'], ora['init'], sd, dims)
    trace = float(np.trace(Q.activation_statistics(Q.random_features(ora['init'], dims=dims))[1]))
    print('init: Frechet %.3e (feature covariance trace %.3e), LPIPS-like mean %.3e, max-norm %.3e' % (r['frechet'], trace, r['lpips_mean'], r['max_rel']))
    assert r['max_rel'] < 1e-3                       # north_star's output bound (eval mode: the f16f6 forward; measured 1e-4)
    assert abs(r['frechet']) < 1e-8 * trace and r['lpips_max'] < 1e-9
    # and the FID-shaped number against the real targets is the same number on both sides
    fa, fb = Q.frechet_between(hip['init'], ora['real'], dims=dims), Q.frechet_between(ora['init'], ora['real'], dims=dims)
    assert abs(fa - fb) < 1e-6 * abs(fb)


@pytest.mark.gpu_slow
def test_after_training_steps_the_hip_path_stays_within_the_oracles_own_noise():
    """K optimiser steps on each side from the same state; the noise floor is the oracle against itself on another thread count.  GAN
    training from random weights amplifies rounding-level differences to O(1) within a handful of steps (Adam's first updates are
    +-lr whatever the gradient's size), so the statement that CAN be asserted is one of ORDER: the HIP path is not further from the
    oracle than a small multiple of what separates two runs of the oracle."""
    import quality_surrogate as S
    n, side, dims, steps = 64, 64, 32, [5]
    cores = len(os.sched_getaffinity(0))
    hip = S.hip_run(n, side, 4, steps)
    ora = S.oracle_run(n, side, 4, steps, min(cores, 16))
    orb = S.oracle_run(n, side, 4, steps, max(1, min(cores, 16) // 2 - 1))
    sd = seeded_state(S.GEN)[3]
    rows = S.report(hip, ora, orb, sd, print, dims)
    r = rows['k5']
    assert r['hip_vs_oracle']['frechet'] < 10 * max(r['oracle_vs_oracle']['frechet'], 1e-4 * rows['feature_trace'])
    assert r['hip_vs_oracle']['lpips_mean'] < 10 * max(r['oracle_vs_oracle']['lpips_mean'], 1e-6)
    lo = min(r['to_real']['oracle'], r['to_real']['oracle_other_threads'])
    hi = max(r['to_real']['oracle'], r['to_real']['oracle_other_threads'])
    assert 0.5 * lo < r['to_real']['hip'] < 2.0 * hi
