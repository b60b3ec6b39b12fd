"""tests/quality_metrics.py itself (the offline stand-ins for FID / LPIPS: SURVEY 8c(iii)): the Frechet distance
against closed forms, the LPIPS-shaped distance's basic properties, and the oracle-against-itself floor on a tiny case."""
import numpy as np
import torch

import quality_metrics as Q
from common import seeded_state


def test_frechet_distance_closed_forms():
    g = np.random.default_rng(3)
    d = 6
    a = g.standard_normal((d, d))
    s1 = a @ a.T + np.eye(d)
    mu1, mu2 = g.standard_normal(d), g.standard_normal(d)
    assert abs(Q.frechet_distance(mu1, s1, mu1, s1)) < 1e-9                                  # identical Gaussians
    assert abs(Q.frechet_distance(mu1, s1, mu2, s1) - float(((mu1 - mu2) ** 2).sum())) < 1e-8     # equal covariances: the mean term
    c = 2.5                                                                                  # commuting covariances: Tr(S1 + c S1 - 2 sqrt(c) S1)
    want = (1 + c - 2 * np.sqrt(c)) * np.trace(s1)
    assert abs(Q.frechet_distance(mu1, s1, mu1, c * s1) - want) < 1e-8 * want + 1e-9
    assert abs(Q.frechet_distance(mu1, s1, mu2, c * s1) - Q.frechet_distance(mu2, c * s1, mu1, s1)) < 1e-8      # symmetric


def test_random_features_are_fixed_and_separate_image_sets():
    g = torch.Generator().manual_seed(1)
    x = torch.rand(80, 3, 32, 32, generator=g) * 2 - 1
    f1, f2 = Q.random_features(x, dims=32), Q.random_features(x.clone(), dims=32)
    assert f1.shape == (80, 32) and np.array_equal(f1, f2)                                   # seeded extractor: a function of the images only
    assert abs(Q.frechet_between(x, x, dims=32)) < 1e-9
    y = (x * 0.5).clamp(-1, 1)
    assert Q.frechet_between(x, y, dims=32) > 1e-3 * np.trace(Q.activation_statistics(f1)[1])


def test_lpips_like_is_zero_on_equal_pairs_and_grows_with_the_perturbation():
    sd = seeded_state('generator_spade_attn')[3]
    g = torch.Generator().manual_seed(2)
    x = torch.rand(4, 3, 32, 32, generator=g) * 2 - 1
    n = torch.randn(4, 3, 32, 32, generator=g)
    assert float(np.abs(Q.lpips_like(x, x, sd)).max()) == 0.0
    d1, d2 = Q.lpips_like(x, x + 0.01 * n, sd), Q.lpips_like(x, x + 0.1 * n, sd)
    assert (d1 > 0).all() and (d2 > 5 * d1).all()
    assert np.allclose(Q.lpips_like(x, x + 0.1 * n, sd), Q.lpips_like(x + 0.1 * n, x, sd))   # symmetric
