"""Offline stand-ins for the reference's image-quality metrics (TEST INFRASTRUCTURE).

north_star asks for "FID/LPIPS within noise" of the CPU reference.  FID and LPIPS need InceptionV3 / AlexNet ImageNet weights and the
LPIPS linear heads (metrics/pytorch_fid/inception.py, metrics/lpips.py:47-57: `lpips_weights.ckpt`), none of which exist offline.
What CAN be measured here is whether the HIP path's images differ from the oracle's by more than the oracle's images differ from
THEMSELVES when nothing but its summation order changes -- with metrics of the same FORM, computed by the same code on both sides:

  * `frechet_distance`: the Frechet distance between two Gaussians fitted to feature vectors -- the formula of
    metrics/pytorch_fid/fid_score.py:146-200 (||mu1 - mu2||^2 + Tr(S1 + S2 - 2 sqrt(S1 S2)), scipy sqrtm, the same eps fallback
    and imaginary-part handling), over `random_features` instead of InceptionV3's pool3: a fixed, seeded, four-layer
    convolutional extractor (He-scaled Gaussian weights, stride 2, ReLU, global average pooling of the last two levels).
  * `lpips_like`: metrics/lpips.py:41-56 -- multi-layer features, unit-normalised along the channels (`normalize`, :8-9), squared
    difference, a non-negative 1x1 weighting, spatial mean, summed over layers -- over the repo's seeded VGG19 surrogate
    (oracle.vgg_features: relu1_1 .. relu5_1) with uniform 1/C weights instead of AlexNet and the trained linear heads.

Neither number is comparable with a published FID / LPIPS; each is meaningful only beside its noise floor, which the callers report.
"""
import numpy as np
import torch
import torch.nn.functional as F
from scipy import linalg

from oracle import hogan_oracle as O

FEATURE_DIMS = 96


def _feature_weights(seed, dims):
    g = torch.Generator().manual_seed(seed)
    chans = [3, 24, 48, dims // 2, dims // 2]
    ws = []
    for ci, co in zip(chans[:-1], chans[1:]):
        ws.append(torch.randn(co, ci, 3, 3, generator=g, dtype=torch.float64) * (2.0 / (9 * ci)) ** 0.5)
    return ws


def random_features(images, seed=1234, dims=FEATURE_DIMS, batch=64):
    """images (N,3,H,W) in [-1,1] -> (N, dims) float64 features of the fixed random extractor."""
    ws = _feature_weights(seed, dims)
    out = []
    for i in range(0, images.shape[0], batch):
        x = images[i:i + batch].double()
        pooled = []
        for k, w in enumerate(ws):
            x = F.relu(F.conv2d(x, w, None, stride=2, padding=1))
            if k >= len(ws) - 2:
                pooled.append(x.mean(dim=(2, 3)))
        out.append(torch.cat(pooled, dim=1))
    return torch.cat(out).numpy()


def activation_statistics(feats):
    """fid_score.py:203-222: mean and covariance (rowvar=False) of the feature vectors."""
    return np.mean(feats, axis=0), np.cov(feats, rowvar=False)


def frechet_distance(mu1, sigma1, mu2, sigma2, eps=1e-6):
    """fid_score.py:146-200, restated."""
    mu1, mu2 = np.atleast_1d(mu1), np.atleast_1d(mu2)
    sigma1, sigma2 = np.atleast_2d(sigma1), np.atleast_2d(sigma2)
    assert mu1.shape == mu2.shape and sigma1.shape == sigma2.shape
    diff = mu1 - mu2
    covmean, _ = linalg.sqrtm(sigma1.dot(sigma2), disp=False)
    if not np.isfinite(covmean).all():
        offset = np.eye(sigma1.shape[0]) * eps
        covmean = linalg.sqrtm((sigma1 + offset).dot(sigma2 + offset))
    if np.iscomplexobj(covmean):
        if not np.allclose(np.diagonal(covmean).imag, 0, atol=1e-3):
            raise ValueError('Imaginary component %g' % np.max(np.abs(covmean.imag)))
        covmean = covmean.real
    return float(diff.dot(diff) + np.trace(sigma1) + np.trace(sigma2) - 2 * np.trace(covmean))


def frechet_between(images_a, images_b, seed=1234, dims=FEATURE_DIMS):
    """(dims must stay below the number of images: a covariance of rank < dims makes sqrtm's answer an artefact of its eps fallback)"""
    fa, fb = random_features(images_a, seed, dims), random_features(images_b, seed, dims)
    return frechet_distance(*activation_statistics(fa), *activation_statistics(fb))


def lpips_like(x, y, sd_vgg, batch=32):
    """(N,) per-pair distances, lpips.py:41-56 in form (see the module docstring)."""
    out = []
    with torch.no_grad():
        for i in range(0, x.shape[0], batch):
            fx, fy = O.vgg_features(sd_vgg, x[i:i + batch]), O.vgg_features(sd_vgg, y[i:i + batch])
            d = 0
            for a, b in zip(fx, fy):
                a = a * torch.rsqrt(torch.sum(a ** 2, dim=1, keepdim=True) + 1e-10)
                b = b * torch.rsqrt(torch.sum(b ** 2, dim=1, keepdim=True) + 1e-10)
                d = d + ((a - b) ** 2).mean(dim=1).mean(dim=(1, 2))          # uniform 1/C "linear head"
            out.append(d.double())
    return torch.cat(out).numpy()
