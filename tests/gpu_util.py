"""Helpers shared by the -m gpu parity tests."""
import torch


def nhwc_cuda(t):
    """CPU NCHW tensor -> contiguous NHWC CUDA tensor (plain torch: test plumbing, not the product path)."""
    return t.permute(0, 2, 3, 1).contiguous().cuda()


def nchw_cpu(t):
    """NHWC CUDA tensor -> CPU NCHW tensor."""
    return t.detach().permute(0, 3, 1, 2).contiguous().cpu()


def rel_err(a, b):
    """max |a-b| / max |b|  (the 'relative fp32' error north_star quotes, at tensor scale)."""
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def rel_l2(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
