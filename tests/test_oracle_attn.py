"""Pins the oracle's restatement of the reference's CUDA-only ops (K1-K4) with the only checks the reference itself
defines for them (its two manual scripts), plus a cross-check against the independent plain-C restatement."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import hogan_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def clib():
    subprocess.run(['make', '-C', os.path.join(ROOT, 'oracle')], check=True, stdout=subprocess.DEVNULL)
    return ctypes.CDLL(os.path.join(ROOT, 'oracle', '_build', 'libhoig_oracle_c.so'))


def _dp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def test_gradcheck_recipe_of_the_reference():
    """thirdparty/block_extractor/test_block_extractor.py:74-78: double source (4,6,14,10), flow = rand*1.8, k=3."""
    torch.manual_seed(0)
    source = torch.rand(4, 6, 14, 10, dtype=torch.double, requires_grad=True)
    flow = (torch.rand(4, 2, 14, 10, dtype=torch.double) * 1.8).requires_grad_(True)
    assert torch.autograd.gradcheck(lambda s, f: O.block_extract(s, f, 3), (source, flow))
    # thirdparty/local_attn_reshape/test_local_attn_reshape.py:66-70
    x = torch.rand(4, 9, 14, 10, dtype=torch.double, requires_grad=True)
    assert torch.autograd.gradcheck(lambda t: O.local_attn_reshape(t, 3), (x,))


def test_zero_flow_extracts_the_clamped_neighbourhood():
    """test_block_extractor.py:46-49 (zero flow): block (yf,xf) of the output is the k x k neighbourhood of the pixel,
    indices clamped at the border."""
    src = torch.randn(2, 3, 7, 5)
    out = O.block_extract(src, torch.zeros(2, 2, 7, 5), 3)
    for yf in range(7):
        for xf in range(5):
            for dy in range(3):
                for dx in range(3):
                    yy, xx = min(max(yf + dy - 1, 0), 6), min(max(xf + dx - 1, 0), 4)
                    assert torch.equal(out[:, :, 3 * yf + dy, 3 * xf + dx], src[:, :, yy, xx])


def test_local_attn_reshape_known_answer():
    """test_local_attn_reshape.py:29-43: channels 0..8 constant -> out[0,0,:3,:3] = [[0,1,2],[3,4,5],[6,7,8]]."""
    inp = torch.arange(9.0).view(1, -1, 1, 1).repeat(2, 1, 10, 10)
    out = O.local_attn_reshape(inp, 3)
    assert out.shape == (2, 1, 30, 30)
    assert out[0, 0, :3, :3].tolist() == [[0, 1, 2], [3, 4, 5], [6, 7, 8]]
    assert torch.equal(out, torch.nn.functional.pixel_shuffle(inp, 3))


@pytest.mark.parametrize('k', [3, 5])
def test_torch_restatement_equals_c_restatement(clib, k):
    rng = np.random.default_rng(0)
    B, C, H, W = 2, 3, 6, 7
    src = rng.standard_normal((B, C, H, W))
    flow = rng.standard_normal((B, 2, H, W)) * 2.5
    flow[0, :, 0, 0] = -30.0                 # far outside: border clamp with un-renormalised weights
    gout = rng.standard_normal((B, C, k * H, k * W))
    out = np.zeros((B, C, k * H, k * W))
    clib.oracle_block_extractor_forward_f64(_dp(src), _dp(flow), _dp(out), B, C, H, W, H, W, k)
    gs, gf = np.zeros_like(src), np.zeros_like(flow)
    clib.oracle_block_extractor_backward_f64(_dp(src), _dp(flow), _dp(gout), _dp(gs), _dp(gf), B, C, H, W, H, W, k)
    ts = torch.from_numpy(src).requires_grad_(True)
    tf = torch.from_numpy(flow).requires_grad_(True)
    to = O.block_extract(ts, tf, k)
    to.backward(torch.from_numpy(gout))
    np.testing.assert_allclose(to.detach().numpy(), out, atol=1e-12)
    np.testing.assert_allclose(ts.grad.numpy(), gs, atol=1e-12)
    np.testing.assert_allclose(tf.grad.numpy(), gf, atol=1e-10)
    a = rng.standard_normal((B, k * k, H, W))
    r = np.zeros((B, 1, k * H, k * W))
    clib.oracle_local_attn_reshape_forward_f64(_dp(a), _dp(r), B, H, W, k)
    np.testing.assert_array_equal(O.local_attn_reshape(torch.from_numpy(a), k).numpy(), r)
    gr = rng.standard_normal(r.shape)
    ga = np.zeros_like(a)
    clib.oracle_local_attn_reshape_backward_f64(_dp(gr), _dp(ga), B, H, W, k)
    ta = torch.from_numpy(a).requires_grad_(True)
    O.local_attn_reshape(ta, k).backward(torch.from_numpy(gr))
    np.testing.assert_array_equal(ta.grad.numpy(), ga)
