"""hoig_conv2d_fwd_wino (hoig_amd/csrc/conv_wino.hip): the 3x3 stride-1 forward through Winograd F(2x2,3x3) on three fp16 terms, against
torch's fp32 convolution (the bound the direct three-term kernel's tests use) and against the direct kernel."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

from gpu_util import rel_err

pytestmark = pytest.mark.gpu
_p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())


def wino_planes(w):
    from hoig_amd import _lib as L
    co, ci = w.shape[0], w.shape[1]
    n = L.lib.hoig_wino_plane_halfs(co, ci)
    assert n == 16 * co * ci
    uh = torch.empty(n, dtype=torch.int16, device='cuda')
    ul = torch.empty(n, dtype=torch.int16, device='cuda')
    L.call('hoig_pack_conv_weight_wino', _p(w), co, ci, _p(uh), _p(ul), torch.cuda.current_stream().cuda_stream)
    return uh, ul


def wino_conv(x, w, bias=None, act=0, slope=0.0):
    from hoig_amd import _lib as L
    B, H, W, Ci = x.shape
    Co = w.shape[0]
    uh, ul = wino_planes(w)
    y = torch.empty(B, H, W, Co, device='cuda')
    d = L.ConvDesc(B, H, W, Ci, H, W, Co, 3, 3, 1, 1, 0, act, slope, L.PREC_BF16X3)
    L.call('hoig_conv2d_fwd_wino', ctypes.byref(d), _p(x), _p(uh), _p(ul), _p(bias), _p(y), torch.cuda.current_stream().cuda_stream)
    return y


CASES = [
    # B, Ci, Co, H, W
    (1, 32, 64, 16, 16),        # one workgroup, one channel block
    (2, 64, 64, 16, 32),        # two tile columns, two channel blocks
    (3, 96, 128, 32, 16),       # two tile rows, two channel tiles, three channel blocks
    (8, 512, 512, 32, 32),      # the step's dominant launch (8 images)
    (2, 128, 256, 64, 64),      # a larger map
]


@pytest.mark.parametrize('B,Ci,Co,H,W', CASES)
def test_wino_forward_against_torch_fp32_and_the_direct_kernel(B, Ci, Co, H, W):
    from hoig_amd import ops, _lib as L
    g = torch.Generator().manual_seed(31 + Ci)
    x = (torch.randn(B, H, W, Ci, generator=g) * 1.5 + 0.3).cuda()
    w = ops.pack_weight((torch.randn(Co, Ci, 3, 3, generator=g) * 0.05).cuda())
    bias = torch.randn(Co, generator=g).cuda()
    y = wino_conv(x, w, bias)
    ya = wino_conv(x, w, bias, act=L.ACT_LRELU, slope=0.2)
    y0 = wino_conv(x, w, None)
    torch.cuda.synchronize()
    yr = F.conv2d(x.permute(0, 3, 1, 2), w, bias, padding=1).permute(0, 2, 3, 1)
    assert rel_err(y, yr) < 3e-4, rel_err(y, yr)
    assert rel_err(ya, F.leaky_relu(yr, 0.2)) < 3e-4
    assert rel_err(y0, yr - bias) < 3e-4
    # the direct three-term kernel on the same operands: the same products to fp32 rounding
    ops.set_precision('bf16x3')
    try:
        with torch.no_grad():
            yd = ops.conv2d(x, w, bias, 1, 1)
    finally:
        ops.set_precision('f32')
    print('wino vs torch %.2e, direct vs torch %.2e, wino vs direct %.2e' % (rel_err(y, yr), rel_err(yd, yr), rel_err(y, yd)))
    assert rel_err(y, yd) < 2e-5


def test_wino_refuses_what_it_has_no_tiling_for():
    from hoig_amd import _lib as L
    x = torch.randn(1, 24, 16, 32, device='cuda')
    w = torch.randn(64, 32, 3, 3, device='cuda')
    uh = torch.empty(16 * 64 * 32, dtype=torch.int16, device='cuda')
    y = torch.empty(1, 24, 16, 64, device='cuda')
    d = L.ConvDesc(1, 24, 16, 32, 24, 16, 64, 3, 3, 1, 1, 0, 0, 0.0, L.PREC_BF16X3)
    assert L.lib.hoig_conv2d_fwd_wino(ctypes.byref(d), _p(x), _p(uh), _p(uh), None, _p(y), 0) == L.EUNSUPPORTED       # H % 16
    d = L.ConvDesc(1, 16, 16, 32, 16, 16, 64, 3, 3, 1, 1, 0, 0, 0.0, L.PREC_F32)
    assert L.lib.hoig_conv2d_fwd_wino(ctypes.byref(d), _p(x), _p(uh), _p(uh), None, _p(y), 0) == L.EUNSUPPORTED       # arithmetic


def test_tuning_key_wino8_routes_the_half_chip_forward_launches_and_leaves_the_backward_alone():
    """hoig_set_tuning('wino8', 1): ops.conv2d sends a three-term 3x3 forward that the direct kernel would run on half the chip and one
    Winograd round covers (8 images of 512 -> 512 at 32 x 32) to hoig_conv2d_fwd_wino -- same outputs to the arithmetic's floor, the
    backward (direct kernels) unchanged; a launch outside that window (16 images) is not touched: equal bits with the key on or off."""
    from hoig_amd import ops, _lib as L
    ops.set_precision('bf16x3:f16x2')
    prev = L.set_tuning('wino8', 0)
    try:
        g = torch.Generator(device='cuda').manual_seed(77)
        w = ops.pack_weight(torch.randn(512, 512, 3, 3, device='cuda', generator=g) * 0.03).requires_grad_(True)
        bias = torch.randn(512, device='cuda', generator=g).requires_grad_(True)
        res = {}
        for B in (8, 16):
            x = torch.randn(B, 32, 32, 512, device='cuda', generator=g)
            gy = torch.randn(B, 32, 32, 512, device='cuda', generator=g)
            for key in (0, 1):
                L.set_tuning('wino8', key)
                xd = x.clone().requires_grad_(True)
                w.grad = bias.grad = None
                y = ops.conv2d(xd, w, bias, 1, 1)
                y.backward(gy)
                torch.cuda.synchronize()
                res[B, key] = (y.detach().clone(), xd.grad.clone(), w.grad.clone(), bias.grad.clone())
        y0, dx0, dw0, db0 = res[8, 0]
        y1, dx1, dw1, db1 = res[8, 1]
        assert not torch.equal(y0, y1), 'the key did not change the forward kernel'
        assert rel_err(y1, y0) < 2e-5
        assert rel_err(dx1, dx0) < 1e-5 and rel_err(dw1, dw0) < 1e-4 and rel_err(db1, db0) < 1e-4      # (fp32 atomics: summation order)
        assert torch.equal(res[16, 0][0], res[16, 1][0])
    finally:
        L.set_tuning('wino8', prev)
        ops.set_precision('f32')

