"""The weight gradient from PRE-SPLIT dy (hoig_amd/csrc/wgrad_dma.hip): `hoig_split_planes_bf16` makes exactly the
two bf16 values the register kernel's in-kernel split makes, and `hoig_conv2d_bwd_weight_split` -- dy tiles copied global -> LDS by
LDS-DMA into double-buffered, XOR-swizzled images -- sums the same products as `hoig_conv2d_bwd_weight`: compared with that kernel (same
arithmetic: agreement to the order of the fp32 atomics) and with a float64 reference of the convolution's weight gradient."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _split_reference(x):
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    return torch.stack([hi, lo], dim=-2)                    # [..., 2, C]


def test_split_planes_are_the_kernels_own_split():
    from hoig_amd import _lib as L
    g = torch.Generator(device='cuda').manual_seed(1)
    for shape in ((3, 5, 7, 64), (1, 32, 32, 512), (2, 1, 3, 4)):
        x = torch.randn(shape, device='cuda', generator=g) * torch.logspace(-6, 3, shape[-1], device='cuda')
        out = torch.empty(shape[:-1] + (2, shape[-1]), dtype=torch.bfloat16, device='cuda')
        L.call('hoig_split_planes_bf16', _p(x), _p(out), x.numel() // shape[-1], shape[-1], torch.cuda.current_stream().cuda_stream)
        want = _split_reference(x)
        assert torch.equal(out.view(torch.int16), want.view(torch.int16)), shape
        # hi + lo reproduces x to 2^-16 relative
        assert ((out[..., 0, :].float() + out[..., 1, :].float() - x).abs() <= x.abs() * 2.0 ** -15).all()


@pytest.mark.parametrize('shape', [(2, 32, 32, 64, 128), (3, 32, 64, 96, 256), (16, 32, 32, 512, 512), (1, 8, 32, 32, 128),
                                   (2, 36, 96, 32, 128)])
@pytest.mark.parametrize('prec', ['f16x2', 'bf16'])
def test_weight_gradient_from_split_dy(shape, prec):
    from hoig_amd import _lib as L, ops
    B, H, W, Ci, Co = shape
    if prec == 'bf16' and Ci * Co > 128 * 128:
        pytest.skip('one large case per arithmetic is enough')
    g = torch.Generator(device='cuda').manual_seed(B * 1000 + Ci)
    x = torch.randn(B, H, W, Ci, device='cuda', generator=g)
    dy = torch.randn(B, H, W, Co, device='cuda', generator=g) * 0.1
    dys = torch.empty(B, H, W, 2, Co, dtype=torch.bfloat16, device='cuda')
    st = torch.cuda.current_stream().cuda_stream
    L.call('hoig_split_planes_bf16', _p(dy), _p(dys), B * H * W, Co, st)
    d = L.ConvDesc(B, H, W, Ci, H, W, Co, 3, 3, 1, 1, 0, L.ACT_NONE, 0.0, ops._PREC[prec])
    dw_ref = torch.zeros(Co, 3, 3, Ci, device='cuda')
    ops.wgrad_call('hoig_conv2d_bwd_weight', d, _p(x), _p(dy), _p(dw_ref), None, st)
    dw = torch.zeros(Co, 3, 3, Ci, device='cuda')
    for _ in range(2):                                       # twice: the kernel ACCUMULATES into dw
        L.call('hoig_conv2d_bwd_weight_split', ctypes.byref(d), _p(x), _p(dys), _p(dw), st)
    torch.cuda.synchronize()
    scale = dw_ref.abs().max().item()
    assert (dw - 2 * dw_ref).abs().max().item() <= 2e-5 * scale, (dw - 2 * dw_ref).abs().max().item() / scale
    if B * H * W * Ci * Co <= 3 * 32 * 64 * 96 * 256:
        want = torch.nn.grad.conv2d_weight(x.cpu().double().permute(0, 3, 1, 2), (Co, Ci, 3, 3), dy.cpu().double().permute(0, 3, 1, 2),
                                           padding=1).permute(0, 2, 3, 1).float().cuda()
        rel = ((dw * 0.5 - want).norm() / want.norm()).item()
        assert rel < (3e-3 if prec == 'f16x2' else 6e-3), rel


def test_split_entry_point_refuses_what_it_has_no_kernel_for():
    from hoig_amd import _lib as L
    x = torch.zeros(1, 32, 32, 32, device='cuda')
    dys = torch.zeros(1, 32, 32, 2, 64, dtype=torch.bfloat16, device='cuda')
    dw = torch.zeros(64, 3, 3, 32, device='cuda')
    st = torch.cuda.current_stream().cuda_stream
    for d in (L.ConvDesc(1, 32, 32, 32, 32, 32, 64, 3, 3, 1, 1, 0, L.ACT_NONE, 0.0, L.PREC_F16X2),           # Co % 128
              L.ConvDesc(1, 32, 32, 32, 32, 32, 128, 3, 3, 1, 1, 0, L.ACT_NONE, 0.0, L.PREC_F32),            # exact arithmetic
              L.ConvDesc(1, 32, 32, 32, 32, 32, 128, 3, 3, 1, 1, 0, L.ACT_NONE, 0.0, L.PREC_BF16X3),         # x split as well
              L.ConvDesc(1, 30, 32, 32, 30, 32, 128, 3, 3, 1, 1, 0, L.ACT_NONE, 0.0, L.PREC_F16X2)):         # rows % 4
        rc = L.lib.hoig_conv2d_bwd_weight_split(ctypes.byref(d), _p(x), _p(dys), _p(dw), st)
        assert rc == L.EUNSUPPORTED, rc


@pytest.mark.parametrize('shape', [(16, 32, 32, 512, 512), (8, 32, 32, 512, 512), (2, 64, 64, 256, 256), (2, 256, 256, 64, 64)])
@pytest.mark.parametrize('with_addend', [False, True])
def test_data_gradient_from_split_dy(shape, with_addend):
    """hoig_conv2d_bwd_data_packed_split against hoig_conv2d_bwd_data_packed(_add) on the same values: the kernel is the same but for
    how the halo reaches LDS, so the results are IDENTICAL (one workgroup per output tile, no atomics)."""
    from hoig_amd import _lib as L, ops
    B, H, W, Ci, Co = shape
    g = torch.Generator(device='cuda').manual_seed(7 + B)
    dy = torch.randn(B, H, W, Co, device='cuda', generator=g) * 0.1
    w = ops.pack_weight(torch.randn(Co, Ci, 3, 3, device='cuda', generator=g) * 0.02)
    add = torch.randn(B, H, W, Ci, device='cuda', generator=g) if with_addend else None
    st = torch.cuda.current_stream().cuda_stream
    dys = torch.empty(B, H, W, 2, Co, dtype=torch.bfloat16, device='cuda')
    L.call('hoig_split_planes_bf16', _p(dy), _p(dys), B * H * W, Co, st)
    thi, tlo = ops._packed_planes(w, False, True)
    d = L.ConvDesc(B, H, W, Ci, H, W, Co, 3, 3, 1, 1, 0, L.ACT_NONE, 0.0, L.PREC_F16X2)
    dx_ref, dx = torch.empty(B, H, W, Ci, device='cuda'), torch.empty(B, H, W, Ci, device='cuda')
    if with_addend:
        L.call('hoig_conv2d_bwd_data_packed_add', ctypes.byref(d), _p(dy), _p(thi), _p(tlo), _p(add), _p(dx_ref), st)
    else:
        L.call('hoig_conv2d_bwd_data_packed', ctypes.byref(d), _p(dy), _p(thi), _p(tlo), _p(dx_ref), st)
    rc = L.lib.hoig_conv2d_bwd_data_packed_split(ctypes.byref(d), _p(dys), _p(thi), _p(tlo), _p(add) if with_addend else None, _p(dx), st)
    if rc == L.EUNSUPPORTED:
        pytest.skip('no pre-split form of the data gradient for this shape (the caller un-splits): %r' % (shape,))
    assert rc == 0, rc
    torch.cuda.synchronize()
    assert torch.equal(dx, dx_ref), (dx - dx_ref).abs().max().item()


@pytest.mark.parametrize('kind', ['in_relu', 'in_affine_residual', 'spade', 'fork'])
@pytest.mark.parametrize('hw', [(32, 32), (64, 64)])
def test_norm_backward_hands_the_convolution_split_planes(kind, hw):
    """conv3x3 -> instance norm (plain / affine + residual / SPADE) with the tuning key `split_grads` on and off: the norm's backward
    writes its dx as bf16 hi | lo planes (one-launch kernel at 32 x 32, three-kernel path at 64 x 64), the convolution's backward reads
    them.  Same arithmetic both ways: dx identical, dW to the order of the atomics; every registered gradient is consumed."""
    from hoig_amd import _lib as L, nn as hnn, ops
    H, W = hw
    B, Ci, Co = 4, 64, 128
    ops.set_precision('bf16x3:f16x2')
    try:
        g = torch.Generator(device='cuda').manual_seed(11)
        tree = hnn.ParamTree({'c.weight': (Co, Ci, 3, 3), 'n.weight': (Co,), 'n.bias': (Co,)}, torch.device('cuda'), {}, {})
        with torch.no_grad():
            tree.flat.copy_(torch.randn(tree.flat.shape, device='cuda', generator=g) * 0.05)
        tree.version += 1
        x0 = torch.randn(B, H, W, Ci, device='cuda', generator=g)
        res = torch.randn(B, H, W, Co, device='cuda', generator=g)
        gb = torch.randn(B, H, W, 2 * Co, device='cuda', generator=g) * 0.3
        gout = torch.randn(B, H, W, Co, device='cuda', generator=g)
        outs = {}
        for split in (1, 0):
            L.set_tuning('split_grads', split)
            tree.flat_grad.zero_()
            x = x0.clone().requires_grad_(True)
            skip = None
            if kind == 'fork':
                y, xr = ops.conv2d_fork(x, tree.P['c.weight'], None, 1, 1, dead_bias=True)
                skip = xr * 2.0                                         # the second reader of x
            else:
                y = ops.conv2d(x, tree.P['c.weight'], None, 1, 1, dead_bias=True)
            assert (getattr(y, '_hoig_split_grad', None) is not None) == bool(split)
            if kind == 'in_affine_residual':
                z = ops.instance_norm(y, tree.P['n.weight'], tree.P['n.bias'], residual=res)
            elif kind == 'spade':
                z = ops.spade_norm_fused(y, gb, act=L.ACT_RELU)
            else:
                z = ops.instance_norm(y, act=L.ACT_RELU)
            loss = (z * gout).sum() + (skip.sum() if skip is not None else 0.0)
            loss.backward()
            ops.join_wgrad_streams()
            ops.check_split_grads_consumed()
            torch.cuda.synchronize()
            outs[split] = (x.grad.clone(), tree.flat_grad.clone())
        dx1, dw1 = outs[1]
        dx0, dw0 = outs[0]
        if H * W <= 1024:
            # (identical where the data gradient has a pre-split kernel; where it un-splits hi + lo and splits again, equal to ~2^-17)
            assert (dx1 - dx0).abs().max().item() <= 4e-6 * dx0.abs().max().item(), (dx1 - dx0).abs().max().item()
        else:
            # larger maps take their statistics from the convolution's epilogue: fp32 atomics, so mean / rstd differ in the last bit from
            # run to run and a ReLU mask recomputed from them flips at elements that sit on zero -- isolated pixels, either way round
            assert ((dx1 - dx0).norm() / dx0.norm()).item() < 1e-2
        assert ((dw1 - dw0).norm() / dw0.norm()).item() < (2e-5 if H * W <= 1024 else 1e-2)
    finally:
        L.set_tuning('split_grads', 1)
        ops.set_precision('f32')


def _tagged_conv(seed=21, B=2, H=32, W=32, Ci=64, Co=128):
    from hoig_amd import nn as hnn, ops
    g = torch.Generator(device='cuda').manual_seed(seed)
    tree = hnn.ParamTree({'c.weight': (Co, Ci, 3, 3)}, torch.device('cuda'), {}, {})
    with torch.no_grad():
        tree.flat.copy_(torch.randn(tree.flat.shape, device='cuda', generator=g) * 0.05)
    tree.version += 1
    x = torch.randn(B, H, W, Ci, device='cuda', generator=g).requires_grad_(True)
    y = ops.conv2d(x, tree.P['c.weight'], None, 1, 1, dead_bias=True)
    assert getattr(y, '_hoig_split_grad', None) is not None              # eligible: the convolution hung a token on its output
    return tree, x, y, g


def test_a_tagged_output_with_a_second_consumer_raises_instead_of_mixing_planes_with_fp32():
    """conv -> norm hands planes over through the bytes of an fp32 tensor.  If the convolution's output has a SECOND consumer
    that is not a norm, autograd sums that consumer's fp32 gradient with the plane bits; the convolution's backward must notice that what
    arrived is not the tensor the norm wrote, and raise -- never produce a gradient from the mixture."""
    from hoig_amd import _lib as L, ops
    ops.set_precision('bf16x3:f16x2')
    try:
        tree, x, y, g = _tagged_conv()
        z = ops.instance_norm(y, act=L.ACT_RELU)
        loss = z.sum() + (y * 0.5).sum()                                 # the second reader: a plain torch op
        with pytest.raises(RuntimeError, match='second consumer'):
            loss.backward()
        ops.join_wgrad_streams()
        torch.cuda.synchronize()
        ops.check_split_grads_consumed()                                 # (the failed hand-off left nothing behind)
    finally:
        ops.set_precision('f32')


def test_two_norms_on_one_tagged_output_fall_back_to_fp32_gradients():
    """Two norms reading one tagged convolution output both count themselves on its token: neither writes planes, the engine sums two
    fp32 gradients and the convolution's backward reads fp32 -- the result of the un-tagged path."""
    from hoig_amd import _lib as L, ops
    ops.set_precision('bf16x3:f16x2')
    try:
        outs = []
        for split in (1, 0):
            L.set_tuning('split_grads', split)
            g = torch.Generator(device='cuda').manual_seed(21)
            from hoig_amd import nn as hnn
            tree = hnn.ParamTree({'c.weight': (128, 64, 3, 3)}, torch.device('cuda'), {}, {})
            with torch.no_grad():
                tree.flat.copy_(torch.randn(tree.flat.shape, device='cuda', generator=g) * 0.05)
            tree.version += 1
            x = torch.randn(2, 32, 32, 64, device='cuda', generator=g).requires_grad_(True)
            gout = torch.randn(2, 32, 32, 128, device='cuda', generator=g)
            y = ops.conv2d(x, tree.P['c.weight'], None, 1, 1, dead_bias=True)
            z = ops.instance_norm(y, act=L.ACT_RELU) + 2.0 * ops.instance_norm(y)
            (z * gout).sum().backward()
            ops.join_wgrad_streams()
            ops.check_split_grads_consumed()
            torch.cuda.synchronize()
            outs.append((x.grad.clone(), tree.flat_grad.clone()))
        # (this small layer's data gradient runs on the flattened-axis kernel, whose atomic epilogue reorders fp32 sums)
        assert (outs[0][0] - outs[1][0]).abs().max().item() <= 4e-6 * outs[1][0].abs().max().item()
        assert ((outs[0][1] - outs[1][1]).norm() / outs[1][1].norm()).item() < 2e-5      # (fp32 atomics order)
    finally:
        L.set_tuning('split_grads', 1)
        ops.set_precision('f32')


def test_an_unconsumed_offer_is_reported_and_released():
    """A norm wrote planes for a convolution whose backward then never ran (the graph was cut): check_split_grads_consumed reports it
    and drops the tensor it pinned."""
    from hoig_amd import _lib as L, ops
    ops.set_precision('bf16x3:f16x2')
    try:
        tree, x, y, g = _tagged_conv()
        z = ops.instance_norm(y, act=L.ACT_RELU)
        torch.autograd.grad(z.sum(), y)                                  # stops at y: the convolution's backward does not run
        with pytest.raises(RuntimeError, match='not consumed'):
            ops.check_split_grads_consumed()
        ops.check_split_grads_consumed()                                 # cleared
    finally:
        ops.set_precision('f32')


@pytest.mark.parametrize('shape', [(8, 32, 32, 512, 512), (2, 64, 64, 128, 256), (4, 32, 32, 128, 1024)])
def test_grouped_launch_equals_the_two_single_launches(shape):
    """hoig_conv2d_*_pair: two convolutions of one descriptor (different tensors, different weights) as ONE grid -- src_model's and
    tsf_model's layer.  Forward and data gradient are those of the single launches bit for bit (one workgroup per output tile either
    way); the weight gradients agree to the order of the atomics."""
    from hoig_amd import _lib as L, ops
    B, H, W, Ci, Co = shape
    g = torch.Generator(device='cuda').manual_seed(5 + B)
    st = torch.cuda.current_stream().cuda_stream
    xs = [torch.randn(B, H, W, Ci, device='cuda', generator=g) for _ in range(2)]
    dys = [torch.randn(B, H, W, Co, device='cuda', generator=g) * 0.1 for _ in range(2)]
    ws = [ops.pack_weight(torch.randn(Co, Ci, 3, 3, device='cuda', generator=g) * 0.02) for _ in range(2)]
    bias = [torch.randn(Co, device='cuda', generator=g) for _ in range(2)]
    adds = [torch.randn(B, H, W, Ci, device='cuda', generator=g) for _ in range(2)]
    fwd_planes = [ops._packed_planes(w, False, False) for w in ws]
    bwd_planes = [ops._packed_planes(w, False, True) for w in ws]
    d = L.ConvDesc(B, H, W, Ci, H, W, Co, 3, 3, 1, 1, 0, L.ACT_NONE, 0.0, L.PREC_BF16X3)
    d2 = L.ConvDesc(B, H, W, Ci, H, W, Co, 3, 3, 1, 1, 0, L.ACT_NONE, 0.0, L.PREC_F16X2)
    # forward
    y_ref = [torch.empty(B, H, W, Co, device='cuda') for _ in range(2)]
    for i in range(2):
        L.call('hoig_conv2d_fwd_packed', ctypes.byref(d), _p(xs[i]), _p(fwd_planes[i][0]), _p(fwd_planes[i][1]), _p(bias[i]), _p(y_ref[i]), st)
    y = [torch.empty_like(t) for t in y_ref]
    rc = L.lib.hoig_conv2d_fwd_packed_pair(ctypes.byref(d), _p(xs[0]), _p(xs[1]), _p(fwd_planes[0][0]), _p(fwd_planes[0][1]),
                                           _p(fwd_planes[1][0]), _p(fwd_planes[1][1]), _p(bias[0]), _p(bias[1]), _p(y[0]), _p(y[1]), st)
    if rc == L.EUNSUPPORTED:
        pytest.skip('no grouped tiling for %r' % (shape,))
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(y[0], y_ref[0]) and torch.equal(y[1], y_ref[1])
    # backward from split dy
    sp = [torch.empty(B, H, W, 2, Co, dtype=torch.bfloat16, device='cuda') for _ in range(2)]
    for i in range(2):
        L.call('hoig_split_planes_bf16', _p(dys[i]), _p(sp[i]), B * H * W, Co, st)
    for with_add in (False, True):
        dx_ref = [torch.empty(B, H, W, Ci, device='cuda') for _ in range(2)]
        for i in range(2):
            if with_add:
                L.call('hoig_conv2d_bwd_data_packed_add', ctypes.byref(d2), _p(dys[i]), _p(bwd_planes[i][0]), _p(bwd_planes[i][1]), _p(adds[i]),
                       _p(dx_ref[i]), st)
            else:
                L.call('hoig_conv2d_bwd_data_packed', ctypes.byref(d2), _p(dys[i]), _p(bwd_planes[i][0]), _p(bwd_planes[i][1]), _p(dx_ref[i]), st)
        dx = [torch.empty_like(t) for t in dx_ref]
        rc = L.lib.hoig_conv2d_bwd_data_packed_split_pair(ctypes.byref(d2), _p(sp[0]), _p(sp[1]), _p(bwd_planes[0][0]), _p(bwd_planes[0][1]),
                                                          _p(bwd_planes[1][0]), _p(bwd_planes[1][1]), _p(adds[0]) if with_add else None,
                                                          _p(adds[1]) if with_add else None, _p(dx[0]), _p(dx[1]), st)
        if rc == L.EUNSUPPORTED:                 # (too few tiles of Ci channels for a grouped tiling: the caller falls back per problem)
            break
        assert rc == 0
        torch.cuda.synchronize()
        assert torch.equal(dx[0], dx_ref[0]) and torch.equal(dx[1], dx_ref[1]), with_add
    dw_ref = [torch.zeros(Co, 3, 3, Ci, device='cuda') for _ in range(2)]
    for i in range(2):
        L.call('hoig_conv2d_bwd_weight_split', ctypes.byref(d2), _p(xs[i]), _p(sp[i]), _p(dw_ref[i]), st)
    dw = [torch.zeros_like(t) for t in dw_ref]
    L.call('hoig_conv2d_bwd_weight_split_pair', ctypes.byref(d2), _p(xs[0]), _p(xs[1]), _p(sp[0]), _p(sp[1]), _p(dw[0]), _p(dw[1]), st)
    torch.cuda.synchronize()
    for i in range(2):
        assert (dw[i] - dw_ref[i]).abs().max().item() <= 2e-5 * dw_ref[i].abs().max().item(), i


def test_step_with_grouped_launches_matches_the_step_without():
    """Trainer at 256 x 256, batch 4, with src_model's / tsf_model's residual-block convolutions as grouped launches (tuning key `pair`)
    and as one launch per sub-network: the forward is identical (the same tiles, issued in one grid), the seven loss terms agree, the
    generator's gradient agrees to the order of the fp32 atomics / mask flips."""
    from common import product_trainer
    from hoig_amd import _lib as L, ops
    ops.set_precision('bf16x3:f16x2')
    res = {}
    try:
        for pair in (1, 0):
            L.set_tuning('pair', pair)
            m = product_trainer('generator_spade_attn', 4, 256)
            with torch.no_grad():
                outs = [o.float().clone() for o in m.forward()]
            m.optimize_parameters()
            torch.cuda.synchronize()
            res[pair] = (outs, dict(m.get_current_errors()), m._net(m._G).flat_grad.clone())
            m.close()
            del m
            torch.cuda.empty_cache()
    finally:
        L.set_tuning('pair', 2)
        ops.set_precision('f32')
    for a, b in zip(res[1][0], res[0][0]):          # (not bitwise: the large maps' norm statistics come from fp32 atomics in both runs)
        assert (a - b).abs().max().item() <= 1e-4 * max(b.abs().max().item(), 1e-3)
    for k, v in res[0][1].items():
        assert abs(res[1][1][k] - v) <= 1e-5 * max(abs(v), 1e-2), (k, res[1][1][k], v)
    g1, g0 = res[1][2], res[0][2]
    assert ((g1 - g0).norm() / g0.norm()).item() < 2e-2


def test_grouped_backward_skips_the_weight_gradient_of_a_frozen_weight():
    """ops.conv2d_pair with one of the two weights frozen (the grouped backward used to compute -- and need a gradient
    buffer for -- both): the live weight's gradient and both data gradients equal the all-live run's, the frozen one's buffer stays
    untouched."""
    from hoig_amd import _lib as L, nn as hnn, ops
    ops.set_precision('bf16x3:f16x2')
    prev = L.set_tuning('pair', 1)
    try:
        g = torch.Generator(device='cuda').manual_seed(23)
        tree = hnn.ParamTree({'a.weight': (256, 64, 3, 3), 'b.weight': (256, 64, 3, 3)}, torch.device('cuda'), {}, {})
        with torch.no_grad():
            tree.flat.copy_(torch.randn(tree.flat.shape, device='cuda', generator=g) * 0.05)
        tree.version += 1
        xa0, xb0 = (torch.randn(8, 32, 32, 64, device='cuda', generator=g) for _ in range(2))
        ga, gb = (torch.randn(8, 32, 32, 256, device='cuda', generator=g) for _ in range(2))
        wa, wb = tree.P['a.weight'], tree.P['b.weight']
        assert ops.pair_ok(xa0, xb0, wa, wb)
        res = []
        for frozen in (False, True):
            tree.flat_grad.zero_()
            wb.requires_grad_(not frozen)
            xa, xb = xa0.clone().requires_grad_(True), xb0.clone().requires_grad_(True)
            ya, yb = ops.conv2d_pair(xa, xb, wa, wb)
            ((ya * ga).sum() + (yb * gb).sum()).backward()
            ops.join_wgrad_streams()
            torch.cuda.synchronize()
            res.append((xa.grad.clone(), xb.grad.clone(), wa.grad.clone(), wb.grad.clone()))
        wb.requires_grad_(True)
        (dxa0, dxb0, dwa0, dwb0), (dxa1, dxb1, dwa1, dwb1) = res
        # (64 output channels: the data gradient runs on a kernel whose atomic epilogue reorders fp32 sums from run to run)
        assert (dxa0 - dxa1).abs().max().item() <= 4e-6 * dxa0.abs().max().item()
        assert (dxb0 - dxb1).abs().max().item() <= 4e-6 * dxb0.abs().max().item()
        assert ((dwa0 - dwa1).norm() / dwa0.norm()).item() < 2e-5 and dwb0.abs().max().item() > 0
        assert dwb1.abs().max().item() == 0.0
    finally:
        L.set_tuning('pair', prev)
        ops.set_precision('f32')
