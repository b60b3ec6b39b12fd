"""The 3x3 stride-1 halo kernel on v_mfma_f32_16x16x32 (hoig_amd/csrc/conv_halo16.hip) against torch's fp32 convolution: forward, data
gradient, bias / activation / addend / statistics epilogues, the two-tensor input and the two-tensor output of the decoder's skip
convolution.  Shapes are chosen so that the launcher picks the 8-row tilings (128- and 64-channel tiles).  (Until round 6 these tests
also compared with the 32x32x16 instantiations the kernel replaced in round 4; those are deleted.)"""
import ctypes

import pytest
import torch
import torch.nn.functional as F

from gpu_util import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture
def tuning():
    from hoig_amd import _lib as L
    yield L


CASES = [
    # B, Ci, Co, H, W                   tiling the launcher picks
    (8, 64, 256, 64, 64),       # 8 x 32 x 128
    (16, 512, 512, 32, 32),     # 8 x 32 x 128: the step's dominant launch
    (8, 96, 128, 64, 64),       # 8 x 32 x 64  (three channel blocks)
    (8, 512, 512, 32, 32),      # 8 x 32 x 64: the 8-image launches of src_model / tsf_model
    (2, 128, 512, 32, 128),     # 8 x 32 x 64, four tile columns
]


@pytest.mark.parametrize('mode', ['bf16x3', 'f16x2'])
@pytest.mark.parametrize('B,Ci,Co,H,W', CASES)
def test_m16_forward_and_data_gradient(tuning, B, Ci, Co, H, W, mode):
    from hoig_amd import ops
    L = tuning
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(B, H, W, Ci, generator=g) * 1.5 + 0.3).cuda()
    w = (torch.randn(Co, Ci, 3, 3, generator=g) * 0.05)
    bias = torch.randn(Co, generator=g).cuda()
    gy = torch.randn(B, H, W, Co, generator=g).cuda()
    addend = torch.randn(B, H, W, Ci, generator=g).cuda()
    wd = ops.pack_weight(w.cuda())
    ops.set_precision(mode)
    try:
        def run():
            # (no activation in the differentiated pass: a LeakyReLU mask that differs between the two kernels at one output near
            # zero would put an isolated large error into the max-norm of the gradient; the activation epilogue is checked forward)
            xd = x.clone().requires_grad_(True)
            x1 = ops.add(xd, xd)
            y, x2 = ops.conv2d_fork(x1, wd.clone().requires_grad_(True), bias, 1, 1)
            ((y * gy).sum() + (x2 * addend).sum()).backward()
            with torch.no_grad():
                ya = ops.conv2d(x1.detach(), wd, bias, 1, 1, act=L.ACT_LRELU, slope=0.2)
            return y.detach(), xd.grad, ya
        y1, dx1, ya1 = run()
        torch.cuda.synchronize()
    finally:
        ops.set_precision('f32')
    # an independent reference: torch fp32 on the device
    xr = (2 * x).permute(0, 3, 1, 2).clone().requires_grad_(True)
    yr = F.conv2d(xr, w.cuda(), bias, padding=1)
    (yr * gy.permute(0, 3, 1, 2)).sum().backward()
    dxr = 2 * (xr.grad.permute(0, 2, 3, 1) + addend)
    yr = yr.detach().permute(0, 2, 3, 1)
    bf, bd = (3e-4, 3e-4) if mode == 'bf16x3' else (1e-3, 8e-3)
    ef, ed, ea = rel_err(y1, yr), rel_err(dx1, dxr), rel_err(ya1, F.leaky_relu(yr, 0.2))
    assert ef < bf and ed < bd and ea < bf, (ef, ed, ea)


@pytest.mark.parametrize('B,C1,C2,Co,H,W', [(4, 256, 256, 256, 32, 128), (8, 64, 128, 256, 64, 64)])
def test_m16_two_tensor_input_and_output(tuning, B, C1, C2, Co, H, W):
    """ops.conv2d_cat2: the halo loader reads [x1 | x2] by channel block, the data gradient writes [dx1 | dx2]."""
    from hoig_amd import ops
    L = tuning
    g = torch.Generator().manual_seed(9)
    x1 = torch.randn(B, H, W, C1, generator=g).cuda()
    x2 = torch.randn(B, H, W, C2, generator=g).cuda()
    w = ops.pack_weight((torch.randn(Co, C1 + C2, 3, 3, generator=g) * 0.05).cuda())
    gy = torch.randn(B, H, W, Co, generator=g).cuda()

    def run():
        a1, a2 = x1.clone().requires_grad_(True), x2.clone().requires_grad_(True)
        y = ops.conv2d_cat2(a1, a2, w.clone().requires_grad_(True), prec=L.PREC_BF16X3)
        y.backward(gy)
        return y.detach(), a1.grad, a2.grad
    y1, d11, d21 = run()
    torch.cuda.synchronize()
    xr = torch.cat([x1, x2], 3).permute(0, 3, 1, 2).clone().requires_grad_(True)
    yr = F.conv2d(xr, w.permute(0, 1, 2, 3).contiguous(), padding=1)
    yr.backward(gy.permute(0, 3, 1, 2))
    dr = xr.grad.permute(0, 2, 3, 1)
    assert rel_err(y1, yr.detach().permute(0, 2, 3, 1)) < 3e-4
    assert rel_err(d11, dr[..., :C1]) < 3e-4 and rel_err(d21, dr[..., C1:]) < 3e-4


@pytest.mark.parametrize('B,Ci,Co,H,W', [(8, 64, 128, 64, 64), (8, 64, 256, 64, 64)])
def test_m16_statistics_epilogue(tuning, B, Ci, Co, H, W):
    """conv -> instance norm with the statistics taken from the convolution's epilogue (hoig_conv2d_fwd_packed_stats)."""
    from hoig_amd import ops
    L = tuning
    g = torch.Generator().manual_seed(13)
    x = (torch.randn(B, H, W, Ci, generator=g) + 0.5).cuda()
    w = ops.pack_weight((torch.randn(Co, Ci, 3, 3, generator=g) * 0.05).cuda())
    ops.set_precision('bf16x3')
    try:
        def run():
            ops._stats_pending.clear()
            h = ops.conv2d(x, w, None, 1, 1, dead_bias=True)
            assert len(ops._stats_pending) == 1
            y = ops.instance_norm(h)
            assert not ops._stats_pending
            return h, y
        h1, y1 = run()
        torch.cuda.synchronize()
    finally:
        ops.set_precision('f32')
    hr = F.conv2d(x.permute(0, 3, 1, 2), w, padding=1).permute(0, 2, 3, 1)
    assert rel_err(h1, hr) < 3e-4
    yr = F.instance_norm(h1.permute(0, 3, 1, 2), eps=1e-5).permute(0, 2, 3, 1)
    assert rel_err(y1, yr) < 1e-4
    for ws in ops._norm_ws.values():
        assert float(ws[:1 << 18].abs().max()) == 0.0


@pytest.mark.parametrize('mode', ['bf16x3', 'f16x2'])
@pytest.mark.parametrize('B,Ci,Co,Hi,Wi,bias', [(8, 512, 128, 40, 40, False),      # the attention's source-side convolution
                                                (8, 128, 128, 36, 36, True),       # ... its target side (output 32 x 32)
                                                (2, 128, 256, 21, 44, True),       # two channel tiles, no split over the channel blocks
                                                (1, 64, 128, 9, 52, False)])       # the widest canvas the kernel takes; one image
def test_flat5_valid_convolution_and_its_data_gradient(B, Ci, Co, Hi, Wi, bias, mode):
    """Tuning key 'flat5' (conv_flat16.hip): valid 5x5 convolutions on the flattened pixel axis, forward (split over the channel
    blocks with the atomic epilogue where the launch has few tiles) and data gradient (N = Ci: needs Ci % 128 == 0, else the generic
    kernel runs), against torch fp32 and against the generic kernel."""
    from hoig_amd import ops, _lib as L
    g = torch.Generator().manual_seed(23)
    x = torch.randn(B, Hi, Wi, Ci, generator=g).cuda()
    w = ops.pack_weight((torch.randn(Co, Ci, 5, 5, generator=g) * 0.03).cuda())
    b = torch.randn(Co, generator=g).cuda() if bias else None
    gy = torch.randn(B, Hi - 4, Wi - 4, Co, generator=g).cuda()
    prev = L.set_tuning('flat5', 0)
    ops.set_precision(mode)
    try:
        res = []
        for v in (0, 1):
            L.set_tuning('flat5', v)
            xd = x.clone().requires_grad_(True)
            y = ops.conv2d(xd, w.clone().requires_grad_(True), b, 1, 0)
            y.backward(gy)
            torch.cuda.synchronize()
            res.append((y.detach(), xd.grad))
    finally:
        ops.set_precision('f32')
        L.set_tuning('flat5', prev)
    xr = x.permute(0, 3, 1, 2).clone().requires_grad_(True)
    yr = F.conv2d(xr, w, b)
    yr.backward(gy.permute(0, 3, 1, 2))
    bf, bd = (3e-4, 3e-4) if mode == 'bf16x3' else (1e-3, 8e-3)
    for tag, (y, dx) in zip(('generic', 'flat'), res):
        ef, ed = rel_err(y, yr.detach().permute(0, 2, 3, 1)), rel_err(dx, xr.grad.permute(0, 2, 3, 1))
        assert ef < bf and ed < bd, (tag, ef, ed)
    assert rel_err(res[1][0], res[0][0]) < 1e-5 and rel_err(res[1][1], res[0][1]) < 1e-4


@pytest.mark.parametrize('mode', ['f16x2', 'bf16x3'])
@pytest.mark.parametrize('B,Ci,Co,Hi,Wi,bias', [(8, 512, 128, 40, 40, False),      # the attention's source side: output 36 x 36
                                                (4, 128, 128, 36, 36, True),       # its target side: output 32 x 32 (the halo kernel's shape)
                                                (2, 64, 192, 13, 66, True),        # the widest canvas of the kernel; three co tiles
                                                (1, 32, 64, 9, 11, False)])        # fewer positions than one pixel tile
def test_wflat5_weight_gradient_of_valid_5x5(B, Ci, Co, Hi, Wi, bias, mode):
    """Tuning key 'wflat5' (wgrad_flat.hip): dW and the bias gradient of valid 5x5 convolutions on the flattened pixel axis, against
    the kernels it replaces and against torch fp32."""
    from hoig_amd import ops, _lib as L
    g = torch.Generator().manual_seed(29)
    x = torch.randn(B, Hi, Wi, Ci, generator=g).cuda()
    w = ops.pack_weight((torch.randn(Co, Ci, 5, 5, generator=g) * 0.03).cuda())
    b = torch.randn(Co, generator=g).cuda() if bias else None
    gy = torch.randn(B, Hi - 4, Wi - 4, Co, generator=g).cuda()
    prev = L.set_tuning('wflat5', 0)
    ops.set_precision(mode)
    try:
        res = []
        for v in (0, 2):                       # 2: the flattened kernel also where the halo kernel could run
            L.set_tuning('wflat5', v)
            wd = w.clone().requires_grad_(True)
            bd = b.clone().requires_grad_(True) if bias else None
            ops.conv2d(x, wd, bd, 1, 0).backward(gy)
            torch.cuda.synchronize()
            res.append((wd.grad.clone(), None if bd is None else bd.grad.clone()))
    finally:
        ops.set_precision('f32')
        L.set_tuning('wflat5', prev)
    wr = w.detach().clone().requires_grad_(True)
    br = b.clone().requires_grad_(True) if bias else None
    F.conv2d(x.permute(0, 3, 1, 2), wr, br).backward(gy.permute(0, 3, 1, 2))
    lim = 8e-3 if mode == 'f16x2' else 3e-4
    for tag, (dw, db) in zip(('halo / generic', 'flat'), res):
        assert rel_err(dw, wr.grad) < lim, (tag, rel_err(dw, wr.grad))
        if db is not None:
            assert rel_err(db, br.grad) < 1e-4, tag
    assert rel_err(res[1][0], res[0][0]) < 1e-4


@pytest.mark.parametrize('B,Co,H,W', [(2, 5, 64, 64),          # one strip of four tiles, four row chunks
                                      (1, 5, 40, 256),         # the step's shape: two overlapping strips of nine tiles; a ragged last chunk
                                      (2, 3, 16, 256),         # bg_model's head (two column tiles)
                                      (1, 4, 23, 272),         # three strips, the last 16 pixels wide
                                      (1, 5, 9, 150),          # a ragged width: the last tile of the second strip is part padding
                                      (1, 3, 7, 16)])          # fewer rows than taps, one tile
def test_head16_forward_of_the_7x7_heads(B, Co, H, W):
    """Tuning key 'head16' (conv_head16.hip): the 7x7 heads' forward with the horizontal taps as MFMA columns, against the exact-fp32
    VALU kernel it replaces and against torch fp32 -- fused heads with one activation per channel (hoig_conv2d_fwd_heads) and a plain
    convolution with bias and tanh (hoig_conv2d_fwd: bg_model's head)."""
    from hoig_amd import ops, _lib as L
    g = torch.Generator().manual_seed(31 + Co)
    x = (torch.randn(B, H, W, 64, generator=g).abs() * 0.7).cuda()          # (post-ReLU features)
    w = ops.pack_weight((torch.randn(Co, 64, 7, 7, generator=g) * 0.02).cuda())
    bias = (torch.randn(Co, generator=g) * 0.1).cuda()
    splits, acts = ((3, Co - 3), (L.ACT_TANH, L.ACT_SIGMOID)) if Co > 3 else ((3,), (L.ACT_TANH,))
    prev = L.set_tuning('head16', 0)
    ops.set_precision('bf16x3')
    try:
        res = []
        for v in (0, 1):
            L.set_tuning('head16', v)
            with torch.no_grad():
                heads = ops.conv_heads(x, w, splits, acts)
                plain = ops.conv2d(x, w, bias, 1, 3, act=L.ACT_TANH)
            torch.cuda.synchronize()
            res.append((torch.cat(heads, dim=-1), plain))
    finally:
        ops.set_precision('f32')
        L.set_tuning('head16', prev)
    pre = F.conv2d(x.permute(0, 3, 1, 2), w, padding=3).permute(0, 2, 3, 1)
    want_heads = torch.cat([torch.tanh(pre[..., :3]), torch.sigmoid(pre[..., 3:])], dim=-1)
    want_plain = torch.tanh(pre + bias)
    for tag, (heads, plain) in zip(('fp32 VALU', 'mfma'), res):
        assert rel_err(heads, want_heads) < 1e-5 and rel_err(plain, want_plain) < 1e-5, (tag, rel_err(heads, want_heads), rel_err(plain, want_plain))
    assert (res[1][0] - res[0][0]).abs().max() < 2e-5 and (res[1][1] - res[0][1]).abs().max() < 2e-5


# B, Ci, Co, H (input), transposed.  Channel counts give 1..6 blocks of 32 gathered channels on either side: the statically walked kernel
# has its own code for the last block (gather), the last two steps (scatter, one step per block) and the two-step blocks of phase (1, 1)
S2_CASES = [(2, 32, 64, 64, False), (2, 64, 128, 64, False), (2, 96, 64, 64, False), (1, 128, 192, 64, False), (2, 160, 128, 64, False),
            (2, 64, 32 * 3, 32, True), (2, 32, 64, 32, True), (1, 128, 64, 64, True), (2, 160, 128, 32, True), (1, 192, 64, 32, True),
            (16, 64, 128, 256, False), (16, 128, 64, 128, True)]


@pytest.mark.parametrize('mode', ['bf16x3', 'f16x2'])
@pytest.mark.parametrize('B,Ci,Co,H,transposed', S2_CASES)
def test_stride2_statically_walked_kernel_is_bit_identical(B, Ci, Co, H, transposed, mode):
    """conv_s2_16.hip (tuning key 's2_pipe' = 2: every launch) against conv_halo_s2_m16_kernel (0): the same products in the same order, only the
    loads are issued earlier -- forward output and data gradient must be IDENTICAL, for Conv2d s2 p1 (forward gathers, data gradient
    scatters) and ConvTranspose2d s2 p1 op1 (the other way round), with the norm-statistics epilogue on; and within the arithmetic's
    bound of torch's fp32 convolution."""
    from hoig_amd import _lib as L, ops
    if B == 16 and mode == 'f16x2':
        pytest.skip('the large cases once')
    ops.set_precision(mode + ':f16x2')
    prev = L.set_tuning('s2_pipe', -1)
    try:
        g = torch.Generator(device='cuda').manual_seed(B * 100 + Ci)
        x0 = torch.randn(B, H, H, Ci, device='cuda', generator=g)
        w0 = torch.randn((Ci, Co, 3, 3) if transposed else (Co, Ci, 3, 3), device='cuda', generator=g) * 0.05
        outs = []
        for v in (0, 2, 3):
            L.set_tuning('s2_pipe', v)
            x = x0.clone().requires_grad_(True)
            w = ops.pack_weight(w0, transposed=transposed).requires_grad_(True)
            if transposed:
                y = ops.conv_transpose2d(x, w, norm_next=True)
            else:
                y = ops.conv2d(x, w, None, 2, 1, dead_bias=True)               # (dead_bias: the statistics epilogue runs)
            gy = torch.randn(y.shape, device='cuda', generator=torch.Generator(device='cuda').manual_seed(7))
            y.backward(gy)
            ops.join_wgrad_streams()
            torch.cuda.synchronize()
            outs.append((y.detach().clone(), x.grad.clone(), gy))
        (y0, dx0, gy), (y1, dx1, _), (y3, dx3, _) = outs
        # (a side whose output channel count is no multiple of 64 runs on another kernel under either key value -- one that sums with
        # atomics where tiles are few: compared to rounding there)
        for a0, a1, n in ((y0, y1, Co), (dx0, dx1, Ci), (y0, y3, Co), (dx0, dx3, Ci)):      # (y3 / dx3: the 8 x 32 tile form)
            if n % 64 == 0:
                assert torch.equal(a0, a1), (a0 - a1).abs().max().item()
            else:
                assert (a0 - a1).abs().max().item() <= 1e-5 * a0.abs().max().item()
        if B <= 2:
            xr = x0.permute(0, 3, 1, 2).clone().requires_grad_(True)
            if transposed:
                yr = F.conv_transpose2d(xr, w0, None, stride=2, padding=1, output_padding=1)
            else:
                yr = F.conv2d(xr, w0, None, stride=2, padding=1)
            yr.backward(gy.permute(0, 3, 1, 2))
            assert rel_err(y1.permute(0, 3, 1, 2), yr) < (3e-4 if mode == 'bf16x3' else 1e-3)
            assert rel_err(dx1.permute(0, 3, 1, 2), xr.grad) < 8e-3
    finally:
        L.set_tuning('s2_pipe', prev)
        ops.set_precision('f32')


@pytest.mark.parametrize('case', ['resblock_32', 'resblock_64', 'decoder_cat'])
def test_inference_norm_applied_by_the_consumers_loader(case):
    """ops.conv2d_after_norm (hoig_conv2d_fwd_packed_normin): conv3x3(relu(IN(x) * gamma + beta)) with the norm folded into the halo
    loader, against the same chain with the norm as a pass of its own -- same arithmetic up to the association of the fold (one FMA
    instead of subtract / multiply / FMA): 2e-5 of the output's scale; the zero frame of the padding must stay zero (the border pixels
    are where a norm applied to the frame would show).  'decoder_cat': the skip operand precedes x along the channels and is read as
    it is (generator.py:298-309)."""
    from hoig_amd import _lib as L, ops
    ops.set_precision('bf16x3:f16x2')
    try:
        g = torch.Generator(device='cuda').manual_seed(21)
        if case == 'decoder_cat':
            B, H, W, C1, C, Co = 8, 64, 64, 128, 128, 128
        elif case == 'resblock_64':
            B, H, W, C1, C, Co = 4, 64, 64, 0, 256, 256
        else:
            B, H, W, C1, C, Co = 16, 32, 32, 0, 128, 256
        x = torch.randn(B, H, W, C, device='cuda', generator=g) * 2.0 + 0.7           # (a mean and a spread for the norm to remove)
        first = torch.randn(B, H, W, C1, device='cuda', generator=g).relu() if C1 else None
        gamma = torch.rand(C, device='cuda', generator=g) + 0.5
        beta = torch.randn(C, device='cuda', generator=g) * 0.3
        w = ops.pack_weight(torch.randn(Co, C1 + C, 3, 3, device='cuda', generator=g) * 0.05)
        bias = torch.randn(Co, device='cuda', generator=g) if case != 'decoder_cat' else None
        with torch.no_grad():
            y = ops.conv2d_after_norm(x, gamma, beta, w, bias, first=first, norm_next=False)
            assert y is not None, 'the layer should be on the 8-row tilings of the 16x16x32 kernel'
            xn = ops.instance_norm(x, gamma, beta, act=L.ACT_RELU)
            if first is not None:
                want = ops.conv2d_cat2(first, xn, w)
            else:
                want = ops.conv2d(xn, w, bias, 1, 1)
        torch.cuda.synchronize()
        scale = want.abs().max().item()
        assert (y - want).abs().max().item() <= 2e-5 * scale, (y - want).abs().max().item() / scale
        border = torch.ones(H, W, dtype=torch.bool, device='cuda')
        border[1:-1, 1:-1] = False
        assert (y[:, border] - want[:, border]).abs().max().item() <= 2e-5 * scale
        # and against torch, in fp64
        xr = torch.nn.functional.instance_norm(x.permute(0, 3, 1, 2).double(), weight=gamma.double(), bias=beta.double(), eps=1e-5).relu()
        if first is not None:
            xr = torch.cat([first.permute(0, 3, 1, 2).double(), xr], 1)
        wd = w.double()                                                                # (packed = logical values, other strides)
        ref = F.conv2d(xr, wd, bias.double() if bias is not None else None, padding=1).permute(0, 2, 3, 1)
        assert rel_err(y.double(), ref) < 3e-4
        # statistics already left by a producing convolution are taken from the accumulators, not recomputed: same result
        with torch.no_grad():
            src = torch.randn(B, H, W, 64, device='cuda', generator=g)
            w0 = ops.pack_weight(torch.randn(C, 64, 3, 3, device='cuda', generator=g) * 0.1)
            raw = ops.conv2d(src, w0, None, 1, 1, dead_bias=True)                     # (offers its sums on maps of > 1024 pixels)
            y1 = ops.conv2d_after_norm(raw, gamma, beta, w, bias, first=first)
            y2 = ops.conv2d_after_norm(raw.clone(), gamma, beta, w, bias, first=first)   # (a copy: nobody offered sums for it)
        assert (y1 - y2).abs().max().item() <= 2e-5 * y2.abs().max().item()
    finally:
        ops.set_precision('f32')


@pytest.mark.parametrize('case', ['resblock_32', 'resblock_64', 'decoder_cat'])
def test_inference_norm_applied_by_the_f6_loader_and_sums_left_by_its_epilogue(case):
    """The same two fusions on eval.py's default arithmetic (f16f6, conv_f6.hip: hoig_conv2d_fwd_f6_ex): conv3x3(relu(IN(x) * gamma + beta))
    with the norm applied when the fp32 halo is converted to fp16 | fp6 | fp6, against the same kernel fed the tensor a norm pass wrote
    (1e-4 of the output's scale: the fold's association moves an input by an ulp, which can move a block's fp6 rounding), against torch
    in fp64 (the arithmetic's 1e-3 bound), the zero frame -- and the channel sums the f6 kernel's epilogue leaves for the NEXT norm."""
    from hoig_amd import _lib as L, ops
    ops.set_precision('f16f6')
    try:
        g = torch.Generator(device='cuda').manual_seed(23)
        if case == 'decoder_cat':
            B, H, W, C1, C, Co = 8, 64, 64, 128, 128, 128
        elif case == 'resblock_64':
            B, H, W, C1, C, Co = 4, 64, 64, 0, 256, 256
        else:
            B, H, W, C1, C, Co = 16, 32, 32, 0, 128, 256
        x = torch.randn(B, H, W, C, device='cuda', generator=g) * 2.0 + 0.7
        first = torch.randn(B, H, W, C1, device='cuda', generator=g).relu() if C1 else None
        gamma = torch.rand(C, device='cuda', generator=g) + 0.5
        beta = torch.randn(C, device='cuda', generator=g) * 0.3
        w = ops.pack_weight(torch.randn(Co, C1 + C, 3, 3, device='cuda', generator=g) * 0.05)
        bias = torch.randn(Co, device='cuda', generator=g) if case != 'decoder_cat' else None
        with torch.no_grad():
            y = ops.conv2d_after_norm(x, gamma, beta, w, bias, first=first, norm_next=False)
            assert y is not None
            xn = ops.instance_norm(x, gamma, beta, act=L.ACT_RELU)
            want = ops.conv2d_cat2(first, xn, w) if first is not None else ops.conv2d(xn, w, bias, 1, 1)
        torch.cuda.synchronize()
        scale = want.abs().max().item()
        assert (y - want).abs().max().item() <= 1e-4 * scale, (y - want).abs().max().item() / scale
        border = torch.ones(H, W, dtype=torch.bool, device='cuda')
        border[1:-1, 1:-1] = False
        assert (y[:, border] - want[:, border]).abs().max().item() <= 1e-4 * scale
        xr = torch.nn.functional.instance_norm(x.permute(0, 3, 1, 2).double(), weight=gamma.double(), bias=beta.double(), eps=1e-5).relu()
        if first is not None:
            xr = torch.cat([first.permute(0, 3, 1, 2).double(), xr], 1)
        ref = F.conv2d(xr, w.double(), bias.double() if bias is not None else None, padding=1).permute(0, 2, 3, 1)
        assert rel_err(y.double(), ref) < 1e-3
        # the kernel itself, not its three-term fallback: call the entry point and require HOIG_OK
        d = L.ConvDesc(B, H, W, C1 + C, H, W, Co, 3, 3, 1, 1, 0, L.ACT_NONE, 0.0, L.PREC_F16F6)
        hi, _ = ops._packed_planes(w, False, False)
        qh, ql = ops._f6_planes(w)
        st = torch.cuda.current_stream().cuda_stream
        _p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
        a, a2 = (first, xn) if first is not None else (xn, None)
        y0 = torch.empty_like(y)
        sums = torch.zeros(B, 2, Co, device='cuda')
        assert L.lib.hoig_conv2d_fwd_f6_ex(ctypes.byref(d), _p(a), C1, _p(a2), _p(hi), _p(qh), _p(ql), _p(bias), None, None, 0, _p(y0), _p(sums),
                                           st) == L.OK
        torch.cuda.synchronize()
        assert torch.equal(y0, want)
        s1 = y0.double().sum((1, 2))
        s2 = (y0.double() ** 2).sum((1, 2))
        assert rel_err(sums[:, 0].double(), s1) < 1e-5 and rel_err(sums[:, 1].double(), s2) < 1e-5
        # through ops: a producing f6 convolution offers its sums (maps of > 1024 pixels), the norm that follows takes them
        if H * W > 1024:
            with torch.no_grad():
                src = torch.randn(B, H, W, 64, device='cuda', generator=g)
                w0 = ops.pack_weight(torch.randn(C, 64, 3, 3, device='cuda', generator=g) * 0.1)
                raw = ops.conv2d(src, w0, None, 1, 1, dead_bias=True)
                n1 = ops.instance_norm(raw, gamma, beta, act=L.ACT_RELU)
                n2 = ops.instance_norm(raw.clone(), gamma, beta, act=L.ACT_RELU)
                raw = ops.conv2d(src, w0, None, 1, 1, dead_bias=True)
                y1 = ops.conv2d_after_norm(raw, gamma, beta, w, bias, first=first)
                y2 = ops.conv2d_after_norm(raw.clone(), gamma, beta, w, bias, first=first)
            assert (n1 - n2).abs().max().item() <= 2e-5 * n2.abs().max().item()
            assert (y1 - y2).abs().max().item() <= 1e-4 * y2.abs().max().item()
    finally:
        ops.set_precision('f32')


@pytest.mark.parametrize('Ci,B', [(3, 2), (8, 3)])
def test_stem_convolution_leaves_channel_sums_for_its_norm(Ci, B):
    """hoig_conv2d_fwd_stats (conv_thin.hip): the 7x7 stems (3 / 8 -> 64 channels at full resolution, generator.py:100,262) accumulate
    the per-image channel sums of their output in the epilogue, over a workgroup's strip of tiles; the instance norm that follows takes
    them instead of re-reading the largest tensor of the network.  Against the entry point without sums (same kernel: equal bits),
    against fp64 sums of the stored tensor, and through ops (conv2d(.., dead_bias=True) -> instance_norm)."""
    from hoig_amd import _lib as L, ops
    ops.set_precision('bf16x3:f16x2')
    try:
        g = torch.Generator(device='cuda').manual_seed(5 + Ci)
        H = W = 128
        x = torch.randn(B, H, W, Ci, device='cuda', generator=g)
        w = ops.pack_weight(torch.randn(64, Ci, 7, 7, device='cuda', generator=g) * 0.1)
        bias = torch.randn(64, device='cuda', generator=g)
        d = L.ConvDesc(B, H, W, Ci, H, W, 64, 7, 7, 1, 3, 0, L.ACT_NONE, 0.0, L.PREC_BF16X3)
        st = torch.cuda.current_stream().cuda_stream
        _p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
        y0, y1 = torch.empty(B, H, W, 64, device='cuda'), torch.empty(B, H, W, 64, device='cuda')
        sums = torch.zeros(B, 2, 64, device='cuda')
        L.call('hoig_conv2d_fwd', ctypes.byref(d), _p(x), _p(w), _p(bias), _p(y0), st)
        assert L.lib.hoig_conv2d_fwd_stats(ctypes.byref(d), _p(x), _p(w), _p(bias), _p(y1), _p(sums), st) == L.OK
        torch.cuda.synchronize()
        assert torch.equal(y0, y1)
        assert rel_err(sums[:, 0].double(), y1.double().sum((1, 2))) < 1e-5
        assert rel_err(sums[:, 1].double(), (y1.double() ** 2).sum((1, 2))) < 1e-5
        yr = F.conv2d(x.permute(0, 3, 1, 2), w, bias, padding=3).permute(0, 2, 3, 1)
        assert rel_err(y1, yr) < 3e-4
        gamma, beta = torch.rand(64, device='cuda', generator=g) + 0.5, torch.randn(64, device='cuda', generator=g)
        with torch.no_grad():
            raw = ops.conv2d(x, w, bias, 1, 3, dead_bias=True)
            n1 = ops.instance_norm(raw, gamma, beta, act=L.ACT_RELU)            # (takes the sums the stem left)
            n2 = ops.instance_norm(raw.clone(), gamma, beta, act=L.ACT_RELU)    # (a copy: its own statistics pass)
        assert torch.equal(raw, y1)
        assert (n1 - n2).abs().max().item() <= 2e-5 * n2.abs().max().item()
    finally:
        ops.set_precision('f32')

