"""GPU parity of every C-ABI operator family against the fp32 CPU oracle arithmetic (torch ATen ops are the
reference's own arithmetic; K1-K4 / attention against oracle/hogan_oracle.py).  Tolerances: fp32-exact MFMA mode
1e-4 relative (summation order differs), far inside the 1e-3 north_star bound."""
import pytest
import torch
import torch.nn.functional as F

from gpu_util import nhwc_cuda, nchw_cpu, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _ops():
    from hoig_amd import ops
    return ops


CONV_CASES = [
    # B, Ci, Co, H, k, stride, pad, bias
    (2, 64, 64, 16, 3, 1, 1, False),
    (2, 128, 256, 16, 3, 2, 1, False),
    (1, 512, 512, 8, 3, 1, 1, True),
    (2, 3, 64, 32, 7, 1, 3, False),
    (2, 8, 64, 16, 7, 1, 3, False),
    (2, 12, 128, 16, 3, 1, 1, True),
    (2, 64, 3, 32, 7, 1, 3, False),
    (2, 128, 1, 16, 7, 1, 3, False),
    (2, 19, 64, 32, 4, 2, 1, True),
    (2, 64, 128, 16, 4, 2, 1, True),
    (2, 256, 256, 9, 4, 1, 1, True),
    (2, 256, 1, 8, 4, 1, 1, True),
    (3, 96, 160, 10, 3, 1, 1, True),
    (2, 128, 25, 8, 1, 1, 0, True),
]


@pytest.mark.parametrize('B,Ci,Co,H,k,stride,pad,bias', CONV_CASES)
def test_conv2d_fwd_bwd(B, Ci, Co, H, k, stride, pad, bias):
    ops = _ops()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, Ci, H, H, generator=g)
    w = torch.randn(Co, Ci, k, k, generator=g) * 0.1
    b = torch.randn(Co, generator=g) if bias else None
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True) if bias else None
    yr = F.conv2d(xr, wr, br, stride=stride, padding=pad)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)

    xd = nhwc_cuda(x).requires_grad_(True)
    wd = ops.pack_weight(w.cuda()).requires_grad_(True)
    bd = b.cuda().requires_grad_(True) if bias else None
    y = ops.conv2d(xd, wd, bd, stride, pad)
    y.backward(nhwc_cuda(gy))
    torch.cuda.synchronize()
    assert rel_err(nchw_cpu(y), yr) < TOL
    assert rel_err(nchw_cpu(xd.grad), xr.grad) < TOL
    assert rel_err(wd.grad, wr.grad) < TOL
    if bias:
        assert rel_err(bd.grad, br.grad) < TOL


@pytest.mark.parametrize('B,Ci,Co,H', [(2, 64, 32, 8), (1, 512, 256, 4), (2, 128, 64, 16), (3, 32, 32, 6)])
def test_conv_transpose2d_fwd_bwd(B, Ci, Co, H):
    ops = _ops()
    g = torch.Generator().manual_seed(2)
    x = torch.randn(B, Ci, H, H, generator=g)
    w = torch.randn(Ci, Co, 3, 3, generator=g) * 0.1
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv_transpose2d(xr, wr, None, stride=2, padding=1, output_padding=1)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    xd = nhwc_cuda(x).requires_grad_(True)
    wd = ops.pack_weight(w.cuda(), transposed=True).requires_grad_(True)
    y = ops.conv_transpose2d(xd, wd)
    y.backward(nhwc_cuda(gy))
    assert rel_err(nchw_cpu(y), yr) < TOL
    assert rel_err(nchw_cpu(xd.grad), xr.grad) < TOL
    assert rel_err(wd.grad, wr.grad) < TOL


@pytest.mark.parametrize('act', ['relu', 'lrelu', 'tanh', 'sigmoid'])
def test_conv_epilogue_activations(act):
    ops = _ops()
    from hoig_amd import _lib as L
    code = dict(relu=L.ACT_RELU, lrelu=L.ACT_LRELU, tanh=L.ACT_TANH, sigmoid=L.ACT_SIGMOID)[act]
    fn = dict(relu=F.relu, lrelu=lambda t: F.leaky_relu(t, 0.2), tanh=torch.tanh, sigmoid=torch.sigmoid)[act]
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 32, 8, 8, generator=g)
    w = torch.randn(32, 32, 3, 3, generator=g) * 0.1
    b = torch.randn(32, generator=g)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = fn(F.conv2d(xr, wr, b, padding=1))
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    xd = nhwc_cuda(x).requires_grad_(True)
    wd = ops.pack_weight(w.cuda()).requires_grad_(True)
    y = ops.conv2d(xd, wd, b.cuda(), 1, 1, code, 0.2)
    y.backward(nhwc_cuda(gy))
    assert rel_err(nchw_cpu(y), yr) < TOL
    assert rel_err(nchw_cpu(xd.grad), xr.grad) < TOL
    assert rel_err(wd.grad, wr.grad) < TOL


@pytest.mark.parametrize('mode', ['plain', 'affine', 'spade'])
@pytest.mark.parametrize('B,C,H,W', [(2, 64, 16, 16), (2, 512, 8, 8), (3, 128, 15, 15), (1, 256, 14, 14)])
def test_instance_norm(mode, B, C, H, W):
    ops = _ops()
    from hoig_amd import _lib as L
    g = torch.Generator().manual_seed(4)
    x = torch.randn(B, C, H, W, generator=g) * 2 + 3          # non-zero mean stresses the variance computation
    xr = x.clone().requires_grad_(True)
    xd = nhwc_cuda(x).requires_grad_(True)
    if mode == 'plain':
        yr = F.leaky_relu(F.instance_norm(xr, eps=1e-5), 0.2)
        y = ops.instance_norm(xd, act=L.ACT_LRELU, slope=0.2)
        extra = []
    elif mode == 'affine':
        w, b = torch.randn(C, generator=g), torch.randn(C, generator=g)
        wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        res = torch.randn(B, C, H, W, generator=g)
        yr = res + F.instance_norm(xr, weight=wr, bias=br, eps=1e-5)
        wd, bd = w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
        y = ops.instance_norm(xd, wd, bd, residual=nhwc_cuda(res))
        extra = [(wd, wr), (bd, br)]
    else:
        ga, be = torch.randn(B, C, H, W, generator=g), torch.randn(B, C, H, W, generator=g)
        gar, ber = ga.clone().requires_grad_(True), be.clone().requires_grad_(True)
        yr = F.relu(F.instance_norm(xr, eps=1e-5) * (1 + gar) + ber)
        gad, bed = nhwc_cuda(ga).requires_grad_(True), nhwc_cuda(be).requires_grad_(True)
        y = ops.spade_norm(xd, gad, bed, act=L.ACT_RELU)
        extra = [(gad, gar), (bed, ber)]
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    y.backward(nhwc_cuda(gy))
    assert rel_err(nchw_cpu(y), yr) < TOL
    assert rel_err(nchw_cpu(xd.grad), xr.grad) < 5 * TOL
    for d, r in extra:
        dg = nchw_cpu(d.grad) if d.grad.dim() == 4 else d.grad
        assert rel_err(dg, r.grad) < 5 * TOL


@pytest.mark.parametrize('fork', [False, True])
@pytest.mark.parametrize('B,C,H,W', [(2, 64, 16, 16), (3, 128, 15, 15), (2, 64, 48, 32)])     # one-launch and streaming kernels
def test_spade_norm_fused_and_its_fork(B, C, H, W, fork):
    """IN(x) * (1 + gamma) + beta -> ReLU from the [.,2C] gamma|beta tensor (spade.py:36); with fork=True the norm's backward
    also adds the gradient of x's second reader (the SPADE residual block's skip, generator.py:63-71)."""
    ops = _ops()
    from hoig_amd import _lib as L
    g = torch.Generator().manual_seed(14)
    x = torch.randn(B, C, H, W, generator=g) * 2 + 1
    ga, be = torch.randn(B, C, H, W, generator=g), torch.randn(B, C, H, W, generator=g)
    xr, gar, ber = (t.clone().requires_grad_(True) for t in (x, ga, be))
    yr = F.relu(F.instance_norm(xr, eps=1e-5) * (1 + gar) + ber)
    gy, gx = torch.randn(yr.shape, generator=g), torch.randn(x.shape, generator=g)
    ((yr * gy).sum() + ((xr * gx).sum() if fork else 0)).backward()
    xd = nhwc_cuda(x).requires_grad_(True)
    gb = torch.cat([nhwc_cuda(ga), nhwc_cuda(be)], dim=3).contiguous().requires_grad_(True)
    if fork:
        y, x2 = ops.spade_norm_fused(xd, gb, act=L.ACT_RELU, fork=True)
        ((y * nhwc_cuda(gy)).sum() + (x2 * nhwc_cuda(gx)).sum()).backward()
    else:
        y = ops.spade_norm_fused(xd, gb, act=L.ACT_RELU)
        y.backward(nhwc_cuda(gy))
    assert rel_err(nchw_cpu(y), yr) < TOL
    assert rel_err(nchw_cpu(xd.grad), xr.grad) < 5 * TOL
    assert rel_err(nchw_cpu(gb.grad[..., :C]), gar.grad) < 5 * TOL and rel_err(nchw_cpu(gb.grad[..., C:]), ber.grad) < 5 * TOL


@pytest.mark.parametrize('fork', [False, True])
@pytest.mark.parametrize('B,C,h', [(2, 128, 8), (1, 256, 6), (2, 512, 4), (1, 128, 32)])
def test_local_attention_vs_oracle(B, C, h, fork):
    """Fused attention (fc1 MFMA + pixel kernel) against the oracle's materialising restatement of
    extract_attn.py:23-29, forward and all gradients.  fork=True: source and target have later readers, whose gradients come
    back through the attention's pass-through outputs and are added by its backward kernels (generator.py:391-392)."""
    ops = _ops()
    from oracle import hogan_oracle as O
    g = torch.Generator().manual_seed(5)
    src = torch.randn(B, C, h, h, generator=g)
    tgt = torch.randn(B, C, h, h, generator=g)
    flow = torch.randn(B, 2, h, h, generator=g) * 1.5
    flow[0, :, 0, 0] = -40.0                                  # far out of range: exercises the border clamp
    sd = {'a.fully_connect_layer.0.weight': torch.randn(128, 2 * C, 5, 5, generator=g) * 0.02,
          'a.fully_connect_layer.0.bias': torch.randn(128, generator=g) * 0.1,
          'a.fully_connect_layer.2.weight': torch.randn(25, 128, 1, 1, generator=g) * 0.3,
          'a.fully_connect_layer.2.bias': torch.randn(25, generator=g) * 0.1}
    ref = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    sr, tr = src.clone().requires_grad_(True), tgt.clone().requires_grad_(True)
    yr = O.extractor_attn(sr, tr, flow, ref, 'a')
    gy = torch.randn(yr.shape, generator=g)
    es, et = torch.randn(src.shape, generator=g), torch.randn(tgt.shape, generator=g)
    ((yr * gy).sum() + (((sr * es).sum() + (tr * et).sum()) if fork else 0)).backward()

    from hoig_amd.nn import split_attn_weight, merge_attn_weight
    sdv, tdv = nhwc_cuda(src).requires_grad_(True), nhwc_cuda(tgt).requires_grad_(True)
    wt_l, ws_l = split_attn_weight(sd['a.fully_connect_layer.0.weight'].cuda())     # storage form of the (128,2C,5,5) weight
    wt = ops.pack_weight(wt_l.contiguous()).requires_grad_(True)
    ws = ops.pack_weight(ws_l.contiguous()).requires_grad_(True)
    b1 = sd['a.fully_connect_layer.0.bias'].cuda().requires_grad_(True)
    w2 = ops.pack_weight(sd['a.fully_connect_layer.2.weight'].cuda()).requires_grad_(True)
    b2 = sd['a.fully_connect_layer.2.bias'].cuda().requires_grad_(True)
    if fork:
        gsv, s1 = ops.attn_source_conv(sdv, ws, fork=True)
        y, s2, t2 = ops.local_attention(s1, tdv, flow.cuda(), wt, ws, b1, w2, b2, gs=gsv, fork=True)
        assert s2.data_ptr() == sdv.data_ptr() and t2.data_ptr() == tdv.data_ptr()
        ((y * nhwc_cuda(gy)).sum() + (s2 * nhwc_cuda(es)).sum() + (t2 * nhwc_cuda(et)).sum()).backward()
    else:
        y = ops.local_attention(sdv, tdv, flow.cuda(), wt, ws, b1, w2, b2)
        y.backward(nhwc_cuda(gy))
    assert rel_err(nchw_cpu(y), yr) < TOL
    assert rel_err(nchw_cpu(sdv.grad), sr.grad) < 5 * TOL
    assert rel_err(nchw_cpu(tdv.grad), tr.grad) < 5 * TOL
    w1g = merge_attn_weight(wt.grad, ws.grad)
    assert rel_err(w1g, ref['a.fully_connect_layer.0.weight'].grad) < 5 * TOL
    assert rel_err(b1.grad, ref['a.fully_connect_layer.0.bias'].grad) < 5 * TOL
    assert rel_err(w2.grad, ref['a.fully_connect_layer.2.weight'].grad) < 5 * TOL
    assert rel_err(b2.grad, ref['a.fully_connect_layer.2.bias'].grad) < 5 * TOL


def test_block_extractor_and_reshape_dropins():
    """The NCHW drop-ins for block_extractor_cuda / local_attn_reshape_cuda against the oracle (K1-K4), incl. the
    reference's own known-answer read-out (test_local_attn_reshape.py:29-43)."""
    ops = _ops()
    from oracle import hogan_oracle as O
    g = torch.Generator().manual_seed(6)
    src = torch.rand(4, 6, 14, 10, generator=g)
    flow = torch.rand(4, 2, 14, 10, generator=g) * 1.8      # the recipe of test_block_extractor.py:74-75
    sr, fr = src.clone().requires_grad_(True), flow.clone().requires_grad_(True)
    yr = O.block_extract(sr, fr, 3)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    sd, fd = src.cuda().requires_grad_(True), flow.cuda().requires_grad_(True)
    y = ops.block_extractor(sd, fd, 3)
    y.backward(gy.cuda())
    assert rel_err(y, yr) < 1e-6
    assert rel_err(sd.grad, sr.grad) < 1e-5
    assert rel_err(fd.grad, fr.grad) < 1e-4
    inp = torch.arange(9.0).view(1, 9, 1, 1).repeat(2, 1, 10, 10)
    out = ops.local_attn_reshape(inp.cuda().requires_grad_(True), 3)
    assert out[0, 0, :3, :3].cpu().tolist() == [[0, 1, 2], [3, 4, 5], [6, 7, 8]]
    assert torch.equal(out.cpu(), O.local_attn_reshape(inp, 3))
    gg = torch.randn(out.shape)
    xin = inp.cuda().requires_grad_(True)
    ops.local_attn_reshape(xin, 3).backward(gg.cuda())
    xr = inp.clone().requires_grad_(True)
    O.local_attn_reshape(xr, 3).backward(gg)
    assert torch.equal(xin.grad.cpu(), xr.grad)


def test_sampling_ops():
    ops = _ops()
    g = torch.Generator().manual_seed(7)
    x = torch.randn(2, 32, 16, 16, generator=g)
    grid = torch.rand(2, 16, 16, 2, generator=g) * 2.4 - 1.2
    grid[0, :4] = -2.0                                        # the -2 sentinel of utils/nmr.py:884
    xr = x.clone().requires_grad_(True)
    yr = F.grid_sample(xr, grid, mode='bilinear', padding_mode='zeros', align_corners=False)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    xd = nhwc_cuda(x).requires_grad_(True)
    y = ops.grid_sample(xd, grid.cuda())
    y.backward(nhwc_cuda(gy))
    assert rel_err(nchw_cpu(y), yr) < 1e-5
    assert rel_err(nchw_cpu(xd.grad), xr.grad) < 1e-5
    T = torch.rand(2, 64, 64, 2, generator=g) * 2 - 1
    for h in (32, 16, 8):
        tr = F.interpolate(T.permute(0, 3, 1, 2), size=(h, h), mode='bilinear', align_corners=True).permute(0, 2, 3, 1)
        td = ops.resize_bilinear_ac(T.cuda(), h, h)
        assert rel_err(td, tr) < 2e-5
        from oracle import hogan_oracle as O
        assert rel_err(ops.attn_flow(td), O.attn_flow(T, h)) < 2e-5
    seg = torch.randn(2, 12, 64, 64, generator=g)
    for h in (32, 16, 8):
        sr = F.interpolate(seg, size=(h, h), mode='nearest')
        assert torch.equal(nchw_cpu(ops.resize_nearest(nhwc_cuda(seg), h, h)), sr)


def test_pointwise_and_losses():
    ops = _ops()
    g = torch.Generator().manual_seed(8)
    B, H, W = 2, 16, 16
    bg, obj, hand = [torch.randn(B, 3, H, W, generator=g) for _ in range(3)]
    mbg, mh = [torch.rand(B, 1, H, W, generator=g) for _ in range(2)]
    refs = [t.clone().requires_grad_(True) for t in (bg, obj, hand, mbg, mh)]
    r = refs
    img_r = r[3] * r[0] + (1 - r[3]) * (r[1] * r[4] + r[2] * (1 - r[4]))
    devs = [nhwc_cuda(t).requires_grad_(True) for t in (bg, obj, hand, mbg, mh)]
    img = ops.compose(*devs)
    real = torch.rand(B, 3, H, W, generator=g) * 2 - 1
    tgt_m = (torch.rand(B, 1, H, W, generator=g) > 0.5).float()
    loss_r = F.l1_loss(img_r, real) * 10 + F.binary_cross_entropy(r[3], tgt_m) * 1.0 + F.mse_loss(r[4], tgt_m) * 0.5 \
        + torch.mean((r[4] - 1) ** 2) * 2.0 \
        + ((r[3][:, :, :, :-1] - r[3][:, :, :, 1:]).abs().mean() + (r[3][:, :, :-1] - r[3][:, :, 1:]).abs().mean()) * 3.0
    loss_r.backward()
    loss = ops.l1_loss(img, nhwc_cuda(real), 10.0) + ops.bce_loss(devs[3], nhwc_cuda(tgt_m), 1.0) \
        + ops.mse_loss(devs[4], nhwc_cuda(tgt_m), 0.5) + ops.lsgan_loss(devs[4], 1.0, 2.0) + ops.tv_loss(devs[3], 3.0)
    loss.backward()
    assert abs(loss.item() - loss_r.item()) < 1e-4 * abs(loss_r.item())
    assert rel_err(nchw_cpu(img), img_r) < 1e-6
    for d, rr in zip(devs, refs):
        assert rel_err(nchw_cpu(d.grad), rr.grad) < 1e-4
    x = torch.randn(2, 8, 6, 6, generator=g)
    assert abs(ops.mean(nhwc_cuda(x)).item() - x.mean().item()) < 1e-6
    # maxpool
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool2d(xr, 2, 2)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    xd = nhwc_cuda(x).requires_grad_(True)
    y = ops.maxpool2(xd)
    y.backward(nhwc_cuda(gy))
    assert torch.equal(nchw_cpu(y), yr) and torch.equal(nchw_cpu(xd.grad), xr.grad)
    # layout + cat
    assert torch.equal(ops.nhwc_to_nchw(ops.nchw_to_nhwc(x.cuda())).cpu(), x)
    a, b = torch.randn(2, 5, 4, 4, generator=g), torch.randn(2, 7, 4, 4, generator=g)
    c = ops.cat_channels([nhwc_cuda(a), nhwc_cuda(b)])
    assert torch.equal(nchw_cpu(c), torch.cat([a, b], 1))
    # the discriminator input (3 image + 16 condition channels; 24 for DexYCB) and other odd / mixed widths, pixel counts that
    # are not a multiple of the 256-pixel workgroup; wide rows take the element-wise kernel
    for c1, c2, hw in ((3, 16, 33), (3, 21, 32), (13, 9, 17), (4, 9, 16), (64, 37, 8)):
        a, b = torch.randn(3, c1, hw, hw, generator=g), torch.randn(3, c2, hw, hw, generator=g)
        c = ops.cat_channels([nhwc_cuda(a), nhwc_cuda(b)])
        assert torch.equal(nchw_cpu(c), torch.cat([a, b], 1)), (c1, c2, hw)


def test_loss_slots_compose_an_objective_like_torch_scalars():
    """ops.LossSlots: terms added into device slots by the loss kernels, one launch for the total, unit gradients -- against the
    same objective composed from the stand-alone losses with torch scalar arithmetic (trainer.py:448-457)."""
    ops = _ops()
    g = torch.Generator().manual_seed(21)
    a, b = torch.randn(2, 16, 16, 8, generator=g).cuda(), torch.randn(2, 16, 16, 8, generator=g).cuda()
    m, t = torch.rand(2, 16, 16, 1, generator=g).cuda(), (torch.rand(2, 16, 16, 1, generator=g) > 0.5).float().cuda()
    d = torch.randn(2, 6, 6, 1, generator=g).cuda()
    res = []
    for slots in (None, ops.LossSlots(['x', 'y', 'z', 'unused'], a.device, extra=['mean_d'])):
        ar, mr, dr = (v.clone().requires_grad_(True) for v in (a, m, d))
        into = (lambda n: slots.term(n)) if slots is not None else (lambda n: None)
        if slots is not None:
            slots.buf.fill_(7.0)           # begin() must clear whatever the previous step left
            slots.begin()
        terms = [ops.l1_loss(ar, b, 2.0, into=into('x')), ops.l1_loss(ar * 1.0, b, 0.5, into=into('x')),     # two terms, one slot
                 ops.bce_loss(mr, t, 3.0, into=into('y')), ops.tv_loss(mr * 1.0, 0.7, into=into('y')),
                 ops.lsgan_loss(dr, -1.0, 0.25, into=into('z'))]
        with torch.no_grad():
            mean_d = ops.mean(d, into=into('mean_d'))
        total = slots.total(*terms) if slots is not None else sum(terms)
        total.backward()
        torch.cuda.synchronize()
        vals = ([float(slots.value(n)) for n in ('x', 'y', 'z')] if slots is not None
                else [float(terms[0] + terms[1]), float(terms[2] + terms[3]), float(terms[4])])
        res.append((float(total), vals, float(mean_d), ar.grad.clone(), mr.grad.clone(), dr.grad.clone()))
    (t0, v0, m0, *g0), (t1, v1, m1, *g1) = res
    assert abs(t0 - t1) <= 1e-6 * abs(t0) and abs(m0 - m1) <= 1e-6 * abs(m0) + 1e-8
    for x, y in zip(v0, v1):
        assert abs(x - y) <= 1e-6 * abs(x)
    for x, y in zip(g0, g1):
        assert torch.equal(x, y)           # the same pre-scaled gradients; the unit factor is exact


def test_fused_adam_matches_torch():
    from hoig_amd.nn import ParamTree, FusedAdam
    shapes = {'a.weight': (8, 4, 3, 3), 'a.bias': (8,), 'b.weight': (5,), 'b.bias': (5,)}
    tree = ParamTree(shapes, torch.device('cuda'))
    g = torch.Generator().manual_seed(9)
    ref = {k: torch.randn(v, generator=g) for k, v in shapes.items()}
    tree.load_state_dict(ref)
    params = [v.clone().requires_grad_(True) for v in ref.values()]
    opt_r = torch.optim.Adam(params, lr=2e-4, betas=(0.5, 0.999))
    opt = FusedAdam(tree, lr=2e-4, betas=(0.5, 0.999))
    for step in range(3):
        grads = [torch.randn(p.shape, generator=g) for p in params]
        for p, gr in zip(params, grads):
            p.grad = gr.clone()
        opt_r.step()
        with torch.no_grad():
            for p, gr in zip(tree.P.values(), grads):
                p.grad.copy_(gr.cuda())
        opt.step()
    sd = tree.state_dict()
    for k, p in zip(shapes, params):
        assert rel_err(sd[k], p) < 1e-6
    osd, osr = opt.state_dict(), opt_r.state_dict()
    assert set(osd['state'].keys()) == set(osr['state'].keys())
    for i in osr['state']:
        assert rel_err(osd['state'][i]['exp_avg'], osr['state'][i]['exp_avg']) < 1e-6
        assert rel_err(osd['state'][i]['exp_avg_sq'], osr['state'][i]['exp_avg_sq']) < 1e-6
        assert float(osd['state'][i]['step']) == float(osr['state'][i]['step'])


def test_fused_adam_with_planes_matches_the_two_launches():
    """hoig_adam_pack_step (FusedAdam.fuse_planes): the optimiser step that also writes the operand planes of the updated conv weights,
    against hoig_adam_step_dev followed by hoig_pack_conv_weights_bf16_all: parameters, both moments and all four planes bit for bit,
    over a buffer with packed weights (forward-only, dgrad-only and both kinds of planes), unpacked weights and biases in between."""
    from hoig_amd import _lib as L
    from hoig_amd.nn import ParamTree, FusedAdam
    shapes = {'s.weight': (64, 8, 7, 7), 's.bias': (64,), 'a.weight': (128, 64, 3, 3), 'a.bias': (128,), 'b.weight': (32, 96, 3, 3),
              'h.weight': (3, 64, 7, 7), 'h.bias': (3,), 'c.weight': (64, 32, 5, 5), 'd.weight': (96, 160, 1, 1), 'n.weight': (96,)}
    g = torch.Generator().manual_seed(17)
    ref = {k: torch.randn(v, generator=g) * 0.05 for k, v in shapes.items()}
    grads = [[torch.randn(v, generator=g) for v in shapes.values()] for _ in range(3)]
    res = []
    if True:
        for fused in (0, 1):
            tree = ParamTree(shapes, torch.device('cuda'))
            tree.load_state_dict(ref)
            tree._ensure_plane_bufs()
            tree._refresh_planes()
            assert tree.fused_step_tables() is not None
            opt = FusedAdam(tree, lr=2e-4, betas=(0.5, 0.999))
            opt.fuse_planes = bool(fused)
            for gs in grads:
                with torch.no_grad():
                    for p_, gr in zip(tree.P.values(), gs):
                        p_.grad.copy_(gr.cuda())
                opt.step(grad_scale=0.5)
                if fused:
                    assert tree._plane_version == tree.version          # nothing left for the next forward to pack
                tree._refresh_planes()
            torch.cuda.synchronize()
            flags = dict(tree._plane_flags)
            res.append((tree.flat.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(), [b.clone() for b in tree._plane_bufs], flags, tree))
    (f0, m0, v0, b0, flags, tree), (f1, m1, v1, b1, _, _) = res
    assert torch.equal(f0, f1) and torch.equal(m0, m1) and torch.equal(v0, v1)
    assert sorted(flags.values()) == [1, 2, 3, 3]                            # c: forward planes only; b: data-gradient planes only
    for name, shp in shapes.items():
        off, n = tree._offsets[name], int(torch.tensor(shp).prod())
        fl = flags.get(off, 0)
        for k in range(4):
            if fl & (1 if k < 2 else 2):
                assert torch.equal(b0[k][off:off + n], b1[k][off:off + n]), (name, k)


BF16_CASES = [
    # B, Ci, Co, H, k, stride, pad, transposed
    (2, 64, 64, 16, 3, 1, 1, False),
    (2, 128, 256, 16, 3, 2, 1, False),
    (1, 512, 512, 8, 3, 1, 1, False),
    (8, 512, 512, 32, 3, 1, 1, False),
    (2, 64, 64, 64, 3, 1, 1, False),
    (3, 32, 192, 32, 3, 1, 1, False),
    (2, 64, 128, 16, 4, 2, 1, False),
    (2, 256, 256, 9, 4, 1, 1, False),
    (3, 96, 160, 10, 3, 1, 1, False),
    (2, 128, 64, 16, 3, 2, 1, True),
    (2, 64, 128, 64, 3, 2, 1, False),      # stride-2 conv on the parity-phase halo kernel (gather fwd, scatter dgrad)
    (2, 128, 64, 32, 3, 2, 1, True),       # ConvTranspose2d on it (scatter fwd, gather dgrad; 64-channel tiles)
    (1, 256, 128, 32, 3, 2, 1, True),
    (1, 512, 256, 8, 3, 2, 1, True),
    (8, 64, 128, 256, 3, 2, 1, False),     # ... at the step's own sizes (thousands of tiles): gather fwd, scatter dgrad, 128-column tiles
    (8, 128, 64, 64, 3, 2, 1, True),       # ... scatter fwd, gather dgrad, 64-column tiles
]


# (forward, data gradient, weight gradient) max-norm relative bounds per arithmetic mode.  bf16x3: fp32-class.  f16x2: the
# weight operand is ONE fp16 (forward, 2^-12 per weight, random: ~1.4e-4 rms of the output) / ONE bf16 (backward: 2^-9 per
# weight or per x element) -- bounds = ~4x the rms estimate, and a floor that proves the two-term kernels really ran.
PREC_BOUNDS = {'bf16x3': (3e-4, 3e-4, 3e-4), 'f16x2': (1e-3, 8e-3, 8e-3)}


@pytest.mark.parametrize('mode', ['bf16x3', 'f16x2'])
@pytest.mark.parametrize('B,Ci,Co,H,k,stride,pad,transposed', BF16_CASES)
def test_conv_bf16x3_fast_path(B, Ci, Co, H, k, stride, pad, transposed, mode):
    """The split 16-bit MFMA paths.  bf16x3 (3 MFMAs per k-step) must stay fp32-class: bound 3e-4 relative, inside
    north_star's 1e-3; f16x2 (2 MFMAs: gathered operand split, weights / x single) within its own error model."""
    ops = _ops()
    from hoig_amd import _lib as L
    prec = {'bf16x3': L.PREC_BF16X3, 'f16x2': L.PREC_F16X2}[mode]
    bf, bd, bw = PREC_BOUNDS[mode]
    ops.set_precision(mode)                     # (the backward launches take the module-level backward precision)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, Ci, H, H, generator=g)
    bias = torch.randn(Co, generator=g)
    if transposed:
        w = torch.randn(Ci, Co, k, k, generator=g) * 0.05
        xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        yr = F.conv_transpose2d(xr, wr, None, stride=2, padding=1, output_padding=1)
    else:
        w = torch.randn(Co, Ci, k, k, generator=g) * 0.05
        xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        yr = F.conv2d(xr, wr, bias, stride=stride, padding=pad)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    xd = nhwc_cuda(x).requires_grad_(True)
    wd = ops.pack_weight(w.cuda(), transposed=transposed).requires_grad_(True)
    try:
        if transposed:
            y = ops.conv_transpose2d(xd, wd, prec=prec)
        else:
            y = ops.conv2d(xd, wd, bias.cuda(), stride, pad, prec=prec)
        y.backward(nhwc_cuda(gy))
    finally:
        ops.set_precision('f32')
    ef, ed, ew = rel_err(nchw_cpu(y), yr), rel_err(nchw_cpu(xd.grad), xr.grad), rel_err(wd.grad, wr.grad)
    print('%s: fwd %.2e dgrad %.2e wgrad %.2e' % (mode, ef, ed, ew))
    assert ef < bf and ed < bd and ew < bw
    if mode == 'f16x2' and Ci > 32 and Co > 32:          # (<= 32 gathered channels: that launch runs on the exact-fp32 kernels)
        assert ef > 2e-5 and ed > 1e-4, 'the two-term kernels did not run (error is at the three-term level)'


@pytest.mark.parametrize('mode', ['f32', 'bf16x3', 'f16x2'])
@pytest.mark.parametrize('B,C,Co,H,W,k,stride', [(8, 512, 512, 32, 32, 3, 1),      # the residual blocks: addend epilogue of the 3x3 kernel
                                                 (8, 128, 64, 64, 64, 3, 1),       # ... its 64-column variant
                                                 (8, 64, 64, 64, 64, 1, 1),        # 1x1 on the halo kernel
                                                 (2, 128, 128, 16, 32, 3, 1),      # few tiles: split-K kernel, separate add
                                                 (4, 64, 128, 64, 64, 3, 2),       # encoder level: the stride-2 kernel's epilogue
                                                 (2, 128, 128, 16, 16, 3, 2),      # no such epilogue: separate add
                                                 (2, 16, 64, 8, 8, 3, 1)])         # thin input
def test_conv_fork_adds_the_other_consumers_gradient(B, C, Co, H, W, k, stride, mode):
    """ops.conv2d_fork(x, w) -> (conv(x), x'): the gradient of x must be dgrad + (gradient that arrived through x'), whichever
    kernel adds it -- and equal to what autograd computes when it sums the two consumers itself (generator.py:29-32)."""
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(Co, C, k, k, generator=g) * 0.05
    ops.set_precision(mode)
    try:
        res = []
        for fork in (False, True):
            xd = nhwc_cuda(x).requires_grad_(True)
            wd = ops.pack_weight(w.cuda()).requires_grad_(True)
            x1 = ops.add(xd, xd)                               # a non-leaf (leaves accumulate into .grad anyway)
            if fork:
                y, x2 = ops.conv2d_fork(x1, wd, None, stride, k // 2)
                assert x2.data_ptr() == x1.data_ptr()
            else:
                y, x2 = ops.conv2d(x1, wd, None, stride, k // 2), x1
            gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(6)).cuda()
            gx = torch.randn(x1.shape, generator=torch.Generator().manual_seed(7)).cuda()
            ((y * gy).sum() + (x2 * gx).sum()).backward()
            torch.cuda.synchronize()
            res.append((y.detach().clone(), xd.grad.clone(), wd.grad.clone()))
        # only the pass-through output used: the gradient passes through untouched
        xd = nhwc_cuda(x).requires_grad_(True)
        _, x2 = ops.conv2d_fork(ops.add(xd, xd), ops.pack_weight(w.cuda()).requires_grad_(True), None, stride, k // 2)
        (x2 * gx).sum().backward()
        assert torch.equal(xd.grad, 2 * gx)
    finally:
        ops.set_precision('f32')
    (y0, dx0, dw0), (y1, dx1, dw1) = res
    assert float((y0 - y1).abs().max()) <= 1e-5 * float(y0.abs().max())         # (split-K launches: fp32 atomics)
    assert float((dx0 - dx1).abs().max()) <= 1e-5 * float(dx0.abs().max())      # same products, one rounding apart
    assert float((dw0 - dw1).abs().max()) <= 1e-4 * float(dw0.abs().max())      # (fp32 atomics)
    # the fork must not have dropped either part: dx = 2 * (dgrad + gx)
    assert rel_err(dx1 - 2 * gx, dx0 - 2 * gx) < 1e-4 and float((dx1 - 2 * gx).abs().max()) > 0


@pytest.mark.parametrize('kind,B,Ci,Co,H,W', [('s1', 8, 64, 128, 64, 64), ('s1', 4, 128, 64, 64, 128), ('s2', 4, 64, 128, 128, 128),
                                              ('s2', 2, 128, 256, 128, 64), ('convT', 4, 128, 64, 64, 64), ('convT', 2, 256, 128, 32, 64),
                                              ('cat2', 4, 64, 64, 64, 128), ('small', 8, 512, 512, 32, 32), ('thin', 2, 8, 64, 64, 64)])
def test_instance_norm_statistics_from_the_convolution_epilogue(kind, B, Ci, Co, H, W):
    """conv -> IN(+ReLU) where the convolution's epilogue leaves the per-image channel sums for the norm (SURVEY 7.4;
    hoig_conv2d_fwd_packed_stats / hoig_inorm_stats_from_sums): same outputs and gradients as the reference ops, the hand-off really
    happens on the layers that have the epilogue, and never leaves the workspace dirty."""
    ops = _ops()
    from hoig_amd import _lib as L
    g = torch.Generator().manual_seed(31)
    x = torch.randn(B, Ci, H, W, generator=g) + 0.5
    x2 = torch.randn(B, Ci, H, W, generator=g) if kind == 'cat2' else None
    ci_w = 2 * Ci if kind == 'cat2' else Ci
    w = torch.randn(*((ci_w, Co, 3, 3) if kind == 'convT' else (Co, ci_w, 3, 3)), generator=g) * 0.05
    aw, ab = torch.randn(Co, generator=g), torch.randn(Co, generator=g)
    xr, wr, awr, abr = (t.clone().requires_grad_(True) for t in (x, w, aw, ab))
    x2r = x2.clone().requires_grad_(True) if x2 is not None else None
    if kind == 'convT':
        hr = F.conv_transpose2d(xr, wr, None, stride=2, padding=1, output_padding=1)
    else:
        hr = F.conv2d(torch.cat([xr, x2r], 1) if x2 is not None else xr, wr, None, stride=2 if kind == 's2' else 1, padding=1)
    yr = F.instance_norm(hr, weight=awr, bias=abr, eps=1e-5)     # (no activation: a ReLU mask one rounding away from the reference's
    gy = torch.randn(yr.shape, generator=g)                      # would put isolated large errors into the max-norm of the gradients)
    yr.backward(gy)
    ops.set_precision('bf16x3')
    try:
        xd = nhwc_cuda(x).requires_grad_(True)
        x2d = nhwc_cuda(x2).requires_grad_(True) if x2 is not None else None
        wd = ops.pack_weight(w.cuda(), transposed=kind == 'convT').requires_grad_(True)
        awd, abd = aw.cuda().requires_grad_(True), ab.cuda().requires_grad_(True)
        ops._stats_pending.clear()
        if kind == 'convT':
            h = ops.conv_transpose2d(xd, wd, norm_next=True)
        elif kind == 'cat2':
            h = ops.conv2d_cat2(xd, x2d, wd, norm_next=True)
        else:
            h = ops.conv2d(xd, wd, None, 2 if kind == 's2' else 1, 1, dead_bias=True)
        offered = len(ops._stats_pending) == 1
        assert offered == (kind != 'small')       # <= 1024-pixel maps keep the norm's own statistics (thin-input layers: hoig_conv2d_fwd_stats, round 6)
        y = ops.instance_norm(h, awd, abd)
        assert not ops._stats_pending
        y.backward(nhwc_cuda(gy))
        torch.cuda.synchronize()
        for ws in ops._norm_ws.values():                         # accumulators are left zero by whoever consumed them
            assert float(ws[:1 << 18].abs().max()) == 0.0
        # sums nobody consumes (the norm is never called) are cleared by the next user of the workspace
        if offered:
            ops.conv2d(xd, wd, None, 2 if kind == 's2' else 1, 1, dead_bias=True) if kind in ('s1', 's2') else None
            z = ops.instance_norm(nhwc_cuda(torch.randn(2, 64, 48, 48, generator=g)))
            torch.cuda.synchronize()
            assert not ops._stats_pending
            assert float(z.mean(dim=(1, 2)).abs().max()) < 1e-4
    finally:
        ops.set_precision('f32')
    assert rel_err(nchw_cpu(y), yr) < 3e-4
    assert rel_err(nchw_cpu(xd.grad), xr.grad) < 3e-3           # (bf16x3 products)
    assert rel_err(wd.grad, wr.grad) < 3e-3
    assert rel_err(awd.grad, awr.grad) < 3e-3 and rel_err(abd.grad, abr.grad) < 3e-3


THIN_CASES = [
    # B, Ci, Co, H, W, k, bias, act       (stride 1, 'same' padding)
    (2, 3, 64, 32, 32, 7, False, 'none'),      # the 7x7 stems (generator.py:100,153)
    (1, 8, 64, 8, 64, 7, True, 'none'),
    (2, 3, 128, 32, 32, 3, True, 'relu'),      # SPADE's shared conv over the 3-channel condition map (spade.py:18)
    (1, 12, 128, 16, 32, 3, True, 'relu'),     # ... over obj_model's 12-channel one
    (1, 3, 64, 64, 64, 3, True, 'relu'),       # VGG19's first layer
    (2, 3, 64, 32, 64, 3, False, 'none'),      # ... whose data gradient (3 outputs) the perceptual loss needs: thin_out_kernel
    (1, 64, 5, 16, 32, 7, False, 'tanh'),      # five output channels (the fused heads' shape)
    (2, 64, 1, 32, 32, 7, True, 'sigmoid'),    # the mask heads (generator.py:219-235)
    (2, 64, 3, 16, 64, 7, True, 'tanh'),       # the image heads
    (1, 128, 1, 32, 32, 7, False, 'none'),     # (two 64-channel groups)
]


@pytest.mark.parametrize('mode', ['bf16x3', 'f16x2'])
@pytest.mark.parametrize('B,Ci,Co,H,W,k,bias,act', THIN_CASES)
def test_conv_thin_channels_on_mfma(B, Ci, Co, H, W, k, bias, act, mode):
    """conv_thin.hip: convolutions with a few channels on one side, taps standing in for the missing channels on the MFMA --
    forward + weight gradient of thin-input layers, data + weight gradient of thin-output layers (their forward stays on the
    fp32 VALU kernel) -- against torch fp32, in the three-term and the two-term arithmetic."""
    ops = _ops()
    from hoig_amd import _lib as L
    prec = {'bf16x3': L.PREC_BF16X3, 'f16x2': L.PREC_F16X2}[mode]
    bf, bd, bw = PREC_BOUNDS[mode]
    acts = {'none': (L.ACT_NONE, lambda t: t), 'relu': (L.ACT_RELU, torch.relu), 'tanh': (L.ACT_TANH, torch.tanh),
            'sigmoid': (L.ACT_SIGMOID, torch.sigmoid)}
    code, fn = acts[act]
    ops.set_precision(mode)
    g = torch.Generator().manual_seed(21)
    x = torch.randn(B, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, k, k, generator=g) * 0.05
    b = torch.randn(Co, generator=g) if bias else None
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = fn(F.conv2d(xr, wr, b, stride=1, padding=k // 2))
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    xd = nhwc_cuda(x).requires_grad_(True)
    wd = ops.pack_weight(w.cuda()).requires_grad_(True)
    try:
        y = ops.conv2d(xd, wd, b.cuda() if bias else None, 1, k // 2, act=code, prec=prec)
        y.backward(nhwc_cuda(gy))
        torch.cuda.synchronize()
    finally:
        ops.set_precision('f32')
    ef, ed, ew = rel_err(nchw_cpu(y), yr), rel_err(nchw_cpu(xd.grad), xr.grad), rel_err(wd.grad, wr.grad)
    print('thin %s: fwd %.2e dgrad %.2e wgrad %.2e' % (mode, ef, ed, ew))
    assert ef < bf
    # behind a ReLU a forward error of 1e-4 flips the mask of the outputs that close to zero, and every flip moves a gradient
    # element by ~1 %: the gradients of those cases are only held to that
    relu = act == 'relu'
    if not (relu and mode == 'f16x2'):           # (two-term forward: ~10x more flipped masks; the smooth cases carry the check)
        assert ew < (5e-2 if relu else bw) and ed < (5e-2 if relu else bd)
    if mode == 'f16x2' and not relu:          # the two-term MFMA kernels really ran (the fp32 VALU / generic kernels sit at 1e-6)
        assert ew > 2e-5
        if Ci <= 16:
            assert ef > 1e-5
        else:
            assert ed > 2e-5


@pytest.mark.parametrize('mode', ['bf16x3', 'f16x2'])
@pytest.mark.parametrize('B,Ci,Co,Hi,Wi,bias', [(2, 128, 128, 40, 40, False), (3, 64, 128, 36, 36, True), (1, 64, 128, 12, 136, True),
                                                (1, 96, 256, 23, 72, False)])
def test_conv_valid_5x5_odd_widths(B, Ci, Co, Hi, Wi, bias, mode):
    """The attention's VALID 5x5 convolutions (extract_attn.py:18 over replicate-padded maps: 40 -> 36 -> 32, 72 -> 68, 136 ->
    132 pixels wide): forward, data gradient (a full correlation) and weight gradient on widths that are not multiples of 32,
    with and without the split-K atomic epilogue."""
    ops = _ops()
    from hoig_amd import _lib as L
    prec = {'bf16x3': L.PREC_BF16X3, 'f16x2': L.PREC_F16X2}[mode]
    bf, bd, bw = PREC_BOUNDS[mode]
    ops.set_precision(mode)
    g = torch.Generator().manual_seed(31)
    x = torch.randn(B, Ci, Hi, Wi, generator=g)
    w = torch.randn(Co, Ci, 5, 5, generator=g) * 0.03
    b = torch.randn(Co, generator=g) if bias else None
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, b)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    xd = nhwc_cuda(x).requires_grad_(True)
    wd = ops.pack_weight(w.cuda()).requires_grad_(True)
    try:
        y = ops.conv2d(xd, wd, b.cuda() if bias else None, 1, 0, prec=prec)
        y.backward(nhwc_cuda(gy))
        torch.cuda.synchronize()
    finally:
        ops.set_precision('f32')
    ef, ed, ew = rel_err(nchw_cpu(y), yr), rel_err(nchw_cpu(xd.grad), xr.grad), rel_err(wd.grad, wr.grad)
    print('valid 5x5 %s: fwd %.2e dgrad %.2e wgrad %.2e' % (mode, ef, ed, ew))
    assert ef < bf and ed < bd and ew < bw


def test_batched_weight_split_matches_per_weight_split():
    """hoig_pack_conv_weights_bf16_all (one launch over a network's flat buffer, LDS-tiled transpose for the data-gradient
    planes) must produce exactly the planes hoig_pack_conv_weight_bf16 produces weight by weight, including for the fused
    SPADE gamma|beta view and a ConvTranspose weight."""
    import ctypes
    from hoig_amd import _lib as L
    from hoig_amd.nn import ParamTree
    from hoig_amd.ops import _p, _st
    shapes = [('a.weight', (64, 96, 3, 3)), ('a.bias', (64,)),
              ('s.mlp_gamma.weight', (48, 128, 3, 3)), ('s.mlp_gamma.bias', (48,)),
              ('s.mlp_beta.weight', (48, 128, 3, 3)), ('s.mlp_beta.bias', (48,)),
              ('up.weight', (128, 64, 3, 3)), ('head.weight', (3, 64, 7, 7)), ('one.weight', (128, 800, 1, 1))]
    tree = ParamTree(shapes, torch.device('cuda'), transposed_names=('up.weight',))
    torch.manual_seed(3)
    tree.flat.normal_()
    tree.version += 1
    checked = 0
    for name, w, transposed in [('a.weight', tree.P['a.weight'], False), ('gb', tree.F['s.mlp_gb.weight'], False),
                                ('up.weight', tree.P['up.weight'], True), ('one.weight', tree.P['one.weight'], False)]:
        ci, co = (w.shape[0], w.shape[1]) if transposed else (w.shape[1], w.shape[0])
        for for_dgrad in (False, True):
            got = tree.packed_planes(w, for_dgrad)
            eligible = co % 32 == 0 and ci % 32 == 0 and (ci > 32 if for_dgrad else co > 32)
            assert (got is not None) == eligible, (name, for_dgrad)
            if got is None:
                continue
            hi = torch.empty(w.numel(), dtype=torch.int16, device='cuda')
            lo = torch.empty_like(hi)
            L.call('hoig_pack_conv_weight_bf16', _p(w), co, w.shape[2] * w.shape[3], ci, 1 if for_dgrad else 0,
                   _p(hi), _p(lo), _st())
            assert torch.equal(got[0], hi) and torch.equal(got[1], lo), (name, for_dgrad)
            checked += 1
    assert checked >= 6
    assert tree.packed_planes(tree.P['head.weight'], False) is None


RECT_CASES = [
    # Ci, Co, H, W, k, stride, pad, transposed
    (64, 128, 16, 64, 3, 1, 1, False),      # halo conv, 8x32 tiles: tiles_x != tiles_y
    (128, 128, 8, 96, 3, 1, 1, False),
    (64, 128, 16, 128, 3, 2, 1, False),     # parity-phase stride-2 kernel (gather fwd / scatter dgrad)
    (128, 64, 8, 64, 3, 2, 1, True),        # ConvTranspose2d on it
    (64, 3, 20, 72, 7, 1, 3, False),        # direct head kernels, ragged 16x64 tiles
    (3, 64, 20, 40, 7, 1, 3, False),        # direct stem forward, ragged tiles
    (8, 64, 12, 72, 7, 1, 3, False),        # two 4-channel groups (accumulating second pass)
]


@pytest.mark.parametrize('Ci,Co,H,W,k,stride,pad,transposed', RECT_CASES)
def test_conv_rectangular_images(Ci, Co, H, W, k, stride, pad, transposed):
    """Non-square feature maps and sizes that do not fill the last tile, through the fast paths (the model itself only
    ever sees squares): forward, data gradient and weight gradient against torch fp32."""
    ops = _ops()
    from hoig_amd import _lib as L
    g = torch.Generator().manual_seed(23)
    B = 2
    x = torch.randn(B, Ci, H, W, generator=g)
    if transposed:
        w = torch.randn(Ci, Co, k, k, generator=g) * 0.05
        xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        yr = F.conv_transpose2d(xr, wr, None, stride=2, padding=1, output_padding=1)
    else:
        w = torch.randn(Co, Ci, k, k, generator=g) * 0.05
        xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        yr = F.conv2d(xr, wr, None, stride=stride, padding=pad)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    xd = nhwc_cuda(x).requires_grad_(True)
    wd = ops.pack_weight(w.cuda(), transposed=transposed).requires_grad_(True)
    if transposed:
        y = ops.conv_transpose2d(xd, wd, prec=L.PREC_BF16X3)
    else:
        y = ops.conv2d(xd, wd, None, stride, pad, prec=L.PREC_BF16X3)
    y.backward(nhwc_cuda(gy))
    assert rel_err(nchw_cpu(y), yr) < 3e-4
    assert rel_err(nchw_cpu(xd.grad), xr.grad) < 3e-4
    assert rel_err(wd.grad, wr.grad) < 3e-4


@pytest.mark.parametrize('B,C1,C2,Co,H,W', [(4, 64, 64, 64, 64, 128), (8, 128, 128, 128, 32, 64), (4, 256, 256, 256, 32, 128),
                                            (8, 64, 192, 128, 16, 256), (2, 96, 64, 64, 8, 32)])
def test_conv_over_two_tensors_equals_conv_of_cat(B, C1, C2, Co, H, W):
    """ops.conv2d_cat2 (the decoder's skip convolution reading [skip | up] from two tensors) against conv2d(cat_channels):
    forward, data and weight gradients equal up to summation order; the small cases fall back to the concatenating path."""
    ops = _ops()
    from hoig_amd import _lib as L
    g = torch.Generator().manual_seed(31)
    x1 = torch.randn(B, H, W, C1, generator=g).cuda()
    x2 = torch.randn(B, H, W, C2, generator=g).cuda()
    w = ops.pack_weight((torch.randn(Co, C1 + C2, 3, 3, generator=g) * 0.05).cuda())
    gy = torch.randn(B, H, W, Co, generator=g).cuda()
    a1, a2, wa = x1.clone().requires_grad_(True), x2.clone().requires_grad_(True), w.clone().requires_grad_(True)
    b1, b2, wb = x1.clone().requires_grad_(True), x2.clone().requires_grad_(True), w.clone().requires_grad_(True)
    ya = ops.conv2d_cat2(a1, a2, wa, prec=L.PREC_BF16X3)
    yb = ops.conv2d(ops.cat_channels([b1, b2]), wb, None, 1, 1, prec=L.PREC_BF16X3)
    ya.backward(gy)
    yb.backward(gy)
    # (same kernels and k order, so normally bit-identical; the split-K fallback of few-tile shapes sums with atomics)
    assert rel_err(ya, yb) < 1e-6
    assert rel_err(a1.grad, b1.grad) < 1e-6 and rel_err(a2.grad, b2.grad) < 1e-6
    assert rel_err(wa.grad, wb.grad) < 1e-5


@pytest.mark.parametrize('B,C1,C2,Co,H,W', [
    (8, 512, 0, 256, 16, 64),        # two tile columns
    (8, 512, 0, 512, 20, 32),        # five tile rows: ten tiles per workgroup
    (8, 256, 256, 256, 32, 32),      # the decoder's skip convolution: x is [skip | up]
    (8, 512, 0, 512, 18, 32)])       # H % 4 != 0: stays on 2-row tiles
def test_wgrad_four_row_tiles(B, C1, C2, Co, H, W):
    """Weight (and bias) gradient of 3x3 stride-1 layers in the two-term arithmetic on wgrad_halo_bf16_kernel<.., TH = 4> (4 x 32
    pixel tiles, dynamic LDS; launch_wgrad_halo picks it when every workgroup keeps >= 8 tiles) against torch fp32 at the f16x2
    bound, on maps whose tile grid is not square / not a power of two, and over a two-tensor input."""
    ops = _ops()
    from hoig_amd import _lib as L
    g = torch.Generator().manual_seed(41)
    Ci = C1 + C2
    x = torch.randn(B, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, 3, 3, generator=g) * 0.05
    bias = torch.randn(Co, generator=g)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, br, stride=1, padding=1)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    ops.set_precision('f16x2')
    try:
        wd = ops.pack_weight(w.cuda()).requires_grad_(True)
        bd = bias.cuda().requires_grad_(True)
        if C2:
            xn = nhwc_cuda(x)
            x1, x2 = xn[..., :C1].contiguous().requires_grad_(True), xn[..., C1:].contiguous().requires_grad_(True)
            y = ops.conv2d_cat2(x1, x2, wd) + bd
        else:
            xd = nhwc_cuda(x).requires_grad_(True)
            y = ops.conv2d(xd, wd, bd, 1, 1)
        y.backward(nhwc_cuda(gy))
    finally:
        ops.set_precision('f32')
    bf, bdg, bw = PREC_BOUNDS['f16x2']
    assert rel_err(nchw_cpu(y), yr) < bf
    assert rel_err(wd.grad, wr.grad) < bw
    assert rel_err(bd.grad.cpu(), br.grad) < 1e-4
    if C2:
        dx = torch.cat([x1.grad, x2.grad], -1)
    else:
        dx = xd.grad
    assert rel_err(nchw_cpu(dx), xr.grad) < bdg


@pytest.mark.parametrize('B,Ci,Co,H,bias,min_tiles,runs_f6', [
    (16, 512, 512, 32, False, 192, True),       # the dominant launch of the step: 128-channel tiles
    (8, 512, 512, 32, False, 192, True),        # src_model / tsf_model: 64-channel tiles (twice the workgroups)
    (8, 128, 1024, 32, True, 192, True),        # SPADE gamma|beta
    (16, 64, 128, 64, False, 192, True),
    (3, 256, 256, 64, True, 192, True),
    (1, 128, 128, 32, True, 192, False),        # 8 workgroups: left to the three-term kernels ...
    (1, 128, 128, 32, True, 1, True),           # ... unless the threshold is lowered (what the parity tests do)
    (2, 64, 384, 64, False, 1, True)])          # three 128-channel tiles -> six 64-channel ones
def test_conv_f16f6_forward(B, Ci, Co, H, bias, min_tiles, runs_f6):
    """hoig_conv2d_fwd_f6 (hi*hi on fp16 + the two cross terms of the split on block-scaled fp6 MFMAs) against torch fp32:
    per layer the cross-term quantisation (4 significant bits on terms that carry 2^-11 of the product) must leave ~2^-15 of
    relative error -- bound 2e-4 -- and MORE than the three-fp16-term kernel's ~1e-6, which proves the fp6 kernel ran; the
    backward of the mode (two bf16 terms) within the f16x2 bounds."""
    ops = _ops()
    from hoig_amd import _lib as L
    g = torch.Generator().manual_seed(13)
    x = torch.randn(B, Ci, H, H, generator=g) * torch.rand(B, Ci, 1, 1, generator=g) * 3.0      # per-channel magnitudes differ
    w = torch.randn(Co, Ci, 3, 3, generator=g) * 0.05
    bv = torch.randn(Co, generator=g) if bias else None
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, bv, stride=1, padding=1)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    ops.set_precision('f16f6')
    old = ops.set_f6_min_tiles(min_tiles)
    try:
        xd = nhwc_cuda(x).requires_grad_(True)
        wd = ops.pack_weight(w.cuda()).requires_grad_(True)
        y = ops.conv2d(xd, wd, bv.cuda() if bias else None, 1, 1)
        y.backward(nhwc_cuda(gy))
    finally:
        ops.set_precision('f32')
        ops.set_f6_min_tiles(old)
    ef, ed, ew = rel_err(nchw_cpu(y), yr), rel_err(nchw_cpu(xd.grad), xr.grad), rel_err(wd.grad, wr.grad)
    print('f16f6: fwd %.2e dgrad %.2e wgrad %.2e' % (ef, ed, ew))
    assert ef < 2e-4 and ed < 8e-3 and ew < 8e-3
    if runs_f6:
        assert ef > 5e-6, 'the fp6 kernel did not run (error at the three-fp16-term level)'
    else:                       # too few 8x32 tiles to fill the chip: the launch runs as three fp16 terms
        assert ef < 5e-6


@pytest.mark.parametrize('B,C1,C2,Co,H', [(8, 64, 64, 64, 64), (4, 128, 128, 128, 64), (2, 256, 256, 256, 32)])
def test_conv_cat_f16f6_forward(B, C1, C2, Co, H):
    """The decoder's skip convolution over [skip | up] (generator.py:305-306) on the fp16 + fp6 forward kernel reading the two
    tensors directly, against torch's conv2d of the concatenation."""
    ops = _ops()
    g = torch.Generator().manual_seed(17)
    x1, x2 = torch.randn(B, C1, H, H, generator=g), torch.randn(B, C2, H, H, generator=g) * 2.0
    w = torch.randn(Co, C1 + C2, 3, 3, generator=g) * 0.05
    yr = F.conv2d(torch.cat([x1, x2], 1), w, None, stride=1, padding=1)
    ops.set_precision('f16f6')
    old = ops.set_f6_min_tiles(1)
    try:
        y = ops.conv2d_cat2(nhwc_cuda(x1), nhwc_cuda(x2), ops.pack_weight(w.cuda()))
    finally:
        ops.set_precision('f32')
        ops.set_f6_min_tiles(old)
    e = rel_err(nchw_cpu(y), yr)
    print('cat f16f6 fwd %.2e' % e)
    assert 5e-6 < e < 2e-4
