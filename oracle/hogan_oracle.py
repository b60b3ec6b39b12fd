"""CPU ORACLE for the HOGAN generator/discriminator hot path.

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this file; the product package
``hoig_amd`` never does (tests/test_host_cpu.py::test_product_never_imports_the_oracle
enforces it).

What it is: a functional, state-dict driven restatement in plain fp32 PyTorch
(CPU) of the reference's algorithm for the path SURVEY.md §8a lists -- the
reference is Python on torch ATen, so torch-CPU fp32 is the reference
arithmetic (conv/instance_norm/grid_sample/interpolate/BCE are the same ATen
operators the reference calls, torch==1.8.0 semantics per requirements.txt:169,
unchanged in torch 2.10 for the arguments used).  Every function cites the
reference file:line it follows (paths relative to /root/reference/HOIG_HOv3).

Pinning status (see DESIGN.md "Oracle"):
  * G / D / SPADE / losses / Adam step: PINNED against the reference's own
    Python executed in the build container (tests/test_oracle_vs_reference.py,
    and the committed vectors in tests/golden/ made by tests/golden/make_golden.py).
  * K1-K4 (block_extractor / local_attn_reshape CUDA kernels): the reference
    cannot run them on CPU and holds no expected tensors; pinned only by the
    three checks its manual scripts define (gradcheck recipe, zero-flow
    neighbourhood identity, the [[0,1,2],[3,4,5],[6,7,8]] read-out) plus a
    cross-check against the independent plain-C restatement oracle/block_ops.c.
    Beyond that: PARITY UNPINNED for K1-K4.
  * VGG19 perceptual loss: structure pinned, ImageNet weights unavailable
    offline -> deterministic surrogate weights on both sides.
"""
from collections import OrderedDict
import math

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------- config

GEN_VARIANTS = {                       # models/networks/__init__.py:11-25
    'generator_base': dict(spade_layers=[0, 0, 0, 0], attn_layers=[]),
    'generator_spade': dict(spade_layers=[1, 1, 0, 0], attn_layers=[]),
    'generator_spade_attn': dict(spade_layers=[1, 1, 0, 0], attn_layers=list(range(1, 10))),
    'generator_spade_attn_tiny': dict(spade_layers=[0, 0, 1, 1], attn_layers=list(range(1, 10))),
}


def make_cfg(gen_name='generator_spade_attn', dataset='hov3', repeat_num=6, conv_dim=64):
    """Channel configuration of trainer.py:258-268 (HOv3) /
    HOIG_DexYCB/models/trainer.py:261-271 (DexYCB)."""
    v = GEN_VARIANTS[gen_name]
    hov3 = dataset == 'hov3'
    return dict(gen_name=gen_name, dataset=dataset, repeat_num=repeat_num, conv_dim=conv_dim, n_down=3,
                spade_layers=list(v['spade_layers']), attn_layers=list(v['attn_layers']),
                bg_dim=8 if hov3 else 13, img_dim=3, obj_dim=3,
                img_cond_dim=3 if hov3 else 9, obj_cond_dim=12,
                armask=hov3, d_input_nc=19 if hov3 else 24, d_ndf=64, d_layers=4)


# --------------------------------------------------------------------------- parameter schema

def _spade_shapes(out, prefix, norm_nc, label_nc):
    out[prefix + '.mlp_shared.0.weight'] = (128, label_nc, 3, 3)          # spade.py:16-19
    out[prefix + '.mlp_shared.0.bias'] = (128,)
    out[prefix + '.mlp_gamma.weight'] = (norm_nc, 128, 3, 3)              # spade.py:21
    out[prefix + '.mlp_gamma.bias'] = (norm_nc,)
    out[prefix + '.mlp_beta.weight'] = (norm_nc, 128, 3, 3)               # spade.py:22
    out[prefix + '.mlp_beta.bias'] = (norm_nc,)


def _in_shapes(out, prefix, c):
    out[prefix + '.weight'] = (c,)
    out[prefix + '.bias'] = (c,)


def _resblock_shapes(out, prefix, c):                                     # generator.py:17-22
    out[prefix + '.main.0.weight'] = (c, c, 3, 3)
    _in_shapes(out, prefix + '.main.1', c)
    out[prefix + '.main.3.weight'] = (c, c, 3, 3)
    _in_shapes(out, prefix + '.main.4', c)


def _spade_resblock_shapes(out, prefix, c, s_dim):                        # generator.py:42-49
    for i in (0, 1):
        out[prefix + '.conv_%d.weight' % i] = (c, c, 3, 3)
        out[prefix + '.conv_%d.bias' % i] = (c,)
    for i in (0, 1):
        _spade_shapes(out, prefix + '.norm_%d' % i, c, s_dim)


def _bg_shapes(out, cfg):                                                 # generator.py:93-127
    d, p = cfg['conv_dim'], 'bg_model.model'
    out[p + '.0.weight'] = (d, cfg['bg_dim'], 7, 7)
    _in_shapes(out, p + '.1', d)
    idx, c = 3, d
    for _ in range(cfg['n_down']):
        out[p + '.%d.weight' % idx] = (2 * c, c, 3, 3)
        _in_shapes(out, p + '.%d' % (idx + 1), 2 * c)
        idx, c = idx + 3, 2 * c
    for _ in range(cfg['repeat_num']):
        _resblock_shapes(out, p + '.%d' % idx, c)
        idx += 1
    for _ in range(cfg['n_down']):
        out[p + '.%d.weight' % idx] = (c, c // 2, 3, 3)                   # ConvTranspose: (Cin, Cout, k, k)
        _in_shapes(out, p + '.%d' % (idx + 1), c // 2)
        idx, c = idx + 3, c // 2
    out[p + '.%d.weight' % idx] = (3, c, 7, 7)


def _unet_shapes(out, cfg, p, c_dim, s_dim, on_obj):                      # generator.py:138-237
    d, nd, rn, sl = cfg['conv_dim'], cfg['n_down'], cfg['repeat_num'], cfg['spade_layers']
    out[p + '.encoders.0.0.weight'] = (d, c_dim, 7, 7)
    _in_shapes(out, p + '.encoders.0.1', d)
    c = d
    for i in range(1, nd + 1):
        if sl[0]:
            out[p + '.encoders.%d.conv.weight' % i] = (2 * c, c, 3, 3)
            _spade_shapes(out, p + '.encoders.%d.norm' % i, 2 * c, s_dim)
        else:
            out[p + '.encoders.%d.0.weight' % i] = (2 * c, c, 3, 3)
            _in_shapes(out, p + '.encoders.%d.1' % i, 2 * c)
        c *= 2
    for i in range(rn):
        use_spade = sl[1] if i < rn // 2 else sl[2]
        if use_spade:
            _spade_resblock_shapes(out, p + '.resnets.%d' % i, c, s_dim)
        else:
            _resblock_shapes(out, p + '.resnets.%d' % i, c)
    dec, skp = OrderedDict(), OrderedDict()
    for i in range(nd):
        if sl[3]:
            dec[p + '.decoders.%d.conv.weight' % i] = (c, c // 2, 3, 3)
            _spade_shapes(dec, p + '.decoders.%d.norm' % i, c // 2, s_dim)
        else:
            dec[p + '.decoders.%d.0.weight' % i] = (c, c // 2, 3, 3)
            _in_shapes(dec, p + '.decoders.%d.1' % i, c // 2)
        skp[p + '.skippers.%d.0.weight' % i] = (c // 2, c, 3, 3)
        _in_shapes(skp, p + '.skippers.%d.1' % i, c // 2)
        c //= 2
    out.update(dec)
    out.update(skp)
    out[p + '.img_reg.0.weight'] = (3, c, 7, 7)
    if not on_obj:
        out[p + '.attetion_reg_hand.0.weight'] = (1, c, 7, 7)
        out[p + '.attetion_reg_bg.0.weight'] = (1, 2 * c, 7, 7)


def unet_num_channel(cfg):
    """ResUnetGenerator.num_channel (generator.py:157,170,182,189)."""
    d, nd, rn = cfg['conv_dim'], cfg['n_down'], cfg['repeat_num']
    nc = {0: d}
    for i in range(nd):
        nc[i + 1] = d * 2 ** (i + 1)
    for i in range(rn):
        nc[i + 1 + nd] = d * 2 ** nd
    return nc


def gen_param_shapes(cfg):
    """Ordered name->shape of Generator.state_dict() (SURVEY.md Appendix A;
    module registration order of generator.py:330-345)."""
    out = OrderedDict()
    _bg_shapes(out, cfg)
    _unet_shapes(out, cfg, 'obj_model', cfg['obj_dim'], cfg['obj_cond_dim'], True)
    _unet_shapes(out, cfg, 'src_model', cfg['img_dim'], cfg['img_cond_dim'], False)
    _unet_shapes(out, cfg, 'tsf_model', cfg['img_dim'], cfg['img_cond_dim'], False)
    nc = unet_num_channel(cfg)
    for L in cfg['attn_layers']:                                          # extract_attn.py:17-21
        c = nc[L]
        out['attn_%d.fully_connect_layer.0.weight' % L] = (128, 2 * c, 5, 5)
        out['attn_%d.fully_connect_layer.0.bias' % L] = (128,)
        out['attn_%d.fully_connect_layer.2.weight' % L] = (25, 128, 1, 1)
        out['attn_%d.fully_connect_layer.2.bias' % L] = (25,)
    return out


def disc_param_shapes(cfg):
    """PatchDiscriminator.state_dict() (discriminator.py:27-52, n_layers=4)."""
    ndf, nl = cfg['d_ndf'], cfg['d_layers']
    out = OrderedDict()
    out['model.0.weight'] = (ndf, cfg['d_input_nc'], 4, 4)
    out['model.0.bias'] = (ndf,)
    idx, prev = 2, 1
    for n in range(1, nl):
        mult = min(2 ** n, 8)
        out['model.%d.weight' % idx] = (ndf * mult, ndf * prev, 4, 4)
        out['model.%d.bias' % idx] = (ndf * mult,)
        idx, prev = idx + 3, mult
    mult = min(2 ** nl, 8)
    out['model.%d.weight' % idx] = (ndf * mult, ndf * prev, 4, 4)
    out['model.%d.bias' % idx] = (ndf * mult,)
    idx += 3
    out['model.%d.weight' % idx] = (1, ndf * mult, 4, 4)
    out['model.%d.bias' % idx] = (1,)
    return out


VGG_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512]
VGG_SLICE_ENDS = [2, 7, 12, 21, 30]                                       # vgg19.py:62 (after-relu ids)


def vgg_param_shapes():
    """torchvision vgg19().features up to index 29 (conv5_1 + relu), keyed as
    Vgg19.state_dict() does (vgg19.py:64-78: slice{k}.{features index})."""
    out, idx, cin, sl = OrderedDict(), 0, 3, 1
    for v in VGG_CFG:
        while idx >= VGG_SLICE_ENDS[sl - 1]:
            sl += 1
        if v == 'M':
            idx += 1
            continue
        out['slice%d.%d.weight' % (sl, idx)] = (v, cin, 3, 3)
        out['slice%d.%d.bias' % (sl, idx)] = (v,)
        cin = v
        idx += 2
    return out


def _normal(seed, index, shape):
    g = np.random.Generator(np.random.Philox(key=[seed & 0xFFFFFFFF, index]))
    return g.standard_normal(size=shape, dtype=np.float32)


def make_weights(shapes, seed=8, mode='init', kind='gan'):
    """Platform-independent deterministic weights (numpy Philox keyed by
    (seed, parameter index)) -- NOT torch.manual_seed, whose stream is not
    guaranteed across builds.

    mode='init'   : the recipe of base_network.py:14-25 -- every Conv/ConvTranspose
                    weight ~ N(0, 0.02), bias 0; InstanceNorm affine stays (1, 0).
    mode='random' : same conv weights but biases ~ N(0, 0.02), IN weight
                    1 + N(0, 0.1), IN bias N(0, 0.1) so every term is exercised.
    kind='vgg'    : He-normal surrogate for the unavailable ImageNet weights.
    """
    sd = OrderedDict()
    for i, (name, shp) in enumerate(shapes.items()):
        z = _normal(seed, i, shp)
        if kind == 'vgg':
            if len(shp) == 4:
                w = z * math.sqrt(2.0 / (shp[1] * 9))
            else:
                w = z * 0.05
        elif len(shp) == 4:
            w = z * 0.02
        elif name.endswith('.bias'):
            is_in = _is_instance_norm_param(name, shapes)
            w = z * (0.1 if is_in else 0.02) if mode == 'random' else np.zeros(shp, np.float32)
        else:                                   # 1-D '.weight' = InstanceNorm scale
            w = 1.0 + z * 0.1 if mode == 'random' else np.ones(shp, np.float32)
        sd[name] = torch.from_numpy(np.ascontiguousarray(w, dtype=np.float32))
    return sd


def _is_instance_norm_param(name, shapes):
    base = name.rsplit('.', 1)[0]
    w = shapes.get(base + '.weight')
    return w is not None and len(w) == 1


# --------------------------------------------------------------------------- layers

def _conv(x, sd, name, stride=1, pad=0):
    return F.conv2d(x, sd[name + '.weight'], sd.get(name + '.bias'), stride=stride, padding=pad)


def _convT(x, sd, name):
    # nn.ConvTranspose2d(k=3, stride=2, padding=1, output_padding=1, bias=False) generator.py:118,201
    return F.conv_transpose2d(x, sd[name + '.weight'], None, stride=2, padding=1, output_padding=1)


def _inorm(x, sd=None, name=None):
    # nn.InstanceNorm2d: eps 1e-5, biased variance, no running stats (generator.py:16; base_network.py:31)
    w = sd[name + '.weight'] if name else None
    b = sd[name + '.bias'] if name else None
    return F.instance_norm(x, weight=w, bias=b, eps=1e-5)


def spade(x, seg, sd, p):
    """spade.py:25-38."""
    normalized = _inorm(x)
    seg = F.interpolate(seg, size=x.shape[2:], mode='nearest')
    actv = F.relu(_conv(seg, sd, p + '.mlp_shared.0', pad=1))
    gamma = _conv(actv, sd, p + '.mlp_gamma', pad=1)
    beta = _conv(actv, sd, p + '.mlp_beta', pad=1)
    return normalized * (1 + gamma) + beta


def residual_block(x, sd, p):
    """generator.py:9-32 (dim_in == dim_out: identity shortcut)."""
    h = F.relu(_inorm(_conv(x, sd, p + '.main.0', pad=1), sd, p + '.main.1'))
    h = _inorm(_conv(h, sd, p + '.main.3', pad=1), sd, p + '.main.4')
    return x + h


def spade_residual_block(x, seg, sd, p):
    """generator.py:63-71."""
    dx = _conv(F.relu(spade(x, seg, sd, p + '.norm_0')), sd, p + '.conv_0', pad=1)
    dx = _conv(F.relu(spade(dx, seg, sd, p + '.norm_1')), sd, p + '.conv_1', pad=1)
    return x + dx


def spade_block(x, seg, sd, p, down=True):
    """generator.py:74-90."""
    x = _conv(x, sd, p + '.conv', stride=2, pad=1) if down else _convT(x, sd, p + '.conv')
    return F.relu(spade(x, seg, sd, p + '.norm'))


def conv_in_relu(x, sd, p, stride=1, pad=1, transposed=False):
    x = _convT(x, sd, p + '.0') if transposed else _conv(x, sd, p + '.0', stride=stride, pad=pad)
    return F.relu(_inorm(x, sd, p + '.1'))


def bg_net(x, sd, cfg):
    """ResNetGenerator.forward generator.py:129-135 (c=None)."""
    p, idx = 'bg_model.model', 0
    x = F.relu(_inorm(_conv(x, sd, p + '.0', pad=3), sd, p + '.1'))
    idx = 3
    for _ in range(cfg['n_down']):
        x = F.relu(_inorm(_conv(x, sd, p + '.%d' % idx, stride=2, pad=1), sd, p + '.%d' % (idx + 1)))
        idx += 3
    for _ in range(cfg['repeat_num']):
        x = residual_block(x, sd, p + '.%d' % idx)
        idx += 1
    for _ in range(cfg['n_down']):
        x = F.relu(_inorm(_convT(x, sd, p + '.%d' % idx), sd, p + '.%d' % (idx + 1)))
        idx += 3
    return torch.tanh(_conv(x, sd, p + '.%d' % idx, pad=3))


def unet_encoder_level(x, seg, sd, cfg, p, i):
    """encoders[i] for i>=1 (generator.py:161-171)."""
    if cfg['spade_layers'][0]:
        return spade_block(x, seg, sd, p + '.encoders.%d' % i, down=True)
    return conv_in_relu(x, sd, p + '.encoders.%d' % i, stride=2)


def unet_resnet(x, seg, sd, cfg, p, i):
    """resnets[i] (generator.py:176-191)."""
    rn = cfg['repeat_num']
    use_spade = cfg['spade_layers'][1] if i < rn // 2 else cfg['spade_layers'][2]
    if use_spade:
        return spade_residual_block(x, seg, sd, p + '.resnets.%d' % i)
    return residual_block(x, sd, p + '.resnets.%d' % i)


def unet_decode(x, enc_outs, seg, sd, cfg, p):
    """ResUnetGenerator.decode generator.py:298-309."""
    nd = cfg['n_down']
    for i in range(nd):
        if cfg['spade_layers'][3]:
            x = spade_block(x, seg, sd, p + '.decoders.%d' % i, down=False)
        else:
            x = conv_in_relu(x, sd, p + '.decoders.%d' % i, transposed=True)
        x = torch.cat([enc_outs[nd - 1 - i], x], dim=1)
        x = conv_in_relu(x, sd, p + '.skippers.%d' % i, stride=1)
    return x


def unet_forward(x, seg, sd, cfg, p):
    """ResUnetGenerator.forward generator.py:261-283."""
    e = conv_in_relu(x, sd, p + '.encoders.0', pad=3)
    enc = [e]
    for i in range(1, cfg['n_down'] + 1):
        e = unet_encoder_level(e, seg, sd, cfg, p, i)
        enc.append(e)
    for i in range(cfg['repeat_num']):
        e = unet_resnet(e, seg, sd, cfg, p, i)
    return unet_decode(e, enc, seg, sd, cfg, p)


# --------------------------------------------------------------------------- K1-K4 + attention

def block_extract(source, flow, k):
    """K1: thirdparty/block_extractor/block_extractor_kernel.cu:20-85.
    out[b,c,y,x], y in [0,k*Hf): yf=y/k, offset=y%k-k/2; sample position =
    pixel index + flow (PIXEL units, channel 1 = y, channel 0 = x) + offset
    (:57-67); bilinear taps with indices clamped to the border and weights NOT
    renormalised (:69-84).  Differentiable through torch indexing, which yields
    exactly the gradients K2 (:89-170) scatters."""
    B, C, H, W = source.shape
    Hf, Wf = flow.shape[2], flow.shape[3]
    dev = source.device
    y = torch.arange(k * Hf, device=dev)
    x = torch.arange(k * Wf, device=dev)
    yf, xf = y // k, x // k
    yo = (y % k - k // 2).to(flow.dtype)
    xo = (x % k - k // 2).to(flow.dtype)
    fy = flow[:, 1][:, yf][:, :, xf] + yo[None, :, None]
    fx = flow[:, 0][:, yf][:, :, xf] + xo[None, None, :]
    dy = fy + yf.to(flow.dtype)[None, :, None]
    dx = fx + xf.to(flow.dtype)[None, None, :]
    fly, flx = torch.floor(dy), torch.floor(dx)
    yT = fly.long().clamp(0, H - 1)
    yB = (fly.long() + 1).clamp(0, H - 1)
    xL = flx.long().clamp(0, W - 1)
    xR = (flx.long() + 1).clamp(0, W - 1)
    xR_P = dx - flx
    xL_P = 1 - xR_P
    yB_P = dy - fly
    yT_P = 1 - yB_P
    flat = source.reshape(B, C, H * W)

    def tap(yy, xx):
        idx = (yy * W + xx).reshape(B, 1, -1).expand(B, C, -1)
        return flat.gather(2, idx).reshape(B, C, k * Hf, k * Wf)

    out = (xL_P * yT_P).unsqueeze(1) * tap(yT, xL)
    out = out + (xR_P * yT_P).unsqueeze(1) * tap(yT, xR)
    out = out + (xL_P * yB_P).unsqueeze(1) * tap(yB, xL)
    out = out + (xR_P * yB_P).unsqueeze(1) * tap(yB, xR)
    return out


def local_attn_reshape(x, k):
    """K3: local_attn_reshape_kernel.cu:52-58: out[b,0,y,x] = in[b,(y%k)*k+x%k,y/k,x/k]."""
    B, C, H, W = x.shape
    assert C == k * k
    return x.reshape(B, k, k, H, W).permute(0, 3, 1, 4, 2).reshape(B, 1, k * H, k * W)


def extractor_attn(source, target, flow, sd, p, k=5):
    """ExtractorAttn.forward extract_attn.py:23-29 (softmax=True, LeakyReLU(0.01))."""
    bs = block_extract(source, flow, k)
    bt = block_extract(target, torch.zeros_like(flow), k)
    a = _conv(torch.cat((bt, bs), 1), sd, p + '.fully_connect_layer.0', stride=k)
    a = F.leaky_relu(a, 0.01)
    a = torch.softmax(_conv(a, sd, p + '.fully_connect_layer.2'), dim=1)
    a = local_attn_reshape(a, k)
    return F.avg_pool2d(a * bs, k, k)


def resize_trans(T, h):
    """Generator.resize_trans generator.py:466-473 (note size=(h, h))."""
    t = F.interpolate(T.permute(0, 3, 1, 2), size=(h, h), mode='bilinear', align_corners=True)
    return t.permute(0, 2, 3, 1)


def attn_flow(T, h):
    """The flow Generator.transform hands to ExtractorAttn (generator.py:484-488):
    idt built from an 'ij' meshgrid of arange(-1,1,2/h), stacked [xx, yy]."""
    t = resize_trans(T, h)
    ax = torch.arange(start=-1.0, end=1.0, step=2.0 / h)
    xx, yy = torch.meshgrid(ax, ax, indexing='ij')
    idt = torch.stack([xx, yy], dim=2).unsqueeze(0).to(T.device)
    return (t - idt).permute(0, 3, 1, 2)


def transform(x, T, sd, cfg, layer_num, y=None):
    """Generator.transform generator.py:480-491."""
    h = x.shape[2]
    if layer_num in cfg['attn_layers']:
        return extractor_attn(x, y, attn_flow(T, h), sd, 'attn_%d' % layer_num)
    return F.grid_sample(x, resize_trans(T, h), mode='bilinear', padding_mode='zeros', align_corners=False)


# --------------------------------------------------------------------------- generator / discriminator

def generator_forward(sd, cfg, bg, src_obj, tsf_obj, src_hand, tsf_hand, T,
                      src_obj_c, src_hand_c, tsf_obj_c, tsf_hand_c, src_armask=None, tsf_armask=None):
    """Generator.forward + infer_front generator.py:347-464 (conds given)."""
    src_bg_in = torch.cat([bg, src_hand_c], dim=1)
    tsf_bg_in = torch.cat([bg, tsf_hand_c], dim=1)
    if src_armask is not None:
        src_bg_in = torch.cat([src_bg_in, src_armask], dim=1)
    if tsf_armask is not None:
        tsf_bg_in = torch.cat([tsf_bg_in, tsf_armask], dim=1)
    src_img_bg = bg_net(src_bg_in, sd, cfg)
    tsf_img_bg = bg_net(tsf_bg_in, sd, cfg)

    nd, rn = cfg['n_down'], cfg['repeat_num']
    sx = conv_in_relu(src_hand, sd, 'src_model.encoders.0', pad=3)
    tx = conv_in_relu(tsf_hand, sd, 'tsf_model.encoders.0', pad=3)
    s_enc, t_enc = [sx], [tx]
    for i in range(1, nd + 1):
        sx = unet_encoder_level(sx, src_hand_c, sd, cfg, 'src_model', i)
        tx = unet_encoder_level(tx, tsf_hand_c, sd, cfg, 'tsf_model', i)
        tx = tx + transform(sx, T, sd, cfg, i, y=tx)
        s_enc.append(sx)
        t_enc.append(tx)
    for i in range(rn):
        sx = unet_resnet(sx, src_hand_c, sd, cfg, 'src_model', i)
        tx = unet_resnet(tx, tsf_hand_c, sd, cfg, 'tsf_model', i)
        tx = tx + transform(sx, T, sd, cfg, i + nd + 1, y=tx)

    sy = unet_forward(src_obj, src_obj_c, sd, cfg, 'obj_model')
    ty = unet_forward(tsf_obj, tsf_obj_c, sd, cfg, 'obj_model')
    sx = unet_decode(sx, s_enc, src_hand_c, sd, cfg, 'src_model')
    tx = unet_decode(tx, t_enc, tsf_hand_c, sd, cfg, 'tsf_model')

    def regress(x, y, p):                                                  # generator.py:311-315
        img = torch.tanh(_conv(x, sd, p + '.img_reg.0', pad=3))
        mh = torch.sigmoid(_conv(x, sd, p + '.attetion_reg_hand.0', pad=3))
        mb = torch.sigmoid(_conv(torch.cat([x, y], dim=1), sd, p + '.attetion_reg_bg.0', pad=3))
        return img, mh, mb

    src_hand_o, src_mask_hand, src_mask_bg = regress(sx, sy, 'src_model')
    tsf_hand_o, tsf_mask_hand, tsf_mask_bg = regress(tx, ty, 'tsf_model')
    src_obj_o = torch.tanh(_conv(sy, sd, 'obj_model.img_reg.0', pad=3))
    tsf_obj_o = torch.tanh(_conv(ty, sd, 'obj_model.img_reg.0', pad=3))
    return (src_img_bg, tsf_img_bg, src_obj_o, src_hand_o, src_mask_bg, src_mask_hand,
            tsf_obj_o, tsf_hand_o, tsf_mask_bg, tsf_mask_hand)


def discriminator_forward(sd, cfg, x):
    """PatchDiscriminator.forward discriminator.py:55-57 (instance norm without affine)."""
    nl = cfg['d_layers']
    x = F.leaky_relu(_conv(x, sd, 'model.0', stride=2, pad=1), 0.2)
    idx = 2
    for _ in range(1, nl):
        x = F.leaky_relu(_inorm(_conv(x, sd, 'model.%d' % idx, stride=2, pad=1)), 0.2)
        idx += 3
    x = F.leaky_relu(_inorm(_conv(x, sd, 'model.%d' % idx, stride=1, pad=1)), 0.2)
    idx += 3
    return _conv(x, sd, 'model.%d' % idx, stride=1, pad=1)


def vgg_features(sd, x):
    """Vgg19.forward vgg19.py:84-91: relu1_1, 2_1, 3_1, 4_1, 5_1."""
    outs, idx, sl = [], 0, 1
    for v in VGG_CFG:
        if idx >= VGG_SLICE_ENDS[sl - 1]:
            outs.append(x)
            sl += 1
        if v == 'M':
            x = F.max_pool2d(x, 2, 2)
            idx += 1
        else:
            x = F.relu(_conv(x, sd, 'slice%d.%d' % (sl, idx), pad=1))
            idx += 2
    outs.append(x)
    return outs


VGG_LOSS_WEIGHTS = [1.0 / 32, 1.0 / 16, 1.0 / 8, 1.0 / 4, 1.0]            # vgg19.py:102


def vgg_loss(sd, x, y):
    """VGGLoss.forward vgg19.py:104-109."""
    fx, fy = vgg_features(sd, x), vgg_features(sd, y)
    loss = 0
    for w, a, b in zip(VGG_LOSS_WEIGHTS, fx, fy):
        loss = loss + w * F.l1_loss(a, b.detach())
    return loss


# --------------------------------------------------------------------------- trainer

class Lambdas(object):
    def __init__(self, **kw):
        # scripts/train_hov3_ddp.sh:24-27 values
        self.D_prob, self.rec, self.tsf, self.mask, self.mask_smooth = 1.0, 10.0, 10.0, 1.0, 1.0
        self.__dict__.update(kw)


def smooth_loss(m):
    """Trainer._compute_loss_smooth trainer.py:479-481."""
    return (m[:, :, :, :-1] - m[:, :, :, 1:]).abs().mean() + (m[:, :, :-1, :] - m[:, :, 1:, :]).abs().mean()


class OracleTrainer(object):
    """Restatement of Trainer (trainer.py:188-591) over explicit state dicts.
    Inputs are the a2 attributes of SURVEY.md §8a (what set_input stages)."""

    def __init__(self, cfg, sd_G, sd_D, sd_vgg, lam=None, mask_bce=True, lr=2e-4, betas=(0.5, 0.999)):
        self.cfg = cfg
        self.G = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in sd_G.items())
        self.D = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in sd_D.items())
        self.vgg = OrderedDict((k, v.clone()) for k, v in sd_vgg.items()) if sd_vgg is not None else None
        self.lam = lam or Lambdas()
        self.mask_bce = mask_bce
        # trainer.py:275-278
        self.opt_G = torch.optim.Adam(list(self.G.values()), lr=lr, betas=betas)
        self.opt_D = torch.optim.Adam(list(self.D.values()), lr=lr, betas=betas)
        self.errors = OrderedDict()

    def set_prepared_input(self, inp):
        self.inp = inp

    def _tsf_cond(self):
        i = self.inp
        parts = [i['input_G_tsf_obj'][:, 3:], i['input_G_tsf_hand'][:, 3:]]
        if self.cfg['armask']:
            parts.append(i['armask_tsf'])
        return torch.cat(parts, dim=1)

    def forward(self):
        """Trainer.forward trainer.py:373-415 (use_spade branch)."""
        i, cfg = self.inp, self.cfg
        so, sh, to, th = i['input_G_src_obj'], i['input_G_src_hand'], i['input_G_tsf_obj'], i['input_G_tsf_hand']
        outs = generator_forward(
            self.G, cfg, i['input_G_bg'], so[:, :3], to[:, :3], sh[:, :3], th[:, :3], i['T'],
            so[:, 3:], sh[:, 3:], to[:, 3:], th[:, 3:],
            i['armask_src'] if cfg['armask'] else None, i['armask_tsf'] if cfg['armask'] else None)
        (src_bg, tsf_bg, src_obj, src_hand, src_mbg, src_mh, tsf_obj, tsf_hand, tsf_mbg, tsf_mh) = outs
        self.g_outs = outs
        fake_src = src_mbg * src_bg + (1 - src_mbg) * (src_obj * src_mh + src_hand * (1 - src_mh))
        fake_tsf = tsf_mbg * tsf_bg + (1 - tsf_mbg) * (tsf_obj * tsf_mh + tsf_hand * (1 - tsf_mh))
        masks_bg = torch.cat([src_mbg, tsf_mbg], dim=0)
        masks_hand = torch.cat([src_mh, tsf_mh], dim=0)
        return src_bg, tsf_bg, fake_src, fake_tsf, masks_bg, masks_hand

    def g_loss(self, fake_src, fake_tsf, masks_bg, masks_hand):
        """Trainer._optimize_G trainer.py:436-457."""
        i, lam = self.inp, self.lam
        d_fake = discriminator_forward(self.D, self.cfg, torch.cat([fake_tsf, self._tsf_cond()], dim=1))
        e = self.errors
        e['g_adv'] = torch.mean((d_fake - 0) ** 2) * lam.D_prob
        e['g_rec'] = F.l1_loss(fake_src, i['real_src']) * lam.rec
        e['g_tsf'] = torch.mean(vgg_loss(self.vgg, fake_tsf, i['real_tsf'])) * lam.tsf
        crt = F.binary_cross_entropy if self.mask_bce else F.mse_loss
        e['g_mask'] = (crt(masks_bg, i['bg_mask']) + crt(masks_hand, i['hand_mask'])) * lam.mask
        e['g_mask_smooth'] = torch.zeros(())
        if lam.mask_smooth != 0:
            e['g_mask_smooth'] = (smooth_loss(masks_bg) + smooth_loss(masks_hand)) * lam.mask_smooth
        return e['g_adv'] + e['g_rec'] + e['g_tsf'] + e['g_mask'] + e['g_mask_smooth']

    def d_loss(self, fake_tsf):
        """Trainer._optimize_D trainer.py:459-474."""
        cond = self._tsf_cond()
        d_real = discriminator_forward(self.D, self.cfg, torch.cat([self.inp['real_tsf'], cond], dim=1))
        d_fake = discriminator_forward(self.D, self.cfg, torch.cat([fake_tsf.detach(), cond], dim=1))
        self.errors['d_real'] = d_real.mean()
        self.errors['d_fake'] = d_fake.mean()
        return (torch.mean((d_real - 1) ** 2) + torch.mean((d_fake + 1) ** 2)) * self.lam.D_prob

    def optimize_parameters(self, trainable=True):
        """Trainer.optimize_parameters trainer.py:417-434."""
        _, _, fake_src, fake_tsf, mbg, mh = self.forward()
        loss_G = self.g_loss(fake_src, fake_tsf, mbg, mh)
        self.opt_G.zero_grad()
        for p in self.D.values():          # the reference lets D grads accumulate here; they are zeroed at :432
            p.grad = None
        loss_G.backward()
        self.opt_G.step()
        self.loss_G = loss_G.detach()
        if trainable:
            loss_D = self.d_loss(fake_tsf)
            self.opt_D.zero_grad()
            loss_D.backward()
            self.opt_D.step()
            self.loss_D = loss_D.detach()

    def get_current_errors(self):
        order = ['g_rec', 'g_tsf', 'g_adv', 'g_mask', 'g_mask_smooth', 'd_real', 'd_fake']   # trainer.py:483-492
        return OrderedDict((k, float(self.errors[k].detach())) for k in order if k in self.errors)
