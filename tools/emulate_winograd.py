"""CPU emulation of Winograd F(2x2,3x3) on split-operand MFMA arithmetic, forward AND gradients, BEFORE a kernel is built for it
(TEST INFRASTRUCTURE; imports the oracle).  VERDICT r5 item 1(a).

Every 32-channel-aligned convolution of the oracle's step runs through an autograd.Function that restates what the HIP path issues:
    forward           three fp16 terms (weights scaled by 2^8): ah*wh + al*wh + ah*wl, fp32 accumulation
    data gradient     two bf16 terms: dy split hi + lo, the weights ONE bf16
    weight gradient   two bf16 terms: dy split hi + lo, x ONE bf16
(`direct`: the shipped default `bf16x3:f16x2`).  The Winograd modes replace the arithmetic of the 3x3 stride-1 pad-1
layers among them:
    V = B^T d B on fp32 (adds), U = G (2^8 g) G^T on fp32 (adds, halvings), BOTH split AFTER their transform, 16 element-wise
    products accumulated over the input channels in fp32, Y = A^T M A on fp32
    wino-f   forward only                  wino-fd  forward and data gradient (the flipped, transposed weights; U as ONE bf16)
    wino-fd3 as wino-fd with the data gradient's U split too (three terms)
Reported, against the exact-fp32 oracle step on the same seeded inputs and weights: the six forward outputs (max-norm relative), and
the rel-L2 error of every gradient tensor of G and D (median / p95 / worst) -- the quantities tests/test_trainer_gpu.py and
tests/test_configs_gpu.py bound at 1e-3 and 1e-2 / 2e-2 / 3e-2.
    python tools/emulate_winograd.py [side] [batch] [min_cin]"""
import os, sys, time
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from common import oracle_trainer

_conv2d, _convT = F.conv2d, F.conv_transpose2d
MODE = ['exact']
MIN_CIN = [32]
SKIP = ('.conv_0.bias', )                                  # biases in front of an instance norm: their true gradient is zero
SKIP_D = ('model.2.bias', 'model.5.bias', 'model.8.bias', 'model.11.bias')      # (tools/precision_frontier.py holds the same list)

BT = torch.tensor([[1., 0., -1., 0.], [0., 1., 1., 0.], [0., -1., 1., 0.], [0., 1., 0., -1.]])
G_ = torch.tensor([[1., 0., 0.], [.5, .5, .5], [.5, -.5, .5], [0., 0., 1.]])
AT = torch.tensor([[1., 1., 1., 0.], [0., 1., -1., -1.]])


def h16(x):
    return x.half().float()


def b16(x):
    return x.bfloat16().float()


def wino_in(x):
    """x (B,C,H,W), H and W even -> V (B,C,H/2,W/2,4,4) = B^T d B over the 4x4 patches (stride 2) of the zero-padded image."""
    xp = F.pad(x, (1, 1, 1, 1))
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2)                   # (B,C,th,tw,4,4)
    t = torch.einsum('ik,bcyxkl->bcyxil', BT, d)            # fp32 adds: each row of B^T has two +-1 entries
    return torch.einsum('bcyxil,jl->bcyxij', t, BT)


def wino_w(w):
    """w (Co,Ci,3,3) -> U (Co,Ci,4,4) = G g G^T."""
    return torch.einsum('ik,ockl,jl->ocij', G_, w, G_)


def wino_out(m):
    """M (B,Co,th,tw,4,4) -> Y (B,Co,2 th,2 tw) = A^T M A."""
    y = torch.einsum('ai,boyxij,cj->boyxac', AT, m, AT)
    B, Co, th, tw = y.shape[:4]
    return y.permute(0, 1, 2, 4, 3, 5).reshape(B, Co, 2 * th, 2 * tw)


def wino_conv(x, w, terms, half):
    """3x3 stride-1 pad-1 convolution through F(2x2,3x3): `terms` = 3 (both operands split), 2 (V split, U one value); `half` = h16
    (weights scaled by 2^8 first) or b16."""
    sc = 256.0 if half is h16 else 1.0
    V, U = wino_in(x), wino_w(w * sc)
    Vh, Uh = half(V), half(U)
    Vl = half(V - Vh)
    mm = lambda v, u: torch.einsum('bcyxij,ocij->boyxij', v, u)
    M = mm(Vh, Uh) + mm(Vl, Uh)
    if terms == 3:
        M = M + mm(Vh, half(U - Uh))
    return wino_out(M) / sc


class SplitConv(torch.autograd.Function):
    """conv2d / conv_transpose2d with the HIP path's arithmetic in all three launches."""

    @staticmethod
    def forward(ctx, x, w, transposed, kw, wino_f, wino_d):
        ctx.save_for_backward(x, w)
        ctx.cfg = (transposed, kw, wino_d)
        conv = _convT if transposed else _conv2d
        if wino_f:
            return wino_conv(x, w, 3, h16)
        ah, wh = h16(x), h16(w * 256.0)
        al, wl = h16(x - ah), h16(w * 256.0 - wh)
        return (conv(ah, wh, None, **kw) + conv(al, wh, None, **kw) + conv(ah, wl, None, **kw)) / 256.0

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        transposed, kw, wino_d = ctx.cfg
        dh = b16(dy)
        dl = b16(dy - dh)
        w1, x1 = b16(w), b16(x)
        if wino_d:                                            # dx = conv(dy, flip(w)^T), pad 1
            wf = w.flip(2, 3).transpose(0, 1).contiguous()
            dx = wino_conv(dy, wf, wino_d, b16)
        elif not transposed:
            gi = lambda d: torch.nn.grad.conv2d_input(x.shape, w1, d, **kw)
            dx = gi(dh) + gi(dl)
        else:                                                 # the data gradient of a transposed convolution is a convolution
            k2 = dict(stride=kw['stride'], padding=kw['padding'])
            dx = _conv2d(dh, w1, None, **k2) + _conv2d(dl, w1, None, **k2)
        if not transposed:
            gw = lambda d: torch.nn.grad.conv2d_weight(x1, w.shape, d, **kw)
            dw = gw(dh) + gw(dl)
        else:                                                 # dW of convT(x -> y) = dW of conv(y -> x) with the roles swapped
            k2 = dict(stride=kw['stride'], padding=kw['padding'])
            gw = lambda d: torch.nn.grad.conv2d_weight(d, w.shape, x1, **k2)
            dw = gw(dh) + gw(dl)
        return dx, dw, None, None, None, None


def _route(x, w, bias, transposed, kw, wdim):
    m = MODE[0]
    conv = _convT if transposed else _conv2d
    if m == 'exact' or x.shape[1] % 32 or w.shape[wdim] % 32:      # first-layer convolutions: exact fp32 on the GPU too
        return conv(x, w, bias, **kw)
    s1 = (not transposed and tuple(w.shape[2:]) == (3, 3) and kw['stride'] in (1, (1, 1)) and kw['padding'] in (1, (1, 1))
          and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0 and x.shape[1] >= MIN_CIN[0])
    wino_f = s1 and m.startswith('wino')
    wino_d = (3 if m == 'wino-fd3' else 2) if (s1 and m in ('wino-fd', 'wino-fd3')) else 0
    y = SplitConv.apply(x, w, transposed, kw, wino_f, wino_d)
    return y if bias is None else y + bias.view(1, -1, 1, 1)


def conv2d(x, w, bias=None, stride=1, padding=0, dilation=1, groups=1):
    return _route(x, w, bias, False, dict(stride=stride, padding=padding), 1)


def convT(x, w, bias=None, stride=1, padding=0, output_padding=0, groups=1, dilation=1):
    return _route(x, w, bias, True, dict(stride=stride, padding=padding, output_padding=output_padding), 0)


def rel_l2(a, b):
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def run(mode, side, batch):
    MODE[0] = mode
    ot = oracle_trainer('generator_spade_attn', batch, side)
    with torch.no_grad():
        outs = [o.clone() for o in ot.forward()]
    # gradients as optimize_parameters forms them, without the optimiser steps
    _, _, fake_src, fake_tsf, mbg, mh = ot.forward()
    lg = ot.g_loss(fake_src, fake_tsf, mbg, mh)
    for p in list(ot.G.values()) + list(ot.D.values()):
        p.grad = None
    lg.backward()
    gG = {('G', k): v.grad.clone() for k, v in ot.G.items() if v.grad is not None and not k.endswith(SKIP)}
    for p in ot.D.values():
        p.grad = None
    ot.d_loss(fake_tsf).backward()
    gD = {('D', k): v.grad.clone() for k, v in ot.D.items() if v.grad is not None and k not in SKIP_D}
    gG.update(gD)
    return outs, gG


def main():
    side = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    if len(sys.argv) > 3:
        MIN_CIN[0] = int(sys.argv[3])
    torch.set_num_threads(8)
    F.conv2d, F.conv_transpose2d = conv2d, convT
    torch.nn.functional.conv2d, torch.nn.functional.conv_transpose2d = conv2d, convT
    # self-check of the transform matrices: fp32 Winograd against the direct convolution
    g = torch.Generator().manual_seed(1)
    x, w = torch.randn(2, 32, 8, 8, generator=g), torch.randn(32, 32, 3, 3, generator=g) * 0.05
    ref = _conv2d(x, w, None, padding=1)
    got = wino_out(torch.einsum('bcyxij,ocij->boyxij', wino_in(x), wino_w(w)))
    assert float((got - ref).abs().max() / ref.abs().max()) < 1e-5
    ref_o, ref_g = run('exact', side, batch)
    print('side %d batch %d, Winograd on the 3x3 stride-1 layers with >= %d input channels; against the exact-fp32 step' % (side, batch, MIN_CIN[0]))
    print('%-9s | %-9s | gradient rel-L2 median / p95 / worst (tensor)' % ('mode', 'fwd max'))
    for m in ['direct', 'wino-f', 'wino-fd', 'wino-fd3']:
        t0 = time.time()
        o, gr = run(m, side, batch)
        ferr = max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(o, ref_o))
        e = sorted((rel_l2(gr[k], ref_g[k]), k[0] + '.' + k[1]) for k in ref_g)
        print('%-9s | %.3e | %.2e / %.2e / %.2e (%s)   [%.0f s]' % (m, ferr, e[len(e) // 2][0], e[int(0.95 * len(e))][0], e[-1][0], e[-1][1],
                                                                  time.time() - t0), flush=True)


if __name__ == '__main__':
    main()
