"""CPU emulation of split-operand MFMA arithmetic in the FORWARD of the generator (TEST INFRASTRUCTURE; imports the oracle).
Every convolution of the oracle's forward is replaced by a sum of convolutions over rounded operand halves, to predict the
forward error of arithmetic modes BEFORE building kernels for them:
    x3     : ah*wh + al*wh + ah*wl          (fp16 halves; what HOIG_PREC_BF16X3 issues in the forward)
    x2     : ah*wh + al*wh                  (HOIG_PREC_F16X2)
    x1     : ah*wh                          (HOIG_PREC_BF16 forward)
    x1+2q8 : ah*wh + Q8(al)*Q8(wh) + Q8(ah)*Q8(wl)    Q8 = OCP e4m3 with a power-of-two scale per 32 channels (MX blocks):
             the two cross terms on v_mfma_scale_f32_32x32x64_f8f6f4 at 2x the bf16 rate -> 2 MFMA units per product
    x1+2q6 : the same with e2m3 (fp6: 4x the bf16 rate -> 1.5 units)
Calibration: x2 and x1 are measured on the GPU (profiles/r02_precision_frontier.txt: 4.6e-3 and 7.2e-3 at 64x64).
    python tools/emulate_split_terms.py [side] [batch]"""
import os, sys
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from common import oracle_trainer

_conv2d, _convT = F.conv2d, F.conv_transpose2d
MODE = ['exact']


def h16(x):
    return x.half().float()


def q_mx(x, dim, kind):
    """Block-scaled quantisation along `dim` in blocks of 32 (power-of-two scale per block), e4m3 or e2m3 elements."""
    x = x.movedim(dim, -1)
    n = x.shape[-1]
    pad = (-n) % 32
    xp = F.pad(x, (0, pad)).reshape(x.shape[:-1] + ((n + pad) // 32, 32))
    amax = xp.abs().amax(-1, keepdim=True).clamp_min(1e-38)
    emax = 7 if kind == 'e4m3' else 2                       # block maximum lands in [128, 256) (e4m3: up to 448) / [4, 8) (e2m3: up to 7.5)
    scale = torch.exp2(torch.floor(torch.log2(amax)) - emax)
    y = xp / scale
    if kind == 'e4m3':
        q = y.to(torch.float8_e4m3fn).float()
    else:                                                   # e2m3: sign, 2 exponent bits (bias 1), 3 mantissa bits; step 0.125 below 1
        a = y.abs().clamp(max=7.5)
        e = torch.floor(torch.log2(a.clamp_min(1e-30))).clamp(min=0, max=2)
        step = torch.exp2(e - 3)
        q = torch.sign(y) * torch.round(a / step) * step
    out = (q * scale).reshape(x.shape[:-1] + (n + pad,))[..., :n]
    return out.movedim(-1, dim)


def split_conv(conv, x, w, bias, kw, wdim):
    m = MODE[0]
    if m == 'exact' or x.shape[1] % 32 or w.shape[wdim] % 32:     # the first-layer convs run in exact fp32 on the GPU too
        return conv(x, w, bias, **kw)
    ah, wh = h16(x), h16(w * 256.0)                         # weights scaled by 2^8 before the split, as the kernels do
    al, wl = h16(x - ah), h16(w * 256.0 - wh)
    y = conv(ah, wh, None, **kw)
    if m == 'x3':
        y = y + conv(al, wh, None, **kw) + conv(ah, wl, None, **kw)
    elif m == 'x2':
        y = y + conv(al, wh, None, **kw)
    elif m.startswith('x1+2q'):
        kind = 'e4m3' if m.endswith('8') else 'e2m3'
        cd = 0 if wdim == 0 else 1                           # the reduction (input-channel) axis of the weight
        y = y + conv(q_mx(al, 1, kind), q_mx(wh, cd, kind), None, **kw) + conv(q_mx(ah, 1, kind), q_mx(wl, cd, kind), None, **kw)
    y = y / 256.0
    return y if bias is None else y + bias.view(1, -1, 1, 1)


def conv2d(x, w, bias=None, stride=1, padding=0, dilation=1, groups=1):
    return split_conv(_conv2d, x, w, bias, dict(stride=stride, padding=padding), 1)


def convT(x, w, bias=None, stride=1, padding=0, output_padding=0, groups=1, dilation=1):
    return split_conv(_convT, x, w, bias, dict(stride=stride, padding=padding, output_padding=output_padding), 0)


def main():
    side = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    torch.set_num_threads(8)
    F.conv2d, F.conv_transpose2d = conv2d, convT
    torch.nn.functional.conv2d, torch.nn.functional.conv_transpose2d = conv2d, convT
    ot = oracle_trainer('generator_spade_attn', batch, side)
    outs = {}
    for m in ['exact', 'x3', 'x2', 'x1', 'x1+2q8', 'x1+2q6']:
        MODE[0] = m
        with torch.no_grad():
            outs[m] = [o.clone() for o in ot.forward()]
    print('side %d batch %d: max over the six forward outputs of max|a-b| / max|b| against the exact-fp32 forward' % (side, batch))
    for m in ['x3', 'x2', 'x1', 'x1+2q8', 'x1+2q6']:
        err = max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(outs[m], outs['exact']))
        print('  %-8s %.3e' % (m, err))


if __name__ == '__main__':
    main()
