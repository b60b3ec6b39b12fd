"""VGG19 perceptual loss on the HIP operators (reference: models/networks/vgg19.py:6-109).

The reference builds ``torchvision.models.vgg19(pretrained=True).features``; those ImageNet weights are not
vendored and there is no network here, so ``Vgg19`` starts from deterministic He-normal surrogate weights and
loads real ones through ``load_state_dict`` / ``load_torchvision_features`` when a checkpoint is supplied
(opt.vgg_weights).  Inputs are fed in [-1,1] without ImageNet normalisation, as the reference does."""
import math

import numpy as np
import torch

from ... import ops
from ..._lib import ACT_RELU
from ...nn import ParamTree
from .schema import vgg_schema, VGG_LAYERS, VGG_SLICE_ENDS
from .generator import to_nhwc


class Vgg19(ParamTree):
    def __init__(self, requires_grad=False, before_relu=False, device=None, seed=10):
        if before_relu:
            raise NotImplementedError('before_relu=True is never used by Trainer (trainer.py:302-304)')
        sch = vgg_schema()
        device = device if device is not None else torch.device('cuda', torch.cuda.current_device())
        super().__init__(sch.shapes, device)
        with torch.no_grad():
            for i, (name, p) in enumerate(self.P.items()):
                g = np.random.Generator(np.random.Philox(key=[seed, i]))
                z = torch.from_numpy(g.standard_normal(size=tuple(p.shape), dtype=np.float32))
                p.copy_(z * (math.sqrt(2.0 / (p.shape[1] * 9)) if p.dim() == 4 else 0.05))
        if not requires_grad:
            self.set_requires_grad(False)                                   # vgg19.py:80-82

    def load_torchvision_features(self, sd):
        """Accepts torchvision's vgg19().features state_dict ('0.weight', '2.weight', ...)."""
        mapped = {}
        for name in self.P:
            sl, idx, kind = name.split('.')
            mapped[name] = sd['%s.%s' % (idx, kind)]
        return self.load_state_dict(mapped)

    def forward_nhwc(self, x):
        outs, idx, sl = [], 0, 1
        prec = ops.subnet_precision('vgg')
        for v in VGG_LAYERS:
            tap = idx >= VGG_SLICE_ENDS[sl - 1]           # x is a slice output (relu{k}_1): the loss reads it AND the next layer
            if tap:
                sl += 1
            if v == 'M':
                if tap:
                    outs.append(x)
                x = ops.maxpool2(x)
                idx += 1
            else:
                p = 'slice%d.%d' % (sl, idx)
                if tap:
                    # (the loss reads the feature through the convolution's pass-through output: its gradient is then added by the
                    # convolution's data-gradient kernel, ops.conv2d_fork, instead of by the autograd engine)
                    x, feat = ops.conv2d_fork(x, self.P[p + '.weight'], self.P[p + '.bias'], 1, 1, ACT_RELU, prec=prec)
                    outs.append(feat)
                else:
                    x = ops.conv2d(x, self.P[p + '.weight'], self.P[p + '.bias'], 1, 1, ACT_RELU, prec=prec)
                idx += 2
        outs.append(x)
        return outs

    def forward(self, X):
        return [o.permute(0, 3, 1, 2) for o in self.forward_nhwc(to_nhwc(X))]


class VGGLoss(object):
    """sum_i w_i * L1(vgg_i(x), vgg_i(y).detach()), w = [1/32, 1/16, 1/8, 1/4, 1] (vgg19.py:94-109)."""
    weights = [1.0 / 32, 1.0 / 16, 1.0 / 8, 1.0 / 4, 1.0]

    def __init__(self, vgg=None, before_relu=False):
        self.vgg = vgg if vgg is not None else Vgg19(before_relu=before_relu)

    def to(self, *a, **k):
        return self

    def cuda(self, *a, **k):
        return self

    def forward_nhwc(self, x, y, scale=1.0, side=None, into=None):
        """`side`: a stream on which the (gradient-free) features of the target `y` are evaluated beside those of `x`.
        `into` = ops.LossSlots.term(name): the five levels are added to that slot; returns their handles (for LossSlots.total)."""
        if side is not None and x.is_cuda:
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side), torch.no_grad():
                ops.test_delay('loss_vgg')
                fy = self.vgg.forward_nhwc(y)
            fx = self.vgg.forward_nhwc(x)
            main.wait_stream(side)
            for t in fy:
                ops.cross_stream(t, main)
        else:
            fx = self.vgg.forward_nhwc(x)
            with torch.no_grad():
                fy = self.vgg.forward_nhwc(y)
        if into is not None:
            return [ops.l1_loss(a, b, scale=w * scale, into=into) for w, a, b in zip(self.weights, fx, fy)]
        loss = 0
        for w, a, b in zip(self.weights, fx, fy):
            loss = loss + ops.l1_loss(a, b, scale=w * scale)
        return loss

    def __call__(self, x, y):
        return self.forward_nhwc(to_nhwc(x), to_nhwc(y))
