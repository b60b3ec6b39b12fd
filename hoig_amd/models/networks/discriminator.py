"""PatchGAN discriminator on the HIP operators (reference: models/networks/discriminator.py:8-57,
n_layers=4, norm_type='instance' i.e. InstanceNorm2d(affine=False), LeakyReLU(0.2), all convs with bias)."""
import torch

from ... import ops
from ..._lib import ACT_NONE, ACT_LRELU
from ...nn import ParamTree
from .schema import discriminator_schema
from .generator import to_nhwc, as_nchw


class PatchDiscriminator(ParamTree):
    def __init__(self, input_nc, ndf=64, n_layers=3, norm_type='batch', use_sigmoid=False, device=None):
        if norm_type != 'instance':
            raise NotImplementedError('normalization layer [%s] is not on the HOGAN path (options default: instance, '
                                      'base_options.py:48)' % norm_type)
        if use_sigmoid:
            raise NotImplementedError('use_sigmoid=True is never used by Trainer (trainer.py:266-268)')
        sch = discriminator_schema(input_nc, ndf, n_layers)
        device = device if device is not None else torch.device('cuda', torch.cuda.current_device())
        super().__init__(sch.shapes, device, sch.transposed)
        self.n_layers = n_layers
        self._name = 'BaseNetwork'

    @property
    def name(self):
        return self._name

    def forward_nhwc(self, x):
        P = self.P
        prec = ops.subnet_precision('d')
        # (19 [DexYCB: 24] input channels: zero-padded to 32 so that the layer runs on the 16-bit kernels, ops.conv2d_padded_in)
        x = ops.conv2d_padded_in(x, P['model.0.weight'], P['model.0.bias'], 2, 1, ACT_LRELU, 0.2, prec=prec)
        idx = 2
        for _ in range(1, self.n_layers):
            x = ops.conv2d(x, P['model.%d.weight' % idx], P['model.%d.bias' % idx], 2, 1, dead_bias=True, prec=prec)
            x = ops.instance_norm(x, act=ACT_LRELU, slope=0.2)
            idx += 3
        x = ops.conv2d(x, P['model.%d.weight' % idx], P['model.%d.bias' % idx], 1, 1, dead_bias=True, prec=prec)
        x = ops.instance_norm(x, act=ACT_LRELU, slope=0.2)
        idx += 3
        return ops.conv2d(x, P['model.%d.weight' % idx], P['model.%d.bias' % idx], 1, 1, ACT_NONE, prec=prec)

    def forward(self, input):
        return as_nchw(self.forward_nhwc(to_nhwc(input)))
