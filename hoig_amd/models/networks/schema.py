"""Parameter schema (names, logical shapes, order) of the hot-path networks, as the reference's
``state_dict()`` lays them out (generator.py:93-345, discriminator.py:27-52,
extract_attn.py:17-21, vgg19.py:56-78).  ``Schema.transposed`` lists the ConvTranspose2d weights, whose logical
shape is (Cin, Cout, k, k)."""
from collections import OrderedDict

GEN_VARIANTS = {                       # models/networks/__init__.py:11-25
    'generator_base': ([0, 0, 0, 0], []),
    'generator_spade': ([1, 1, 0, 0], []),
    'generator_spade_attn': ([1, 1, 0, 0], [1, 2, 3, 4, 5, 6, 7, 8, 9]),
    'generator_spade_attn_tiny': ([0, 0, 1, 1], [1, 2, 3, 4, 5, 6, 7, 8, 9]),
}


class Schema(object):
    def __init__(self):
        self.shapes = OrderedDict()
        self.transposed = []

    def conv(self, name, cout, cin, k, bias=False):
        self.shapes[name + '.weight'] = (cout, cin, k, k)
        if bias:
            self.shapes[name + '.bias'] = (cout,)

    def convT(self, name, cin, cout, k):
        self.shapes[name + '.weight'] = (cin, cout, k, k)
        self.transposed.append(name + '.weight')

    def affine(self, name, c):
        self.shapes[name + '.weight'] = (c,)
        self.shapes[name + '.bias'] = (c,)

    def spade(self, name, norm_nc, label_nc):
        self.conv(name + '.mlp_shared.0', 128, label_nc, 3, bias=True)
        self.conv(name + '.mlp_gamma', norm_nc, 128, 3, bias=True)
        self.conv(name + '.mlp_beta', norm_nc, 128, 3, bias=True)

    def resblock(self, name, c):
        self.conv(name + '.main.0', c, c, 3)
        self.affine(name + '.main.1', c)
        self.conv(name + '.main.3', c, c, 3)
        self.affine(name + '.main.4', c)

    def spade_resblock(self, name, c, s_dim):
        self.conv(name + '.conv_0', c, c, 3, bias=True)
        self.conv(name + '.conv_1', c, c, 3, bias=True)
        self.spade(name + '.norm_0', c, s_dim)
        self.spade(name + '.norm_1', c, s_dim)


class GeneratorConfig(object):
    """Generator(bg_dim, img_dim, obj_dim, img_cond_dim, obj_cond_dim, conv_dim, repeat_num, spade_layers,
    attn_layers) of generator.py:320-345."""

    def __init__(self, gen_name, bg_dim, img_dim, obj_dim, img_cond_dim, obj_cond_dim, conv_dim=64, repeat_num=6):
        if gen_name not in GEN_VARIANTS:
            raise ValueError('Network %s not recognized.' % gen_name)
        self.gen_name = gen_name
        self.spade_layers, self.attn_layers = [list(v) for v in GEN_VARIANTS[gen_name]]
        self.bg_dim, self.img_dim, self.obj_dim = bg_dim, img_dim, obj_dim
        self.img_cond_dim, self.obj_cond_dim = img_cond_dim, obj_cond_dim
        self.conv_dim, self.repeat_num, self.n_down = conv_dim, repeat_num, 3

    def num_channel(self, layer):
        """ResUnetGenerator.num_channel (generator.py:157,170,182,189)."""
        return self.conv_dim * 2 ** min(layer, self.n_down)


def _bg(s, c):
    p, d = 'bg_model.model', c.conv_dim
    s.conv(p + '.0', d, c.bg_dim, 7)
    s.affine(p + '.1', d)
    idx, ch = 3, d
    for _ in range(c.n_down):
        s.conv(p + '.%d' % idx, 2 * ch, ch, 3)
        s.affine(p + '.%d' % (idx + 1), 2 * ch)
        idx, ch = idx + 3, 2 * ch
    for _ in range(c.repeat_num):
        s.resblock(p + '.%d' % idx, ch)
        idx += 1
    for _ in range(c.n_down):
        s.convT(p + '.%d' % idx, ch, ch // 2, 3)
        s.affine(p + '.%d' % (idx + 1), ch // 2)
        idx, ch = idx + 3, ch // 2
    s.conv(p + '.%d' % idx, 3, ch, 7)


def _unet(s, c, p, c_dim, s_dim, on_obj):
    d, sl = c.conv_dim, c.spade_layers
    s.conv(p + '.encoders.0.0', d, c_dim, 7)
    s.affine(p + '.encoders.0.1', d)
    ch = d
    for i in range(1, c.n_down + 1):
        if sl[0]:
            s.conv(p + '.encoders.%d.conv' % i, 2 * ch, ch, 3)
            s.spade(p + '.encoders.%d.norm' % i, 2 * ch, s_dim)
        else:
            s.conv(p + '.encoders.%d.0' % i, 2 * ch, ch, 3)
            s.affine(p + '.encoders.%d.1' % i, 2 * ch)
        ch *= 2
    for i in range(c.repeat_num):
        if (sl[1] if i < c.repeat_num // 2 else sl[2]):
            s.spade_resblock(p + '.resnets.%d' % i, ch, s_dim)
        else:
            s.resblock(p + '.resnets.%d' % i, ch)
    top = ch
    for i in range(c.n_down):           # all decoders are registered before the skippers (generator.py:214-215)
        if sl[3]:
            s.convT(p + '.decoders.%d.conv' % i, ch, ch // 2, 3)
            s.spade(p + '.decoders.%d.norm' % i, ch // 2, s_dim)
        else:
            s.convT(p + '.decoders.%d.0' % i, ch, ch // 2, 3)
            s.affine(p + '.decoders.%d.1' % i, ch // 2)
        ch //= 2
    ch = top
    for i in range(c.n_down):
        s.conv(p + '.skippers.%d.0' % i, ch // 2, ch, 3)
        s.affine(p + '.skippers.%d.1' % i, ch // 2)
        ch //= 2
    s.conv(p + '.img_reg.0', 3, ch, 7)
    if not on_obj:
        s.conv(p + '.attetion_reg_hand.0', 1, ch, 7)
        s.conv(p + '.attetion_reg_bg.0', 1, 2 * ch, 7)


def generator_schema(cfg):
    s = Schema()
    _bg(s, cfg)
    _unet(s, cfg, 'obj_model', cfg.obj_dim, cfg.obj_cond_dim, True)
    _unet(s, cfg, 'src_model', cfg.img_dim, cfg.img_cond_dim, False)
    _unet(s, cfg, 'tsf_model', cfg.img_dim, cfg.img_cond_dim, False)
    for layer in cfg.attn_layers:
        c = cfg.num_channel(layer)
        s.conv('attn_%d.fully_connect_layer.0' % layer, 128, 2 * c, 5, bias=True)
        s.conv('attn_%d.fully_connect_layer.2' % layer, 25, 128, 1, bias=True)
    return s


def discriminator_schema(input_nc, ndf=64, n_layers=4):
    s = Schema()
    s.conv('model.0', ndf, input_nc, 4, bias=True)
    idx, prev = 2, 1
    for n in range(1, n_layers):
        mult = min(2 ** n, 8)
        s.conv('model.%d' % idx, ndf * mult, ndf * prev, 4, bias=True)
        idx, prev = idx + 3, mult
    mult = min(2 ** n_layers, 8)
    s.conv('model.%d' % idx, ndf * mult, ndf * prev, 4, bias=True)
    idx += 3
    s.conv('model.%d' % idx, 1, ndf * mult, 4, bias=True)
    return s


VGG_LAYERS = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512]
VGG_SLICE_ENDS = [2, 7, 12, 21, 30]       # vgg19.py:62


def vgg_schema():
    s, idx, cin, sl = Schema(), 0, 3, 1
    for v in VGG_LAYERS:
        while idx >= VGG_SLICE_ENDS[sl - 1]:
            sl += 1
        if v == 'M':
            idx += 1
            continue
        s.conv('slice%d.%d' % (sl, idx), v, cin, 3, bias=True)
        cin, idx = v, idx + 2
    return s
