"""HOGAN generator on the HIP operators (NHWC, fp32 activations in HBM).

Mirrors the reference's ``Generator`` interface (models/networks/generator.py:318-491): same constructor
arguments, ``forward`` signature, ``init_weights`` and parameter names; the computation is a functional
pass over a flat parameter store, every operator a hand-written gfx950 kernel (hoig_amd/ops.py).
"""
import os

import torch

from ... import ops
from ..._lib import ACT_NONE, ACT_RELU, ACT_TANH, ACT_SIGMOID
from ...nn import ParamTree
from .schema import GeneratorConfig, generator_schema


def to_nhwc(t):
    """NCHW-shaped CUDA tensor -> contiguous NHWC tensor (free when `t` already is a channels-last view)."""
    if t is None:
        return None
    v = t.permute(0, 2, 3, 1)
    if v.is_contiguous():
        return v
    return ops.nchw_to_nhwc(t)


def as_nchw(t):
    """NHWC tensor presented with the reference's NCHW shape (a view, no copy)."""
    return t.permute(0, 3, 1, 2)


# HOIG_STREAMS=0: every chain of the step on the caller's stream -- per-kernel profiles only add up that
# way.  Default: the sub-networks that do not read each other run on HIP streams of their own ('Concurrent
# chains'); inside a captured step those forks are the hipGraph's parallel branches.
_FORK_STREAMS = os.environ.get('HOIG_STREAMS', '1') == '1'


def forks_streams():
    """True if Generator.forward runs bg_model / obj_model / src_model on branch streams (their backward then runs there too)."""
    return _FORK_STREAMS


class Generator(ParamTree):
    def __init__(self, bg_dim, img_dim, obj_dim, img_cond_dim=0, obj_cond_dim=0, conv_dim=64, repeat_num=6,
                 gen_name='generator_spade_attn', device=None):
        self.cfg = GeneratorConfig(gen_name, bg_dim, img_dim, obj_dim, img_cond_dim, obj_cond_dim, conv_dim, repeat_num)
        sch = generator_schema(self.cfg)
        device = device if device is not None else torch.device('cuda', torch.cuda.current_device())
        # weights stored as the two halves of their input channels (hoig_amd.nn): the attention's first conv ([target | source],
        # two 5x5 convolutions over different tensors) and the background-mask heads over cat[x, obj features]
        # (generator.py:315: evaluated as conv(x) + conv(y), which needs no concatenated 128-channel full-resolution tensor)
        split = ['attn_%d.fully_connect_layer.0.weight' % l for l in self.cfg.attn_layers]
        split += ['%s.attetion_reg_bg.0.weight' % m for m in ('src_model', 'tsf_model')]
        # obj_model's last feature map feeds its own image head and the y halves of BOTH background-mask heads (generator.py:
        # 311-315, 449-456): the three 7x7 weights are kept back to back so they can run as one convolution (heads_y, below)
        heads_y = ['obj_model.img_reg.0.weight'] + ['%s.attetion_reg_bg.0.weight#s' % m for m in ('src_model', 'tsf_model')]
        super().__init__(sch.shapes, device, sch.transposed, split, adjacent=[heads_y])
        self._name = 'generator'
        self._seg_cache = {}
        # the three heads that read the decoder's last feature map -- img_reg (3), attetion_reg_hand (1) and the x half of
        # attetion_reg_bg (1) -- are registered one after the other (generator.py:219-235), so their packed weights are ONE
        # (5, C, 7, 7) block of the flat store: evaluate them as one convolution (ops.conv_heads)
        for m in ('src_model', 'tsf_model'):
            names = [m + '.img_reg.0.weight', m + '.attetion_reg_hand.0.weight', m + '.attetion_reg_bg.0.weight#t']
            self.fuse_conv_weights(m + '.heads_x.weight', names)
        self.fuse_conv_weights('obj_model.heads_y.weight', heads_y)

    @property
    def name(self):
        return self._name

    def _branch_streams(self, device):
        if getattr(self, '_streams', None) is None:
            self._streams = tuple(ops.new_stream(device, role) for role in ('g_bg', 'g_obj', 'g_src'))
        return self._streams

    # ---- building blocks -------------------------------------------------------------------
    def _conv(self, x, name, stride=1, pad=1, act=ACT_NONE, to_norm=False):
        # to_norm: the output goes straight into an instance norm, which cancels the bias (ops.conv2d dead_bias)
        return ops.conv2d(x, self.P[name + '.weight'], self.P.get(name + '.bias'), stride, pad, act, dead_bias=to_norm)

    def _convT(self, x, name):
        return ops.conv_transpose2d(x, self.P[name + '.weight'], norm_next=True)      # (always followed by a norm: generator.py:118,201)

    def _in(self, x, name, act=ACT_NONE, residual=None):
        return ops.instance_norm(x, self.P[name + '.weight'], self.P[name + '.bias'], act=act, residual=residual)

    def _seg_at(self, seg, h, w):
        key = (id(seg), h, w)
        if key not in self._seg_cache:
            self._seg_cache[key] = ops.resize_nearest(seg, h, w)          # spade.py:30
        return self._seg_cache[key]

    def _spade(self, x, seg, name, act, fork=False):
        """spade.py:25-38 (+ the ReLU that always follows it, generator.py:66-67,88).  fork=True -> (y, x') for an x that has a
        second reader, which must read x' (ops.spade_norm_fused)."""
        s = self._seg_at(seg, x.shape[1], x.shape[2])
        actv = self._conv(s, name + '.mlp_shared.0', act=ACT_RELU)
        if name + '.mlp_gb.weight' in self.F:
            # mlp_gamma and mlp_beta are two 3x3 convs over the same activation (spade.py:33-34): run them as ONE conv with
            # 2C outputs (their weights sit back to back in the flat store) and modulate from the [.,2C] result in place
            gb = ops.conv2d(actv, self.F[name + '.mlp_gb.weight'], self.F[name + '.mlp_gb.bias'], 1, 1)
            return ops.spade_norm_fused(x, gb, act=act, fork=fork)
        gamma = self._conv(actv, name + '.mlp_gamma')
        beta = self._conv(actv, name + '.mlp_beta')
        y = ops.spade_norm(x, gamma, beta, act=act)
        return (y, x) if fork else y

    def _conv_fork(self, x, name, stride=1, pad=1):
        """-> (conv(x), x'): x has a second reader (a skip connection), which must read x' (ops.conv2d_fork)."""
        return ops.conv2d_fork(x, self.P[name + '.weight'], self.P.get(name + '.bias'), stride, pad, dead_bias=True)

    def _conv_in_relu(self, x, name, stride=1, pad=1, transposed=False, fork=False):
        if fork:
            h, x = self._conv_fork(x, name + '.0', stride, pad)
            return self._in(h, name + '.1', act=ACT_RELU), x
        h = self._convT(x, name + '.0') if transposed else self._conv(x, name + '.0', stride, pad, to_norm=True)
        return self._in(h, name + '.1', act=ACT_RELU)

    def _resblock(self, x, name):                                          # generator.py:9-32
        if not torch.is_grad_enabled():
            # inference: the first norm + ReLU are applied by the second convolution's loader (ops.conv2d_after_norm)
            with ops.small_map_sums():             # (the loader norm below needs h's statistics: the convolution's epilogue leaves them)
                h = self._conv(x, name + '.main.0', to_norm=True)
            y = ops.conv2d_after_norm(h, self.P[name + '.main.1.weight'], self.P[name + '.main.1.bias'], self.P[name + '.main.3.weight'],
                                      self.P.get(name + '.main.3.bias'), norm_next=True)
            if y is None:
                y = self._conv(self._in(h, name + '.main.1', act=ACT_RELU), name + '.main.3', to_norm=True)
            return self._in(y, name + '.main.4', residual=x)
        # (x has two readers, the first conv and the skip: conv2d_fork routes the skip's gradient through the conv's backward)
        h, x = self._conv_fork(x, name + '.main.0')
        h = self._in(h, name + '.main.1', act=ACT_RELU)
        return self._in(self._conv(h, name + '.main.3', to_norm=True), name + '.main.4', residual=x)

    def _spade_resblock(self, x, seg, name):                               # generator.py:63-71
        h, x = self._spade(x, seg, name + '.norm_0', ACT_RELU, fork=True)      # (the skip below reads x through the norm's fork)
        dx = self._conv(h, name + '.conv_0', to_norm=True)
        dx = self._conv(self._spade(dx, seg, name + '.norm_1', ACT_RELU), name + '.conv_1')
        return ops.add(x, dx)

    # ---- src_model's and tsf_model's residual blocks in lock-step (ops.conv2d_pair): the two sub-networks have one architecture and
    # separate weights (generator.py:379-464), 8 images each at the bench's batch -- two half-chip launches per layer on two streams.
    # Their 3x3 512 -> 512 convolutions run as GROUPED launches on the caller's stream instead (one grid over both problems: forward,
    # data gradient, weight gradient); everything between them -- norms, SPADE's small convolutions -- stays on the two streams.
    # `sy` = (on_src, to_main, to_src): the src stream's context and the two hand-overs (no-ops when everything is on one stream).
    def _resblock_pair(self, sx, tx, ns, nt, sy):
        on_src, to_main, to_src = sy
        P = self.P
        hs, ht, sx, tx = ops.conv2d_pair(to_main(sx), tx, P[ns + '.main.0.weight'], P[nt + '.main.0.weight'], dead_bias=True, fork=True)
        to_src(hs); to_src(sx)
        with on_src():
            hs = self._in(hs, ns + '.main.1', act=ACT_RELU)
        ht = self._in(ht, nt + '.main.1', act=ACT_RELU)
        hs, ht = ops.conv2d_pair(to_main(hs), ht, P[ns + '.main.3.weight'], P[nt + '.main.3.weight'], dead_bias=True)
        to_src(hs)
        with on_src():
            sx = self._in(hs, ns + '.main.4', residual=sx)
        return sx, self._in(ht, nt + '.main.4', residual=tx)

    def _spade_resblock_pair(self, sx, tx, segs, segt, ns, nt, sy):
        on_src, to_main, to_src = sy
        P = self.P
        with on_src():
            hs, sx = self._spade(sx, segs, ns + '.norm_0', ACT_RELU, fork=True)
        ht, tx = self._spade(tx, segt, nt + '.norm_0', ACT_RELU, fork=True)
        ds, dt = ops.conv2d_pair(to_main(hs), ht, P[ns + '.conv_0.weight'], P[nt + '.conv_0.weight'], P.get(ns + '.conv_0.bias'),
                                 P.get(nt + '.conv_0.bias'), dead_bias=True)
        to_src(ds)
        with on_src():
            hs = self._spade(ds, segs, ns + '.norm_1', ACT_RELU)
        ht = self._spade(dt, segt, nt + '.norm_1', ACT_RELU)
        ds, dt = ops.conv2d_pair(to_main(hs), ht, P[ns + '.conv_1.weight'], P[nt + '.conv_1.weight'], P.get(ns + '.conv_1.bias'),
                                 P.get(nt + '.conv_1.bias'))
        to_src(ds)
        with on_src():
            sx = ops.add(sx, ds)
        return sx, ops.add(tx, dt)

    def _resnet_pair(self, sx, tx, segs, segt, i, sy):
        """-> (src_model's, tsf_model's residual block i), or None where the grouped kernels do not cover the layer."""
        c = self.cfg
        ns, nt = 'src_model.resnets.%d' % i, 'tsf_model.resnets.%d' % i
        spade = c.spade_layers[1] if i < c.repeat_num // 2 else c.spade_layers[2]
        wkey = '.conv_0.weight' if spade else '.main.0.weight'
        if not ops.pair_ok(sx, tx, self.P[ns + wkey], self.P[nt + wkey]):
            return None
        if spade:
            return self._spade_resblock_pair(sx, tx, segs, segt, ns, nt, sy)
        return self._resblock_pair(sx, tx, ns, nt, sy)

    def _spade_block(self, x, seg, name, down, fork=False):                # generator.py:74-90
        if fork:
            h, x = self._conv_fork(x, name + '.conv', stride=2)
            return self._spade(h, seg, name + '.norm', ACT_RELU), x
        h = self._conv(x, name + '.conv', stride=2, to_norm=True) if down else self._convT(x, name + '.conv')
        return self._spade(h, seg, name + '.norm', ACT_RELU)

    def _bg_net_steps(self, x, out):                                       # generator.py:93-135
        """bg_model as a Python generator that yields after every level / block (out[0] = the result): forward_nhwc advances
        the sub-networks round-robin, so that the host issues -- and autograd later replays -- the concurrent chains interleaved
        instead of one whole chain after the other."""
        c, p = self.cfg, 'bg_model.model'
        x = self._in(self._conv(x, p + '.0', pad=3, to_norm=True), p + '.1', act=ACT_RELU)
        yield
        idx = 3
        for _ in range(c.n_down):
            x = self._in(self._conv(x, p + '.%d' % idx, stride=2, to_norm=True), p + '.%d' % (idx + 1), act=ACT_RELU)
            idx += 3
            yield
        for _ in range(c.repeat_num):
            x = self._resblock(x, p + '.%d' % idx)
            idx += 1
            yield
        for _ in range(c.n_down):
            x = self._in(self._convT(x, p + '.%d' % idx), p + '.%d' % (idx + 1), act=ACT_RELU)
            idx += 3
            yield
        out[0] = self._conv(x, p + '.%d' % idx, pad=3, act=ACT_TANH)

    def _bg_net(self, x):
        out = [None]
        for _ in self._bg_net_steps(x, out):
            pass
        return out[0]

    def _enc_level(self, x, seg, p, i):
        """-> (encoder level i of x, x'): x is also the skip connection of a decoder level, which must read x'."""
        if self.cfg.spade_layers[0]:
            return self._spade_block(x, seg, p + '.encoders.%d' % i, True, fork=True)
        return self._conv_in_relu(x, p + '.encoders.%d' % i, stride=2, fork=True)

    def _resnet(self, x, seg, p, i):
        c = self.cfg
        if (c.spade_layers[1] if i < c.repeat_num // 2 else c.spade_layers[2]):
            return self._spade_resblock(x, seg, p + '.resnets.%d' % i)
        return self._resblock(x, p + '.resnets.%d' % i)

    def _decode_level(self, x, enc, seg, p, i):                            # generator.py:298-309
        nd = self.cfg.n_down
        name = p + '.skippers.%d' % i
        if self.cfg.spade_layers[3]:
            x = self._spade_block(x, seg, p + '.decoders.%d' % i, False)
        elif not torch.is_grad_enabled() and self.P.get(name + '.0.bias') is None:
            # inference: the up-sampled operand stays raw; its norm + ReLU are applied by the skip convolution's loader
            dn = p + '.decoders.%d' % i
            xr = self._convT(x, dn + '.0')
            y = ops.conv2d_after_norm(xr, self.P[dn + '.1.weight'], self.P[dn + '.1.bias'], self.P[name + '.0.weight'], None,
                                      first=enc[nd - 1 - i], norm_next=True)
            if y is not None:
                return self._in(y, name + '.1', act=ACT_RELU)
            x = self._in(xr, dn + '.1', act=ACT_RELU)
        else:
            x = self._conv_in_relu(x, p + '.decoders.%d' % i, transposed=True)
        # cat[skip, up] -> conv3x3 -> IN -> ReLU: the convolution reads the two tensors directly (ops.conv2d_cat2)
        if self.P.get(name + '.0.bias') is None:
            return self._in(ops.conv2d_cat2(enc[nd - 1 - i], x, self.P[name + '.0.weight'], norm_next=True), name + '.1', act=ACT_RELU)
        return self._conv_in_relu(ops.cat_channels([enc[nd - 1 - i], x]), name)

    def _decode(self, x, enc, seg, p):
        for i in range(self.cfg.n_down):
            x = self._decode_level(x, enc, seg, p, i)
        return x

    def _unet_steps(self, x, seg, p, out):                                 # generator.py:261-283 (see _bg_net_steps)
        e = self._conv_in_relu(x, p + '.encoders.0', pad=3)
        yield
        enc = [e]
        for i in range(1, self.cfg.n_down + 1):
            e, enc[-1] = self._enc_level(e, seg, p, i)
            enc.append(e)
            yield
        for i in range(self.cfg.repeat_num):
            e = self._resnet(e, seg, p, i)
            yield
        for i in range(self.cfg.n_down):
            e = self._decode_level(e, enc, seg, p, i)
            if i + 1 < self.cfg.n_down:
                yield
        out[0] = e

    def _unet(self, x, seg, p):
        out = [None]
        for _ in self._unet_steps(x, seg, p, out):
            pass
        return out[0]

    # ---- feature warping (generator.py:466-491) ------------------------------------------------
    def _tscale(self, T, h):
        key = ('T', id(T), h)
        if key not in self._seg_cache:
            self._seg_cache[key] = ops.resize_bilinear_ac(T, h, h)         # resize_trans: size=(h, h)
        return self._seg_cache[key]

    # The source features of a level have up to four readers (the next src level, the skip, the attention's source convolution
    # and its weighted average) and the target features two (the attention and the sum that follows it).  Each reader but the
    # last hands the tensor on as a pass-through output (ops.conv2d_fork), so autograd sees a CHAIN of single consumers and every
    # backward kernel adds the gradient that arrived behind it -- no gradient sums by the autograd engine.
    def _attn_source(self, x, layer):
        """-> (Gs, x'): the source half of layer `layer`'s attention (ops.attn_source_conv; None where the layer warps by
        grid_sample), evaluated by the caller on src_model's stream right after the source features exist, and the source
        features for their later readers."""
        if layer not in self.cfg.attn_layers:
            return None, x
        return ops.attn_source_conv(x, self.P['attn_%d.fully_connect_layer.0.weight#s' % layer], fork=True)

    def _transform(self, x, T, layer, y=None, gs=None):
        """-> (x warped to the target frame, x', y'): x', y' are x and y for their later readers."""
        h = x.shape[1]
        ts = self._tscale(T, h)
        if layer in self.cfg.attn_layers:
            key = ('F', id(T), h)
            if key not in self._seg_cache:
                self._seg_cache[key] = ops.attn_flow(ts)
            p = 'attn_%d.fully_connect_layer' % layer
            return ops.local_attention(x, y, self._seg_cache[key], self.P[p + '.0.weight#t'], self.P[p + '.0.weight#s'],
                                       self.P[p + '.0.bias'], self.P[p + '.2.weight'], self.P[p + '.2.bias'], gs=gs, fork=True)
        return ops.grid_sample(x, ts), x, y

    # ---- public forward: reference signature (generator.py:347-376), NCHW in / NCHW-shaped out ---
    def forward(self, bg_inputs, src_obj_inputs, tsf_obj_inputs, src_hand_inputs, tsf_hand_inputs, T,
                src_obj_conds=None, src_hand_conds=None, tsf_obj_conds=None, tsf_hand_conds=None,
                src_armask=None, tsf_armask=None):
        if src_obj_conds is None or src_hand_conds is None or tsf_obj_conds is None or tsf_hand_conds is None:
            raise NotImplementedError('Generator.forward without conds (use_spade=False) is unreachable in the reference: '
                                      'Trainer.forward passes arguments in an order that mismatches obj_dim/img_dim '
                                      '(trainer.py:395-398 vs :263-264)')
        outs = self.forward_nhwc(to_nhwc(bg_inputs), to_nhwc(src_obj_inputs), to_nhwc(tsf_obj_inputs),
                                 to_nhwc(src_hand_inputs), to_nhwc(tsf_hand_inputs), T.contiguous(),
                                 to_nhwc(src_obj_conds), to_nhwc(src_hand_conds), to_nhwc(tsf_obj_conds),
                                 to_nhwc(tsf_hand_conds), to_nhwc(src_armask), to_nhwc(tsf_armask))
        return tuple(as_nchw(o) for o in outs)

    @staticmethod
    def stack_inputs(bg, src_obj, tsf_obj, src_hand_c, tsf_hand_c, src_obj_c, tsf_obj_c, src_armask=None, tsf_armask=None):
        """The batch-stacked inputs of the two shared-weight sub-networks (bg_model over [src | tsf] backgrounds, obj_model over
        [src | tsf] objects and their condition maps).  They depend on the batch only, so a caller that steps several times on a
        batch -- or stages its inputs once per iteration (Trainer.set_input) -- makes them once and passes them as `stacked`."""
        src_bg_in = [bg, src_hand_c] + ([src_armask] if src_armask is not None else [])
        tsf_bg_in = [bg, tsf_hand_c] + ([tsf_armask] if tsf_armask is not None else [])
        return (torch.cat([ops.cat_channels(src_bg_in), ops.cat_channels(tsf_bg_in)], dim=0),
                torch.cat([src_obj, tsf_obj], dim=0), torch.cat([src_obj_c, tsf_obj_c], dim=0))

    def forward_nhwc(self, bg, src_obj, tsf_obj, src_hand, tsf_hand, T, src_obj_c, src_hand_c, tsf_obj_c, tsf_hand_c,
                     src_armask=None, tsf_armask=None, stacked=None):
        c = self.cfg
        self._seg_cache = {}
        ops.attn_index_clear()
        # bg_model runs on the src and tsf inputs with SHARED weights (generator.py:367-369): one pass over the two
        # batches stacked (instance norm is per sample, so the result is identical) -> twice the tiles per launch at the
        # 32x32 bottleneck and one weight-gradient accumulation instead of two
        nb = bg.shape[0]
        # The three sub-networks that do not depend on each other until the heads -- bg_model, obj_model, and the src/tsf
        # pair -- run on separate HIP streams: their kernels fill each other's tails (autograd replays each branch's
        # backward on its own stream too).  The weight split is refreshed first, on the main stream.
        main = torch.cuda.current_stream()
        fork = _FORK_STREAMS and bg.is_cuda
        self.refresh_planes()
        if stacked is None:
            stacked = self.stack_inputs(bg, src_obj, tsf_obj, src_hand_c, tsf_hand_c, src_obj_c, tsf_obj_c, src_armask, tsf_armask)
        bg_in, obj_in, obj_c = stacked
        if fork:
            s_bg, s_obj, s_src = self._branch_streams(bg.device)
            s_bg.wait_stream(main)
            s_obj.wait_stream(main)
            s_src.wait_stream(main)
            for st_, role in ((s_bg, 'g_bg'), (s_obj, 'g_obj'), (s_src, 'g_src')):      # (test hook: ops.test_delay)
                with torch.cuda.stream(st_):
                    ops.test_delay(role)
            bg_out, obj_out = [None], [None]
            branches = [(self._bg_net_steps(bg_in, bg_out), s_bg), (self._unet_steps(obj_in, obj_c, 'obj_model', obj_out), s_obj)]
            for t_, st_ in ((bg_in, s_bg), (obj_in, s_obj), (obj_c, s_obj)):
                ops.cross_stream(t_, st_)
        else:
            branches = []
            bg_both = self._bg_net(bg_in)

        def advance(all_the_way=False):                      # one level / block of bg_model and obj_model on their streams
            for gen, st_ in branches:
                with torch.cuda.stream(st_):
                    if all_the_way:
                        for _ in gen:
                            pass
                    else:
                        next(gen, None)

        # infer_front (generator.py:379-464).  src_model never reads tsf_model's features, so it runs AHEAD on a stream of its own;
        # tsf_model waits, level by level, for the src features it warps in.
        import contextlib
        fork_src = fork
        on_src = (lambda: torch.cuda.stream(s_src)) if fork_src else contextlib.nullcontext

        def src_ready(t_):                                   # the main stream may read a tensor made on the src stream
            if fork_src:
                main.wait_stream(s_src)
                ops.cross_stream(t_, main)
            return t_

        with on_src():
            sx = self._conv_in_relu(src_hand, 'src_model.encoders.0', pad=3)
        tx = self._conv_in_relu(tsf_hand, 'tsf_model.encoders.0', pad=3)
        advance()
        s_enc, t_enc = [sx], [tx]
        for i in range(1, c.n_down + 1):
            with on_src():
                sx, s_enc[-1] = self._enc_level(sx, src_hand_c, 'src_model', i)
                gs, sx = self._attn_source(sx, i)           # (the attention's source convolution rides on the src stream too)
            tx, t_enc[-1] = self._enc_level(tx, tsf_hand_c, 'tsf_model', i)
            warped, sx, tx = self._transform(src_ready(sx), T, i, y=tx, gs=None if gs is None else src_ready(gs))
            tx = ops.add(tx, warped)
            s_enc.append(sx)
            t_enc.append(tx)
            advance()
        def to_src(t_):                                      # the src stream may read a tensor made on the main stream
            if fork_src:
                s_src.wait_stream(main)
                ops.cross_stream(t_, s_src)
            return t_

        for i in range(c.repeat_num):
            pair = self._resnet_pair(sx, tx, src_hand_c, tsf_hand_c, i, (on_src, src_ready, to_src))
            if pair is not None:
                sx, tx = pair
                with on_src():
                    gs, sx = self._attn_source(sx, i + c.n_down + 1)
            else:
                with on_src():
                    sx = self._resnet(sx, src_hand_c, 'src_model', i)
                    gs, sx = self._attn_source(sx, i + c.n_down + 1)
                tx = self._resnet(tx, tsf_hand_c, 'tsf_model', i)
            warped, sx, tx = self._transform(src_ready(sx), T, i + c.n_down + 1, y=tx, gs=None if gs is None else src_ready(gs))
            tx = ops.add(tx, warped)
            advance()
        advance(True)                                        # (their decoders: issued before the join below)
        if fork:
            with torch.cuda.stream(s_bg):
                bg_both = ops.delay_backward(bg_out[0], 'g_bg')
            with torch.cuda.stream(s_obj):
                obj_both = ops.delay_backward(obj_out[0], 'g_obj')

        # obj_model likewise serves both the src and the tsf object (generator.py:449-450): one stacked pass
        if not fork:
            obj_both = self._unet(obj_in, obj_c, 'obj_model')
        # The object branch's feature map has THREE readers -- its image head over both halves, and the y half of the src / tsf
        # background-mask head over one half each.  Fused: one 5-channel 7x7 convolution over the stacked map (the mask columns of
        # the other half are computed and dropped: 2 of 5 columns of a convolution that is bound by reading its input), so the
        # backward is ONE data gradient instead of three plus two full-resolution zero-padded slice gradients and their sums.
        # It runs where the map was made (the object branch's stream), ahead of the joins below.
        fused_y = self.F.get('obj_model.heads_y.weight') if obj_both.is_cuda else None
        if fused_y is not None:
            with (torch.cuda.stream(s_obj) if fork else contextlib.nullcontext()):
                obj_o, my_src, my_tsf = ops.conv_heads(obj_both, fused_y, (3, 1, 1), (ACT_TANH, ACT_NONE, ACT_NONE))
        if fork:
            main.wait_stream(s_bg)
            main.wait_stream(s_obj)
            ops.cross_stream(bg_both, main)
            ops.cross_stream(obj_both, main)
            if fused_y is not None:
                ops.cross_stream(obj_o, main)
                ops.cross_stream(my_tsf, main)
                ops.cross_stream(my_src, s_src)
        src_img_bg, tsf_img_bg = bg_both[:nb], bg_both[nb:]
        sy, ty = obj_both[:nb], obj_both[nb:]
        if fork_src:
            s_src.wait_stream(s_obj)                         # the src heads read the object branch's output
            ops.cross_stream(obj_both, s_src)
        with on_src():
            sx = self._decode(sx, s_enc, src_hand_c, 'src_model')
        tx = self._decode(tx, t_enc, tsf_hand_c, 'tsf_model')

        if fused_y is not None:
            my = {'src_model': my_src[:nb], 'tsf_model': my_tsf[nb:]}

        def regress(x, y, p):                                              # generator.py:311-315
            fused = self.F.get(p + '.heads_x.weight')
            if fused is not None and x.is_cuda:
                img, mh, mbt = ops.conv_heads(x, fused, (3, 1, 1), (ACT_TANH, ACT_SIGMOID, ACT_NONE))
            else:
                img = self._conv(x, p + '.img_reg.0', pad=3, act=ACT_TANH)
                mh = self._conv(x, p + '.attetion_reg_hand.0', pad=3, act=ACT_SIGMOID)
                mbt = ops.conv2d(x, self.P[p + '.attetion_reg_bg.0.weight#t'], None, 1, 3)
            mby = my[p] if fused_y is not None else ops.conv2d(y, self.P[p + '.attetion_reg_bg.0.weight#s'], None, 1, 3)
            mb = ops.add_act(mbt, mby, ACT_SIGMOID)
            return img, mh, mb

        with on_src():
            src_hand_o, src_mask_hand, src_mask_bg = regress(sx, sy, 'src_model')
            src_hand_o = ops.delay_backward(src_hand_o, 'g_src')
        tsf_hand_o, tsf_mask_hand, tsf_mask_bg = regress(tx, ty, 'tsf_model')
        if fork_src:
            main.wait_stream(s_src)
            for t_ in (src_hand_o, src_mask_hand, src_mask_bg):
                ops.cross_stream(t_, main)
        if fused_y is None:
            obj_o = self._conv(obj_both, 'obj_model.img_reg.0', pad=3, act=ACT_TANH)
        src_obj_o, tsf_obj_o = obj_o[:nb], obj_o[nb:]
        self._seg_cache = {}
        return (src_img_bg, tsf_img_bg, src_obj_o, src_hand_o, src_mask_bg, src_mask_hand,
                tsf_obj_o, tsf_hand_o, tsf_mask_bg, tsf_mask_hand)
