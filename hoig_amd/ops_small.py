"""Element-wise, channel-bookkeeping and sampling operators (split off hoig_amd/ops.py in round 6; re-exported there)."""
import contextlib
import ctypes

import torch
from torch.autograd import Function

from . import _lib as L
from ._lib import call, ConvDesc
from . import ops as _o          # (names of the core module are read at call time: _chk, _p, _st, conv2d, packed_strides, precision)

# ------------------------------------------------------------------------------------------------- small ops
class _Add(Function):
    @staticmethod
    def forward(ctx, a, b):
        _o._chk(a); _o._chk(b)
        assert a.shape == b.shape and a.is_contiguous() and b.is_contiguous()
        y = torch.empty_like(a)
        call('hoig_add', _o._p(a), _o._p(b), _o._p(y), a.numel(), _o._st())
        return y

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


def add(a, b):
    return _Add.apply(a, b)


class _AddAct(Function):
    """act(a + b) -- the epilogue of a convolution evaluated as the sum of two convolutions over the halves of its input."""

    @staticmethod
    def forward(ctx, a, b, act, slope):
        _o._chk(a); _o._chk(b)
        assert a.shape == b.shape and a.is_contiguous() and b.is_contiguous()
        y = torch.empty_like(a)
        call('hoig_add_act', _o._p(a), _o._p(b), _o._p(y), act, slope, a.numel(), _o._st())
        ctx.cfg = (act, slope)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, = ctx.saved_tensors
        act, slope = ctx.cfg
        g = torch.empty_like(y)
        call('hoig_act_bwd', _o._p(y), _o._p(dy.contiguous()), _o._p(g), act, slope, y.numel(), _o._st())
        return g, g, None, None


def add_act(a, b, act, slope=0.0):
    return _AddAct.apply(a, b, act, slope)


def _copy_channels(x, y, x_off, y_off, n, accumulate=False):
    npix = x.numel() // x.shape[-1]
    call('hoig_copy_channels', _o._p(x), _o._p(y), npix, x.shape[-1], x_off, y.shape[-1], y_off, n, 1 if accumulate else 0,
         _o._st())


class _Cat(Function):
    @staticmethod
    def forward(ctx, *xs):
        for t in xs:
            _o._chk(t)
            assert t.is_contiguous()
        cs = [t.shape[-1] for t in xs]
        y = torch.empty(xs[0].shape[:-1] + (sum(cs),), dtype=xs[0].dtype, device=xs[0].device)
        if len(xs) == 2:
            call('hoig_cat2_channels', _o._p(xs[0]), cs[0], _o._p(xs[1]), cs[1], _o._p(y), y.numel() // y.shape[-1], _o._st())
        else:
            off = 0
            for t, c in zip(xs, cs):
                _copy_channels(t, y, 0, off, c)
                off += c
        ctx.cs = cs
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        outs, off = [], 0
        for i, c in enumerate(ctx.cs):
            if ctx.needs_input_grad[i]:
                # a strided VIEW: where autograd sums it with another gradient of the same tensor (an encoder output also
                # feeds the next level) the add reads it in place and the slice copy never happens; single consumers
                # make it contiguous themselves
                outs.append(dy[..., off:off + c])
            else:
                outs.append(None)
            off += c
        return tuple(outs)


def cat_channels(xs):
    return _Cat.apply(*xs)


class _PadChannels(Function):
    """x [.., C] -> [.., C'] with zeros behind (C' > C): puts a 19- / 24-channel tensor (the discriminator's input, discriminator.py:29)
    on the 16-bit convolution kernels, which want multiples of 32 channels."""

    @staticmethod
    def forward(ctx, x, c_to):
        _o._chk(x)
        assert x.is_contiguous() and c_to > x.shape[-1]
        y = torch.zeros(x.shape[:-1] + (c_to,), dtype=x.dtype, device=x.device)
        _copy_channels(x, y, 0, 0, x.shape[-1])
        ctx.c = x.shape[-1]
        return y

    @staticmethod
    def backward(ctx, dy):
        return dy[..., :ctx.c], None          # (a strided view: see _Cat.backward)


class _PadConvIn(Function):
    """conv weight (Co, Ci, R, S) over packed storage -> (Co, Ci', R, S), zero input channels behind; the gradient of the real channels
    goes back to the parameter (autograd adds it into its flat gradient view)."""

    @staticmethod
    def forward(ctx, w, c_to):
        co, ci, r, s_ = w.shape
        assert tuple(w.stride()) == _o.packed_strides(w.shape, False) and c_to > ci
        out = torch.empty_strided((co, c_to, r, s_), _o.packed_strides((co, c_to, r, s_), False), dtype=w.dtype, device=w.device)
        out.zero_()
        out[:, :ci].copy_(w)
        ctx.ci = ci
        return out

    @staticmethod
    def backward(ctx, dw):
        return dw[:, :ctx.ci], None


def conv2d_padded_in(x, w, b, stride, pad, act=L.ACT_NONE, slope=0.0, prec=None, to=32):
    """conv2d for a layer whose input channel count is no multiple of 32, on the 16-bit kernels: input and weight are zero-padded to `to`
    channels (one fill + one copy of the input; 68 % more multiply-adds on a kernel that runs 4-6x faster than the exact-fp32 one the
    layer took before).  In exact-fp32 arithmetic the plain convolution."""
    p = _o.precision if prec is None else prec
    if p == L.PREC_F32 or x.shape[-1] % 32 == 0 or not x.is_cuda:
        return _o.conv2d(x, w, b, stride, pad, act, slope, prec=prec)
    return _o.conv2d(_PadChannels.apply(x, to), _PadConvIn.apply(w, to), b, stride, pad, act, slope, prec=prec)


def slice_channels(x, a, b):
    """x[..., a:b] as a new contiguous NHWC tensor (inputs only; no gradient)."""
    y = torch.empty(x.shape[:-1] + (b - a,), dtype=x.dtype, device=x.device)
    _copy_channels(x, y, a, 0, b - a)
    return y


def nchw_to_nhwc(x):
    _o._chk(x)
    x = x.contiguous()
    B, C, H, W = x.shape
    y = torch.empty((B, H, W, C), dtype=x.dtype, device=x.device)
    call('hoig_nchw_to_nhwc', _o._p(x), _o._p(y), B, C, H, W, _o._st())
    return y


def nhwc_to_nchw(x):
    _o._chk(x)
    x = x.contiguous()
    B, H, W, C = x.shape
    y = torch.empty((B, C, H, W), dtype=x.dtype, device=x.device)
    call('hoig_nhwc_to_nchw', _o._p(x), _o._p(y), B, C, H, W, _o._st())
    return y


class _MaxPool(Function):
    @staticmethod
    def forward(ctx, x):
        _o._chk(x)
        B, H, W, C = x.shape
        y = torch.empty((B, H // 2, W // 2, C), dtype=x.dtype, device=x.device)
        call('hoig_maxpool2_fwd', _o._p(x), _o._p(y), B, H, W, C, _o._st())
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, = ctx.saved_tensors
        B, H, W, C = x.shape
        dx = torch.empty_like(x)
        call('hoig_maxpool2_bwd', _o._p(x), None, _o._p(dy.contiguous()), _o._p(dx), B, H, W, C, _o._st())
        return dx


def maxpool2(x):
    return _MaxPool.apply(x)


# ------------------------------------------------------------------------------------------------- sampling
def resize_bilinear_ac(x, ho, wo):
    B, Hi, Wi, C = x.shape
    y = torch.empty((B, ho, wo, C), dtype=x.dtype, device=x.device)
    call('hoig_resize_bilinear_ac', _o._p(x.contiguous()), _o._p(y), B, Hi, Wi, C, ho, wo, _o._st())
    return y


def resize_nearest(x, ho, wo):
    B, Hi, Wi, C = x.shape
    if (Hi, Wi) == (ho, wo):
        return x
    y = torch.empty((B, ho, wo, C), dtype=x.dtype, device=x.device)
    call('hoig_resize_nearest', _o._p(x.contiguous()), _o._p(y), B, Hi, Wi, C, ho, wo, _o._st())
    return y


def attn_flow(tscale):
    B, h = tscale.shape[0], tscale.shape[1]
    flow = torch.empty((B, 2, h, h), dtype=tscale.dtype, device=tscale.device)
    call('hoig_attn_flow', _o._p(tscale.contiguous()), _o._p(flow), B, h, _o._st())
    return flow


class _GridSample(Function):
    @staticmethod
    def forward(ctx, x, grid):
        _o._chk(x); _o._chk(grid)
        B, H, W, C = x.shape
        Ho, Wo = grid.shape[1], grid.shape[2]
        y = torch.empty((B, Ho, Wo, C), dtype=x.dtype, device=x.device)
        call('hoig_grid_sample_fwd', _o._p(x), _o._p(grid), _o._p(y), B, H, W, C, Ho, Wo, _o._st())
        ctx.save_for_backward(grid)
        ctx.shape = (B, H, W, C, Ho, Wo)
        return y

    @staticmethod
    def backward(ctx, dy):
        grid, = ctx.saved_tensors
        B, H, W, C, Ho, Wo = ctx.shape
        dx = torch.zeros((B, H, W, C), dtype=dy.dtype, device=dy.device)
        call('hoig_grid_sample_bwd', _o._p(grid), _o._p(dy.contiguous()), _o._p(dx), B, H, W, C, Ho, Wo, _o._st())
        return dx, None


def grid_sample(x, grid):
    return _GridSample.apply(x, grid.contiguous())


_f6_cache = {}
