"""Image composition, the loss terms and the eval output conversion (split off hoig_amd/ops.py in round 6; re-exported there)."""
import contextlib
import ctypes

import torch
from torch.autograd import Function

from . import _lib as L
from ._lib import call, ConvDesc
from . import ops as _o          # (names of the core module are read at call time: _chk, _p, _st)

# ------------------------------------------------------------------------------------------------- compose / losses
class _Compose(Function):
    @staticmethod
    def forward(ctx, bg, obj, hand, mbg, mh):
        for t in (bg, obj, hand, mbg, mh):
            _o._chk(t)
        C = bg.shape[-1]
        npix = bg.numel() // C
        img = torch.empty_like(bg)
        call('hoig_compose_fwd', _o._p(bg), _o._p(obj), _o._p(hand), _o._p(mbg), _o._p(mh), _o._p(img), npix, C, _o._st())
        ctx.save_for_backward(bg, obj, hand, mbg, mh)
        return img

    @staticmethod
    def backward(ctx, dimg):
        bg, obj, hand, mbg, mh = ctx.saved_tensors
        C = bg.shape[-1]
        npix = bg.numel() // C
        dbg, dobj, dhand = torch.empty_like(bg), torch.empty_like(obj), torch.empty_like(hand)
        dmbg, dmh = torch.empty_like(mbg), torch.empty_like(mh)
        call('hoig_compose_bwd', _o._p(bg), _o._p(obj), _o._p(hand), _o._p(mbg), _o._p(mh), _o._p(dimg.contiguous()), _o._p(dbg), _o._p(dobj),
             _o._p(dhand), _o._p(dmbg), _o._p(dmh), npix, C, _o._st())
        return dbg, dobj, dhand, dmbg, dmh


def compose(bg, obj, hand, mbg, mh):
    """mbg*bg + (1-mbg)*(obj*mh + hand*(1-mh))  (trainer.py:400-401)."""
    return _Compose.apply(bg, obj, hand, mbg, mh)


class LossSlots(object):
    """The scalar terms of ONE objective (trainer.py:448-457: loss_G = g_adv + g_rec + g_tsf + g_mask + g_mask_smooth) as fp32
    slots of one small device buffer.  The loss kernels add their scaled value straight into a slot (`into=slots.term(name)`;
    several calls may share a slot: the five VGG levels of g_tsf), `total()` sums the slots in one launch, and differentiating the
    total hands every term the constant 1 -- the host composes the objective without the per-term scalar multiplies, adds, fills
    and gradient scalings torch would launch (42 of them per step).  `extra` names report-only slots (means) outside the total.
    Usage per step: begin() -> the loss calls -> total(*handles).backward(); value(name) reads a slot (0-dim view, no launch)."""

    def __init__(self, names, device, extra=()):
        self.names = list(names)
        self.extra = list(extra)
        self.buf = torch.zeros(len(self.names) + 1 + len(self.extra), dtype=torch.float32, device=device)
        self.one = torch.ones((), dtype=torch.float32, device=device)

    def begin(self):
        self.buf.zero_()

    def _index(self, name):
        return self.names.index(name) if name in self.names else len(self.names) + 1 + self.extra.index(name)

    def term(self, name):
        return (self, self._index(name))

    def value(self, name):
        return self.buf[self._index(name)]

    def total(self, *handles):
        return _LossRoot.apply(self, *handles)


class _LossRoot(Function):
    """Sum of a LossSlots' objective slots.  Its backward hands each term the constant 1 whatever gradient arrives: the terms
    (`into=` losses) store their gradient pre-scaled and ignore it anyway -- the total is meant to be differentiated as it is."""

    @staticmethod
    def forward(ctx, slots, *handles):
        k = len(slots.names)
        call('hoig_sum', _o._p(slots.buf), slots.buf.data_ptr() + 4 * k, k, _o._st())
        ctx.one, ctx.n = slots.one, len(handles)
        return slots.buf[k]

    @staticmethod
    def backward(ctx, g):
        return (None,) + (ctx.one,) * ctx.n


class _MeanLoss(Function):
    """scale * mean(loss(pred, target)); the kernel produces the sum and the pre-scaled gradient in one pass.
    `into` = LossSlots.term(name): the value is a term of that objective (see LossSlots)."""

    @staticmethod
    def forward(ctx, pred, target, kind, tconst, scale, into=None):
        _o._chk(pred); _o._chk(target)
        pred = pred.contiguous()
        n = pred.numel()
        need = pred.requires_grad
        dpred = torch.empty_like(pred) if need else None
        ctx.save_for_backward(dpred)
        ctx.term = into is not None
        if into is not None:
            slots, k = into
            call('hoig_loss_accumulate', kind, _o._p(pred), _o._p(target), tconst, scale / n, slots.buf.data_ptr() + 4 * k, _o._p(dpred), n,
                 _o._st())
            return slots.buf[k]
        out = torch.zeros(1, dtype=torch.float32, device=pred.device)
        call('hoig_loss_fwd_bwd', kind, _o._p(pred), _o._p(target), tconst, scale / n, _o._p(out), _o._p(dpred), n, _o._st())
        return out[0] * (scale / n)

    @staticmethod
    def backward(ctx, g):
        dpred, = ctx.saved_tensors
        if ctx.term:
            return dpred, None, None, None, None, None
        return (dpred * g if dpred is not None else None), None, None, None, None, None


def l1_loss(pred, target, scale=1.0, into=None):
    return _MeanLoss.apply(pred, target.contiguous(), L.LOSS_L1, 0.0, scale, into)


def mse_loss(pred, target, scale=1.0, into=None):
    return _MeanLoss.apply(pred, target.contiguous(), L.LOSS_MSE, 0.0, scale, into)


def bce_loss(pred, target, scale=1.0, into=None):
    return _MeanLoss.apply(pred, target.contiguous(), L.LOSS_BCE, 0.0, scale, into)


def lsgan_loss(pred, target_value, scale=1.0, into=None):
    """mean((x - y)^2) * scale with a constant target (trainer.py:476-477)."""
    return _MeanLoss.apply(pred, None, L.LOSS_MSE, float(target_value), scale, into)


class _TV(Function):
    """Trainer._compute_loss_smooth (trainer.py:479-481) on a single-channel NHWC map."""

    @staticmethod
    def forward(ctx, m, scale, into=None):
        _o._chk(m)
        m = m.contiguous()
        B, H, W, C = m.shape
        assert C == 1
        nx, ny = B * H * (W - 1), B * (H - 1) * W
        need = m.requires_grad
        dm = torch.empty_like(m) if need else None
        ctx.save_for_backward(dm)
        ctx.term = into is not None
        if into is not None:                  # a term of an objective: see LossSlots
            slots, k = into
            call('hoig_tv_accumulate', _o._p(m), scale / nx, scale / ny, slots.buf.data_ptr() + 4 * k, _o._p(dm), B, H, W, _o._st())
            return slots.buf[k]
        out = torch.zeros(2, dtype=torch.float32, device=m.device)
        call('hoig_tv_fwd_bwd', _o._p(m), scale / nx, scale / ny, _o._p(out), _o._p(dm), B, H, W, _o._st())
        return out[0] * (scale / nx) + out[1] * (scale / ny)

    @staticmethod
    def backward(ctx, g):
        dm, = ctx.saved_tensors
        if ctx.term:
            return dm, None, None
        return (dm * g if dm is not None else None), None, None


def tv_loss(m, scale=1.0, into=None):
    return _TV.apply(m, scale, into)


def mean(x, into=None):
    """into = LossSlots.term(name) of a report-only slot: the mean is added there (no result tensor)."""
    if into is not None:
        slots, k = into
        call('hoig_sum_scaled', _o._p(x.contiguous()), 1.0 / x.numel(), slots.buf.data_ptr() + 4 * k, x.numel(), _o._st())
        return slots.buf[k]
    out = torch.zeros(1, dtype=torch.float32, device=x.device)
    call('hoig_sum', _o._p(x.contiguous()), _o._p(out), x.numel(), _o._st())
    return out[0] / x.numel()


def tensor2im_u8(x_nhwc, nrow, unnormalize=True):
    """utils/util.py:249-264 for a batch grid: uint8 CHW of make_grid(nrow, padding=0)."""
    B, H, W, C = x_nhwc.shape
    ncol = min(nrow, B)
    nrw = (B + ncol - 1) // ncol
    out = torch.empty((C, nrw * H, ncol * W), dtype=torch.uint8, device=x_nhwc.device)
    call('hoig_tensor2im_u8', _o._p(x_nhwc.contiguous()), _o._p(out), B, H, W, C, nrow, 1 if unnormalize else 0, _o._st())
    return out
