"""Instance / SPADE normalisation operators (split off hoig_amd/ops.py in round 6; re-exported there: `ops.instance_norm`, ...)."""
import contextlib
import ctypes

import torch
from torch.autograd import Function

from . import _lib as L
from ._lib import call, ConvDesc
from . import ops as _o          # (names of the core module are read at call time: _chk, _claim_split, _grad_epoch, _grad_target, _offer_split, _p, _st, _writes_split)

# ------------------------------------------------------------------------------------------------- instance norm
_norm_ws = {}


_stats_pending = {}     # (device, stream) -> (data_ptr, B, HW, C) of the tensor whose sums a convolution left in that workspace


def _norm_workspace(nfloats, device, take=None):
    """One zero-initialised instance-norm workspace per (device, stream): the kernels leave their accumulators zeroed
    (include/hoig_kernels.h), so it is never memset again.  A convolution whose output goes straight into an instance norm may
    have left that tensor's sums in the accumulators (_conv_fwd_raw): `take` = (data_ptr, B, HW, C) of the tensor the caller is
    about to normalise -> (workspace, True) if they are its sums.  Sums nobody asked for (the norm took another path) are cleared
    before anyone else uses the accumulators."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ws = _norm_ws.get(key)
    pend = _stats_pending.pop(key, None)
    if pend is not None and pend == take:
        return ws, True
    if ws is None or ws.numel() < nfloats:
        ws = torch.zeros(max(nfloats, 1 << 20), dtype=torch.float32, device=device)
        _norm_ws[key] = ws
    elif pend is not None:
        ws[:pend[1] * 2 * pend[3]].zero_()
    return (ws, False) if take is not None else ws


_small_sums = [False]


@contextlib.contextmanager
def small_map_sums():
    """Inside: convolutions whose output feeds a norm leave their channel sums for maps of <= 1024 pixels too -- the caller knows the
    reader is ops.conv2d_after_norm (inference: the norm is folded into the next convolution's loader and needs the statistics, not the
    one-launch norm kernel's pass; round 6: 15 statistics launches per forward gone)."""
    prev, _small_sums[0] = _small_sums[0], True
    try:
        yield
    finally:
        _small_sums[0] = prev


def _conv_stats_workspace(y):
    """Accumulators for the statistics of `y` (a convolution output about to be written), or None when its instance norm would
    not read them: maps of <= 1024 pixels take the one-launch norm kernel, which computes its own (but see small_map_sums)."""
    B, H, W_, C = y.shape
    if (H * W_ <= 1024 and not _small_sums[0]) or C % 4 or B * 2 * C > (1 << 18):        # (1 << 18: the accumulator pool, norm.hip ACC_POOL)
        return None
    return _norm_workspace(L.lib.hoig_inorm_workspace_bytes(B, H * W_, C) // 4, y.device)


def _stats_drop(x):
    """Nobody will take the sums the producer of `x` left in the accumulators: clear them now (_norm_workspace does when asked)."""
    key = (x.device, torch.cuda.current_stream(x.device).cuda_stream)
    pend = _stats_pending.get(key)
    if pend is not None and pend[0] == x.data_ptr():
        _norm_workspace(1, x.device)


def _stats_offer(y):
    B, H, W_, C = y.shape
    _stats_pending[(y.device, torch.cuda.current_stream(y.device).cuda_stream)] = (y.data_ptr(), B, H * W_, C)


class _INorm(Function):
    @staticmethod
    def forward(ctx, x, p0, p1, mode, act, slope, residual, eps):
        _o._chk(x, 'x')
        assert x.is_contiguous()
        B, H, W, C = x.shape
        HW = H * W
        if act != L.ACT_NONE and residual is not None:
            raise ValueError('activation + residual in one instance-norm epilogue is not defined')
        mean = torch.empty(B * C, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        y = torch.empty_like(x)
        # maps of <= 1024 pixels: statistics + apply in one launch from one read of x
        rc = L.EUNSUPPORTED if HW > 1024 else L.lib.hoig_inorm_fwd_fused(_o._p(x), mode, _o._p(p0), _o._p(p1), C, act, slope, _o._p(residual),
                                                                         eps, _o._p(y), _o._p(mean), _o._p(rstd), B, HW, C, _o._st())
        if rc == L.EUNSUPPORTED:
            ws, have = _norm_workspace(L.lib.hoig_inorm_workspace_bytes(B, HW, C) // 4, x.device, take=(x.data_ptr(), B, HW, C))
            if have:              # the convolution that made x left its sums in the accumulators: no pass over x for them
                call('hoig_inorm_stats_from_sums', B, HW, C, eps, _o._p(mean), _o._p(rstd), _o._p(ws), _o._st())
            else:
                call('hoig_inorm_stats', _o._p(x), B, HW, C, eps, _o._p(mean), _o._p(rstd), _o._p(ws), _o._st())
            call('hoig_inorm_apply', _o._p(x), _o._p(mean), _o._p(rstd), mode, _o._p(p0), _o._p(p1), act, slope, _o._p(residual), _o._p(y),
                 B, HW, C, _o._st())
        else:
            L.check(rc, 'hoig_inorm_fwd_fused')
        ctx.cfg = (mode, act, slope, B, HW, C, residual is not None)
        ctx.split_tok = _o._claim_split(x)          # x is the output of a convolution whose backward reads split dy
        # (Leaky)ReLU after a plain / affine norm: the backward recomputes the activation mask from x instead of reading y
        y_free = act in (L.ACT_RELU, L.ACT_LRELU) and mode in (0, 1)
        ctx.save_for_backward(x, mean, rstd, p0, p1, y if (act != L.ACT_NONE and not y_free) else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        _o._grad_epoch()
        x, mean, rstd, p0, p1, y = ctx.saved_tensors
        mode, act, slope, B, HW, C, has_res = ctx.cfg
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dp0 = dp1 = r0 = r1 = None
        if mode == 1:
            dp0, ret0 = _o._grad_target(p0)
            dp1, ret1 = _o._grad_target(p1)
            r0, r1 = (dp0 if ret0 else None), (dp1 if ret1 else None)
        elif mode == 2:
            dp0, dp1 = torch.empty_like(x), torch.empty_like(x)
            r0, r1 = dp0, dp1
        split_dx = _o._writes_split(ctx.split_tok)
        sfx = '_split' if split_dx else ''
        p1m = _o._p(p1) if mode == 1 else None
        rc = getattr(L.lib, 'hoig_inorm_bwd_fused_add' + sfx)(_o._p(x), _o._p(mean), _o._p(rstd), mode, _o._p(p0), p1m, C, _o._p(y), _o._p(dy), act, slope,
                                                              None, _o._p(dx), _o._p(dp0), _o._p(dp1), B, HW, C, _o._st())
        if rc == L.EUNSUPPORTED:
            ws = _norm_workspace(L.lib.hoig_inorm_workspace_bytes(B, HW, C) // 4, x.device)
            call('hoig_inorm_bwd_add_ld' + sfx, _o._p(x), _o._p(mean), _o._p(rstd), mode, _o._p(p0), p1m, C, _o._p(y), _o._p(dy), act, slope, None, _o._p(dx),
                 _o._p(dp0), _o._p(dp1), B, HW, C, _o._p(ws), _o._st())
        else:
            L.check(rc, 'hoig_inorm_bwd_fused_add' + sfx)
        if split_dx:
            _o._offer_split(ctx.split_tok, dx)
        return dx, r0, r1, None, None, None, (dy if has_res else None), None


def instance_norm(x, weight=None, bias=None, act=L.ACT_NONE, slope=0.0, residual=None, eps=1e-5):
    mode = 1 if weight is not None else 0
    return _INorm.apply(x, weight, bias, mode, act, slope, residual, eps)


def spade_norm(x, gamma, beta, act=L.ACT_NONE, slope=0.0, eps=1e-5):
    """IN(x) * (1 + gamma) + beta (spade.py:36), optionally followed by an activation."""
    return _INorm.apply(x, gamma, beta, 2, act, slope, None, eps)


class _SpadeFused(Function):
    """IN(x) * (1 + gamma) + beta with gamma | beta side by side in ONE tensor gb [B,H,W,2C] (the output of the fused
    gamma|beta convolution); the backward writes dgamma | dbeta straight into the matching [.,2C] gradient."""

    @staticmethod
    def forward(ctx, x, gb, act, slope, eps, fork=False):
        _o._chk(x); _o._chk(gb)
        assert x.is_contiguous() and gb.is_contiguous()
        B, H, W, C = x.shape
        assert gb.shape[-1] == 2 * C
        HW = H * W
        mean = torch.empty(B * C, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        y = torch.empty_like(x)
        rc = L.EUNSUPPORTED if HW > 1024 else L.lib.hoig_inorm_fwd_fused(_o._p(x), 2, _o._p(gb), gb.data_ptr() + 4 * C, 2 * C, act, slope,
                                                                         None, eps, _o._p(y), _o._p(mean), _o._p(rstd), B, HW, C, _o._st())
        if rc == L.EUNSUPPORTED:
            ws, have = _norm_workspace(L.lib.hoig_inorm_workspace_bytes(B, HW, C) // 4, x.device, take=(x.data_ptr(), B, HW, C))
            if have:
                call('hoig_inorm_stats_from_sums', B, HW, C, eps, _o._p(mean), _o._p(rstd), _o._p(ws), _o._st())
            else:
                call('hoig_inorm_stats', _o._p(x), B, HW, C, eps, _o._p(mean), _o._p(rstd), _o._p(ws), _o._st())
            call('hoig_inorm_apply_ld', _o._p(x), _o._p(mean), _o._p(rstd), 2, _o._p(gb), gb.data_ptr() + 4 * C, 2 * C, act, slope, None,
                 _o._p(y), B, HW, C, _o._st())
        else:
            L.check(rc, 'hoig_inorm_fwd_fused')
        ctx.cfg = (act, slope, B, HW, C)
        ctx.split_tok = _o._claim_split(x)          # (see _INorm.forward)
        ctx.save_for_backward(x, mean, rstd, gb, y if act != L.ACT_NONE else None)
        if fork:                              # (y, x): see _Conv.forward
            ctx.set_materialize_grads(False)
            return y, x
        return y

    @staticmethod
    def backward(ctx, dy, dxr=None):
        x, mean, rstd, gb, y = ctx.saved_tensors
        act, slope, B, HW, C = ctx.cfg
        if dy is None:
            return dxr, None, None, None, None, None
        dy = dy.contiguous()
        add = dxr.contiguous() if dxr is not None else None
        dx = torch.empty_like(x)
        dgb = torch.empty_like(gb)
        split_dx = _o._writes_split(ctx.split_tok)
        sfx = '_split' if split_dx else ''
        rc = getattr(L.lib, 'hoig_inorm_bwd_fused_add' + sfx)(_o._p(x), _o._p(mean), _o._p(rstd), 2, _o._p(gb), None, 2 * C, _o._p(y), _o._p(dy), act, slope,
                                                              _o._p(add), _o._p(dx), _o._p(dgb), dgb.data_ptr() + 4 * C, B, HW, C, _o._st())
        if rc == L.EUNSUPPORTED:
            ws = _norm_workspace(L.lib.hoig_inorm_workspace_bytes(B, HW, C) // 4, x.device)
            call('hoig_inorm_bwd_add_ld' + sfx, _o._p(x), _o._p(mean), _o._p(rstd), 2, _o._p(gb), None, 2 * C, _o._p(y), _o._p(dy), act, slope, _o._p(add),
                 _o._p(dx), _o._p(dgb), dgb.data_ptr() + 4 * C, B, HW, C, _o._p(ws), _o._st())
        else:
            L.check(rc, 'hoig_inorm_bwd_fused_add' + sfx)
        if split_dx:
            _o._offer_split(ctx.split_tok, dx)
        return dx, dgb, None, None, None, None


def spade_norm_fused(x, gb, act=L.ACT_NONE, slope=0.0, eps=1e-5, fork=False):
    """fork=True -> (y, x') for an x with a second consumer, which must read x' (see conv2d_fork): its gradient is then added by
    the norm's backward kernel."""
    if fork and not x.requires_grad:
        return _SpadeFused.apply(x, gb, act, slope, eps), x
    return _SpadeFused.apply(x, gb, act, slope, eps, fork)
