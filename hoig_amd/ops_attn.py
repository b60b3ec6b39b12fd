"""The local-attention operators and the reference's two CUDA-extension drop-ins (split off hoig_amd/ops.py in round 6; re-exported there)."""
import contextlib
import ctypes

import torch
from torch.autograd import Function

from . import _lib as L
from ._lib import call, ConvDesc
from . import ops as _o          # (names of the core module are read at call time: _bwd_descs, _chk, _conv_dgrad_raw, _conv_fwd_raw, _grad_epoch, _grad_target, _p, _st, _wgrad_hold, _wgrad_side_stream, packed_strides, precision, wgrad_call)

_attn_index = {}


def attn_index_clear():
    """Forget the pixel indices of the previous forward (called when a new forward starts: flows change per batch)."""
    _attn_index.clear()


def _attn_pixel_index(flow, B, H, W):
    """The bucket index of a flow field (hoig_attn_build_index), built once per flow tensor: every attention layer of a
    resolution shares its flow (generator.py:480-491), and the backward of each needs the same index."""
    key = (flow.data_ptr(), B, H, W, torch.cuda.current_stream().cuda_stream)
    hit = _attn_index.get(key)
    if hit is None:
        idx = torch.empty(L.lib.hoig_attn_index_ints(B, H, W), dtype=torch.int32, device=flow.device)
        call('hoig_attn_build_index', _o._p(flow), _o._p(idx), B, H, W, _o._st())
        hit = _attn_index[key] = (idx, flow)           # (the flow is held so that its address is not reused meanwhile)
    return hit[0]


class _AttnSourceConv(Function):
    """The source half of ExtractorAttn's first layer: Gs = conv5x5(replicate_pad(source, 4), ws) on the grid [-2, H+1]^2 (see
    _LocalAttn).  A Function of its own because it depends on the SOURCE features only: the generator evaluates it on the
    stream of src_model, ahead of the tsf chain that consumes it."""

    @staticmethod
    def forward(ctx, source, ws, prec, fork=False):
        _o._chk(source)
        _o._chk(ws)
        B, H, W, C = source.shape
        assert tuple(ws.shape) == (128, C, 5, 5) and tuple(ws.stride()) == _o.packed_strides(ws.shape, False)
        spad = torch.empty((B, H + 8, W + 8, C), dtype=source.dtype, device=source.device)
        call('hoig_replicate_pad_fwd', _o._p(source), _o._p(spad), B, H, W, C, 4, _o._st())
        d_s = ConvDesc(B, H + 8, W + 8, C, H + 4, W + 4, 128, 5, 5, 1, 0, 0, L.ACT_NONE, 0.0, prec)
        gs = torch.empty((B, H + 4, W + 4, 128), dtype=source.dtype, device=source.device)
        _o._conv_fwd_raw(d_s, spad, ws, None, gs)
        ctx.save_for_backward(ws, spad)
        ctx.descs = _o._bwd_descs(d_s)
        ctx.shape = (B, H, W, C)
        if fork:                              # (gs, source): see _Conv.forward
            ctx.set_materialize_grads(False)
            return gs, source
        return gs

    @staticmethod
    def backward(ctx, dgs, dsrc_r=None):
        _o._grad_epoch()
        ws, spad = ctx.saved_tensors
        ds_dg, ds_wg = ctx.descs
        B, H, W, C = ctx.shape
        if dgs is None:
            return dsrc_r, None, None, None
        dgs = dgs.contiguous()
        gw, ret_w = _o._grad_target(ws)
        side = _o._wgrad_side_stream(dgs.device) if not ret_w else None
        if side is not None:          # (see _Conv.backward)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                _o.wgrad_call('hoig_conv2d_bwd_weight', ds_wg, _o._p(spad), _o._p(dgs), _o._p(gw), None, _o._st())
            _o._wgrad_hold(side, (spad, dgs))
        else:
            _o.wgrad_call('hoig_conv2d_bwd_weight', ds_wg, _o._p(spad), _o._p(dgs), _o._p(gw), None, _o._st())
        dsrc = None
        if ctx.needs_input_grad[0]:
            dspad = torch.empty_like(spad)
            _o._conv_dgrad_raw(ds_dg, dgs, ws, dspad)
            dsrc = torch.empty((B, H, W, C), dtype=dgs.dtype, device=dgs.device)
            call('hoig_replicate_pad_bwd_add', _o._p(dspad), _o._p(dsrc_r.contiguous() if dsrc_r is not None else None), _o._p(dsrc),
                 B, H, W, C, 4, _o._st())                                                           # (writes every element)
        return dsrc, (gw if ret_w else None), None, None


class _LocalAttn(Function):
    """ExtractorAttn.forward (extract_attn.py:23-29) without any 25x-sized tensor (hoig_amd/csrc/attn.hip):
    Gt = conv5x5(replicate_pad(target, 2), wt) + b1 ; Gs = conv5x5(replicate_pad(source, 4), ws) on the grid [-2, H+1]^2
    (_AttnSourceConv, passed in) ; hidden = Gt + bilinear(Gs at pixel + flow) ; LeakyReLU ; conv1x1 128->25 ; softmax ;
    (1/25) sum_q a_q S_q read from the source's 6x6 footprint.  `wt`, `ws` (128,C,5,5) are the two halves of the reference's
    (128,2C,5,5) weight (hoig_amd.nn.split_attn_weight)."""

    @staticmethod
    def forward(ctx, source, target, flow, gs, wt, b1, w2, b2, prec, fork=False):
        for t in (source, target, flow, gs, wt, b1, w2, b2):
            _o._chk(t)
        B, H, W, C = source.shape
        assert tuple(wt.shape) == (128, C, 5, 5) and tuple(wt.stride()) == _o.packed_strides(wt.shape, False)
        assert tuple(gs.shape) == (B, H + 4, W + 4, 128)
        M = B * H * W
        # the backward's gather kernels (hoig_attn_src_gather / hoig_attn_build_index) cover less than the forward does: say so
        # here, before a forward that could not be differentiated (the reference's layers have C = 256 / 512, M <= 131072)
        if source.requires_grad and (C % 64 or M >= (1 << 20) or W + 8 >= 2048):
            raise NotImplementedError('local_attention: the backward needs C %% 64 == 0, fewer than 2^20 pixels per batch and '
                                      'W < 2040 (got C=%d, B*H*W=%d, W=%d)' % (C, M, W))
        dev, dt = source.device, source.dtype
        tpad = torch.empty((B, H + 4, W + 4, C), dtype=dt, device=dev)
        call('hoig_replicate_pad_fwd', _o._p(target), _o._p(tpad), B, H, W, C, 2, _o._st())
        d_t = ConvDesc(B, H + 4, W + 4, C, H, W, 128, 5, 5, 1, 0, 0, L.ACT_NONE, 0.0, prec)
        gt = torch.empty((M, 128), dtype=dt, device=dev)
        _o._conv_fwd_raw(d_t, tpad, wt, b1, gt)
        hidden = torch.empty_like(gt)
        attn = torch.empty((M, 25), dtype=dt, device=dev)
        out = torch.empty_like(source)
        kf = torch.empty((M, 36), dtype=dt, device=dev) if source.requires_grad else None
        call('hoig_attn_pixel_fwd', _o._p(gt), _o._p(gs), _o._p(flow), _o._p(w2), _o._p(b2), _o._p(source), _o._p(hidden), _o._p(attn), _o._p(out),
             _o._p(kf), B, H, W, C, _o._st())
        ctx.save_for_backward(source, flow, wt, b1, w2, b2, tpad, hidden, attn, kf)
        ctx.descs = _o._bwd_descs(d_t)
        ctx.shape = (B, H, W, C)
        if fork:
            # (out, source, target): both feature maps have further readers (the next layer of their chain; the sum
            # `target + out`), which read these pass-through outputs so that their gradients come back through this node and
            # are added by its own kernels (see _Conv.forward)
            ctx.set_materialize_grads(False)
            return out, source, target
        return out

    @staticmethod
    def backward(ctx, dout, dsrc_r=None, dtgt_r=None):
        _o._grad_epoch()
        source, flow, wt, b1, w2, b2, tpad, hidden, attn, kf = ctx.saved_tensors
        dt_dg, dt_wg = ctx.descs
        B, H, W, C = ctx.shape
        if dout is None:
            return (dsrc_r, dtgt_r) + (None,) * 8
        dout = dout.contiguous()
        gs = [_o._grad_target(p) for p in (wt, b1, w2, b2)]
        dhid = torch.empty_like(hidden)                       # = dGt
        e_ws = torch.empty((B * H * W, 36), dtype=dout.dtype, device=dout.device)
        call('hoig_attn_pixel_bwd', _o._p(hidden), _o._p(attn), _o._p(w2), _o._p(source), _o._p(flow), _o._p(dout), _o._p(dhid), _o._p(gs[2][0]),
             _o._p(gs[3][0]), _o._p(e_ws), B, H, W, C, _o._st())
        index = _attn_pixel_index(flow, B, H, W)
        dgs = None
        if ctx.needs_input_grad[3]:
            dgs = torch.empty((B, H + 4, W + 4, 128), dtype=dout.dtype, device=dout.device)
            call('hoig_attn_gs_gather', _o._p(index), _o._p(flow), _o._p(dhid), _o._p(dgs), B, H, W, _o._st())
        side = _o._wgrad_side_stream(dout.device) if not any(r for _, r in gs) else None
        if side is not None:          # (see _Conv.backward)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                _o.wgrad_call('hoig_conv2d_bwd_weight', dt_wg, _o._p(tpad), _o._p(dhid), _o._p(gs[0][0]), _o._p(gs[1][0]), _o._st())
            _o._wgrad_hold(side, (tpad, dhid))
        else:
            _o.wgrad_call('hoig_conv2d_bwd_weight', dt_wg, _o._p(tpad), _o._p(dhid), _o._p(gs[0][0]), _o._p(gs[1][0]), _o._st())
        dtgt = dsrc = None
        if ctx.needs_input_grad[1]:
            dtpad = torch.empty_like(tpad)
            _o._conv_dgrad_raw(dt_dg, dhid, wt, dtpad)
            dtgt = torch.empty((B, H, W, C), dtype=dout.dtype, device=dout.device)
            call('hoig_replicate_pad_bwd_add', _o._p(dtpad), _o._p(dtgt_r.contiguous() if dtgt_r is not None else None), _o._p(dtgt),
                 B, H, W, C, 2, _o._st())
        if ctx.needs_input_grad[0]:                           # the weighted average's part (Gs's part comes from _AttnSourceConv)
            dsrc = torch.empty((B, H, W, C), dtype=dout.dtype, device=dout.device)
            call('hoig_attn_src_gather', _o._p(index), _o._p(kf), _o._p(dout), _o._p(dsrc_r.contiguous() if dsrc_r is not None else None),
                 _o._p(dsrc), B, H, W, C, _o._st())
        rets = [g if r else None for g, r in gs]
        return dsrc, dtgt, None, dgs, rets[0], rets[1], rets[2], rets[3], None, None


def _attn_prec(prec):
    return _o.precision if prec is None else prec


def attn_source_conv(source, ws, prec=None, fork=False):
    """Gs of local_attention(): the part that depends on the source features and the source half of the weight only.
    fork=True -> (Gs, source'): later readers of `source` must read source' (see conv2d_fork)."""
    if fork and not source.requires_grad:
        return _AttnSourceConv.apply(source, ws, _attn_prec(prec)), source
    return _AttnSourceConv.apply(source, ws, _attn_prec(prec), fork)


def local_attention(source, target, flow, wt, ws, b1, w2, b2, prec=None, gs=None, fork=False):
    """`gs`: attn_source_conv(source, ws) if the caller has evaluated it already (on another stream).
    fork=True -> (out, source', target'): later readers of the two feature maps must read those (see conv2d_fork)."""
    prec = _attn_prec(prec)
    if gs is None:
        gs = _AttnSourceConv.apply(source, ws, prec)
    if fork and not (source.requires_grad and target.requires_grad):
        return _LocalAttn.apply(source, target, flow.contiguous(), gs, wt, b1, w2, b2, prec), source, target
    return _LocalAttn.apply(source, target, flow.contiguous(), gs, wt, b1, w2, b2, prec, fork)


# stand-alone equivalents of the reference's two extension modules (NCHW, caller-visible semantics of
# block_extractor.py:5-54 / local_attn_reshape.py:5-46)
class _BlockExtractor(Function):
    @staticmethod
    def forward(ctx, source, flow, k):
        _o._chk(source); _o._chk(flow)
        assert source.is_contiguous() and flow.is_contiguous() and flow.shape[1] == 2
        B, C, Hs, Ws = source.shape
        Hf, Wf = flow.shape[2], flow.shape[3]
        out = source.new_zeros((B, C, k * Hf, k * Wf))
        call('hoig_block_extractor_forward', _o._p(source), _o._p(flow), _o._p(out), B, C, Hs, Ws, Hf, Wf, k, _o._st())
        ctx.save_for_backward(source, flow)
        ctx.k = k
        return out

    @staticmethod
    def backward(ctx, g):
        source, flow = ctx.saved_tensors
        B, C, Hs, Ws = source.shape
        Hf, Wf = flow.shape[2], flow.shape[3]
        gs, gf = torch.zeros_like(source), torch.zeros_like(flow)
        call('hoig_block_extractor_backward', _o._p(source), _o._p(flow), _o._p(g.contiguous()), _o._p(gs), _o._p(gf), B, C, Hs, Ws,
             Hf, Wf, ctx.k, _o._st())
        return gs, gf, None


def block_extractor(source, flow, kernel_size):
    return _BlockExtractor.apply(source.contiguous(), flow.contiguous(), kernel_size)


class _LocalAttnReshape(Function):
    @staticmethod
    def forward(ctx, x, k):
        _o._chk(x)
        B, C, Hs, Ws = x.shape
        assert C == k * k
        out = x.new_zeros((B, 1, k * Hs, k * Ws))
        call('hoig_local_attn_reshape_forward', _o._p(x), _o._p(out), B, Hs, Ws, k, _o._st())
        ctx.cfg = (B, Hs, Ws, k)
        return out

    @staticmethod
    def backward(ctx, g):
        B, Hs, Ws, k = ctx.cfg
        gi = g.new_zeros((B, k * k, Hs, Ws))
        call('hoig_local_attn_reshape_backward', _o._p(g.contiguous()), _o._p(gi), B, Hs, Ws, k, _o._st())
        return gi, None


def local_attn_reshape(x, kernel_size):
    return _LocalAttnReshape.apply(x.contiguous(), kernel_size)
