"""Autograd-visible operators of the HOGAN path; every one is a call into libhoig_hip.so.

Tensors are fp32 CUDA tensors in NHWC ("channels-last") layout, contiguous.
Convolution weights keep the reference's LOGICAL shapes -- Conv2d (Co,Ci,R,S),
ConvTranspose2d (Ci,Co,R,S) -- over PACKED storage [Co][R][S][Ci] (see
``hoig_amd.nn``).  Parameters that live in a flat buffer (``_hoig_flat``) get
their weight gradients accumulated in place by the kernels; for ordinary
tensors the gradient is returned through autograd.
"""
import collections
import contextlib
import ctypes
import os

import torch
from torch.autograd import Function

from . import _lib as L
from ._lib import call, ConvDesc

# arithmetic of the MFMA contractions (include/hoig_kernels.h HOIG_PREC_*).  'f16x3' is the name of what 'bf16x3' has always
# been in the forward (fp16 halves; the backward splits on bf16); 'f16' / 'bf16' likewise name the single-pass mode.
_PREC = {'f32': L.PREC_F32, 'bf16x3': L.PREC_BF16X3, 'f16x3': L.PREC_BF16X3, 'f16x2': L.PREC_F16X2, 'bf16x2': L.PREC_F16X2,
         'bf16': L.PREC_BF16, 'f16': L.PREC_BF16, 'f16f6': L.PREC_F16F6}
precision = precision_dgrad = precision_wgrad = L.PREC_F32


def set_precision(name):
    """'f32' | 'bf16x3' | 'f16x2' | 'bf16', or '<forward>:<backward>', or '<forward>:<data gradient>:<weight gradient>'
    (e.g. 'bf16x3:f16x2': forward launches on three MFMA terms, data and weight gradients on two)."""
    global precision, precision_dgrad, precision_wgrad
    parts = name.split(':')
    if len(parts) > 3:
        raise ValueError('precision %r' % name)
    precision = _PREC[parts[0]]
    if precision == L.PREC_F16F6 and len(parts) == 1:
        parts = [parts[0], 'f16x2']                      # a forward-only arithmetic: the backward defaults to two bf16 terms
    if L.PREC_F16F6 in [_PREC[q] for q in parts[1:]]:
        raise ValueError('f16f6 is a forward arithmetic')
    precision_dgrad = _PREC[parts[1]] if len(parts) > 1 else precision
    precision_wgrad = _PREC[parts[2]] if len(parts) > 2 else precision_dgrad


set_precision(os.environ.get('HOIG_PRECISION', 'f32'))


@contextlib.contextmanager
def inference_forward_precision(name='f16f6'):
    """Forward-only launches (eval mode under torch.no_grad(): eval.py:59-65) on a cheaper forward arithmetic than the training
    forward's.  north_star bounds the OUTPUTS (1e-3); `f16f6` -- fp16 hi*hi plus the two cross terms on block-scaled fp6 MFMAs, 1.6
    MFMA units per product instead of 3 -- stays inside that bound by 5x at 256 x 256 (tests/test_configs_gpu.py) and was kept out of
    the TRAINING default only for what its forward error does to the gradients; a forward without a backward
    has none.  Applies only where the module-level forward is the three-term default: an explicit choice ('f32', 'f16x2', ...) stays."""
    global precision
    if name in (None, '', 'same') or precision != L.PREC_BF16X3 or torch.is_grad_enabled():
        yield False
        return
    prev, precision = precision, _PREC[name]
    try:
        yield True
    finally:
        precision = prev

# Forward arithmetic per sub-network: 'vgg' and 'd' do not sit inside the 45-layer generator chain whose error
# growth forces three forward terms.  name -> HOIG_PREC_* of that sub-network's convolution FORWARDS (their backward then follows
# the forward's arithmetic: _bwd_descs); names without an entry use the module-level `precision`.
_subnet_prec = {}


def set_subnet_precision(mapping):
    """{'vgg': 'f16x2', 'd': 'f16x2'} or 'vgg=f16x2,d=f16x2' (HOIG_PRECISION_MAP); None / '' clears."""
    _subnet_prec.clear()
    if not mapping:
        return
    if isinstance(mapping, str):
        mapping = dict(kv.split('=') for kv in mapping.split(',') if kv)
    for k, v in mapping.items():
        _subnet_prec[k] = _PREC[v]


def subnet_precision(name):
    """Forward arithmetic of sub-network `name`, or None for the module default; an exact-fp32 run keeps every sub-network exact."""
    return None if precision == L.PREC_F32 else _subnet_prec.get(name)


set_subnet_precision(os.environ.get('HOIG_PRECISION_MAP'))


def _bwd_descs(d):
    """(data-gradient, weight-gradient) descriptors of a convolution: same problem, the backward arithmetic modes.  An
    exact-fp32 forward keeps its backward exact (first-layer convolutions are routed to f32 per call)."""
    same = d.precision == precision or (precision == L.PREC_F16F6 and d.precision == L.PREC_BF16X3)   # (an f16f6 launch that ran as x3)
    if not same:                          # a per-call precision override: the backward follows the forward
        return d, d
    out = []
    for prec in (precision_dgrad, precision_wgrad):
        if (d.precision == prec and prec != L.PREC_F16F6) or d.precision == L.PREC_F32:
            out.append(d)
        else:
            b = ConvDesc.from_buffer_copy(d)
            b.precision = prec
            out.append(b)
    return tuple(out)


def _st():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


def _chk(t, name='tensor'):
    if t is None:
        return
    if not t.is_cuda:
        raise NotImplementedError('%s: hoig_amd ops run on the HIP device only (no CPU path), as the reference ops '
                                  'do (block_extractor.py:23-24)' % name)
    if t.dtype != torch.float32:
        raise TypeError('%s must be float32' % name)


_index_cache = collections.OrderedDict()


def device_index(rows, device):
    """int64 index tensor of the Python ints `rows` on `device` WITHOUT a host wait.  ``torch.tensor(rows, device=...)`` is a pageable
    host-to-device copy: the host blocks until everything queued on the current stream before it has run, which throws away the
    run-ahead the step relies on.  Here the values go through pinned memory with a non-blocking copy, and the handful of
    row sets a loader produces (samples grouped by object id) are cached per stream."""
    dev = torch.device(device)
    key = (dev, torch.cuda.current_stream(dev).cuda_stream, tuple(int(r) for r in rows))
    hit = _index_cache.get(key)
    if hit is not None:
        _index_cache.move_to_end(key)
        return hit
    t = torch.tensor(key[2], dtype=torch.int64).pin_memory().to(dev, non_blocking=True)
    _index_cache[key] = t
    if len(_index_cache) > 1024:
        _index_cache.popitem(last=False)
    return t


def packed_strides(shape, transposed):
    """Strides of a logical conv weight over packed [Co][R][S][Ci] storage."""
    if transposed:
        ci, co, r, s = shape
        return (1, r * s * ci, s * ci, ci)
    co, ci, r, s = shape
    return (r * s * ci, 1, s * ci, ci)


def pack_weight(w, transposed=False):
    """Return `w` (logical NCHW-style conv weight) re-laid over packed storage."""
    out = torch.empty_strided(tuple(w.shape), packed_strides(w.shape, transposed), dtype=w.dtype, device=w.device)
    out.copy_(w)
    return out


_grad_observer = None


def set_grad_observer(fn):
    """`fn(p)` is called whenever a backward is about to accumulate into the flat-buffer gradient of parameter `p` (the kernels that do
    so are launched before that backward returns, on the current stream or on the weight-gradient side stream), and `fn(None)` when the
    next such backward begins (_grad_epoch).  hoig_amd/ddp.py's bucketed exchange uses it to learn when a slice of the gradient buffer
    has received its last contribution.  Returns the previous observer."""
    global _grad_observer
    prev, _grad_observer = _grad_observer, fn
    return prev


def _grad_epoch():
    """FIRST statement of every backward that accumulates into flat-buffer gradients: tells the observer that a new backward
    begins, i.e. that the kernels of every gradient write announced so far have been issued (a backward may announce several writes
    before it launches the first of them: bias + weight, the four parameters of the attention)."""
    if _grad_observer is not None:
        _grad_observer(None)


def _grad_target(p):
    """(buffer to accumulate into, returned-through-autograd?)"""
    if getattr(p, '_hoig_flat', False):
        if p.grad is None:
            raise RuntimeError('flat parameter without a gradient view')
        if _grad_observer is not None:
            _grad_observer(p)
        return p.grad, False
    if p.dim() == 4:
        g = torch.empty_strided(tuple(p.shape), p.stride(), dtype=p.dtype, device=p.device).zero_()
    else:
        g = torch.zeros_like(p)
    return g, True


# ------------------------------------------------------------------------------------------------- conv
_pack_cache = {}


def _packed_planes(w, transposed, for_dgrad):
    """bf16 hi/lo planes of a packed conv weight (hoig_pack_conv_weight_bf16), cached per optimiser step for
    parameters that live in a flat buffer (their owner bumps ``version`` whenever the weights change)."""
    if transposed:
        ci, co, r, s = w.shape
    else:
        co, ci, r, s = w.shape
    owner = getattr(w, '_hoig_owner', None)
    if owner is not None:
        planes = owner.packed_planes(w, for_dgrad)      # all weights of the network split in one launch per step
        if planes is not None:
            return planes
    key = (w.data_ptr(), for_dgrad)
    ver = owner.version if owner is not None else None
    hit = _pack_cache.get(key)
    if hit is not None and ver is not None and hit[0] == ver:
        return hit[1], hit[2]
    if hit is not None and hit[1].numel() == w.numel():
        hi, lo = hit[1], hit[2]
    else:
        hi = torch.empty(w.numel(), dtype=torch.int16, device=w.device)
        lo = torch.empty(w.numel(), dtype=torch.int16, device=w.device)
    call('hoig_pack_conv_weight_bf16', _p(w), co, r * s, ci, 1 if for_dgrad else 0, _p(hi), _p(lo), _st())
    if ver is not None:
        _pack_cache[key] = (ver, hi, lo)
    return hi, lo


_stream_banks = {}
_streams_made = _streams_destroyed = 0
# Hardware-queue class of each stream ROLE of the step.  The runtime gives every new HIP stream the least-used of its (four)
# hardware queues, so entry k of a bank of streams created back to back sits on queue k mod 4 (class 0 = the queue of the caller's
# default stream), and WHICH roles share a hardware queue is worth up to 20 % of the step time
# (profiles/r03_stream_queue_map.txt, one box: the three branch chains on three different queues 77-78 ms, every side role on one
# queue 79-88 ms, the assignment below 72.9 ms): kernels of different queues interleave at workgroup granularity, and four heavy
# chains doing that to each other are slower than two pairs.  Streams of ONE class execute in order: a stalled role stalls its
# class mates (tests/test_stream_order_gpu.py moves a role to a class of its own where it needs one role late).  The loader's
# stream (data/device_stage.py) has a role of its own (it shared 'opt'): class 2, whose members -- the loss chains, the D
# step, the weight gradients -- are idle during the generator's forward, which is when batch i + 1 is staged.
_QUEUE_OF_ROLE = {'opt': 3, 'g_bg': 1, 'g_obj': 3, 'g_src': 1, 'loss_adv': 2, 'loss_vgg': 2, 'd': 2, 'wgrad': 2, 'loader': 2}


def new_stream(device=None, role='opt'):
    """A HIP stream for stream role `role` that no other stream object of the process aliases.  torch.cuda.Stream() takes its
    streams round-robin from a pool of 32 per device: in a process that has created more than that (a test session with a dozen
    Trainers) two roles of the step end up on ONE HIP stream, and a stream that waits for itself inside a capture crashes
    hipStreamEndCapture (ROCm 7.2: unbounded recursion over the capture's parallel streams).  The library creates the streams in
    banks of 32, back to back; torch wraps the one this role's queue class asks for.  The owner hands a stream back with
    release_stream() when it goes away (Trainer.close): the next owner of that queue class takes it over."""
    global _streams_made
    dev = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    bank = _stream_banks.get(dev)
    if bank is None:
        bank = _stream_banks[dev] = {'free': [[], [], [], []]}
    q = _QUEUE_OF_ROLE[role]
    if not bank['free'][q]:                     # (first use, or every stream of the class is in use: 32 more, same queue classes)
        with torch.cuda.device(dev):
            for k in range(32):
                h = ctypes.c_void_p()
                call('hoig_stream_create', ctypes.byref(h))
                bank['free'][k % 4].append(h.value)
                _streams_made += 1
    h = bank['free'][q].pop(0)
    s = torch.cuda.ExternalStream(h, device=dev)
    s._hoig_class = q
    return s


def release_stream(s):
    """Hand a stream made by new_stream() back to its bank (work still queued on it simply precedes the next owner's)."""
    q = getattr(s, '_hoig_class', None)
    if q is None:
        return
    s._hoig_class = None
    _stream_banks[s.device]['free'][q].insert(0, s.cuda_stream)     # (first out again: the same few handles, and their scratch, stay in use)


def stream_census():
    """(HIP streams this module has created and not destroyed, of which idle in the banks, bytes of registered scratch)."""
    idle = sum(len(f) for b in _stream_banks.values() for f in b['free'])
    return _streams_made - _streams_destroyed, idle, sum(t.numel() for t in _scratch.values())


def destroy_idle_streams():
    """hoig_stream_destroy every stream that sits unused in a bank, after the device has drained (tests; an embedding process
    that wants its HIP streams back)."""
    global _streams_destroyed
    for dev, bank in _stream_banks.items():
        torch.cuda.synchronize(dev)
        for f in bank['free']:
            while f:
                h = f.pop()
                _scratch.pop((dev, h), None)
                call('hoig_stream_destroy', h)          # (also forgets the stream's scratch registration)
                _streams_destroyed += 1


# Per-stream scratch of the kernels that reduce partial sums through memory (include/hoig_kernels.h hoig_stream_scratch_set): the
# library allocates nothing, so the caller -- this module -- registers one block per stream that launches weight gradients.
_scratch = {}


def _ensure_scratch(nbytes):
    cur = torch.cuda.current_stream()
    key = (cur.device, cur.cuda_stream)
    if nbytes > 0 and key not in _scratch:
        # (never freed while registered; inside a capture it comes from the graph's private pool, which outlives every replay)
        t = torch.empty(L.lib.hoig_stream_scratch_bytes(), dtype=torch.uint8, device=cur.device)
        _scratch[key] = t
        call('hoig_stream_scratch_set', cur.cuda_stream, t.data_ptr(), t.numel())


def wgrad_call(name, d, *args):
    """Weight-gradient entry point `name` for layer `d` (a ConvDesc) on the current stream; if the layer's kernel reduces partial sums
    through memory (the thin-channel layers), that stream's scratch block is registered first."""
    _ensure_scratch(L.lib.hoig_conv2d_bwd_weight_scratch_bytes(ctypes.byref(d)))
    call(name, ctypes.byref(d), *args)


# ---- test hook (tests/test_stream_order_gpu.py): stream role -> cycles to idle.  A role that finds its name here sleeps that long
# where it forks off the caller's stream and again where its backward begins (once per step each), so that a consumer that is not
# ordered behind it reads stale data and a parity check fails instead of the bug hiding behind timing.  Roles: g_bg, g_obj, g_src
# (the generator's branch streams), loss_adv, loss_vgg, d, wgrad (the weight-gradient side stream), opt (the optimiser side stream).
_TEST_DELAYS = {}
_test_fired = set()


def test_step_begins():
    _test_fired.clear()


def test_delay(role, where='fwd'):
    """Idle the CURRENT stream if `role` is being delayed (at most once per step and `where`)."""
    cycles = _TEST_DELAYS.get(role)
    if not cycles or (role, where) in _test_fired:
        return
    _test_fired.add((role, where))
    torch.cuda._sleep(int(cycles))


class _DelayBackward(Function):
    @staticmethod
    def forward(ctx, x, role):
        ctx.role = role
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        test_delay(ctx.role, 'bwd')          # (autograd runs this on the stream of the forward: the role's own)
        return g, None


def delay_backward(x, role):
    """Identity; with the test hook armed for `role`, the backward that starts at `x` begins with the role's delay."""
    if _TEST_DELAYS.get(role) and torch.is_tensor(x) and x.requires_grad:
        return _DelayBackward.apply(x, role)
    return x


_wgrad_streams = {}
# weight gradients on a side stream, beside the data-gradient chain: '1' always, '0' never, 'auto' (default) wherever the
# backward is ONE chain -- the generator's backward already runs as three concurrent chains (its forward forks onto branch
# streams, HOIG_STREAMS), and a fourth stream of one-workgroup-per-CU kernels beside them costs 1 % (measured)
_WGRAD_MODE = os.environ.get('HOIG_WGRAD_STREAM', 'auto')
_WGRAD_SIDE = _WGRAD_MODE != '0'
_wgrad_side_paused = False


def pause_wgrad_side(paused):
    """Trainer: True around a backward that already runs on several streams (only honoured in 'auto' mode)."""
    global _wgrad_side_paused
    _wgrad_side_paused = bool(paused) and _WGRAD_MODE == 'auto'


def _wgrad_side_stream(device):
    if not _WGRAD_SIDE or _wgrad_side_paused:
        return None
    s = _wgrad_streams.get(device)
    if s is None:
        s = _wgrad_streams[device] = new_stream(device, 'wgrad')
    if _capture_depth:
        _capture_forked.add(s)
    return s


_wgrad_pending = collections.deque()

# --- hipGraph capture (Trainer._graph_step): while a step is being CAPTURED nothing executes, memory comes from the graph's
# private pool (no record_stream there) and events may not be queried.  A tensor that is used on a stream other than the one it
# was allocated on is then simply kept referenced until the capture ends: the pool never hands its block to another tensor of
# the capture, which is what record_stream() guarantees in eager mode.
_capture_depth = 0
_capture_hold = []


_capture_forked = set()        # weight-gradient side streams that have joined the capture (and must be joined back before it ends)


_has_gpu = None


def capturing():
    """True while a hipGraph is being captured -- through graph_capture() below, or by a caller's own torch.cuda.graph()."""
    global _has_gpu
    if _capture_depth > 0:
        return True
    if _has_gpu is None:
        _has_gpu = torch.cuda.is_available()          # (hoig_amd.ddp's exchange also runs on CPU tensors over gloo)
    return _has_gpu and torch.cuda.is_current_stream_capturing()


def begin_capture():
    global _capture_depth
    _capture_depth += 1


def end_capture():
    global _capture_depth
    _capture_depth -= 1
    if _capture_depth == 0:
        _capture_hold.clear()
        _capture_forked.clear()
        _wgrad_pending.clear()


class graph_capture(object):
    """``with ops.graph_capture(graph, pool=None):`` = ``torch.cuda.graph`` plus this module's capture bookkeeping."""

    def __init__(self, graph, pool=None):
        # thread-local error mode: the capture still records every launch into the capturing streams (the autograd thread's
        # included), but a HIP call made by ANOTHER thread meanwhile -- ProcessGroupNCCL's watchdog polling the events of earlier
        # collectives -- is not an error that takes the process down (measured: 'operation not permitted when stream is capturing'
        # from the watchdog thread under the default global mode)
        self._ctx = torch.cuda.graph(graph, pool=pool, capture_error_mode='thread_local')

    def __enter__(self):
        begin_capture()
        try:
            return self._ctx.__enter__()
        except BaseException:
            end_capture()
            raise

    def __exit__(self, *exc):
        try:
            if exc[0] is None:
                join_wgrad_streams()
            return self._ctx.__exit__(*exc)
        finally:
            end_capture()


def cross_stream(t, stream):
    """Tensor `t`, allocated on another stream, is (about to be) used on `stream`: keep its memory out of the allocator's hands
    until that use has run."""
    if t is None:
        return t
    if capturing():
        _capture_hold.append(t)
    else:
        t.record_stream(stream)
    return t


def _wgrad_hold(side, tensors):
    """Keep `tensors` (the operands of a weight gradient just launched on `side`) referenced until that launch has run.
    Autograd accumulates a multi-consumer gradient IN PLACE into an incoming gradient tensor when nobody else references it
    (InputBuffer::accumulate): the dy of a residual add reaches both the convolution and the block input's accumulation
    buffer, and once the convolution's backward has returned, the accumulation may overwrite dy while the side stream is
    still reading it.  A live reference keeps the accumulation out of place; record_stream() alone only guards reuse after
    free."""
    if capturing():                     # (a captured event cannot be queried; the references live until the capture ends)
        _capture_hold.append(tensors)
        return
    for t in tensors:
        t.record_stream(side)
    ev = torch.cuda.Event()
    ev.record(side)
    _wgrad_pending.append((ev, tensors))
    while _wgrad_pending and _wgrad_pending[0][0].query():
        _wgrad_pending.popleft()


def wgrad_side_streams():
    """The weight-gradient side streams in use (hoig_amd/ddp.py records events on them)."""
    return list(_wgrad_streams.values())


def join_wgrad_streams():
    """Make the current stream wait for every weight gradient launched on a side stream (called before anything reads a
    flat gradient buffer: the optimiser step, the gradient all-reduce)."""
    if capturing():                 # only streams that forked INTO the capture may be waited for inside it
        for s in list(_capture_forked):
            torch.cuda.current_stream(s.device).wait_stream(s)
        _capture_forked.clear()
        return
    for s in _wgrad_streams.values():
        torch.cuda.current_stream(s.device).wait_stream(s)
    _wgrad_pending.clear()          # later work on this stream is ordered after the side stream's reads


# --- pre-split gradients (round 5; include/hoig_kernels.h 'PRE-SPLIT gradients').  The two-term backward arithmetic multiplies
# bf16(dy) and bf16(dy - bf16(dy)); the convolution kernels used to make that split of every dy tile in every workgroup that loads it.
# Where a convolution's output goes straight into an instance norm, the norm's backward kernel WRITES its dx as those two planes
# (per pixel [hi: C bf16][lo: C bf16], in the bytes of the fp32 tensor autograd passes along) and the convolution's weight- and
# data-gradient kernels copy them to LDS (LDS-DMA / 16-B pieces) without touching the VALU.
# The hand-off is a TOKEN both ends hold (round 6, ADVICE r5: the round-5 form looked gradients up by data_ptr in a module-level table):
#   * the convolution makes one token per eligible output, keeps it on its backward node and hangs it on the output tensor;
#   * every norm that reads that tensor object counts itself on the token (`readers`) and keeps the token; its backward writes planes
#     only if it was the ONLY such reader, and then puts the tensor it wrote on the token (`offered`);
#   * the convolution's backward takes `offered` and accepts planes only if the gradient it was handed IS that tensor.  Anything else
#     -- autograd summed the planes with another consumer's fp32 gradient, a hook cloned them -- has already mixed plane bits with
#     fp32 values: that is an error, raised here, never a silently wrong gradient.  No offer: the gradient is fp32, as before.
class _SplitToken(object):
    __slots__ = ('readers', 'offered', '__weakref__')

    def __init__(self):
        self.readers, self.offered = 0, None


_split_live = set()        # tokens whose offer has not been taken yet (emptied by the consumers; check_split_grads_consumed)


def _split_backward_ok(d, w, live_bias, transposed):
    """May the backward of convolution `d` read pre-split dy?  (the layers hoig_conv2d_bwd_weight_split covers; the data gradient
    falls back to an un-split copy where its kernel has no pre-split form)"""
    if not L.lib.hoig_set_tuning(b'split_grads', -1) or transposed or live_bias or d.act != L.ACT_NONE:
        return False
    if not getattr(w, '_hoig_flat', False) or (d.R, d.S, d.stride, d.pad) != (3, 3, 1, 1) or (d.Ho, d.Wo) != (d.Hi, d.Wi):
        return False
    if d.Co % 128 or d.Ci % 32 or d.Wo % 32 or d.Ho % 8:
        return False
    dg, wg = _bwd_descs(d)
    return dg.precision in (L.PREC_F16X2, L.PREC_BF16) and wg.precision in (L.PREC_F16X2, L.PREC_BF16)


def _tag_split(y, slot=0):
    """Hang a fresh token on convolution output `y` and on its backward node (slot: which output of a grouped launch)."""
    tok = _SplitToken()
    toks = getattr(y.grad_fn, 'split_toks', None)
    if toks is None:
        toks = y.grad_fn.split_toks = {}
    toks[slot] = tok
    y._hoig_split_grad = tok
    return y


def _claim_split(x):
    """A norm's forward: -> the token of `x` (counted as one more reader), or None if x is not a tagged convolution output."""
    tok = getattr(x, '_hoig_split_grad', None)
    if tok is not None:
        tok.readers += 1
    return tok


def _writes_split(tok):
    """A norm's backward: does it write its dx as planes?  Only as the tagged tensor's single norm reader."""
    return tok is not None and tok.readers == 1


def _offer_split(tok, dx):
    tok.offered = dx
    _split_live.add(tok)


def _take_split(ctx, dy, slot=0):
    """A convolution's backward: True if `dy` holds split planes.  Raises if planes were written for this convolution and `dy` is
    not the tensor they were written into."""
    tok = getattr(ctx, 'split_toks', {}).get(slot)
    if tok is None or tok.offered is None:
        return False
    off, tok.offered = tok.offered, None
    _split_live.discard(tok)
    if dy is not off and (dy.data_ptr() != off.data_ptr() or dy.shape != off.shape):
        raise RuntimeError('a norm wrote this convolution\'s incoming gradient as bf16 hi | lo planes, but the gradient that arrived is '
                           'another tensor: the tagged output had a second consumer (autograd summed plane bits with an fp32 gradient) '
                           'or a hook replaced it.  Give the second consumer its own copy, or set the tuning key split_grads=0.')
    return True


def check_split_grads_consumed():
    """Every gradient written as planes must have been read by the convolution it was written for: an offer left over means the
    convolution's backward never ran while something else may have read the planes as fp32 (Trainer calls this after each backward)."""
    if _split_live:
        n = len(_split_live)
        for tok in list(_split_live):
            tok.offered = None
        _split_live.clear()
        raise RuntimeError('%d pre-split gradient tensor(s) were not consumed by a convolution backward' % n)


def _unsplit(dys):
    """split planes -> fp32 (hi + lo) in a new tensor: for a consumer without a pre-split form"""
    out = torch.empty_like(dys)
    call('hoig_unsplit_planes_bf16', _p(dys), _p(out), dys.numel() // dys.shape[-1], dys.shape[-1], _st())
    return out


class _Conv(Function):
    @staticmethod
    def forward(ctx, x, w, b, stride, pad, transposed, act, slope, out_hw, prec, dead_bias=False, fork=False):
        _chk(x, 'x'); _chk(w, 'w'); _chk(b, 'bias')
        assert x.is_contiguous() and x.dim() == 4
        B, Hi, Wi, Ci = x.shape
        if transposed:
            ci_w, Co, R, S = w.shape
        else:
            Co, ci_w, R, S = w.shape
        assert ci_w == Ci, 'channel mismatch %d vs %d' % (ci_w, Ci)
        assert tuple(w.stride()) == packed_strides(w.shape, transposed), 'conv weight is not in packed layout'
        if out_hw is None:
            if transposed:
                raise ValueError('out_hw required for transposed conv')
            out_hw = ((Hi + 2 * pad - R) // stride + 1, (Wi + 2 * pad - S) // stride + 1)
        Ho, Wo = out_hw
        y = torch.empty((B, Ho, Wo, Co), dtype=x.dtype, device=x.device)
        d = ConvDesc(B, Hi, Wi, Ci, Ho, Wo, Co, R, S, stride, pad, 1 if transposed else 0, act, slope, prec)
        _conv_fwd_raw(d, x, w, b, y, transposed, norm_next=dead_bias)
        ctx.d = d
        ctx.d_dg, ctx.d_wg = _bwd_descs(d)
        ctx.transposed = transposed
        ctx.has_bias = b is not None and not dead_bias
        ctx.save_for_backward(x, w, b, y if act != L.ACT_NONE else None)
        ctx.fork = fork
        ctx.split_ok = dead_bias and _split_backward_ok(d, w, ctx.has_bias, transposed)      # (dead_bias: the output's one reader is a norm)
        if fork:
            # (y, x): the caller hands this second output to x's OTHER consumer, so that autograd sees x consumed once -- by
            # this node, whose backward receives both gradients and lets the data-gradient kernel add the other one in its
            # epilogue, instead of the engine summing two full tensors in a pass of its own
            ctx.set_materialize_grads(False)
            return y, x
        return y

    @staticmethod
    def backward(ctx, dy, dxr=None):
        _grad_epoch()
        x, w, b, y = ctx.saved_tensors
        d = ctx.d
        if dy is None:                      # (fork: only the pass-through output was used)
            return (dxr,) + (None,) * 11
        dy = dy.contiguous()
        if _take_split(ctx, dy):
            return _Conv._backward_split(ctx, dy, dxr, x, w)
        dw_ret = db_ret = None
        db, ret_b = None, False
        if ctx.needs_input_grad[1] and ctx.has_bias and ctx.needs_input_grad[2]:
            db, ret_b = _grad_target(b)
            db_ret = db if ret_b else None
        if d.act != L.ACT_NONE:
            g = torch.empty_like(dy)
            if db is not None:       # one pass: activation backward + the bias gradient (column sums of g)
                call('hoig_act_bwd_colsum', _p(y), _p(dy), _p(g), _p(db), d.act, d.slope, dy.numel() // d.Co, d.Co, _st())
                db = None            # done: the weight-gradient call below must not sum again
            else:
                call('hoig_act_bwd', _p(y), _p(dy), _p(g), d.act, d.slope, dy.numel(), _st())
        else:
            g = dy
        if ctx.needs_input_grad[1]:
            dw, ret_w = _grad_target(w)
            side = _wgrad_side_stream(x.device) if not (ret_w or (db is not None and ret_b)) else None
            if side is not None:
                # weight gradients that accumulate straight into a network's flat gradient buffer have no consumer before
                # the optimiser: run them on a side stream, concurrently with the data-gradient chain on the main stream
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    test_delay('wgrad')
                    wgrad_call('hoig_conv2d_bwd_weight', ctx.d_wg, _p(x), _p(g), _p(dw), _p(db), _st())
                _wgrad_hold(side, (x, g))
            else:
                wgrad_call('hoig_conv2d_bwd_weight', ctx.d_wg, _p(x), _p(g), _p(dw), _p(db), _st())
            dw_ret = dw if ret_w else None
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _conv_dgrad_raw(ctx.d_dg, g, w, dx, ctx.transposed, addend=dxr.contiguous() if dxr is not None else None)
        return dx, dw_ret, db_ret, None, None, None, None, None, None, None, None, None


    @staticmethod
    def _backward_split(ctx, dys, dxr, x, w):
        """dy arrived as bf16 hi | lo planes (written by the backward of the norm that reads this convolution's output)."""
        d_dg, d_wg = ctx.d_dg, ctx.d_wg
        unsplit = None
        if ctx.needs_input_grad[1]:
            dw, ret_w = _grad_target(w)
            side = _wgrad_side_stream(x.device) if not ret_w else None
            if side is not None:
                side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
                if side is not None:
                    test_delay('wgrad')
                rc = L.lib.hoig_conv2d_bwd_weight_split(ctypes.byref(d_wg), _p(x), _p(dys), _p(dw), _st())
                if rc == L.EUNSUPPORTED:
                    unsplit = _unsplit(dys)
                    wgrad_call('hoig_conv2d_bwd_weight', d_wg, _p(x), _p(unsplit), _p(dw), None, _st())
                else:
                    L.check(rc, 'hoig_conv2d_bwd_weight_split')
            if side is not None:
                _wgrad_hold(side, (x, dys) if unsplit is None else (x, dys, unsplit))
                unsplit = None                      # (made on the side stream: the data gradient makes its own if it needs one)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            add = dxr.contiguous() if dxr is not None else None
            hi, lo = _packed_planes(w, False, True)
            rc = L.lib.hoig_conv2d_bwd_data_packed_split(ctypes.byref(d_dg), _p(dys), _p(hi), _p(lo), _p(add), _p(dx), _st())
            if rc == L.EUNSUPPORTED:
                _conv_dgrad_raw(d_dg, unsplit if unsplit is not None else _unsplit(dys), w, dx, False, addend=add)
            else:
                L.check(rc, 'hoig_conv2d_bwd_data_packed_split')
        return (dx, (dw if (ctx.needs_input_grad[1] and ret_w) else None)) + (None,) * 10


def conv2d(x, w, b=None, stride=1, pad=0, act=L.ACT_NONE, slope=0.0, prec=None, dead_bias=False):
    """dead_bias=True: the output feeds an instance normalisation directly, whose mean subtraction makes the bias
    gradient exactly zero in exact arithmetic (the reference computes ~1e-10 rounding noise there); the column sum of dy
    is skipped and the bias gradient left at zero."""
    y = _Conv.apply(x, w, b, stride, pad, False, act, slope, None, precision if prec is None else prec, dead_bias)
    return _tag_split(y) if getattr(y.grad_fn, 'split_ok', False) else y


def conv2d_fork(x, w, b=None, stride=1, pad=0, act=L.ACT_NONE, slope=0.0, prec=None, dead_bias=False):
    """-> (conv2d(x, ...), x') for a tensor x with a SECOND consumer (the skip of a residual block: generator.py:29-32).  x' is
    x; the second consumer must read x' so that its gradient comes back through this node, which adds it in the epilogue of its
    data-gradient kernel (hoig_conv2d_bwd_data_packed_add) -- one extra read instead of the autograd engine's three-pass sum."""
    if not x.requires_grad:
        return conv2d(x, w, b, stride, pad, act, slope, prec, dead_bias), x
    y, xr = _Conv.apply(x, w, b, stride, pad, False, act, slope, None, precision if prec is None else prec, dead_bias, True)
    return (_tag_split(y) if getattr(y.grad_fn, 'split_ok', False) else y), xr


def pair_ok(xa, xb, wa, wb, prec=None):
    """May conv3x3(xa, wa) and conv3x3(xb, wb) run as grouped launches (ops.conv2d_pair)?  The layers every one of the three grouped
    kernels covers: same shapes, flat weights, stride-1 "same" 3x3, Wi % 32 == 0, Hi % 8 == 0, Ci % 32 == 0, Co % 128 == 0, a 16-bit
    forward and a two- or one-term backward (pre-split dy)."""
    prec = precision if prec is None else prec
    mode = L.lib.hoig_set_tuning(b'pair', -1)           # 0: never, 1: always, 2: in captured steps only (include/hoig_kernels.h)
    if not mode or (mode == 2 and not capturing()) or prec in (L.PREC_F32, L.PREC_F16F6) or not xa.is_cuda:
        return False
    if xa.shape != xb.shape or wa.shape != wb.shape or tuple(wa.shape[2:]) != (3, 3) or wa.shape[1] != xa.shape[-1]:
        return False
    if not (getattr(wa, '_hoig_flat', False) and getattr(wb, '_hoig_flat', False)):
        return False
    B, H, W_, Ci = xa.shape
    Co = wa.shape[0]
    if W_ % 32 or H % 8 or Ci % 32 or Co % 128 or 2 * B * (H // 8) * (W_ // 32) * (Co // 128) < 96:
        return False
    return precision_dgrad in (L.PREC_F16X2, L.PREC_BF16) and precision_wgrad in (L.PREC_F16X2, L.PREC_BF16) and prec == precision


class _ConvPair(Function):
    """(conv3x3(xa, wa) + ba, conv3x3(xb, wb) + bb) -- src_model's and tsf_model's layer (generator.py:379-464: one architecture, two
    parameter sets, 8 images each at the bench's batch) -- as GROUPED launches: forward, data gradient and weight gradient each run as
    one grid over both problems (hoig_conv2d_*_pair), which fills the chip where the two single launches each covered half of it.
    The backward reads pre-split dy: a gradient that arrives as planes (from a norm's backward: _take_split) is used as it is, an fp32
    one is split once (hoig_split_planes_bf16), after its bias gradient has been taken.  fork: as _Conv (xa, xb have second readers)."""

    @staticmethod
    def forward(ctx, xa, xb, wa, wb, ba, bb, prec, dead_bias, fork):
        for t in (xa, xb, wa, wb, ba, bb):
            _chk(t)
        assert xa.is_contiguous() and xb.is_contiguous()
        B, H, W_, Ci = xa.shape
        Co = wa.shape[0]
        d = ConvDesc(B, H, W_, Ci, H, W_, Co, 3, 3, 1, 1, 0, L.ACT_NONE, 0.0, prec)
        ya, yb = torch.empty((B, H, W_, Co), dtype=xa.dtype, device=xa.device), torch.empty((B, H, W_, Co), dtype=xa.dtype, device=xa.device)
        live = ba is not None and not dead_bias
        (ha, la), (hb, lb) = _packed_planes(wa, False, False), _packed_planes(wb, False, False)
        rc = L.lib.hoig_conv2d_fwd_packed_pair(ctypes.byref(d), _p(xa), _p(xb), _p(ha), _p(la), _p(hb), _p(lb), _p(ba) if live else None,
                                               _p(bb) if live else None, _p(ya), _p(yb), _st())
        if rc == L.EUNSUPPORTED:              # (no grouped tiling for this shape: one after the other)
            _conv_fwd_raw(d, xa, wa, ba if live else None, ya)
            _conv_fwd_raw(d, xb, wb, bb if live else None, yb)
        else:
            L.check(rc, 'hoig_conv2d_fwd_packed_pair')
        ctx.d_dg, ctx.d_wg = _bwd_descs(d)
        ctx.live_bias = live
        ctx.save_for_backward(xa, xb, wa, wb, ba if live else None, bb if live else None)
        ctx.split_ok = dead_bias and _split_backward_ok(d, wa, live, False)
        ctx.fork = fork
        if fork:
            ctx.set_materialize_grads(False)
            return ya, yb, xa, xb
        return ya, yb

    @staticmethod
    def backward(ctx, dya, dyb, dxra=None, dxrb=None):
        _grad_epoch()
        xa, xb, wa, wb, ba, bb = ctx.saved_tensors
        if dya is None or dyb is None:
            raise RuntimeError('conv2d_pair: both outputs must be used (a grouped launch has no half)')
        d_dg, d_wg = ctx.d_dg, ctx.d_wg
        B, H, W_, Co = dya.shape
        npix = B * H * W_
        dys, db_ret = [], [None, None]
        for i, (dy, b) in enumerate(((dya, ba), (dyb, bb))):
            dy = dy.contiguous()
            if _take_split(ctx, dy, i):
                dys.append(dy)
                continue
            if ctx.live_bias and ctx.needs_input_grad[4 + i]:
                db, ret_b = _grad_target(b)
                call('hoig_colsum_accum', _p(dy), _p(db), npix, Co, _st())
                db_ret[i] = db if ret_b else None
            sp = torch.empty_like(dy)
            call('hoig_split_planes_bf16', _p(dy), _p(sp), npix, Co, _st())
            dys.append(sp)
        # weight gradients: only for the weights that need one (frozen weights: ADVICE r5); a shape the LDS-DMA kernel refuses degrades to
        # the un-split kernel per problem, as _Conv._backward_split does
        need = (bool(ctx.needs_input_grad[2]), bool(ctx.needs_input_grad[3]))
        dwa, ret_a = _grad_target(wa) if need[0] else (None, False)
        dwb, ret_b = _grad_target(wb) if need[1] else (None, False)
        if need[0] or need[1]:
            side = _wgrad_side_stream(xa.device) if not (ret_a or ret_b) else None
            if side is not None:
                side.wait_stream(torch.cuda.current_stream())
            held = [xa, xb, dys[0], dys[1]]
            with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
                if side is not None:
                    test_delay('wgrad')
                rc = L.EUNSUPPORTED
                if need[0] and need[1]:
                    rc = L.lib.hoig_conv2d_bwd_weight_split_pair(ctypes.byref(d_wg), _p(xa), _p(xb), _p(dys[0]), _p(dys[1]), _p(dwa), _p(dwb), _st())
                if rc == L.EUNSUPPORTED:
                    for x, sp, dw, nd in ((xa, dys[0], dwa, need[0]), (xb, dys[1], dwb, need[1])):
                        if not nd:
                            continue
                        rc1 = L.lib.hoig_conv2d_bwd_weight_split(ctypes.byref(d_wg), _p(x), _p(sp), _p(dw), _st())
                        if rc1 == L.EUNSUPPORTED:
                            plain = _unsplit(sp)
                            held.append(plain)
                            wgrad_call('hoig_conv2d_bwd_weight', d_wg, _p(x), _p(plain), _p(dw), None, _st())
                        else:
                            L.check(rc1, 'hoig_conv2d_bwd_weight_split')
                else:
                    L.check(rc, 'hoig_conv2d_bwd_weight_split_pair')
            if side is not None:
                _wgrad_hold(side, tuple(held))
        dxa = dxb = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            dxa, dxb = torch.empty_like(xa), torch.empty_like(xb)
            adds = [None if t is None else t.contiguous() for t in (dxra, dxrb)]
            if (adds[0] is None) != (adds[1] is None):
                adds = [t if t is not None else torch.zeros_like(xa) for t in adds]
            (ha, la), (hb, lb) = _packed_planes(wa, False, True), _packed_planes(wb, False, True)
            rc = L.lib.hoig_conv2d_bwd_data_packed_split_pair(ctypes.byref(d_dg), _p(dys[0]), _p(dys[1]), _p(ha), _p(la), _p(hb), _p(lb),
                                                              _p(adds[0]), _p(adds[1]), _p(dxa), _p(dxb), _st())
            if rc == L.EUNSUPPORTED:
                for sp, hi, lo, add, dx, w in ((dys[0], ha, la, adds[0], dxa, wa), (dys[1], hb, lb, adds[1], dxb, wb)):
                    rc1 = L.lib.hoig_conv2d_bwd_data_packed_split(ctypes.byref(d_dg), _p(sp), _p(hi), _p(lo), _p(add), _p(dx), _st())
                    if rc1 == L.EUNSUPPORTED:
                        _conv_dgrad_raw(d_dg, _unsplit(sp), w, dx, False, addend=add)
                    else:
                        L.check(rc1, 'hoig_conv2d_bwd_data_packed_split')
            else:
                L.check(rc, 'hoig_conv2d_bwd_data_packed_split_pair')
        return dxa, dxb, (dwa if (need[0] and ret_a) else None), (dwb if (need[1] and ret_b) else None), db_ret[0], db_ret[1], None, None, None


def conv2d_pair(xa, xb, wa, wb, ba=None, bb=None, prec=None, dead_bias=False, fork=False):
    """-> (ya, yb) [+ (xa', xb') with fork=True: see conv2d_fork]: the two 3x3 convolutions as grouped launches.  Callers check
    pair_ok() first; anything else goes through conv2d / conv2d_fork twice."""
    out = _ConvPair.apply(xa, xb, wa, wb, ba, bb, precision if prec is None else prec, dead_bias, fork)
    if getattr(out[0].grad_fn, 'split_ok', False):
        _tag_split(out[0], 0)
        _tag_split(out[1], 1)
    return out


class _ConvCat2(Function):
    """conv3x3(cat[x1, x2]) for the decoder's skip convolutions (generator.py:305-306) WITHOUT the concatenated tensor: the
    halo kernels read their input halo from x1 or x2 by channel block (forward, weight gradient) and write the data
    gradient as two tensors.  Only taken when every launch is on the 16-bit halo path (`conv2d_cat2` checks); otherwise the
    caller concatenates."""

    @staticmethod
    def forward(ctx, x1, x2, w, prec, norm_next=False):
        B, H, W_, C1 = x1.shape
        C2 = x2.shape[-1]
        Co = w.shape[0]
        assert tuple(w.stride()) == packed_strides(w.shape, False), 'conv weight is not in packed layout'
        d = ConvDesc(B, H, W_, C1 + C2, H, W_, Co, 3, 3, 1, 1, 0, L.ACT_NONE, 0.0, prec)
        y = torch.empty((B, H, W_, Co), dtype=x1.dtype, device=x1.device)
        hi, lo = _packed_planes(w, False, False)
        rc = L.EUNSUPPORTED
        ws = _conv_stats_workspace(y) if norm_next else None
        if prec == L.PREC_F16F6 and (C1 + C2) % 64 == 0 and Co % 64 == 0 and H % 8 == 0:
            qh, ql = _f6_planes(w)
            rc = L.lib.hoig_conv2d_fwd_f6_ex(ctypes.byref(d), _p(x1), C1, _p(x2), _p(hi), _p(qh), _p(ql), None, None, None, 0, _p(y), _p(ws),
                                             _st())
            if rc != L.EUNSUPPORTED and ws is not None:
                L.check(rc, 'hoig_conv2d_fwd_f6_ex')
                _stats_offer(y)
        d = _x3(d)
        if ws is not None and rc == L.EUNSUPPORTED:          # (see _conv_fwd_raw)
            rc = L.lib.hoig_conv2d_cat_fwd_packed_stats(ctypes.byref(d), _p(x1), C1, _p(x2), _p(hi), _p(lo), None, _p(y), _p(ws), _st())
            if rc != L.EUNSUPPORTED:
                L.check(rc, 'hoig_conv2d_cat_fwd_packed_stats')
                _stats_offer(y)
        if rc == L.EUNSUPPORTED:
            rc = L.lib.hoig_conv2d_cat_fwd_packed(ctypes.byref(d), _p(x1), C1, _p(x2), _p(hi), _p(lo), None, _p(y), _st())
        L.check(rc, 'hoig_conv2d_cat_fwd')
        ctx.d_dg, ctx.d_wg = _bwd_descs(d)
        ctx.save_for_backward(x1, x2, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        _grad_epoch()
        x1, x2, w = ctx.saved_tensors
        C1 = x1.shape[-1]
        dy = dy.contiguous()
        dw, ret_w = _grad_target(w)
        side = _wgrad_side_stream(dy.device) if not ret_w else None
        if side is not None:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                test_delay('wgrad')
                wgrad_call('hoig_conv2d_cat_bwd_weight', ctx.d_wg, _p(x1), C1, _p(x2), _p(dy), _p(dw), None, _st())
            _wgrad_hold(side, (x1, x2, dy))
        else:
            wgrad_call('hoig_conv2d_cat_bwd_weight', ctx.d_wg, _p(x1), C1, _p(x2), _p(dy), _p(dw), None, _st())
        dx1, dx2 = torch.empty_like(x1), torch.empty_like(x2)
        hi, lo = _packed_planes(w, False, True)
        L.check(L.lib.hoig_conv2d_cat_bwd_data_packed(ctypes.byref(ctx.d_dg), _p(dy), _p(hi), _p(lo), _p(dx1), C1, _p(dx2), _st()),
                'hoig_conv2d_cat_bwd_data_packed')
        return dx1, dx2, (dw if ret_w else None), None, None


def conv2d_cat2(x1, x2, w, prec=None, norm_next=False):
    """conv2d(cat_channels([x1, x2]), w, None, 1, 1) (3x3, no bias); without the concatenation when the shapes are on the
    16-bit halo path, else through cat_channels.  norm_next: the output goes straight into an instance norm (_conv_fwd_raw)."""
    prec = precision if prec is None else prec
    B, H, W_, C1 = x1.shape
    C2 = x2.shape[-1]
    Co = w.shape[0]
    ok = (prec != L.PREC_F32 and x1.is_cuda and tuple(w.shape[2:]) == (3, 3) and w.shape[1] == C1 + C2 and
          C1 % 64 == 0 and C2 % 64 == 0 and Co % 64 == 0 and W_ % 32 == 0 and H % 4 == 0 and
          B * (H // 4) * (W_ // 32) * ((Co + 127) // 128) >= 160 and B * (H // 4) * (W_ // 32) * ((C1 + C2 + 127) // 128) >= 160 and
          x1.shape[:3] == x2.shape[:3] and x1.is_contiguous() and x2.is_contiguous())
    if not ok:
        return conv2d(cat_channels([x1, x2]), w, None, 1, 1, prec=prec, dead_bias=norm_next)
    return _ConvCat2.apply(x1, x2, w, prec, norm_next)


def conv2d_after_norm(xraw, gamma, beta, w, b=None, first=None, norm_next=False, eps=1e-5):
    """INFERENCE only (no autograd): conv3x3_s1_p1(cat[first, relu(IN(xraw) * gamma + beta)], w) + b WITHOUT the norm's pass over the
    tensor -- the second convolution's halo loader applies the norm, folded to one FMA per element, and the ReLU while it converts the
    raw tensor for the MFMAs (hoig_conv2d_fwd_packed_normin; the reference chains generator.py:16-22 and :298-309).  The statistics
    are the ones the producing convolution left in the stream's accumulators where it could (_conv_fwd_raw), else one read of xraw.
    `first` (optional) = an already-activated tensor that precedes xraw along the channels (the decoder's skip operand).
    -> y, or None where the layer is not on that kernel (the caller then normalises in a pass of its own, as in training)."""
    assert not torch.is_grad_enabled(), 'conv2d_after_norm has no backward'
    _chk(xraw, 'x'); _chk(w, 'w')
    B, H, W_, C = xraw.shape
    HW = H * W_
    C1 = first.shape[-1] if first is not None else 0
    Co = w.shape[0]
    if (precision == L.PREC_F32 or tuple(w.shape[2:]) != (3, 3) or w.shape[1] != C1 + C or C % 32 or C1 % 32 or Co % 64
            or W_ % 32 or H % 8 or not xraw.is_contiguous() or (first is not None and (not first.is_contiguous() or first.shape[:3] != xraw.shape[:3]))
            or not L.lib.hoig_set_tuning(b'norm_in', -1)):
        _stats_drop(xraw)
        return None
    dev = xraw.device
    mean = torch.empty(B * C, dtype=torch.float32, device=dev)
    rstd = torch.empty_like(mean)
    ws, have = _norm_workspace(L.lib.hoig_inorm_workspace_bytes(B, HW, C) // 4, dev, take=(xraw.data_ptr(), B, HW, C))
    if have:
        call('hoig_inorm_stats_from_sums', B, HW, C, eps, _p(mean), _p(rstd), _p(ws), _st())
    else:
        call('hoig_inorm_stats', _p(xraw), B, HW, C, eps, _p(mean), _p(rstd), _p(ws), _st())
    Cg = C1 + C
    fold = torch.empty(2, B, Cg, dtype=torch.float32, device=dev)
    if C1:
        fold[0, :, :C1] = 1.0
        fold[1, :, :C1] = 0.0
    call('hoig_inorm_fold', _p(mean), _p(rstd), _p(gamma), _p(beta), B, C, fold.data_ptr() + 4 * C1, fold.data_ptr() + 4 * (B * Cg + C1), Cg,
         _st())
    y = torch.empty((B, H, W_, Co), dtype=xraw.dtype, device=dev)
    d = ConvDesc(B, H, W_, Cg, H, W_, Co, 3, 3, 1, 1, 0, L.ACT_NONE, 0.0, precision)
    hi, lo = _packed_planes(w, False, False)
    sws = _conv_stats_workspace(y) if norm_next else None
    a, a2 = (first, xraw) if first is not None else (xraw, None)
    rc = L.EUNSUPPORTED
    if precision == L.PREC_F16F6 and Cg % 64 == 0:          # eval.py's default arithmetic: the same two fusions on conv_f6.hip
        qh, ql = _f6_planes(w)
        rc = L.lib.hoig_conv2d_fwd_f6_ex(ctypes.byref(d), _p(a), C1, _p(a2), _p(hi), _p(qh), _p(ql), _p(b), fold.data_ptr(),
                                         fold.data_ptr() + 4 * B * Cg, C1, _p(y), _p(sws), _st())
    if rc == L.EUNSUPPORTED:
        d = _x3(d)
        rc = L.lib.hoig_conv2d_fwd_packed_normin(ctypes.byref(d), _p(a), C1, _p(a2), _p(hi), _p(lo), _p(b), fold.data_ptr(),
                                                 fold.data_ptr() + 4 * B * Cg, C1, _p(y), _p(sws), _st())
    if rc == L.EUNSUPPORTED:
        return None
    L.check(rc, 'hoig_conv2d_fwd_packed_normin')
    if sws is not None:
        _stats_offer(y)
    return y


class _ConvHeads(Function):
    """The generator's image / mask heads over ONE feature map as one 7x7 convolution with a per-channel activation
    (generator.py:311-315: img_reg -> tanh, attetion_reg_hand -> sigmoid, and the x half of attetion_reg_bg, whose sigmoid
    follows the sum with its y half): `w` is the fused (Co_total, Ci, 7, 7) view of the heads' adjacent weights, `splits` the
    channel counts and `acts` the activation of each head.  One read of x forward; backward ONE weight-gradient and ONE
    data-gradient launch over the concatenated head gradients (three of each, and two full-resolution gradient sums, before)."""

    @staticmethod
    def forward(ctx, x, w, splits, acts, prec):
        _chk(x, 'x'); _chk(w, 'w')
        assert x.is_contiguous() and tuple(w.stride()) == packed_strides(w.shape, False)
        B, H, W_, Ci = x.shape
        Co, _, R, S = w.shape
        assert sum(splits) == Co and len(splits) == len(acts)
        assert Co <= 16, 'the activation codes of the fused heads are 4 bits x 16 channels (hoig_conv2d_fwd_heads)'
        if Co > 16:
            raise ValueError('conv_heads: %d output channels, the per-channel activation codes hold 16' % Co)
        # (f16f6 is an arithmetic of the 3x3 stride-1 layers; every other layer of such a forward runs on three fp16 terms -- without this
        #  mapping the heads of an eval-mode forward missed the MFMA head kernel, which asks for BF16X3, and ran on the fp32 VALU one)
        d = _x3(ConvDesc(B, H, W_, Ci, H, W_, Co, R, S, 1, R // 2, 0, L.ACT_NONE, 0.0, prec))
        codes, ch = 0, 0
        for n, a in zip(splits, acts):
            for _ in range(n):
                codes |= (a & 15) << (4 * ch)
                ch += 1
        y = torch.empty((B, H, W_, Co), dtype=x.dtype, device=x.device)
        call('hoig_conv2d_fwd_heads', ctypes.byref(d), _p(x), _p(w), None, _p(y), codes, _st())
        outs, off = [], 0
        for n in splits:
            o = torch.empty((B, H, W_, n), dtype=x.dtype, device=x.device)
            _copy_channels(y, o, off, 0, n)
            outs.append(o)
            off += n
        ctx.d_dg, ctx.d_wg = _bwd_descs(d)
        ctx.cfg = (tuple(splits), tuple(acts))
        ctx.save_for_backward(x, w, *[o if a != L.ACT_NONE else None for o, a in zip(outs, acts)])
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        _grad_epoch()
        x, w = ctx.saved_tensors[:2]
        ys = ctx.saved_tensors[2:]
        splits, acts = ctx.cfg
        B, H, W_, _ = x.shape
        Co = w.shape[0]
        g = torch.empty((B, H, W_, Co), dtype=x.dtype, device=x.device)
        off = 0
        for n, a, yv, dy in zip(splits, acts, ys, douts):
            if dy is None:
                dy = torch.zeros((B, H, W_, n), dtype=x.dtype, device=x.device)
            dy = dy.contiguous()
            if a != L.ACT_NONE:
                t = torch.empty_like(dy)
                call('hoig_act_bwd', _p(yv), _p(dy), _p(t), a, 0.0, dy.numel(), _st())
                dy = t
            _copy_channels(dy, g, 0, off, n)
            off += n
        dw_ret = None
        if ctx.needs_input_grad[1]:
            dw, ret_w = _grad_target(w)
            wgrad_call('hoig_conv2d_bwd_weight', ctx.d_wg, _p(x), _p(g), _p(dw), None, _st())
            dw_ret = dw if ret_w else None
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _conv_dgrad_raw(ctx.d_dg, g, w, dx)
        return dx, dw_ret, None, None, None


def conv_heads(x, w, splits, acts, prec=None):
    """-> one tensor per head (len(splits) of them), activations applied."""
    return _ConvHeads.apply(x, w, tuple(splits), tuple(acts), precision if prec is None else prec)


def conv_transpose2d(x, w, stride=2, pad=1, output_padding=1, prec=None, norm_next=False):
    """nn.ConvTranspose2d(k, stride, padding, output_padding, bias=False) (generator.py:118,201).  norm_next=True: the output goes
    straight into an instance norm (the statistics then come from this convolution's epilogue, see _conv_fwd_raw)."""
    B, Hi, Wi, _ = x.shape
    R, S = w.shape[2], w.shape[3]
    out_hw = ((Hi - 1) * stride - 2 * pad + R + output_padding, (Wi - 1) * stride - 2 * pad + S + output_padding)
    return _Conv.apply(x, w, None, stride, pad, True, L.ACT_NONE, 0.0, out_hw, precision if prec is None else prec, norm_next)


def set_f6_min_tiles(n):
    """Launches of the f16f6 forward with fewer workgroups than `n` run as three fp16 terms (default 192: they would not fill
    the chip).  Returns the previous value.  Parity tests and tools pass 1 to exercise the fp6 kernel at small sizes."""
    return L.lib.hoig_set_f6_min_tiles(int(n))


def _f6_planes(w):
    """fp6 records of a 3x3 conv weight's hi / lo halves (hoig_pack_conv_weight_f6), re-made when the owner's weights change."""
    co, ci, r, s = w.shape
    owner = getattr(w, '_hoig_owner', None)
    if owner is not None and hasattr(owner, 'packed_f6'):
        planes = owner.packed_f6(w)          # all weights of the network in one launch per optimiser step
        if planes is not None:
            return planes
    ver = owner.version if owner is not None else None
    key = (w.data_ptr(), tuple(w.shape))
    hit = _f6_cache.get(key)
    if hit is not None and ver is not None and hit[0] == ver:
        return hit[1], hit[2]
    if hit is not None:
        qh, ql = hit[1], hit[2]
    else:
        n = L.lib.hoig_f6_plane_bytes(co, r * s, ci)
        qh = torch.empty(n, dtype=torch.uint8, device=w.device)
        ql = torch.empty(n, dtype=torch.uint8, device=w.device)
    call('hoig_pack_conv_weight_f6', _p(w), co, r * s, ci, _p(qh), _p(ql), _st())
    if ver is not None:
        _f6_cache[key] = (ver, qh, ql)
    return qh, ql


_wino_cache = {}


def _wino_planes(w):
    """Winograd-transformed fp16 hi / lo planes of a 3x3 conv weight (hoig_pack_conv_weight_wino), re-made when the owner's weights change."""
    co, ci = w.shape[0], w.shape[1]
    owner = getattr(w, '_hoig_owner', None)
    ver = owner.version if owner is not None else None
    key = (w.data_ptr(), tuple(w.shape))
    hit = _wino_cache.get(key)
    if hit is not None and ver is not None and hit[0] == ver:
        return hit[1], hit[2]
    if hit is not None:
        uh, ul = hit[1], hit[2]
    else:
        n = L.lib.hoig_wino_plane_halfs(co, ci)
        uh = torch.empty(n, dtype=torch.int16, device=w.device)
        ul = torch.empty(n, dtype=torch.int16, device=w.device)
    call('hoig_pack_conv_weight_wino', _p(w), co, ci, _p(uh), _p(ul), _st())
    if ver is not None:
        _wino_cache[key] = (ver, uh, ul)
    return uh, ul


def _x3(d):
    """The descriptor of a launch that the fp6 forward kernel does not cover: the same arithmetic on three fp16 terms."""
    if d.precision != L.PREC_F16F6:
        return d
    c = ConvDesc.from_buffer_copy(d)
    c.precision = L.PREC_BF16X3
    return c


def _conv_fwd_raw(d, x, w, b, y, transposed=False, norm_next=False):
    """y = conv(x, w) (+bias, activation) on the kernel the precision mode selects.  norm_next: y goes straight into an instance
    norm -- where the layer's kernel can, it also leaves the per-image channel sums of y in the stream's norm workspace
    (hoig_conv2d_fwd_packed_stats; hoig_conv2d_fwd_f6_ex on the f16f6 arithmetic) and says so (_stats_offer); the norm then skips
    its statistics pass."""
    ws = None
    if norm_next and d.precision != L.PREC_F32 and d.Ci % 32 == 0 and d.Co % 32 == 0 and d.Co > 32:
        ws = _conv_stats_workspace(y)
    elif norm_next and d.precision != L.PREC_F32 and not transposed and d.Ci <= 12 and d.Co % 64 == 0 and d.stride == 1:
        ws0 = _conv_stats_workspace(y)              # the 7x7 stems (thin input): hoig_conv2d_fwd_stats
        if ws0 is not None:
            rc = L.lib.hoig_conv2d_fwd_stats(ctypes.byref(_x3(d)), _p(x), _p(w), _p(b), _p(y), _p(ws0), _st())
            if rc != L.EUNSUPPORTED:
                L.check(rc, 'hoig_conv2d_fwd_stats')
                _stats_offer(y)
                return
    if d.precision == L.PREC_F16F6:
        if (not transposed and d.R == 3 and d.S == 3 and d.stride == 1 and d.pad == 1 and d.Ci % 64 == 0 and d.Co % 64 == 0
                and d.Hi % 8 == 0 and d.Wi % 32 == 0):
            hi, _ = _packed_planes(w, False, False)
            qh, ql = _f6_planes(w)
            rc = L.lib.hoig_conv2d_fwd_f6_ex(ctypes.byref(d), _p(x), 0, None, _p(hi), _p(qh), _p(ql), _p(b), None, None, 0, _p(y), _p(ws),
                                             _st())
            if rc != L.EUNSUPPORTED:
                L.check(rc, 'hoig_conv2d_fwd_f6_ex')
                if ws is not None:
                    _stats_offer(y)
                return
        d = _x3(d)
    if (ws is None and d.precision == L.PREC_BF16X3 and not transposed and d.R == 3 and d.S == 3 and d.stride == 1 and d.pad == 1
            and d.Ci % 32 == 0 and d.Co % 64 == 0 and d.Hi % 16 == 0 and d.Wi % 16 == 0 and L.lib.hoig_set_tuning(b'wino8', -1) == 1
            and d.B * (d.Hi // 16) * (d.Wi // 16) * (d.Co // 64) <= 256 and d.B * (d.Hi // 8) * ((d.Wi + 31) // 32) * ((d.Co + 127) // 128) <= 128):
        # tuning key wino8 (off by default): half-chip launches of the direct kernel that ONE round of the Winograd kernel covers
        uh, ul = _wino_planes(w)
        rc = L.lib.hoig_conv2d_fwd_wino(ctypes.byref(d), _p(x), _p(uh), _p(ul), _p(b), _p(y), _st())
        if rc != L.EUNSUPPORTED:
            L.check(rc, 'hoig_conv2d_fwd_wino')
            return
    if ws is not None:
        hi, lo = _packed_planes(w, transposed, False)
        rc = L.lib.hoig_conv2d_fwd_packed_stats(ctypes.byref(d), _p(x), _p(hi), _p(lo), _p(b), _p(y), _p(ws), _st())
        if rc != L.EUNSUPPORTED:
            L.check(rc, 'hoig_conv2d_fwd_packed_stats')
            _stats_offer(y)
            return
    if d.precision != L.PREC_F32 and d.Ci % 32 == 0 and d.Co % 32 == 0 and d.Co > 32:
        hi, lo = _packed_planes(w, transposed, False)
        rc = L.lib.hoig_conv2d_fwd_packed(ctypes.byref(d), _p(x), _p(hi), _p(lo), _p(b), _p(y), _st())
        if rc != L.EUNSUPPORTED:
            L.check(rc, 'hoig_conv2d_fwd_packed')
            return
    call('hoig_conv2d_fwd', ctypes.byref(d), _p(x), _p(w), _p(b), _p(y), _st())


def _conv_dgrad_raw(d, g, w, dx, transposed=False, addend=None):
    """dx = data gradient (+ addend: in the kernel's epilogue where it has one, by a separate add otherwise)."""
    packed = d.precision != L.PREC_F32 and d.Co % 32 == 0 and d.Ci % 32 == 0 and d.Ci > 32
    if packed:
        hi, lo = _packed_planes(w, transposed, True)
        if addend is not None:
            rc = L.lib.hoig_conv2d_bwd_data_packed_add(ctypes.byref(d), _p(g), _p(hi), _p(lo), _p(addend), _p(dx), _st())
            if rc != L.EUNSUPPORTED:
                L.check(rc, 'hoig_conv2d_bwd_data_packed_add')
                return
    out = dx if addend is None else torch.empty_like(dx)
    rc = L.lib.hoig_conv2d_bwd_data_packed(ctypes.byref(d), _p(g), _p(hi), _p(lo), _p(out), _st()) if packed else L.EUNSUPPORTED
    if rc == L.EUNSUPPORTED:
        call('hoig_conv2d_bwd_data', ctypes.byref(d), _p(g), _p(w), _p(out), _st())
    else:
        L.check(rc, 'hoig_conv2d_bwd_data_packed')
    if addend is not None:
        call('hoig_add', _p(out), _p(addend), _p(dx), dx.numel(), _st())


# ------------------------------------------------------------------------------------------------- the other operator families
# (round 6: split by family; their names -- private ones too, the tests reach for some -- stay reachable as ops.<name>)
def _reexport():
    import importlib
    g = globals()
    for mod in ('ops_norm', 'ops_small', 'ops_attn', 'ops_loss'):
        m = importlib.import_module('.' + mod, __package__)
        for k, v in vars(m).items():
            if not k.startswith('__') and k not in ('_o', 'L', 'call', 'ConvDesc', 'Function', 'torch', 'ctypes', 'contextlib'):
                g[k] = v


_reexport()
