"""Data-parallel gradient exchange for the flat-parameter networks (one process per GPU, RCCL over xGMI).

Replaces the reference's two ``DistributedDataParallel`` wrappers (models/trainer.py:237-239,250-252;
``dist.init_process_group(backend='nccl')`` at train_ddp.py:28): same semantics -- parameters broadcast from rank 0
at construction, gradients averaged over ranks once per optimiser step -- but designed for this machine:
  * gradients already sit in ONE contiguous fp32 buffer, so the exchange is a few LARGE all-reduces (default
    64 MiB slices; xGMI is point-to-point, ring collectives are per-link bound, so fewer/larger messages win over
    DDP's 25 MB buckets) with no flatten/unflatten copies;
  * the reference also all-reduces D's gradients during the G step and then discards them (trainer.py:432 zeroes
    them); here D is frozen during the G step, so that exchange does not exist;
  * the exchange + the fused Adam of G run on a side HIP stream and overlap with the whole D step on the main
    stream (the D step needs neither G's gradients nor G's new weights: trainer.py:460 detaches the fake image).
The averaging factor 1/world is folded into the Adam kernel (grad_scale), not a separate pass.
Works on CPU tensors with the gloo backend too (tests/test_ddp_gloo.py, world_size 2).

Two exchange modes (`opt.ddp_mode`, env HOIG_DDP_MODE; VERDICT r5 item 6 -- NEITHER has been measured on more than one GPU):
  * 'after' (default): every slice is queued when the backward has ended, on the side stream, and hides behind the D step;
  * 'bucket': a slice is queued DURING the backward, as soon as it has received its last contribution -- what the reference's DDP
    reducer hooks do with 25-MB buckets (trainer.py:426,433) -- on a communication stream that waits for exactly the streams that
    wrote into it.  Which contribution is a slice's last is LEARNED: the first backward of a step signature counts the gradient
    writes per slice (ops.set_grad_observer) and exchanges the old way; later backwards count down.  A write that arrives for a
    slice already on the wire means the pattern changed under the same signature: that is raised, never summed twice.
Same slices, same SUM, same order of Adam updates: the two modes are bit-identical (tests/test_ddp_gloo.py, tests/test_ddp_gpu.py).
"""
import os

import torch
import torch.distributed as dist


def _is_dist():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


# HOIG_DDP_PAYLOAD=bf16: exchange the gradients as bf16 (half the bytes per xGMI link: G's 734 MB become 367 MB).  The slice is
# rounded to bf16 before the SUM and widened back into the fp32 gradient buffer after it, so every rank still applies the same
# (rounded) average; each add of the ring rounds to 8 significant bits, i.e. a relative error of ~2^-8 * sqrt(world) per element
# on top of the two-term backward's own 6e-3 -- an option for link-bound configurations, not the default (fp32, exact sum order
# aside).
_PAYLOAD = os.environ.get('HOIG_DDP_PAYLOAD', 'f32')
_MODE = os.environ.get('HOIG_DDP_MODE', 'after')


class GradSync(object):
    def __init__(self, flat_param, flat_grad, bucket_bytes=64 << 20, group=None, force=None, payload=None, mode=None):
        """`force` (default: env HOIG_DDP_FORCE=1): run the collectives even in a world of one rank, so that the whole
        exchange path -- RCCL all-reduce per slice, stream ordering, sliced Adam -- executes on a single-GPU box
        (tests/test_ddp_rccl_gpu.py); a one-rank SUM leaves the gradients unchanged."""
        self.flat_param, self.flat_grad = flat_param, flat_grad
        self.group = group
        inited = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if inited else 1
        if force is None:
            force = os.environ.get('HOIG_DDP_FORCE', '0') == '1'
        self.active = self.world > 1 or (bool(force) and inited)
        n = flat_grad.numel()
        per = max(1, bucket_bytes // 4)
        self.slices = [(s, min(n, s + per)) for s in range(0, n, per)]
        self.payload = (payload or _PAYLOAD).lower()
        if self.payload not in ('f32', 'bf16'):
            raise ValueError('gradient payload %r (f32 | bf16)' % self.payload)
        self._wire = None           # bf16 staging buffer of the whole gradient (allocated on first use)
        self.mode = (mode or _MODE).lower()
        if self.mode not in ('after', 'bucket'):
            raise ValueError('ddp_mode %r (after | bucket)' % self.mode)
        # 'bucket' state: learned write counts per step signature, the running backward's countdown and early launches
        self._learned = {}
        self._run = None
        self._comm = None           # communication stream of the early launches (CUDA tensors only)
        self.early_launches = 0     # (statistics: slices that went on the wire before the backward had ended, last backward)
        # test hook (HOIG_DDP_CHECK=1, one rank): the sum of every early slice when it went on the wire; verify_early_slices() compares
        # it with the slice after the backward -- a write that landed after the launch shows as a difference
        self._check = {} if os.environ.get('HOIG_DDP_CHECK', '0') == '1' else None
        self.check_failures = 0

    def _submit_one(self, i):
        """Queue the SUM all-reduce of slice i behind the current stream; returns its completion callback."""
        a, b = self.slices[i]
        if self.payload == 'f32':
            return dist.all_reduce(self.flat_grad[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True).wait
        if self._wire is None:
            self._wire = torch.empty(self.flat_grad.numel(), dtype=torch.bfloat16, device=self.flat_grad.device)
        self._wire[a:b].copy_(self.flat_grad[a:b])
        h = dist.all_reduce(self._wire[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

        def done():
            h.wait()
            self.flat_grad[a:b].copy_(self._wire[a:b])
        return done

    def _submit(self):
        """Queue the SUM all-reduce of every slice not yet on the wire; returns one completion callback per slice, in slice order."""
        early, self._early = getattr(self, '_early', None) or {}, None
        return [early[i] if i in early else self._submit_one(i) for i in range(len(self.slices))]

    # ---- 'bucket' mode: slices go on the wire while the backward is still running
    def _slices_of(self, p):
        g = p.grad
        off = (g.data_ptr() - self.flat_grad.data_ptr()) // self.flat_grad.element_size()
        if off < 0 or off >= self.flat_grad.numel():
            return ()                                   # a parameter of another network
        per = self.slices[0][1] - self.slices[0][0]
        return range(off // per, (off + p.numel() - 1) // per + 1)

    def begin_backward(self, key):
        """Call right before the backward whose gradients this object exchanges; `key`: anything that identifies the shape of that
        backward (a change of key starts a new learning pass).  No-op unless the mode is 'bucket' and the exchange is active."""
        if self.mode != 'bucket' or not self.active:
            return
        from . import ops
        if ops.capturing():
            return                                      # (a captured step exchanges between its two graphs: Trainer._graph_step)
        learned = self._learned.get(key)
        self._run = dict(key=key, counts=[0] * len(self.slices), left=None if learned is None else list(learned), pending=[],
                         launched={}, prev_observer=None, writers=[set() for _ in self.slices])
        self.early_launches = 0
        self._run['prev_observer'] = ops.set_grad_observer(self._on_grad)

    def _on_grad(self, p):
        """ops' observer: p = a parameter whose flat gradient the running backward function is about to write; None = a new backward
        function begins, so the kernels of every write announced so far have been issued (ops._grad_epoch)."""
        run = self._run
        if p is None:
            self._retire_pending()
            if run['prev_observer'] is not None:
                run['prev_observer'](None)
            return
        touched = self._slices_of(p)
        if not touched:
            if run['prev_observer'] is not None:
                run['prev_observer'](p)
            return
        if run['left'] is not None:
            for i in touched:
                if i in run['launched']:
                    raise RuntimeError('ddp_mode=bucket: slice %d of the gradient buffer received a contribution after it had gone on '
                                       'the wire -- the backward changed under an unchanged step signature' % i)
        stream = torch.cuda.current_stream(self.flat_grad.device) if self.flat_grad.is_cuda else None
        run['pending'].append((touched, stream))

    def _retire_pending(self):
        run = self._run
        pending, run['pending'] = run['pending'], []
        seen = []
        for touched, stream in pending:
            for i in touched:
                run['counts'][i] += 1
                if stream is not None:
                    run['writers'][i].add(stream)
                if run['left'] is not None:
                    run['left'][i] -= 1
                    if i not in seen:
                        seen.append(i)
        for i in seen:                                  # (after the whole batch: a backward announces its writes, then launches them)
            if run['left'][i] == 0:
                self._launch_early(i)
            elif run['left'][i] < 0:
                raise RuntimeError('ddp_mode=bucket: slice %d of the gradient buffer received more contributions than the learning '
                                   'pass counted -- the backward changed under an unchanged step signature' % i)

    def _launch_early(self, i):
        """Slice i is final: every kernel that writes into it has been issued -- on the streams of the chains whose layers share the
        slice, or on a weight-gradient side stream."""
        writers = self._run['writers'][i]
        if not self.flat_grad.is_cuda:                  # CPU tensors (gloo): nothing to order
            self._run['launched'][i] = self._submit_one(i)
        else:
            from . import ops
            if self._comm is None:
                self._comm = ops.new_stream(self.flat_grad.device, 'opt')
            # wait_stream = "everything issued on that stream so far", which contains the slice's writes (and, harmlessly, whatever
            # the chain has queued since)
            for s in list(writers) + ops.wgrad_side_streams():
                self._comm.wait_stream(s)
            with torch.cuda.stream(self._comm):
                if self._check is not None:             # (test hook: what the slice held when it went on the wire)
                    a, b = self.slices[i]
                    self._check[i] = self.flat_grad[a:b].double().sum()
                self._run['launched'][i] = self._submit_one(i)
        self.early_launches += 1

    def end_backward(self):
        """Call when the backward has returned (and its streams have been joined): retires the last write, stores what was learned,
        and hands the early launches to the next _submit()."""
        run, self._run = self._run, None
        if run is None:
            return
        from . import ops
        self._run = run
        try:
            self._retire_pending()
        finally:
            self._run = None
            ops.set_grad_observer(run['prev_observer'])
        if run['left'] is None:
            self._learned[run['key']] = run['counts']
        elif any(run['left']):
            # fewer writes than learned (a branch of the network did not run): forget, exchange the rest the old way, learn again
            self._learned.pop(run['key'], None)
        self._early = run['launched']

    def verify_early_slices(self):
        """HOIG_DDP_CHECK=1, world of one rank (the SUM is the identity): every slice that went on the wire early must still hold what
        it held then.  Host-synchronising; called by the tests after the step's exchange has been waited for."""
        if self._check is None:
            return 0
        bad = 0
        for i, want in self._check.items():
            a, b = self.slices[i]
            bad += int(self.flat_grad[a:b].double().sum().item() != want.item())
        self._check.clear()
        self.check_failures += bad
        return bad

    def broadcast_params(self, src=0):
        """DDP-constructor behaviour: make rank 0's (unseeded, CPU-RNG) initialisation the one everybody uses
        (trainer.py:233-239)."""
        if self.active:
            dist.broadcast(self.flat_param, src=src, group=self.group)

    def all_reduce_grads(self):
        """SUM over ranks, in place; returns the scale (1/world) the optimiser must apply."""
        if self.active:
            from .ops import join_wgrad_streams
            join_wgrad_streams()
            for wait in self._submit():
                wait()
        return 1.0 / self.world

    def iter_all_reduce(self):
        """All slices are submitted at once; yields each slice's (begin, end) once its exchange has been waited for (on NCCL
        / RCCL `wait()` only orders the current stream behind the collective), so a consumer can start on slice i while
        slices i+1.. are still in flight."""
        if not self.active:
            yield (0, self.flat_grad.numel())
            return
        from .ops import join_wgrad_streams
        join_wgrad_streams()
        for wait, ab in zip(self._submit(), self.slices):
            wait()
            yield ab


class FlatDDP(object):
    """Stands where ``DistributedDataParallel(net)`` stands in the reference: exposes ``.module``, forwards calls,
    and saves with the ``module.`` key prefix the reference's DDP checkpoints carry (trainer.py:555-556,
    base_model.py:108-116)."""

    def __init__(self, module, bucket_bytes=64 << 20, payload=None, mode=None):
        self.module = module
        self.sync = GradSync(module.flat, module.flat_grad, bucket_bytes, payload=payload, mode=mode)
        self.sync.broadcast_params(0)

    def forward(self, *a, **k):
        return self.module.forward(*a, **k)

    __call__ = forward

    def forward_nhwc(self, *a, **k):
        return self.module.forward_nhwc(*a, **k)

    def parameters(self):
        return self.module.parameters()

    def train(self, mode=True):
        self.module.train(mode)
        return self

    def eval(self):
        self.module.eval()
        return self

    def state_dict(self):
        return self.module.state_dict(prefix='module.')

    def load_state_dict(self, sd, strict=True):
        return self.module.load_state_dict({(k[7:] if k.startswith('module.') else k): v for k, v in sd.items()}, strict)

    def __getattr__(self, name):
        return getattr(self.__dict__['module'], name)
