"""Parameter storage for the HIP path.

All parameters of a network live in ONE flat fp32 buffer in HBM (and their
gradients / Adam moments in three more), so that
  * the optimiser is a single fused kernel launch over the whole network
    (`hoig_adam_step`) instead of 425 per-tensor launches (trainer.py:275-278,
    425-434),
  * the data-parallel gradient exchange is a handful of large contiguous RCCL
    all-reduces (hoig_amd/ddp.py) instead of DDP's 25 MB buckets,
  * wgrad kernels accumulate straight into the gradient buffer.
Each parameter is an ``nn.Parameter`` VIEW of the flat buffer; ``state_dict()`` /
``load_state_dict()`` present the reference's names and logical NCHW shapes
 so checkpoints interoperate (models/base_model.py:78-124).
Storage layouts differ from the reference where the kernels want it:
  * conv weights are packed [Co][R][S][Ci] (ConvTranspose2d too);
  * the first attention conv (128, 2C, 5, 5) (extract_attn.py:18) is stored as TWO packed 5x5 weights, the target half
    ``<name>#t`` = w[:, :C] and the source half ``<name>#s`` = w[:, C:], both (128, C, 5, 5), because the path runs them as
    two 5x5 convolutions over different tensors (the padded target; the padded source, read back bilinearly).
"""
from collections import OrderedDict

import torch
import torch.nn as nn

from . import _lib as L
from .ops import packed_strides, _p, _st, join_wgrad_streams


def _numel(shape):
    n = 1
    for s in shape:
        n *= s
    return n


def split_attn_weight(w):
    """reference (128, 2C, 5, 5) -> target half w[:, :C], source half w[:, C:] (the reference concatenates
    [target, source] along channels: extract_attn.py:25-26)."""
    c = w.shape[1] // 2
    return w[:, :c], w[:, c:]


def merge_attn_weight(wt, ws):
    return torch.cat([wt, ws], dim=1)


class PendingUpdate(object):
    """An optimiser step of a network (gradient exchange, Adam, operand packing) queued on a side stream.  EVERY stream that
    touches the network's buffers afterwards waits for it -- once per stream: the object remembers who has -- and it stays in
    place until the next step replaces it (a one-shot event, cleared by whichever stream asked first, left the other reader
    streams unordered)."""

    def __init__(self, event):
        self.event = event
        self._waited = set()

    def wait(self, stream=None):
        stream = torch.cuda.current_stream() if stream is None else stream
        if stream.cuda_stream not in self._waited:
            stream.wait_event(self.event)
            self._waited.add(stream.cuda_stream)


class ParamTree(nn.Module):
    """A module tree generated from dotted parameter names; parameters are views of flat buffers."""

    def __init__(self, shapes, device, transposed_names=(), split_names=(), adjacent=()):
        super().__init__()
        self._ref_shapes = OrderedDict(shapes)            # reference names -> reference shapes
        self._split = set(split_names)
        tset = set(transposed_names)
        internal = OrderedDict()                           # internal name -> (shape, transposed)
        for name, shp in self._ref_shapes.items():
            if name in self._split:
                n, c2, r, s = shp
                internal[name + '#t'] = ((n, c2 // 2, r, s), False)
                internal[name + '#s'] = ((n, c2 // 2, r, s), False)
            else:
                internal[name] = (tuple(shp), name in tset)
        self._internal = internal
        # memory order: the reference order, except that SPADE's mlp_gamma / mlp_beta conv weights (and their biases) are
        # placed back to back so the pair is ONE packed conv weight [2C][3][3][128] (fused views in self.F, below)
        order = [n for n in internal if '.mlp_beta.' not in n]
        fused = []
        for n in list(order):
            if n.endswith('.mlp_gamma.weight'):
                pre = n[:-len('.mlp_gamma.weight')]
                order.insert(order.index(n) + 1, pre + '.mlp_beta.weight')
                fused.append(pre)
            elif n.endswith('.mlp_gamma.bias'):
                order.insert(order.index(n) + 1, n.replace('.mlp_gamma.', '.mlp_beta.'))
        # ... and `adjacent` groups of (internal) names, which the caller wants back to back behind the group's first member
        # (conv weights of different modules that read the same tensor: fuse_conv_weights)
        for group in adjacent:
            for n in group[1:]:
                order.remove(n)
            at = order.index(group[0]) + 1
            order[at:at] = list(group[1:])
        assert sorted(order) == sorted(internal), 'allocation order lost a parameter'
        total = 0
        self._offsets = OrderedDict()
        for name in order:
            shp = internal[name][0]
            self._offsets[name] = total
            total += (_numel(shp) + 3) // 4 * 4           # keep every parameter 16-byte aligned
        self.flat = torch.zeros(total, dtype=torch.float32, device=device)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=device)
        self.version = 0          # bumped whenever the weights change (packed bf16 planes are cached per version)
        self.P = OrderedDict()
        gviews = self.views_of(self.flat_grad)
        for name, view in self.views_of(self.flat).items():
            p = nn.Parameter(view, requires_grad=True)
            p.grad = gviews[name]
            p._hoig_flat = True
            p._hoig_transposed = internal[name][1]
            p._hoig_owner = self
            self.P[name] = p
            self._register(name, p)
        # fused gamma|beta views (aliases of the two adjacent parameters; not separate entries of P / state_dict)
        self.F = OrderedDict()
        for pre in fused:
            cg = internal[pre + '.mlp_gamma.weight'][0]
            if _numel(cg) % 4 or cg[0] % 4:
                continue
            off_w, off_b = self._offsets[pre + '.mlp_gamma.weight'], self._offsets[pre + '.mlp_gamma.bias']
            assert self._offsets[pre + '.mlp_beta.weight'] == off_w + _numel(cg)
            assert self._offsets[pre + '.mlp_beta.bias'] == off_b + cg[0]
            shp = (2 * cg[0],) + tuple(cg[1:])
            for key, o, sh, st in ((pre + '.mlp_gb.weight', off_w, shp, packed_strides(shp, False)),
                                   (pre + '.mlp_gb.bias', off_b, (2 * cg[0],), (1,))):
                v = nn.Parameter(self.flat.as_strided(sh, st, o), requires_grad=True)
                v.grad = self.flat_grad.as_strided(sh, st, o)
                v._hoig_flat, v._hoig_transposed, v._hoig_owner = True, False, self
                self.F[key] = v
        self._init_runtime_state()

    def fuse_conv_weights(self, key, names):
        """A (sum Co, Ci, R, S) view over conv weights that sit back to back in the flat store (same Ci, R, S; none transposed):
        registered in self.F like the fused SPADE gamma|beta weight, with the matching gradient view.  Returns False -- and
        registers nothing -- when the members are not adjacent."""
        shapes = [self._internal[n][0] for n in names]
        if any(self._internal[n][1] for n in names) or len({tuple(sh[1:]) for sh in shapes}) != 1:
            return False
        off = self._offsets[names[0]]
        nxt = off
        for n, sh in zip(names, shapes):
            if self._offsets[n] != nxt:
                return False
            nxt += _numel(sh)
        shp = (sum(sh[0] for sh in shapes),) + tuple(shapes[0][1:])
        st = packed_strides(shp, False)
        v = nn.Parameter(self.flat.as_strided(shp, st, off), requires_grad=True)
        v.grad = self.flat_grad.as_strided(shp, st, off)
        v._hoig_flat, v._hoig_transposed, v._hoig_owner = True, False, self
        self.F[key] = v
        return True

    def _init_runtime_state(self):
        self._pending = None      # event of an optimiser step still running on a side stream (Trainer._step)
        self._f6 = None
        self._plane_bufs = None
        self._plane_version = -1
        self._plane_flags = {}
        self._plane_tiles = 0

    # --- split-bf16 operand planes of every conv weight, refreshed by ONE kernel launch per weight version
    def _build_plane_table(self):
        rows = []
        used = [(n, self._internal[n][0]) for n in self._offsets
                if len(self._internal[n][0]) == 4 and not (('.mlp_gamma.' in n or '.mlp_beta.' in n)
                                                            and n.replace('.mlp_gamma.', '.mlp_gb.').replace(
                                                                '.mlp_beta.', '.mlp_gb.') in self.F)]
        used = [(self._offsets[n], shp, self._internal[n][1]) for n, shp in used]
        for key, v in self.F.items():
            if v.dim() == 4:
                used.append((v.storage_offset(), tuple(v.shape), False))
        for off, shp, transposed in used:
            ci, co = (shp[0], shp[1]) if transposed else (shp[1], shp[0])
            flags = 0
            if ci % 32 == 0 and co % 32 == 0:      # same conditions as ops._conv_fwd_raw / _conv_dgrad_raw
                flags = (1 if co > 32 else 0) | (2 if ci > 32 else 0)
            if flags:
                rows.append([off, co, shp[2] * shp[3], ci, flags, self._plane_tiles])
                self._plane_tiles += ((co + 31) // 32) * ((ci + 31) // 32)
                self._plane_flags[off] = flags
        return rows

    def refresh_planes(self):
        """Make the operand planes of the current weights on the CURRENT stream (creating the buffers on first use): a forward
        that forks onto several streams calls this first, so that no chain makes them under another chain's feet."""
        self.wait_pending()
        from . import ops
        if ops.precision != L.PREC_F32:
            self._ensure_plane_bufs()
        self._refresh_planes()
        if ops.precision == L.PREC_F16F6:       # (the fp6 records too)
            self._refresh_f6()

    def _ensure_plane_bufs(self):
        if self._plane_bufs is None:
            rows = self._build_plane_table()
            if not rows:
                self._plane_bufs = ()
                return
            self._plane_table = torch.tensor(rows, dtype=torch.int64, device=self.flat.device)
            self._plain_chunks = self._build_plain_chunks(rows)
            self._plane_bufs = tuple(torch.empty(self.flat.numel(), dtype=torch.int16, device=self.flat.device)
                                     for _ in range(4))

    def _build_plain_chunks(self, rows):
        """The parts of the flat buffer that no row of the plane table covers, as (first element, count <= 1024) chunks for
        hoig_adam_pack_step; None if two rows overlap (the fused step is then not used)."""
        spans = sorted((r[0], r[0] + r[1] * r[2] * r[3]) for r in rows)
        chunks, at = [], 0
        for a, b in spans + [(self.flat.numel(), self.flat.numel())]:
            if a < at:
                return None
            while at < a:
                n = min(1024, a - at)
                chunks.append([at, n])
                at += n
            at = b
        if any(c[0] % 4 or c[1] % 4 for c in chunks):
            return None
        return torch.tensor(chunks if chunks else [[0, 0]], dtype=torch.int64, device=self.flat.device), len(chunks)

    def fused_step_tables(self):
        """(plane table, tile count, plain chunks, chunk count, the four plane buffers) for hoig_adam_pack_step, or None while the
        planes do not exist (fp32 mode, before the first forward) or cannot be updated in the optimiser's launch."""
        if not self._plane_bufs or getattr(self, '_plain_chunks', None) is None:
            return None
        return (self._plane_table, self._plane_tiles) + self._plain_chunks + (self._plane_bufs,)

    def _refresh_planes(self):
        """Split the current weights into bf16 planes (one launch) if they changed since the last split.  Called lazily by
        packed_planes(); a forward that forks onto several streams calls it FIRST, on the main stream, so that no branch
        races the split."""
        if self._plane_bufs and self._plane_version != self.version:
            b = self._plane_bufs
            L.call('hoig_pack_conv_weights_bf16_all', _p(self.flat), _p(self._plane_table), self._plane_table.shape[0],
                   self._plane_tiles, _p(b[0]), _p(b[1]), _p(b[2]), _p(b[3]), _st())
            self._plane_version = self.version

    def packed_planes(self, w, for_dgrad):
        """(hi, lo) bf16 planes of conv weight `w` (a view of self.flat) for the forward (for_dgrad=False) or data-
        gradient GEMM, or None if `w` is not in the table (hoig_pack_conv_weights_bf16_all, include/hoig_kernels.h)."""
        self._ensure_plane_bufs()
        if not self._plane_bufs:
            return None
        off = w.storage_offset()
        if not (self._plane_flags.get(off, 0) & (2 if for_dgrad else 1)) or w.data_ptr() != self.flat.data_ptr() + 4 * off:
            return None
        self.refresh_planes()
        n = w.numel()
        hi, lo = (self._plane_bufs[2], self._plane_bufs[3]) if for_dgrad else (self._plane_bufs[0], self._plane_bufs[1])
        return hi[off:off + n], lo[off:off + n]

    # --- fp6 records of the 3x3 weights (ops._f6_planes): all of the network's in one launch per weight version
    def _build_f6_table(self):
        used = [(self._offsets[n], self._internal[n][0]) for n in self._offsets
                if len(self._internal[n][0]) == 4 and not self._internal[n][1] and not (
                    ('.mlp_gamma.' in n or '.mlp_beta.' in n)
                    and n.replace('.mlp_gamma.', '.mlp_gb.').replace('.mlp_beta.', '.mlp_gb.') in self.F)]
        used += [(v.storage_offset(), tuple(v.shape)) for v in self.F.values() if v.dim() == 4]
        rows, byte_off, task0, where = [], 0, 0, {}
        for off, (co, ci, r, s_) in used:
            if r != 3 or s_ != 3 or ci % 64 or co % 64:
                continue
            nbytes = L.lib.hoig_f6_plane_bytes(co, 9, ci)
            rows.append([off, co, 9, ci, byte_off, task0])
            where[off] = (byte_off, nbytes, (co, ci))
            byte_off += (nbytes + 255) // 256 * 256
            task0 += co * 9 * (ci // 32)
        return rows, byte_off, task0, where

    def _refresh_f6(self):
        if self._f6 is None:
            rows, nbytes, ntasks, where = self._build_f6_table()
            self._f6 = dict(where=where, ntasks=ntasks, version=-1, bufs=None, nbytes=nbytes,
                            table=torch.tensor(rows, dtype=torch.int64, device=self.flat.device) if rows else None)
        f = self._f6
        if f['table'] is None:
            return f
        if f['bufs'] is None:
            f['bufs'] = (torch.empty(f['nbytes'], dtype=torch.uint8, device=self.flat.device),
                         torch.empty(f['nbytes'], dtype=torch.uint8, device=self.flat.device))
        if f['version'] != self.version:
            self.wait_pending()
            L.call('hoig_pack_conv_weights_f6_all', _p(self.flat), _p(f['table']), f['table'].shape[0], f['ntasks'],
                   _p(f['bufs'][0]), _p(f['bufs'][1]), _st())
            f['version'] = self.version
        return f

    def packed_f6(self, w):
        """(q_hi, q_lo) fp6 record arrays of conv weight `w` (a view of self.flat; hoig_pack_conv_weights_f6_all), or None if
        `w` is not a 3x3 weight with Ci, Co multiples of 64."""
        off = w.storage_offset()
        f = self._f6
        if f is not None and (f['table'] is None or off not in f['where']):
            return None
        if w.data_ptr() != self.flat.data_ptr() + 4 * off:
            return None
        f = self._refresh_f6()
        if f['table'] is None or off not in f['where']:
            return None
        a, n, shp = f['where'][off]
        if tuple(w.shape[:2]) != shp:                 # (mlp_gamma alone starts where the fused gamma|beta weight does)
            return None
        return f['bufs'][0][a:a + n], f['bufs'][1][a:a + n]

    def set_pending(self, event):
        """An update of this network's buffers is in flight on another stream (None: no longer); readers call wait_pending()."""
        self._pending = PendingUpdate(event) if event is not None else None

    def wait_pending(self, stream=None):
        if self._pending is not None:
            self._pending.wait(stream)

    def _register(self, dotted, p):
        parts = dotted.split('.')
        mod = self
        for part in parts[:-1]:
            if not hasattr(mod, part) or not isinstance(getattr(mod, part), nn.Module):
                mod.add_module(part, nn.Module())
            mod = getattr(mod, part)
        mod.register_parameter(parts[-1].replace('#', '_'), p)

    def views_of(self, flat):
        """internal name -> strided view of `flat` (same layout as the parameters)."""
        out = OrderedDict()
        for name, (shp, transposed) in self._internal.items():
            off, n = self._offsets[name], _numel(shp)
            if len(shp) == 4:
                out[name] = flat.as_strided(shp, packed_strides(shp, transposed), off)
            else:
                out[name] = flat[off:off + n].view(shp)
        return out

    # --- reference-compatible (de)serialisation of any flat buffer (weights, Adam moments)
    def export_dict(self, flat, prefix=''):
        self.wait_pending()
        v = self.views_of(flat)
        out = OrderedDict()
        for name in self._ref_shapes:
            if name in self._split:
                t = merge_attn_weight(v[name + '#t'].detach(), v[name + '#s'].detach())
            else:
                t = v[name].detach()
            out[prefix + name] = t.clone(memory_format=torch.contiguous_format)
        return out

    def import_dict(self, flat, sd, strict=True):
        missing = [k for k in self._ref_shapes if k not in sd]
        unexpected = [k for k in sd if k not in self._ref_shapes]
        if strict and (missing or unexpected):
            raise RuntimeError('load_state_dict: missing %s unexpected %s' % (missing[:5], unexpected[:5]))
        self.wait_pending()
        v = self.views_of(flat)
        with torch.no_grad():
            for name, shp in self._ref_shapes.items():
                if name not in sd:
                    continue
                src = sd[name]
                if tuple(src.shape) != tuple(shp):
                    raise RuntimeError('size mismatch for %s: %s vs %s' % (name, tuple(src.shape), tuple(shp)))
                src = src.to(device=flat.device, dtype=flat.dtype)
                if name in self._split:
                    wt, ws = split_attn_weight(src)
                    v[name + '#t'].copy_(wt)
                    v[name + '#s'].copy_(ws)
                else:
                    v[name].copy_(src)

    def state_dict(self, *args, **kwargs):
        return self.export_dict(self.flat, kwargs.get('prefix', ''))

    def load_state_dict(self, sd, strict=True):
        self.import_dict(self.flat, sd, strict)
        self.version += 1
        return self

    def ref_names(self):
        return list(self._ref_shapes.keys())

    def parameters(self, recurse=True):
        return iter(self.P.values())

    def named_parameters(self, prefix='', recurse=True, remove_duplicate=True):
        for k, v in self.P.items():
            yield (prefix + ('.' if prefix else '') + k, v)

    def zero_grad(self, set_to_none=False):
        join_wgrad_streams()
        self.flat_grad.zero_()

    def set_requires_grad(self, flag):
        for p in list(self.P.values()) + list(self.F.values()):
            p.requires_grad_(flag)

    def cuda(self, device=None):      # storage is created on the target device; moving would break the flat views
        return self

    def init_weights(self, generator=None):
        """NetworkBase.init_weights (base_network.py:14-25): every Conv* weight ~ N(0, 0.02), conv bias 0;
        InstanceNorm affine parameters stay (1, 0)."""
        sd = OrderedDict()
        for name, shp in self._ref_shapes.items():
            if len(shp) == 4:
                t = torch.empty(tuple(shp), dtype=torch.float32)
                t.normal_(0.0, 0.02, generator=generator)
            elif name.endswith('.bias'):
                t = torch.zeros(tuple(shp))
            else:
                t = torch.ones(tuple(shp))
            sd[name] = t
        return self.load_state_dict(sd)


class FusedAdam(object):
    """torch.optim.Adam semantics (lr, betas, eps=1e-8, no weight decay / amsgrad: trainer.py:275-278) as ONE
    kernel over the network's flat buffers.  ``state_dict()`` / ``load_state_dict()`` use torch.optim.Adam's
    per-parameter layout over the REFERENCE parameter list so optimiser checkpoints interoperate
    (base_model.py:78-90).

    The schedule lives in DEVICE memory ({lr, beta1, beta2, eps, step}: hoig_adam_tick advances the step count and derives the
    bias corrections there), so a step is a pure function of device state and can be captured in a hipGraph
    (Trainer._graph_step).  ``param_groups`` / ``step_count`` are the host's view: a change made on the host (a new learning
    rate, a loaded checkpoint) is uploaded before the next step; a replayed step is reported with ``replayed()``."""

    def __init__(self, tree, lr, betas=(0.9, 0.999), eps=1e-8):
        self.tree = tree
        self.param_groups = [dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=0, amsgrad=False,
                                  params=list(range(len(tree.ref_names()))))]
        self.exp_avg = torch.zeros_like(tree.flat)
        self.exp_avg_sq = torch.zeros_like(tree.flat)
        self.step_count = 0
        self.fuse_planes = True         # the update writes the operand planes of the new weights too (hoig_adam_pack_step); False: two launches
        self._state = torch.zeros(5, dtype=torch.float64, device=tree.flat.device)
        self._derived = torch.zeros(8, dtype=torch.float32, device=tree.flat.device)
        self._on_device = None          # the host values self._state was last written from

    def zero_grad(self, set_to_none=False):
        join_wgrad_streams()
        self.tree.flat_grad.zero_()

    def _host_state(self):
        g = self.param_groups[0]
        return (float(g['lr']), float(g['betas'][0]), float(g['betas'][1]), float(g['eps']), float(self.step_count))

    def sync_state(self):
        """Upload the host's schedule if it differs from what the device holds (construction, update_learning_rate, a loaded
        checkpoint).  Never inside a capture: a captured step must start from the steady state."""
        want = self._host_state()
        if want != self._on_device:
            from . import ops
            if ops.capturing():
                raise RuntimeError('FusedAdam: the optimiser schedule changed while a hipGraph was being captured')
            self._state.copy_(torch.tensor(want, dtype=torch.float64))
            self._on_device = want

    def replayed(self):
        """A captured step of this optimiser has been replayed: the device advanced its step count, follow it."""
        self.step_count += 1
        self._on_device = self._on_device[:4] + (float(self.step_count),)
        # (tree.version stays: it only keys the operand-plane caches, which the captured step re-packs itself)

    def step(self, grad_scale=1.0, ready=None):
        """One Adam step over the flat buffers.  `ready` (DDP): an iterator of (begin, end) element ranges whose gradients
        have just been exchanged -- the update of a range is launched as soon as it is yielded, so Adam pipelines behind
        the sliced all-reduce instead of waiting for the last slice (GradSync.iter_all_reduce)."""
        self.sync_state()
        join_wgrad_streams()
        L.call('hoig_adam_tick', _p(self._state), _p(self._derived), _st())
        self.step_count += 1
        self._on_device = self._on_device[:4] + (float(self.step_count),)
        fl, gr, m, v = self.tree.flat, self.tree.flat_grad, self.exp_avg, self.exp_avg_sq
        fused = self.tree.fused_step_tables() if (ready is None and self.fuse_planes) else None
        if fused is not None:
            # the update and the operand planes of the updated weights in one launch (include/hoig_kernels.h)
            table, ntiles, plain, nplain, b = fused
            L.call('hoig_adam_pack_step', _p(fl), _p(gr), _p(m), _p(v), _p(self._derived), grad_scale, _p(table), table.shape[0],
                   ntiles, _p(plain), nplain, _p(b[0]), _p(b[1]), _p(b[2]), _p(b[3]), _st())
            self.tree.version += 1
            self.tree._plane_version = self.tree.version
            return
        for a, b in (ready if ready is not None else [(0, fl.numel())]):
            L.call('hoig_adam_step_dev', fl.data_ptr() + 4 * a, gr.data_ptr() + 4 * a, m.data_ptr() + 4 * a,
                   v.data_ptr() + 4 * a, b - a, _p(self._derived), grad_scale, _st())
        self.tree.version += 1

    def state_dict(self):
        state = {}
        if self.step_count > 0:
            ms = self.tree.export_dict(self.exp_avg)
            vs = self.tree.export_dict(self.exp_avg_sq)
            for i, name in enumerate(self.tree.ref_names()):
                state[i] = dict(step=torch.tensor(float(self.step_count)), exp_avg=ms[name], exp_avg_sq=vs[name])
        groups = [dict((k, v) for k, v in g.items()) for g in self.param_groups]
        return dict(state=state, param_groups=groups)

    def load_state_dict(self, sd):
        st = sd['state']
        names = self.tree.ref_names()
        ms, vs, steps = {}, {}, set()
        for i, name in enumerate(names):
            if i in st:
                ms[name], vs[name] = st[i]['exp_avg'], st[i]['exp_avg_sq']
                s = st[i]['step']
                steps.add(int(s.item()) if torch.is_tensor(s) else int(s))
        if len(steps) > 1:
            raise RuntimeError('FusedAdam: per-parameter step counts differ: %s' % sorted(steps))
        self.tree.import_dict(self.exp_avg, ms, strict=False)
        self.tree.import_dict(self.exp_avg_sq, vs, strict=False)
        self.step_count = steps.pop() if steps else 0
        g = sd['param_groups'][0]
        self.param_groups[0].update(lr=g['lr'], betas=tuple(g['betas']), eps=g['eps'])
