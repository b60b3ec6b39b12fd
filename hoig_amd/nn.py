"""Parameter storage for the HIP path.

All parameters of a network live in ONE flat fp32 buffer in HBM (and their
gradients / Adam moments in three more), so that
  * the optimiser is a single fused kernel launch over the whole network
    (`hoig_adam_step`) instead of 425 per-tensor launches (trainer.py:275-278,
    425-434),
  * the data-parallel gradient exchange is a handful of large contiguous RCCL
    all-reduces (hoig_amd/ddp.py) instead of DDP's 25 MB buckets,
  * wgrad kernels accumulate straight into the gradient buffer.
Each parameter is an ``nn.Parameter`` VIEW of the flat buffer that keeps the
reference's logical name and shape so ``state_dict()`` /
``load_state_dict()`` interoperate with reference checkpoints
(models/base_model.py:78-124); conv weights are stored packed [Co][R][S][Ci].
"""
from collections import OrderedDict

import torch
import torch.nn as nn

from . import _lib as L
from .ops import packed_strides, _p, _st


def _numel(shape):
    n = 1
    for s in shape:
        n *= s
    return n


class ParamTree(nn.Module):
    """A module tree generated from dotted parameter names; parameters are views of flat buffers."""

    def __init__(self, shapes, device, transposed_names=()):
        super().__init__()
        self._shapes = OrderedDict(shapes)
        total = 0
        offsets = OrderedDict()
        for name, shp in self._shapes.items():
            offsets[name] = total
            total += (_numel(shp) + 3) // 4 * 4          # keep every parameter 16-byte aligned
        self.flat = torch.zeros(total, dtype=torch.float32, device=device)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=device)
        self.P = OrderedDict()
        self.version = 0                 # bumped whenever the weights change (packed bf16 planes are cached per version)
        self._offsets = offsets
        tset = set(transposed_names)
        for name, shp in self._shapes.items():
            off, n = offsets[name], _numel(shp)
            if len(shp) == 4:
                st = packed_strides(shp, name in tset)
                view = self.flat.as_strided(shp, st, off)
                gview = self.flat_grad.as_strided(shp, st, off)
            else:
                view = self.flat[off:off + n].view(shp)
                gview = self.flat_grad[off:off + n].view(shp)
            p = nn.Parameter(view, requires_grad=True)
            p.grad = gview
            p._hoig_flat = True
            p._hoig_transposed = name in tset
            p._hoig_owner = self
            self.P[name] = p
            self._register(name, p)

    def _register(self, dotted, p):
        parts = dotted.split('.')
        mod = self
        for part in parts[:-1]:
            if not hasattr(mod, part) or not isinstance(getattr(mod, part), nn.Module):
                mod.add_module(part, nn.Module())
            mod = getattr(mod, part)
        mod.register_parameter(parts[-1], p)

    # --- reference-compatible (de)serialisation: contiguous NCHW tensors under the reference names
    def state_dict(self, *args, **kwargs):
        out = OrderedDict()
        prefix = kwargs.get('prefix', '')
        for name, p in self.P.items():
            out[prefix + name] = p.detach().clone(memory_format=torch.contiguous_format)
        return out

    def load_state_dict(self, sd, strict=True):
        missing = [k for k in self.P if k not in sd]
        unexpected = [k for k in sd if k not in self.P]
        if strict and (missing or unexpected):
            raise RuntimeError('load_state_dict: missing %s unexpected %s' % (missing[:5], unexpected[:5]))
        with torch.no_grad():
            for name, p in self.P.items():
                if name in sd:
                    src = sd[name]
                    if tuple(src.shape) != tuple(p.shape):
                        raise RuntimeError('size mismatch for %s: %s vs %s' % (name, tuple(src.shape), tuple(p.shape)))
                    p.copy_(src.to(device=p.device, dtype=p.dtype))
        self.version += 1
        return self

    def parameters(self, recurse=True):
        return iter(self.P.values())

    def named_parameters(self, prefix='', recurse=True, remove_duplicate=True):
        for k, v in self.P.items():
            yield (prefix + ('.' if prefix else '') + k, v)

    def zero_grad(self, set_to_none=False):
        self.flat_grad.zero_()

    def set_requires_grad(self, flag):
        for p in self.P.values():
            p.requires_grad_(flag)

    def cuda(self, device=None):      # storage is created on the target device; moving would break the flat views
        return self

    def init_weights(self, generator=None):
        """NetworkBase.init_weights (base_network.py:14-25): every Conv* weight ~ N(0, 0.02), conv bias 0;
        InstanceNorm affine parameters stay (1, 0)."""
        with torch.no_grad():
            for name, p in self.P.items():
                if p.dim() == 4:
                    tmp = torch.empty(tuple(p.shape), dtype=torch.float32)
                    tmp.normal_(0.0, 0.02, generator=generator)
                    p.copy_(tmp)
                elif name.endswith('.bias'):
                    p.zero_()
                else:
                    p.fill_(1.0)
        self.version += 1
        return self


class FusedAdam(object):
    """torch.optim.Adam semantics (lr, betas, eps=1e-8, no weight decay / amsgrad: trainer.py:275-278) as ONE
    kernel over the network's flat buffers.  ``state_dict()`` / ``load_state_dict()`` use torch.optim.Adam's
    per-parameter layout so optimiser checkpoints interoperate (base_model.py:78-90)."""

    def __init__(self, tree, lr, betas=(0.9, 0.999), eps=1e-8):
        self.tree = tree
        self.param_groups = [dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=0, amsgrad=False,
                                  params=list(range(len(tree.P))))]
        self.exp_avg = torch.zeros_like(tree.flat)
        self.exp_avg_sq = torch.zeros_like(tree.flat)
        self.step_count = 0

    def zero_grad(self, set_to_none=False):
        self.tree.flat_grad.zero_()

    def step(self, grad_scale=1.0):
        g = self.param_groups[0]
        self.step_count += 1
        L.call('hoig_adam_step', _p(self.tree.flat), _p(self.tree.flat_grad), _p(self.exp_avg), _p(self.exp_avg_sq),
               self.tree.flat.numel(), g['lr'], g['betas'][0], g['betas'][1], g['eps'], self.step_count,
               grad_scale, _st())
        self.tree.version += 1

    def _views(self, flat):
        out = []
        for name, shp in self.tree._shapes.items():
            off, n = self.tree._offsets[name], _numel(shp)
            p = self.tree.P[name]
            if len(shp) == 4:
                out.append(flat.as_strided(shp, p.stride(), off))
            else:
                out.append(flat[off:off + n].view(shp))
        return out

    def state_dict(self):
        state = {}
        if self.step_count > 0:
            for i, (m, v) in enumerate(zip(self._views(self.exp_avg), self._views(self.exp_avg_sq))):
                state[i] = dict(step=torch.tensor(float(self.step_count)),
                                exp_avg=m.detach().clone(memory_format=torch.contiguous_format),
                                exp_avg_sq=v.detach().clone(memory_format=torch.contiguous_format))
        groups = [dict((k, v) for k, v in g.items()) for g in self.param_groups]
        return dict(state=state, param_groups=groups)

    def load_state_dict(self, sd):
        st = sd['state']
        with torch.no_grad():
            ms, vs = self._views(self.exp_avg), self._views(self.exp_avg_sq)
            steps = set()
            for i, (m, v) in enumerate(zip(ms, vs)):
                if i in st:
                    m.copy_(st[i]['exp_avg'].to(m.device))
                    v.copy_(st[i]['exp_avg_sq'].to(v.device))
                    s = st[i]['step']
                    steps.add(int(s.item()) if torch.is_tensor(s) else int(s))
            if len(steps) > 1:
                raise RuntimeError('FusedAdam: per-parameter step counts differ: %s' % sorted(steps))
            self.step_count = steps.pop() if steps else 0
        g = sd['param_groups'][0]
        self.param_groups[0].update(lr=g['lr'], betas=tuple(g['betas']), eps=g['eps'])
