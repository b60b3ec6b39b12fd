"""``Trainer`` -- the HOGAN model object the reference drivers (train.py, train_ddp.py, eval.py) talk to.

Same public surface as the reference's ``class Trainer(BaseModel)`` (models/trainer.py:188-591): ``set_input /
set_train / set_eval / forward / optimize_parameters / _optimize_G / _optimize_D / get_current_errors /
get_current_scalars / get_current_visuals / save / load / update_learning_rate``; ``HOGANModel``, ``backward_G`` and
``backward_D`` are aliases for the names BASELINE.json's north_star uses.  Underneath everything is new: NHWC fp32
tensors in HBM, every operator a hand-written gfx950 kernel (hoig_amd/ops.py), one flat buffer per network, fused
Adam, RCCL gradient exchange overlapped with the D step (hoig_amd/ddp.py).

``set_input`` takes the raw dataloader batch train_ddp.py:92 passes (imageA/B, maskA/B, manoA/B) when the caller supplies the MANO
model and the per-object renderer buffers -- assets the reference does not ship (``opt.mano_model``, ``opt.object_assets``;
hoig_amd/hand_recovery.py runs ``HandRecoveryFlow.forward``, trainer.py:46-145, as three device stages) -- or images + rasteriser
outputs, or the prepared tensors that stage produces (the a2 surface: a dict with the keys of ``hoig_amd.synthetic.make_inputs``).
"""
import math
import os
import sys
from collections import OrderedDict

import numpy as np
import torch
import torch.distributed as dist

from .. import _lib, ops
from ..options import is_dexycb
from ..ddp import FlatDDP
from ..nn import FusedAdam
from .base_model import BaseModel
from .networks import NetworksFactory
from .networks.generator import to_nhwc, as_nchw, forks_streams as generator_forks_streams
from .networks.vgg19 import Vgg19, VGGLoss

# HOIG_GRAPH=1 (or opt.hip_graph=True): replay the training step as a captured hipGraph after two eager iterations per batch shape.
# Off by default: on this ROCm the replay costs the host 4 ms instead of 28 ms per step, but the graph's internal stream mapping
# overlaps the chains less than the eager streams do -- 80.2 / 81.7 ms per replayed step against 74.8 / 79.5 ms eager, two boxes
#  -- and the step is GPU-bound either way.
_GRAPH = os.environ.get('HOIG_GRAPH', '0') == '1'
_GRAPH_WARMUP = 2         # eager steps per input signature before the capture (lazily made buffers and planes then exist)
_GRAPH_MAX_SIGNATURES = 3
# test hook (tests/test_trainer_gpu.py): cycles the optimiser side stream idles before each step's exchange + Adam, so that a
# reader stream that is not ordered behind it shows up as a parity failure instead of hiding behind timing
_TEST_SIDE_DELAY = 0
# test hook (tests/test_ddp_rccl_gpu.py): a list that receives (tag, timing event) marks of the step's phases, to check that G's
# exchange + Adam on the side stream really finish inside the D step they are meant to hide behind
_TEST_TRACE = None


def _mark(tag, stream=None):
    if _TEST_TRACE is not None and not ops.capturing():
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(stream if stream is not None else torch.cuda.current_stream())
        _TEST_TRACE.append((tag, ev))

PREPARED_KEYS = ['input_G_bg', 'input_G_src_obj', 'input_G_tsf_obj', 'input_G_src_hand', 'input_G_tsf_hand', 'T',
                 'real_src', 'real_tsf', 'bg_mask', 'hand_mask']
RASTER_KEYS = ['src_img', 'ref_img', 'src_faces', 'src_fim', 'src_wim', 'ref_fim', 'ref_wim', 'tables']
RAW_KEYS = ['imageA', 'imageB', 'manoA', 'manoB']          # what train_ddp.py:92 passes (trainer.py:324-331)


def _labelcolormap(n):
    """utils/util.py:22-44 (non-cityscapes branch)."""
    cmap = np.zeros((n, 3), dtype=np.uint8)
    for i in range(n):
        r = g = b = 0
        ident = i + 1
        for j in range(7):
            bits = [(ident >> y) & 1 for y in range(7, -1, -1)]
            r ^= bits[-1] << (7 - j)
            g ^= bits[-2] << (7 - j)
            b ^= bits[-3] << (7 - j)
            ident >>= 3
        cmap[i] = (r & 0xFF, g & 0xFF, b & 0xFF)
    return cmap


class Trainer(BaseModel):
    _use_graph = False
    _static_inputs = None
    _sig = None

    def __init__(self, opt, use_ddp=False):
        super(Trainer, self).__init__(opt, use_ddp)
        self._name = 'Trainer'
        if not torch.cuda.is_available():
            raise RuntimeError('hoig_amd.Trainer needs an MI355X: the product path has no CPU fallback')
        if use_ddp:
            self.device = torch.device('cuda:{}'.format(opt.local_rank))
        else:
            self.device = torch.device('cuda', torch.cuda.current_device())
        torch.cuda.set_device(self.device)
        self._dexycb = is_dexycb(opt)
        self._world = dist.get_world_size() if (use_ddp and dist.is_initialized()) else 1
        # G's gradient exchange + Adam run on a side stream beside the D step; D's run there beside the next forward of G
        self._side = ops.new_stream(self.device, 'opt')
        self._loss_streams = None
        self._d_stream = None
        self._use_graph = bool(getattr(opt, 'hip_graph', _GRAPH))
        self._graphs = {}                # (trainable, input signature, arithmetic) -> captured step
        self._graph_pool = None
        self._static_inputs = {}         # input signature -> the staged tensors a captured step reads
        self._sig = None

        self._init_create_networks(use_ddp=use_ddp)
        if self._is_train:
            self._init_train_vars()
            self._init_losses(use_ddp=use_ddp)

        if self._opt.load_path != 'None' and self._opt.load_path is not None:
            self._load_params(self._G, self._opt.load_path, need_module=False)
        elif not self._is_train or self._opt.load_epoch > 0:
            self.load()

        self._cmap = _labelcolormap(16)
        self._init_prefetch_inputs()

    def close(self):
        """Hand the step's HIP streams back to the library's banks (ops.release_stream): a process that builds Trainer after Trainer
        keeps a flat number of streams.  Called by __del__; the object must not be used afterwards."""
        gen = self._net(self._G) if getattr(self, '_G', None) is not None else None
        held = [getattr(self, '_side', None), getattr(self, '_d_stream', None)]
        held += list(getattr(self, '_loss_streams', None) or ()) + list(getattr(gen, '_streams', None) or ())
        self._graphs = {}                # (captured steps hold the streams they were captured on)
        self._side = self._d_stream = self._loss_streams = None
        if gen is not None:
            gen._streams = None
        for st in held:
            if st is not None:
                ops.release_stream(st)

    def __del__(self):
        try:
            self.close()
        except Exception:                # (interpreter shutdown: modules may be gone)
            pass

    # ------------------------------------------------------------------ construction (trainer.py:217-322)
    def _init_create_networks(self, use_ddp=False):
        self._hdr = None          # hoig_amd.hand_recovery.HandRecoveryFlow, built on the first raw batch (set_raw_input)
        self._G = self._create_generator()
        self._G.init_weights()
        payload = getattr(self._opt, 'ddp_payload', None)        # None: HOIG_DDP_PAYLOAD / fp32 (hoig_amd/ddp.py)
        mode = getattr(self._opt, 'ddp_mode', None)              # None: HOIG_DDP_MODE / 'after'; 'bucket': slices go on the wire during
        if use_ddp:                                              # G's backward (D's 28 MB are one slice: nothing to bucket)
            self._G = FlatDDP(self._G, payload=payload, mode=mode)
        self._D = self._create_discriminator()
        self._D.init_weights()
        if use_ddp:
            self._D = FlatDDP(self._D, payload=payload)

    def _create_generator(self):
        if not self._opt.use_spade:
            raise NotImplementedError('use_spade=False is unreachable in the reference (argument order / channel '
                                      'mismatch at trainer.py:395-398); pass --use_spade as every script does')
        bg_dim, cond = (13, 9) if self._dexycb else (8, 3)     # HOIG_DexYCB/models/trainer.py:263-264 / HOv3 :260-261
        return NetworksFactory.get_by_name(self._opt.gen_name, bg_dim=bg_dim, img_dim=3, obj_dim=3, img_cond_dim=cond,
                                           obj_cond_dim=12, repeat_num=self._opt.repeat_num,
                                           conv_dim=getattr(self._opt, 'conv_dim', 64))

    def _create_discriminator(self):
        return NetworksFactory.get_by_name('discriminator_patch_gan', input_nc=24 if self._dexycb else 19,
                                           norm_type=self._opt.norm_type, ndf=64, n_layers=4, use_sigmoid=False)

    @staticmethod
    def _net(x):
        return x.module if isinstance(x, FlatDDP) else x

    def _init_train_vars(self):
        self._current_lr_G = self._opt.lr_G
        self._current_lr_D = self._opt.lr_D
        self._optimizer_G = FusedAdam(self._net(self._G), lr=self._current_lr_G,
                                      betas=(self._opt.G_adam_b1, self._opt.G_adam_b2))
        self._optimizer_D = FusedAdam(self._net(self._D), lr=self._current_lr_D,
                                      betas=(self._opt.D_adam_b1, self._opt.D_adam_b2))

    def _init_prefetch_inputs(self):
        self._real_src = self._real_tsf = self._bg_mask = self._hand_mask = None
        self._input_G_bg = self._input_G_src_obj = self._input_G_tsf_obj = None
        self._input_G_src_hand = self._input_G_tsf_hand = self._T = None
        self._armask_src = self._armask_tsf = None

    def _init_losses(self, use_ddp=False):
        """trainer.py:302-304 builds ``Vgg19()`` = torchvision's ImageNet-pretrained VGG19.  Those weights are looked for at
        ``opt.vgg_weights`` (a ``vgg19().features`` or full-model state_dict), then in torch hub's cache, where torchvision
        itself would have put them.  Without them the perceptual loss would be computed against a random feature extractor:
        that is refused unless the caller opts in with ``opt.vgg_surrogate=True`` (benchmarks and parity tests, where both
        sides carry the same deterministic surrogate weights)."""
        vgg_net = Vgg19()
        vgg_path = getattr(self._opt, 'vgg_weights', None)
        if not vgg_path:
            hub = os.path.join(torch.hub.get_dir(), 'checkpoints', 'vgg19-dcbb9e9d.pth')
            vgg_path = hub if os.path.exists(hub) else None
        if vgg_path:
            sd = torch.load(vgg_path, map_location='cpu')
            if any(k.startswith('features.') for k in sd):
                sd = {k[len('features.'):]: v for k, v in sd.items() if k.startswith('features.')}
            vgg_net.load_torchvision_features(sd)
        elif self._opt.use_vgg and not getattr(self._opt, 'vgg_surrogate', False):
            raise RuntimeError(
                'Trainer: --use_vgg needs the ImageNet VGG19 weights (the reference downloads them through torchvision, '
                'models/networks/vgg19.py:56); none found at opt.vgg_weights or in %s.  Pass --vgg_weights <vgg19 .pth>, or '
                'opt in to deterministic surrogate weights with opt.vgg_surrogate=True (benchmarks / parity tests only: the '
                'perceptual loss is then NOT the reference\'s).' % os.path.join(torch.hub.get_dir(), 'checkpoints'))
        elif self._opt.use_vgg:
            print('hoig_amd.Trainer: WARNING: VGG19 perceptual loss runs on SURROGATE (random He-normal) weights '
                  '(opt.vgg_surrogate=True)', file=sys.stderr)
        if self._opt.use_vgg:
            self._crt_tsf = VGGLoss(vgg=vgg_net)
        # the terms of the two objectives live in device slots that the loss kernels add into (ops.LossSlots)
        self._g_terms = ops.LossSlots(['g_adv', 'g_rec', 'g_tsf', 'g_mask', 'g_mask_smooth'], self.device)
        self._d_terms = ops.LossSlots(['d'], self.device, extra=['d_real', 'd_fake'])
        z = lambda: torch.zeros((), device=self.device)
        self._loss_g_rec, self._loss_g_tsf, self._loss_g_adv = z(), z(), z()
        self._loss_g_smooth, self._loss_g_mask, self._loss_g_mask_smooth = z(), z(), z()
        self._d_real, self._d_fake = z(), z()

    # ------------------------------------------------------------------ inputs
    def set_input(self, input):
        """trainer.py:324-362.  Three forms of `input`: the raw dataloader batch (imageA/B, manoA/B, and maskA/B on the HOv3 copy: needs opt.mano_model and
        opt.object_assets, see set_raw_input), images + rasteriser outputs (set_rasterised_input), or the prepared tensors
        (set_prepared_input: the synthetic-benchmark surface)."""
        if all(k in input for k in PREPARED_KEYS):
            return self.set_prepared_input(input)
        if all(k in input for k in RASTER_KEYS):
            return self.set_rasterised_input(input)
        if all(k in input for k in RAW_KEYS):
            return self.set_raw_input(input)
        raise KeyError('Trainer.set_input: expected the raw batch %s, images + rasteriser outputs %s, or the prepared tensors %s'
                       % (RAW_KEYS, RASTER_KEYS, PREPARED_KEYS))

    def set_raw_input(self, inp):
        """The batch train_ddp.py:92 passes, through HandRecoveryFlow on the device (hoig_amd/hand_recovery.py: MANO layer ->
        projection -> rasteriser -> tensor stage; trainer.py:324-362).  The MANO model and the per-object renderer buffers are
        the caller's (opt.mano_model, opt.object_assets): the reference builds them from assets it does not ship."""
        if self._hdr is None:
            from ..hand_recovery import HandRecoveryFlow
            try:
                self._hdr = HandRecoveryFlow(self._opt, device=self.device)
            except ValueError as ex:
                raise NotImplementedError('Trainer.set_input(raw batch): %s' % ex)
        with torch.no_grad():
            src_img, tsf_img = inp['imageA'].to(self.device, non_blocking=True), inp['imageB'].to(self.device, non_blocking=True)
            out = self._hdr(src_img, tsf_img, inp['manoA'], inp['manoB'])
            from .. import input_prep as IP
            return self.set_prepared_input(IP.to_prepared(out, src_img.float(), tsf_img.float(), inp.get('maskA'), inp.get('maskB')))

    def set_rasterised_input(self, inp):
        """Raw images + the rasteriser's outputs (``render_fim_wim``: face vertices, face index / weight maps) and the
        per-sample object tables: runs the tensor stage of HandRecoveryFlow.forward (trainer.py:46-145) on the device
        (hoig_amd.input_prep) and stages the result as Trainer.set_input does (trainer.py:346-362).  `tables`: one
        input_prep.ObjectTables (or dict of the object's MANORenderer buffers) per sample; `maskA` / `maskB`: the HOv3
        arm masks (trainer.py:329-337)."""
        from .. import input_prep as IP
        dev = self.device
        with torch.no_grad():
            t = {k: (v.to(dev, non_blocking=True) if torch.is_tensor(v) else v) for k, v in inp.items()}
            tabs = [tb if isinstance(tb, IP.ObjectTables) else IP.ObjectTables(tb, dev) for tb in t['tables']]
            out = IP.prepare_inputs(t['src_img'], t['ref_img'], t['src_faces'], t['src_fim'], t['src_wim'], t['ref_fim'],
                                    t['ref_wim'], tabs, bg_both=bool(getattr(self._opt, 'bg_both', False)), dexycb=self._dexycb,
                                    # training: no host wait inside the loop (reported one batch late, or by
                                    # _flush_input_checks); a single eval / test batch is checked before it runs
                                    validate='deferred' if self._is_train else True)
            return self.set_prepared_input(IP.to_prepared(out, t['src_img'].float(), t['ref_img'].float(),
                                                          t.get('maskA'), t.get('maskB')))

    @staticmethod
    def _flush_input_checks():
        """Face-index ranges of batches staged with validate='deferred' that nobody has looked at yet (the last batch of a run):
        raises the IndexError the reference's indexing would have raised.  Called where the host waits for the
        device anyway: get_current_errors, get_current_visuals, save."""
        from .. import input_prep as IP
        IP.flush_range_checks(wait=True)

    def stage_input(self, inp):
        """The device work of staging the a2 attributes (what trainer.py:346-362 assigns): NCHW tensors (T: B,S,S,2), CPU or device ->
        the dict of NHWC tensors the step reads, on the CURRENT stream, without touching the model's state (`set_prepared_input`
        adopts it at once).  Round 6 also ran it one batch ahead on the loader's stream: 67.38 -> 67.32 ms on the loader-fed step,
        i.e. nothing -- these launches cost GPU time beside a full chip, not latency -- so the loader does not
        (profiles/r06_loader_step.txt)."""
        dev = self.device
        with torch.no_grad():
            t = {k: v.to(dev, non_blocking=True).float() for k, v in inp.items() if torch.is_tensor(v)}
            n = {}
            for k in ['input_G_bg', 'input_G_src_obj', 'input_G_tsf_obj', 'input_G_src_hand', 'input_G_tsf_hand',
                      'real_src', 'real_tsf', 'bg_mask', 'hand_mask', 'armask_src', 'armask_tsf']:
                if k in t:
                    n[k] = to_nhwc(t[k])
            n['T'] = t['T'].contiguous()
            if self._dexycb:
                n.pop('armask_src', None)
                n.pop('armask_tsf', None)
            elif 'armask_src' not in n:
                raise KeyError('HOv3 inputs need armask_src / armask_tsf (trainer.py:329-337)')
            for side in ('src', 'tsf'):
                for part in ('obj', 'hand'):
                    full = n['input_G_%s_%s' % (side, part)]
                    n['%s_%s_rgb' % (side, part)] = ops.slice_channels(full, 0, 3)            # trainer.py:377-385
                    n['%s_%s_cond' % (side, part)] = ops.slice_channels(full, 3, full.shape[-1])
            cond = [n['tsf_obj_cond'], n['tsf_hand_cond']] + ([n['armask_tsf']] if 'armask_tsf' in n else [])
            n['tsf_cond'] = ops.cat_channels(cond)                                            # trainer.py:437,460
            # everything that depends on the batch only is made here, once per batch, not once per step: the stacked inputs of
            # the shared-weight sub-networks and the discriminator's REAL input (trainer.py:460-464)
            n['bg_in'], n['obj_in'], n['obj_c'] = self._net(self._G).stack_inputs(
                n['input_G_bg'], n['src_obj_rgb'], n['tsf_obj_rgb'], n['src_hand_cond'], n['tsf_hand_cond'], n['src_obj_cond'],
                n['tsf_obj_cond'], n.get('armask_src'), n.get('armask_tsf'))
            n['d_real_in'] = ops.cat_channels([n['real_tsf'], n['tsf_cond']])
        return n

    def set_prepared_input(self, inp):
        """Stage the a2 attributes (what trainer.py:346-362 assigns).  NCHW tensors (T: B,S,S,2), CPU or device."""
        return self._adopt_staged(self.stage_input(inp))

    def _adopt_staged(self, n):
        with torch.no_grad():
            n = self._n = self._stage_static(n)
            # reference-named NCHW views
            self._input_G_bg = as_nchw(n['input_G_bg'])
            self._input_G_src_obj, self._input_G_tsf_obj = as_nchw(n['input_G_src_obj']), as_nchw(n['input_G_tsf_obj'])
            self._input_G_src_hand = as_nchw(n['input_G_src_hand'])
            self._input_G_tsf_hand = as_nchw(n['input_G_tsf_hand'])
            self._T = n['T']
            self._real_src, self._real_tsf = as_nchw(n['real_src']), as_nchw(n['real_tsf'])
            self._bg_mask, self._hand_mask = as_nchw(n['bg_mask']), as_nchw(n['hand_mask'])
            if 'armask_src' in n:
                self._armask_src, self._armask_tsf = as_nchw(n['armask_src']), as_nchw(n['armask_tsf'])

    def _stage_static(self, n):
        """A captured step reads its inputs from fixed addresses: the first batch of a shape becomes that shape's staging set
        (cloned: some entries may alias the caller's tensors), later batches are copied into it (46 MB at 256x256, batch 8)."""
        sig = tuple(sorted((k, tuple(v.shape)) for k, v in n.items()))
        self._sig = sig
        if not (self._use_graph and self._is_train):
            return n
        st = self._static_inputs.get(sig)
        if st is None:
            if len(self._static_inputs) >= _GRAPH_MAX_SIGNATURES:
                return n
            st = self._static_inputs[sig] = {k: v.clone() for k, v in n.items()}
        else:
            for k, v in n.items():
                st[k].copy_(v)
        return st

    def set_train(self):
        self._G.train()
        self._D.train()
        self._is_train = True

    def set_eval(self):
        self._G.eval()
        self._is_train = False

    # ------------------------------------------------------------------ forward (trainer.py:373-415)
    def _wait_g(self):
        """G's previous optimiser step (exchange + Adam + operand planes on the side stream) before the current stream reads G."""
        self._net(self._G).wait_pending()

    def _wait_d(self):
        """D's weights are next needed by the discriminator pass of the G loss, a whole generator forward after the
        D step ended: its exchange (28 MB) + Adam have that long to finish on the side stream."""
        self._net(self._D).wait_pending()

    def forward(self, keep_data_for_visuals=False, return_estimates=False):
        if not self._is_train and not torch.is_grad_enabled():
            # generator-only inference (eval.py:59-65; BASELINE.json configs[4]): no backward follows, so the forward may run on the
            # arithmetic that is bounded by north_star's output tolerance alone (opt.eval_precision, default 'f16f6'; 'same': the
            # training forward's)
            with ops.inference_forward_precision(getattr(self._opt, 'eval_precision', os.environ.get('HOIG_EVAL_PRECISION', 'f16f6'))):
                return self._forward(keep_data_for_visuals)
        return self._forward(keep_data_for_visuals)

    def _forward(self, keep_data_for_visuals=False):
        self._wait_g()
        n = self._n
        outs = self._G.forward_nhwc(n['input_G_bg'], n['src_obj_rgb'], n['tsf_obj_rgb'], n['src_hand_rgb'],
                                    n['tsf_hand_rgb'], n['T'], n['src_obj_cond'], n['src_hand_cond'],
                                    n['tsf_obj_cond'], n['tsf_hand_cond'], n.get('armask_src'), n.get('armask_tsf'),
                                    stacked=(n['bg_in'], n['obj_in'], n['obj_c']))
        (src_bg, tsf_bg, src_obj, src_hand, src_mbg, src_mh, tsf_obj, tsf_hand, tsf_mbg, tsf_mh) = outs
        fake_src = ops.compose(src_bg, src_obj, src_hand, src_mbg, src_mh)
        fake_tsf = ops.compose(tsf_bg, tsf_obj, tsf_hand, tsf_mbg, tsf_mh)
        masks_bg = torch.cat([src_mbg, tsf_mbg], dim=0)
        masks_hand = torch.cat([src_mh, tsf_mh], dim=0)
        if keep_data_for_visuals:
            self.visual_imgs(outs, fake_src, fake_tsf, masks_bg, masks_hand)
        return (as_nchw(src_bg), as_nchw(tsf_bg), as_nchw(fake_src), as_nchw(fake_tsf), as_nchw(masks_bg),
                as_nchw(masks_hand))

    # ------------------------------------------------------------------ one GAN iteration (trainer.py:417-434)
    def optimize_parameters(self, trainable=True, keep_data_for_visuals=False):
        """forward -> G loss -> zero / backward / Adam(G) -> if `trainable`: D loss -> zero / backward / Adam(D).  With
        opt.hip_graph / HOIG_GRAPH=1: after _GRAPH_WARMUP eager iterations on a batch shape the whole iteration is captured in a
        hipGraph and replayed from then on (one host call per step instead of ~1 500 launches); iterations that keep data for
        visuals run eagerly."""
        if not self._is_train:
            return
        if self._use_graph and not keep_data_for_visuals:
            self._graph_step(bool(trainable))
        else:
            self._eager_step(trainable, keep_data_for_visuals)

    def _sync_active(self):
        return isinstance(self._G, FlatDDP) and self._G.sync.active

    def _eager_step(self, trainable=True, keep_data_for_visuals=False):
        ops.test_step_begins()
        # The D step reads the fake image and D's weights only.  Issued BEFORE G's backward (tuning key 'd_early') its four
        # milliseconds of kernels run beside the first, thin part of that backward (the loss chains' data gradients) instead of being
        # issued -- and largely executed -- after it: the host spends ~20 ms inside loss_G.backward().  Not under a gradient exchange:
        # there the D step is what G's all-reduce + Adam hide behind, and it stays after G's backward.
        early = bool(trainable and generator_forks_streams() and not self._sync_active() and _lib.set_tuning('d_early', -1))
        fake_tsf_imgs, ev_fwd = self._phase_g(keep_data_for_visuals, d_early=early)
        self._step(self._G, self._optimizer_G, overlap=trainable)
        if trainable:
            _mark('d_phase_begin')
            if early:
                torch.cuda.current_stream().wait_stream(self._d_stream)
            else:
                self._phase_d(fake_tsf_imgs, ev_fwd)
            _mark('d_phase_end')
            self._wait_g()            # G's update has had the whole D step to finish; later readers need no special care
            self._step(self._D, self._optimizer_D, overlap=True)

    def _phase_g(self, keep_data_for_visuals=False, d_early=False):
        """Forward, the G loss and its backward (trainer.py:419-427).  Returns the fake target image and the event that marks the
        end of the forward (what the D step waits for)."""
        _, _, fake_src_imgs, fake_tsf_imgs, fake_masks_bg, fake_masks_hand = \
            self.forward(keep_data_for_visuals=keep_data_for_visuals)
        ev_fwd = torch.cuda.Event()
        ev_fwd.record(torch.cuda.current_stream())
        netD = self._net(self._D)
        netD.set_requires_grad(False)       # the reference computes D grads here and zeroes them at :432
        loss_G = self._optimize_G(fake_src_imgs, fake_tsf_imgs, fake_masks_bg, fake_masks_hand)
        self._optimizer_G.zero_grad()
        if d_early:
            # (G's graph above was recorded with D frozen: its backward still skips D's weight gradients, and D's own gradients
            # below land in D's buffer, which that backward never touches; D's Adam stays behind G's backward and step)
            netD.set_requires_grad(True)
            self._phase_d(fake_tsf_imgs, ev_fwd, join=False)
        ops.pause_wgrad_side(generator_forks_streams())     # G's backward is several concurrent chains already
        bucketed = self._sync_active()
        if bucketed:               # (ddp_mode 'bucket': counts / counts down the gradient writes per slice; a no-op in mode 'after')
            self._G.sync.begin_backward((self._sig, ops.precision, ops.precision_dgrad, ops.precision_wgrad))
        try:
            loss_G.backward()
        finally:
            if bucketed:
                self._G.sync.end_backward()
        ops.check_split_grads_consumed()
        ops.pause_wgrad_side(False)
        self._join_backward_streams()
        netD.set_requires_grad(True)
        return fake_tsf_imgs, ev_fwd

    def _join_backward_streams(self):
        """Autograd replays every chain's backward on the stream of its forward and, when it returns, has synchronised the
        caller's stream only with the streams of gradient-accumulation LEAVES.  This path has none (weight gradients are
        accumulated by the kernels themselves, the inputs need no gradient), so the tails of the branch chains -- the first
        layers' weight gradients on the bg / obj / src streams, VGG's and D's data gradients on the loss streams -- would be
        ordered before nothing: join them, so that zero_grad / the optimiser step (and the end of a captured graph) follow
        them."""
        main = torch.cuda.current_stream()
        streams = list(getattr(self._net(self._G), '_streams', None) or ()) + list(self._loss_streams or ())
        for st in streams:
            main.wait_stream(st)

    def _phase_d(self, fake_tsf_imgs, ev_fwd=None, join=True):
        """The D loss and its backward (trainer.py:429-433).  It reads the fake image and D's weights, nothing of G's backward:
        given the forward's event it runs on a stream of its own that only waits for the generator's forward, i.e. BESIDE G's
        backward chains.  D's Adam still follows G's backward through D: it is queued on the side stream behind G's step, which
        waited for every backward stream."""
        if ev_fwd is not None and generator_forks_streams():
            main = torch.cuda.current_stream()
            if self._d_stream is None:
                self._d_stream = ops.new_stream(self.device, 'd')
            self._d_stream.wait_event(ev_fwd)
            ops.cross_stream(fake_tsf_imgs, self._d_stream)
            with torch.cuda.stream(self._d_stream):
                ops.test_delay('d')
                self._d_backward(fake_tsf_imgs)
            if join:
                main.wait_stream(self._d_stream)
        else:
            self._d_backward(fake_tsf_imgs)

    def _d_backward(self, fake_tsf_imgs):
        loss_D = self._optimize_D(fake_tsf_imgs)          # (waits for D's previous update on THIS stream: _wait_d)
        self._optimizer_D.zero_grad()
        loss_D.backward()
        ops.check_split_grads_consumed()

    def _step(self, net, optimizer, overlap):
        """gradient exchange (RCCL, under DDP) + fused Adam + the operand planes of the new weights.  With `overlap` they run on
        the side stream: G's beside the D step that follows on the main stream (which never touches G's parameters: 734 MB of
        exchange, 5 GB of Adam traffic), D's beside the next generator forward; every later reader of the network waits for the
        event (ParamTree.wait_pending, _wait_g / _wait_d)."""
        ddp = isinstance(net, FlatDDP) and net.sync.active
        tree = self._net(net)

        def run():
            if ddp:           # Adam of slice i runs while slices i+1.. are still being exchanged
                optimizer.step(grad_scale=1.0 / net.sync.world, ready=net.sync.iter_all_reduce())
            else:
                optimizer.step()
            tree.refresh_planes()      # off the next forward's critical path

        if not overlap:
            tree.wait_pending()
            return run()
        main = torch.cuda.current_stream()
        self._side.wait_stream(main)
        which = 'g' if net is self._G else 'd'
        with torch.cuda.stream(self._side):
            if _TEST_SIDE_DELAY and not ops.capturing():
                torch.cuda._sleep(int(_TEST_SIDE_DELAY))
            ops.test_delay('opt', which)
            tree.set_pending(None)     # (this stream IS the writer: the previous step ran here too)
            _mark('step_%s_begin' % which)
            run()
            _mark('step_%s_end' % which)
            ev = torch.cuda.Event()
            ev.record(self._side)
        tree.set_pending(ev)

    # ------------------------------------------------------------------ the captured step
    def _graph_key(self, trainable):
        return (trainable, self._sig, ops.precision, ops.precision_dgrad, ops.precision_wgrad, self._sync_active())

    def _graph_step(self, trainable):
        key = self._graph_key(trainable)
        st = self._graphs.get(key)
        if st is None:
            if len(self._graphs) >= _GRAPH_MAX_SIGNATURES or self._static_inputs.get(self._sig) is not self._n:
                return self._eager_step(trainable)
            st = self._graphs[key] = dict(seen=0, graphs=None)
        if st['graphs'] is None:
            if st['seen'] < _GRAPH_WARMUP:
                st['seen'] += 1
                return self._eager_step(trainable)
            self._capture_step(st, trainable)
        self._replay_step(st, trainable)

    def _quiesce(self):
        """Nothing of a previous step may still be in flight when a capture starts (events recorded outside a capture cannot be
        waited for inside it)."""
        self._wait_g()
        self._wait_d()
        torch.cuda.synchronize(self.device)
        self._net(self._G).set_pending(None)
        self._net(self._D).set_pending(None)

    def _capture(self, body):
        graph = torch.cuda.CUDAGraph()
        with ops.graph_capture(graph, pool=self._graph_pool):
            out = body()
        if self._graph_pool is None:
            self._graph_pool = graph.pool()
        return graph, out

    def _capture_step(self, st, trainable):
        """Single GPU: the whole iteration is ONE graph (its streams become the graph's parallel branches; the optimiser steps run
        inside it from device-resident schedules).  Under DDP the RCCL exchange stays outside: graph 1 = forward + G loss + G
        backward, then G's exchange + Adam eagerly on the side stream, beside graph 2 = the D loss and backward, then D's."""
        self._quiesce()
        opts = (self._optimizer_G, self._optimizer_D)
        for o in opts:
            o.sync_state()
        before = [o.step_count for o in opts]
        if not self._sync_active():
            def body():
                self._eager_step(trainable)
                if trainable:         # (the optimiser side stream forked into the capture: join it back)
                    torch.cuda.current_stream().wait_stream(self._side)
            graph, _ = self._capture(body)
            st['graphs'] = (graph,)
            # the capture ran the host side of one step without executing it: the replay that follows does
            for o, b in zip(opts, before):
                o.step_count = b
                o._on_device = o._on_device[:4] + (float(b),)
        else:
            g1, fake = self._capture(lambda: self._phase_g()[0])
            g2 = self._capture(lambda: self._phase_d(fake))[0] if trainable else None
            st['graphs'] = (g1, g2)
            st['fake'] = fake
        self._net(self._G).set_pending(None)
        self._net(self._D).set_pending(None)

    def _replay_step(self, st, trainable):
        for tree in (self._net(self._G), self._net(self._D), self._crt_tsf.vgg if self._opt.use_vgg else None):
            if tree is not None:
                tree.refresh_planes()        # (a no-op unless weights were loaded since the last step)
        if len(st['graphs']) == 1:
            self._optimizer_G.sync_state()
            if trainable:
                self._optimizer_D.sync_state()
            st['graphs'][0].replay()
            self._optimizer_G.replayed()
            if trainable:
                self._optimizer_D.replayed()
            return
        g1, g2 = st['graphs']
        self._wait_g()
        self._wait_d()
        g1.replay()
        self._step(self._G, self._optimizer_G, overlap=trainable)
        if trainable:
            _mark('d_phase_begin')
            g2.replay()
            _mark('d_phase_end')
            self._wait_g()
            self._step(self._D, self._optimizer_D, overlap=True)

    def _optimize_G(self, fake_src_imgs, fake_tsf_imgs, fake_masks_bg, fake_masks_hand):
        """trainer.py:436-457."""
        o, n = self._opt, self._n
        fake_src, fake_tsf = to_nhwc(fake_src_imgs), to_nhwc(fake_tsf_imgs)
        mbg, mh = to_nhwc(fake_masks_bg), to_nhwc(fake_masks_hand)
        # The adversarial term (D on the fake) and the perceptual term (VGG on the fake, VGG on the target) do not read each other:
        # three chains on three streams, like the generator's sub-networks (their backward replays there too); HOIG_STREAMS=0: one after
        # the other on the caller's stream.
        fork = fake_tsf.is_cuda and generator_forks_streams()
        T = self._g_terms
        into = T.term
        T.begin()                          # (on the caller's stream, before the loss streams fork from it)
        if fork:
            main = torch.cuda.current_stream()
            # operand planes are (re)made lazily by whoever asks first: make D's and VGG's here, on the caller's stream, so that
            # the chains below only read them (VGG's weights never change: this matters on the first step and after a load)
            self._net(self._D).refresh_planes()
            self._crt_tsf.vgg.refresh_planes()
            if self._loss_streams is None:
                self._loss_streams = (ops.new_stream(self.device, 'loss_adv'), ops.new_stream(self.device, 'loss_vgg'))
            s_adv, s_vgg = self._loss_streams
            s_adv.wait_stream(main)
            with torch.cuda.stream(s_adv):
                ops.test_delay('loss_adv')
                self._wait_d()
                d_fake = self._D.forward_nhwc(ops.cat_channels([fake_tsf, n['tsf_cond']]))
                d_fake = ops.delay_backward(d_fake, 'loss_adv')
                self._loss_g_adv = ops.lsgan_loss(d_fake, 0.0, o.lambda_D_prob, into=into('g_adv'))
            ops.cross_stream(fake_tsf, s_adv)
        else:
            s_vgg = None
            self._wait_d()
            d_fake = self._D.forward_nhwc(ops.cat_channels([fake_tsf, n['tsf_cond']]))
            self._loss_g_adv = ops.lsgan_loss(d_fake, 0.0, o.lambda_D_prob, into=into('g_adv'))
        self._loss_g_rec = ops.l1_loss(fake_src, n['real_src'], o.lambda_rec, into=into('g_rec'))
        # the reference uses self._crt_tsf in both branches of `if use_vgg` (:443-446); it only exists with --use_vgg
        tsf = self._crt_tsf.forward_nhwc(fake_tsf, n['real_tsf'], o.lambda_tsf, side=s_vgg, into=into('g_tsf'))
        if fork:
            main.wait_stream(s_adv)
        crt = ops.bce_loss if o.mask_bce else ops.mse_loss
        masks = [crt(mbg, n['bg_mask'], o.lambda_mask, into=into('g_mask')), crt(mh, n['hand_mask'], o.lambda_mask, into=into('g_mask'))]
        smooth = []
        if o.lambda_mask_smooth != 0:
            smooth = [ops.tv_loss(mbg, o.lambda_mask_smooth, into=into('g_mask_smooth')),
                      ops.tv_loss(mh, o.lambda_mask_smooth, into=into('g_mask_smooth'))]
        # every term already sits, scaled, in its slot: one launch sums them; the reported values are views of the slots
        self._loss_g_tsf, self._loss_g_mask = T.value('g_tsf'), T.value('g_mask')
        self._loss_g_mask_smooth = T.value('g_mask_smooth')
        return T.total(self._loss_g_adv, self._loss_g_rec, *(tsf + masks + smooth))

    def _optimize_D(self, fake_tsf_imgs):
        """trainer.py:459-474."""
        o, n = self._opt, self._n
        fake_tsf = to_nhwc(fake_tsf_imgs).detach()
        # the reference runs D twice (trainer.py:464-465); instance norm is per sample, so one stacked pass is identical
        nb = fake_tsf.shape[0]
        self._wait_d()
        d_both = self._D.forward_nhwc(torch.cat([n['d_real_in'], ops.cat_channels([fake_tsf, n['tsf_cond']])], dim=0))
        d_real, d_fake = d_both[:nb], d_both[nb:]
        T = self._d_terms
        T.begin()
        loss_real = ops.lsgan_loss(d_real, 1.0, o.lambda_D_prob, into=T.term('d'))
        loss_fake = ops.lsgan_loss(d_fake, -1.0, o.lambda_D_prob, into=T.term('d'))
        with torch.no_grad():
            self._d_real = ops.mean(d_real, into=T.term('d_real'))
            self._d_fake = ops.mean(d_fake, into=T.term('d_fake'))
        return T.total(loss_real, loss_fake)

    backward_G = _optimize_G        # north_star vocabulary
    backward_D = _optimize_D

    def _compute_loss_D(self, x, y):
        return ops.lsgan_loss(to_nhwc(x), float(y), 1.0)

    def _compute_loss_smooth(self, mat):
        return ops.tv_loss(to_nhwc(mat), 1.0)

    # ------------------------------------------------------------------ reporting (trainer.py:483-551)
    def get_current_errors(self):
        self._flush_input_checks()
        return OrderedDict([('g_rec', self._loss_g_rec.item()), ('g_tsf', self._loss_g_tsf.item()),
                            ('g_adv', self._loss_g_adv.item()), ('g_mask', self._loss_g_mask.item()),
                            ('g_mask_smooth', self._loss_g_mask_smooth.item()), ('d_real', self._d_real.item()),
                            ('d_fake', self._d_fake.item())])

    def get_current_scalars(self):
        return OrderedDict([('lr_G', self._current_lr_G), ('lr_D', self._current_lr_D)])

    def get_current_visuals(self):
        self._flush_input_checks()
        keys = [('1_real_img', '_vis_input'), ('2_input_src_obj', '_vis_src_obj'), ('2_input_src_hand', '_vis_src_hand'),
                ('2_input_tsf_obj', '_vis_tsf_obj'), ('2_input_tsf_hand', '_vis_tsf_hand'),
                ('3_fake_src_bg', '_vis_fake_src_bg'), ('4_fake_tsf_bg', '_vis_fake_tsf_bg'),
                ('5_fake_src_color', '_vis_fake_src_color'), ('6_fake_tsf_color', '_vis_fake_tsf_color'),
                ('7_src_seg', '_vis_src_seg'), ('8_ref_seg', '_vis_ref_seg'), ('10_fake_tsf', '_vis_fake_tsf'),
                ('11_fake_src', '_vis_fake_src'), ('12_fake_mask_bg', '_vis_mask_bg'),
                ('13_fake_mask_hand', '_vis_mask_hand'), ('14_batch_real_img', '_vis_batch_real'),
                ('15_batch_fake_img', '_vis_batch_fake'), ('16_batch_src_img', '_vis_batch_src')]
        return OrderedDict((k, getattr(self, a)) for k, a in keys)

    @staticmethod
    def _im(x_nhwc, idx=0, unnormalize=True):
        """utils/util.py:249-264 tensor2im: CHW uint8 of sample `idx`, or of the padding-0 grid when idx < 0."""
        if idx >= 0:
            return ops.tensor2im_u8(x_nhwc[idx:idx + 1].contiguous(), 1, unnormalize).cpu().numpy()
        nrow = int(math.sqrt(x_nhwc.shape[0]))
        return ops.tensor2im_u8(x_nhwc, nrow, unnormalize).cpu().numpy()

    def _seg_vis(self, seg_nhwc):
        """trainer.py:542-545: label = (any channel != 0) * (argmax + 1), colourised, then tensor2im of sample 0."""
        s = seg_nhwc[0]
        lab = ((s.sum(-1) != 0).long() * (s.argmax(-1) + 1)).cpu().numpy()
        col = np.zeros(lab.shape + (3,), np.uint8)
        for label in range(len(self._cmap)):
            col[lab == label] = self._cmap[label]
        img = col.transpose(2, 0, 1).astype(np.float32)
        img += 1.0
        img /= 2.0
        img *= 255.0
        return img.astype(np.uint8)

    @torch.no_grad()
    def visual_imgs(self, outs, fake_src, fake_tsf, masks_bg, masks_hand):
        n = self._n
        (src_bg, tsf_bg, src_obj, src_hand, src_mbg, src_mh, tsf_obj, tsf_hand, tsf_mbg, tsf_mh) = outs
        ids = masks_bg.shape[0] // 2
        one = torch.ones_like(src_mbg)
        # fg = -m_bg + (1-m_bg)*(...) = compose with a background of -1   (trainer.py:405-406)
        src_fg = ops.compose(-one.expand_as(src_bg).contiguous(), src_obj, src_hand, src_mbg, src_mh)
        tsf_fg = ops.compose(-one.expand_as(tsf_bg).contiguous(), tsf_obj, tsf_hand, tsf_mbg, tsf_mh)
        self._vis_input = self._im(n['real_src'])
        self._vis_src_obj = self._im(n['src_obj_rgb'])
        self._vis_src_hand = self._im(n['src_hand_rgb'])
        self._vis_tsf_obj = self._im(n['tsf_obj_rgb'])
        self._vis_tsf_hand = self._im(n['tsf_hand_rgb'])
        self._vis_fake_src_bg = self._im(src_bg)
        self._vis_fake_tsf_bg = self._im(tsf_bg)
        self._vis_fake_src_color = self._im(src_fg)
        self._vis_fake_tsf_color = self._im(tsf_fg)
        self._vis_fake_src = self._im(fake_src)
        self._vis_fake_tsf = self._im(fake_tsf)
        self._vis_mask_bg = self._im(masks_bg, idx=ids, unnormalize=False)
        self._vis_mask_hand = self._im(masks_hand, idx=ids, unnormalize=False)
        src_seg = torch.cat([n['input_G_src_hand'][..., 6:], n['input_G_src_obj'][..., 3:]], dim=-1)
        tsf_seg = torch.cat([n['input_G_tsf_hand'][..., 6:], n['input_G_tsf_obj'][..., 3:]], dim=-1)
        self._vis_src_seg = self._seg_vis(src_seg)
        self._vis_ref_seg = self._seg_vis(tsf_seg)
        self._vis_batch_src = self._im(n['real_src'], idx=-1)
        self._vis_batch_real = self._im(n['real_tsf'], idx=-1)
        self._vis_batch_fake = self._im(fake_tsf, idx=-1)

    # ------------------------------------------------------------------ checkpoints / schedule (trainer.py:553-591)
    def save(self, label):
        self._flush_input_checks()
        self._wait_g()
        self._wait_d()
        torch.cuda.synchronize()
        self._save_network(self._G, 'G', label)
        self._save_network(self._D, 'D', label)
        self._save_optimizer(self._optimizer_G, 'G', label)
        self._save_optimizer(self._optimizer_D, 'D', label)

    def load(self):
        """HOIG_HOv3/models/trainer.py:562-575; the HOIG_DexYCB copy (its trainer.py:558-573) differs twice: it hands the file's
        keys to the (DDP-wrapped) networks as saved (`need_module=True`), and after restoring the optimisers it REPLAYS the
        linear learning-rate decay from the initial rate for the epochs past `nepochs_no_decay` (the rate in the optimiser
        file is overridden)."""
        load_epoch = self._opt.load_epoch
        need_module = self._dexycb
        self._load_network(self._G, 'G', load_epoch, need_module=need_module)
        if self._is_train:
            self._load_network(self._D, 'D', load_epoch, need_module=need_module)
            self._load_optimizer(self._optimizer_G, 'G', load_epoch)
            self._load_optimizer(self._optimizer_D, 'D', load_epoch)
            no_decay = getattr(self._opt, 'nepochs_no_decay', None)
            if self._dexycb and no_decay is not None and load_epoch > no_decay:
                for _ in range(no_decay, load_epoch):
                    self.update_learning_rate()

    def update_learning_rate(self):
        """One linear decay step towards final_lr for both optimisers (trainer.py:577-591).  The new rates reach the device-
        resident schedules before the next step (FusedAdam.sync_state); rank 0 reports."""
        quiet = dist.is_available() and dist.is_initialized() and dist.get_rank() != 0
        for tag, optimizer, lr0 in (('G', self._optimizer_G, self._opt.lr_G), ('D', self._optimizer_D, self._opt.lr_D)):
            old = getattr(self, '_current_lr_' + tag)
            new = old - (lr0 - self._opt.final_lr) / self._opt.nepochs_decay
            setattr(self, '_current_lr_' + tag, new)
            for group in optimizer.param_groups:
                group['lr'] = new
            if not quiet:
                print('update %s learning rate: %f -> %f' % (tag, old, new))


HOGANModel = Trainer
