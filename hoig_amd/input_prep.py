"""Input preparation after the rasteriser, on the device (tensor stage).

Stands where the body of ``HandRecoveryFlow.forward`` (HOIG_HOv3/models/trainer.py:46-145) stands once
``MANORenderer.render_fim_wim`` (neural_renderer's rasteriser, utils/nmr.py:496-513; needs the MANO model and the YCB
meshes, not part of this build) has produced, per sample and view, the projected face vertices, the face index map and
the barycentric weight map.  Everything downstream -- encode_fim / encode_sem (nmr.py:567-595), cal_bc_transform
(nmr.py:874-968), get_texture_backward_warp (nmr.py:973-1058), sample_from_texture_dense (nmr.py:1068-1100), the two
``F.grid_sample`` calls and the 3x3 / 15x15 ``util.morph`` erosions with the channel bookkeeping of trainer.py:103-145 -- runs
as five HIP launches per batch (round 6; three kernels per sample plus one per batch before: hoig_amd/csrc/input_prep.hip); the reference runs ~60 small torch ops per
sample in a Python loop.  HOv3 and DexYCB channel layouts; 256 x 256 only, as the reference hard-wires it.

``prepare_inputs`` returns the reference's 12-tuple (NCHW).  ``to_prepared`` turns it into the dict ``Trainer.set_input``
stages (trainer.py:346-362).  No CPU path: tensors must live on the HIP device.
"""
import torch

from . import _lib as L

S, TEX_W = 256, 640
MAX_BATCH = 32                      # HOIG_PREP_MAX_BATCH (include/hoig_kernels.h): the batched entry points' limit
_TABLE_KEYS = ('map_fn', 'sem_full', 'fim_uv', 'wim_uv', 'faces_uv_coord', 'obj_tex_img')


class ObjectTables(object):
    """One object's MANORenderer buffers (nmr.py:295-406) on the device: map_fn (F+1,3), sem_full (F+1,1), fim_uv
    (1,256,640) int, wim_uv (1,256,640,3), faces_uv_coord (1,F,3,2), obj_tex_img (256,256,3)."""

    def __init__(self, tables, device):
        self.n_faces = int(tables['map_fn'].shape[0]) - 1
        f32 = lambda k: tables[k].to(device=device, dtype=torch.float32).contiguous()
        self.map_fn, self.sem_full, self.wim_uv = f32('map_fn'), f32('sem_full'), f32('wim_uv')
        self.faces_uv_coord, self.obj_tex_img = f32('faces_uv_coord'), f32('obj_tex_img')
        self.fim_uv = tables['fim_uv'].to(device=device, dtype=torch.int32).contiguous()
        if tuple(self.map_fn.shape) != (self.n_faces + 1, 3) or self.sem_full.numel() != self.n_faces + 1:
            raise ValueError('map_fn must be (F+1,3) and sem_full (F+1,1)')
        if self.fim_uv.numel() != S * TEX_W or self.wim_uv.numel() != S * TEX_W * 3:
            raise ValueError('fim_uv / wim_uv must cover the 256 x 640 atlas')
        if self.faces_uv_coord.numel() != self.n_faces * 6 or self.obj_tex_img.numel() != S * S * 3:
            raise ValueError('faces_uv_coord must be (1,F,3,2) and obj_tex_img (256,256,3)')
        lo, hi = int(self.fim_uv.min()), int(self.fim_uv.max())          # once per object, at construction
        if lo < -1 or hi >= self.n_faces:
            raise IndexError('fim_uv holds face indices in [%d, %d]; the object has %d faces' % (lo, hi, self.n_faces))


def _dev(t, dtype, name):
    if not t.is_cuda:
        raise NotImplementedError('%s: hoig_amd.input_prep runs on the HIP device only (no CPU path)' % name)
    return t.to(dtype).contiguous()


def _check_ranges(rng, n_faces):
    for i, nf in enumerate(n_faces):
        for name, lo, hi in (('src_fim', rng[i][0], rng[i][1]), ('ref_fim', rng[i][2], rng[i][3])):
            if lo < -1 or hi >= nf:
                raise IndexError('%s[%d] holds face indices in [%d, %d]; the sample has %d faces' % (name, i, lo, hi, nf))


_range_checks = []


def _deferred_range_check(rng, n_faces):
    """validate='deferred' (the training loop, Trainer.set_rasterised_input): the batch's index ranges go to pinned host memory
    with an asynchronous copy and are examined when the NEXT batch is prepared (or by flush_range_checks()), by which time the
    copy has long finished -- the host never waits for the device inside the step.  An out-of-range index is still reported,
    one batch late, as the IndexError the reference's indexing would raise."""
    flush_range_checks(wait=False)
    host = torch.empty(tuple(rng.shape), dtype=rng.dtype, pin_memory=True)
    host.copy_(rng, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    _range_checks.append((ev, host, n_faces))


def flush_range_checks(wait=True):
    """Examine the index ranges of batches prepared with validate='deferred' (all of them if `wait`, else those whose copy
    has completed)."""
    while _range_checks and (wait or _range_checks[0][0].query()):
        ev, host, n_faces = _range_checks.pop(0)
        ev.synchronize()
        _check_ranges(host.tolist(), n_faces)


def prepare_inputs(src_img, ref_img, src_faces, src_fim, src_wim, ref_fim, ref_wim, tables, bg_both=False, dexycb=False,
                   validate=True):
    """trainer.py:46-145 after the rasteriser.  src_img / ref_img (B,3,256,256); src_faces (B,F,3,3) as returned by
    render_fim_wim for the SOURCE view (rows beyond a sample's own face count are ignored); *_fim (B,256,256) integer,
    *_wim (B,256,256,3); tables: one ObjectTables per sample; dexycb: the DexYCB copy's hand inputs (12 channels: + the six
    hand-part one-hots, HOIG_DexYCB/models/trainer.py:131,135); validate (default on): check that every face index addresses
    its sample's tables, as the reference's indexing would raise IndexError (the kernels index unchecked: an out-of-range face
    would be a device memory fault) -- one small reduction and one host read per batch; 'deferred': the same check without the
    host waiting for the device (reported when the next batch is prepared; _deferred_range_check); False only for inputs
    that come straight from hoig_amd.raster, whose indices are in range by construction.
    Returns (input_G_src_bg, input_G_tsf_bg | None, input_G_src_obj, input_G_tsf_obj, input_G_src_hand, input_G_ref_hand,
    T_hand, src_crop_mask_bg, ref_crop_mask_bg, src_crop_mask_hand, ref_crop_mask_hand, None)."""
    B = int(src_img.shape[0])
    if tuple(src_img.shape) != (B, 3, S, S) or tuple(ref_img.shape) != (B, 3, S, S):
        raise ValueError('images must be (B,3,256,256): the reference hard-wires 256 (nmr.py:1012,1040,1070)')
    if len(tables) != B:
        raise ValueError('one ObjectTables per sample')
    src_img, ref_img = _dev(src_img, torch.float32, 'src_img'), _dev(ref_img, torch.float32, 'ref_img')
    src_faces = _dev(src_faces, torch.float32, 'src_faces')
    src_fim, ref_fim = _dev(src_fim, torch.int32, 'src_fim'), _dev(ref_fim, torch.int32, 'ref_fim')
    src_wim, ref_wim = _dev(src_wim, torch.float32, 'src_wim'), _dev(ref_wim, torch.float32, 'ref_wim')
    if validate:
        # one reduction for the whole batch: [B,4] = (min, max) of the two index maps of every sample
        rng = torch.stack([src_fim.amin(dim=(1, 2)), src_fim.amax(dim=(1, 2)), ref_fim.amin(dim=(1, 2)),
                           ref_fim.amax(dim=(1, 2))], dim=1)
        n_faces = [tb.n_faces for tb in tables]
        if validate == 'deferred':
            _deferred_range_check(rng, n_faces)
            # the kernels index unchecked: until the report arrives, keep a bad index from becoming a memory fault
            top = torch.tensor(n_faces, dtype=torch.int32).to(src_fim.device, non_blocking=True).view(B, 1, 1) - 1
            src_fim = torch.minimum(src_fim, top).clamp_min_(-1)
            ref_fim = torch.minimum(ref_fim, top).clamp_min_(-1)
        else:
            _check_ranges(rng.cpu().tolist(), n_faces)
    dev = src_img.device
    new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    cond_s, cond_r, rend_s, rend_r = new(B, 3, S, S), new(B, 3, S, S), new(B, 3, S, S), new(B, 3, S, S)
    seg_s, seg_r, hr_s, hr_r = new(B, S, S), new(B, S, S), new(B, S, S), new(B, S, S)
    T_raw = new(B, S, S, 2)
    for i, tb in enumerate(tables):
        if src_faces.shape[1] < tb.n_faces:
            raise ValueError('src_faces has fewer rows than sample %d has faces' % i)
    if B <= MAX_BATCH:
        # one call per stage for the whole batch (round 6): the per-sample tables travel as host arrays of device pointers
        import ctypes
        vp = lambda ts: ctypes.cast((ctypes.c_void_p * B)(*[t.data_ptr() for t in ts]), ctypes.c_void_p)
        nf = ctypes.cast((ctypes.c_int * B)(*[tb.n_faces for tb in tables]), ctypes.c_void_p)
        occ = torch.empty(B, S * TEX_W, dtype=torch.uint8, device=dev)
        tex = new(B, 3, S, TEX_W)
        fstride = int(src_faces.stride(0))
        L.call('hoig_prep_texture_batched', B, src_img.data_ptr(), src_faces.data_ptr(), fstride, src_fim.data_ptr(),
               vp([tb.fim_uv for tb in tables]), vp([tb.wim_uv for tb in tables]), vp([tb.obj_tex_img for tb in tables]),
               occ.data_ptr(), tex.data_ptr(), st)
        maps = (vp([tb.map_fn for tb in tables]), vp([tb.sem_full for tb in tables]), vp([tb.faces_uv_coord for tb in tables]))
        for fim, wim, cond, seg, hr, rend, T in ((src_fim, src_wim, cond_s, seg_s, hr_s, rend_s, None),
                                                 (ref_fim, ref_wim, cond_r, seg_r, hr_r, rend_r, T_raw)):
            L.call('hoig_prep_lookup_batched', B, fim.data_ptr(), wim.data_ptr(), maps[0], maps[1], maps[2], nf, tex.data_ptr(),
                   src_faces.data_ptr(), fstride, cond.data_ptr(), seg.data_ptr(), hr.data_ptr(), rend.data_ptr(),
                   None if T is None else T.data_ptr(), st)
    else:
        occ = torch.empty(S * TEX_W, dtype=torch.uint8, device=dev)
        tex = new(3, S, TEX_W)
        for i in range(B):
            tb = tables[i]
            L.call('hoig_prep_texture', src_img[i].data_ptr(), src_faces[i].data_ptr(), src_fim[i].data_ptr(), tb.fim_uv.data_ptr(),
                   tb.wim_uv.data_ptr(), tb.obj_tex_img.data_ptr(), occ.data_ptr(), tex.data_ptr(), st)
            for fim, wim, cond, seg, hr, rend, T in ((src_fim, src_wim, cond_s, seg_s, hr_s, rend_s, None),
                                                     (ref_fim, ref_wim, cond_r, seg_r, hr_r, rend_r, T_raw)):
                L.call('hoig_prep_lookup', fim[i].data_ptr(), wim[i].data_ptr(), tb.map_fn.data_ptr(), tb.sem_full.data_ptr(),
                       tb.faces_uv_coord.data_ptr(), tb.n_faces, tex.data_ptr(), src_faces[i].data_ptr(), cond[i].data_ptr(),
                       seg[i].data_ptr(), hr[i].data_ptr(), rend[i].data_ptr(), None if T is None else T[i].data_ptr(), st)
    src_bg = new(B, 4, S, S)
    tsf_bg = new(B, 4, S, S) if bg_both else None
    hc = 12 if dexycb else 6
    src_obj, tsf_obj, src_hand, ref_hand = new(B, 15, S, S), new(B, 15, S, S), new(B, hc, S, S), new(B, hc, S, S)
    T_hand = new(B, S, S, 2)
    smb, rmb, smh, rmh = new(B, 1, S, S), new(B, 1, S, S), new(B, 1, S, S), new(B, 1, S, S)
    L.call('hoig_prep_assemble', B, src_img.data_ptr(), ref_img.data_ptr(), cond_s.data_ptr(), cond_r.data_ptr(),
           seg_s.data_ptr(), seg_r.data_ptr(), hr_s.data_ptr(), hr_r.data_ptr(), rend_s.data_ptr(), rend_r.data_ptr(),
           T_raw.data_ptr(), src_bg.data_ptr(), None if tsf_bg is None else tsf_bg.data_ptr(), src_obj.data_ptr(),
           tsf_obj.data_ptr(), src_hand.data_ptr(), ref_hand.data_ptr(), hc, T_hand.data_ptr(), smb.data_ptr(), rmb.data_ptr(),
           smh.data_ptr(), rmh.data_ptr(), st)
    return src_bg, tsf_bg, src_obj, tsf_obj, src_hand, ref_hand, T_hand, smb, rmb, smh, rmh, None


def to_prepared(out, src_img, ref_img, armask_src=None, armask_tsf=None):
    """What Trainer.set_input assigns from that tuple (trainer.py:346-362), keyed by models.trainer.PREPARED_KEYS."""
    d = dict(input_G_bg=out[0] if out[1] is None else torch.cat([out[0], out[1]], dim=0),
             input_G_src_obj=out[2], input_G_tsf_obj=out[3], input_G_src_hand=out[4], input_G_tsf_hand=out[5], T=out[6],
             bg_mask=torch.cat((out[7], out[8]), dim=0), hand_mask=torch.cat((out[9], out[10]), dim=0),
             real_src=src_img, real_tsf=ref_img)
    if armask_src is not None:
        d['armask_src'], d['armask_tsf'] = armask_src, armask_tsf
    return d
