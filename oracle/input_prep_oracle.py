"""CPU restatement (TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this) of the input-preparation stage that FOLLOWS the rasteriser in the reference: ``HandRecoveryFlow.forward``
(HOIG_HOv3/models/trainer.py:46-145) with the MANORenderer helpers it calls (utils/nmr.py) and ``util.morph``
(utils/util.py:142-158).  Plain torch on CPU; HOv3 channel layout, or the DexYCB copy's with dexycb=True.

Pinned: tests/test_input_prep_oracle.py compares it bit-for-bit with the reference's own code run in the build container
(oracle/ref_harness.py::reference_input_prep -- only ``render_fim_wim`` is replaced, by seeded synthetic rasteriser
outputs) and, everywhere, with the committed fixture tests/golden/input_prep_256.npz made by that run
(tests/golden/make_golden_input_prep.py).

Inputs (one batch; the reference loops over samples because every sample has its own object mesh, trainer.py:63-64):
  src_img, ref_img (B,3,256,256) in [-1,1]; src_faces (B,F,3,3): projected face vertices returned by render_fim_wim (only
  x, y are used, y gets negated: trainer.py:67-68); src_fim / ref_fim (B,256,256) int32 face index maps (-1 = no face; ids
  < 1538 are hand faces, trainer.py:73); src_wim / ref_wim (B,256,256,3) barycentric weights; per sample the object's tables
  (nmr.py:295-406): map_fn (F+1,3), sem_full (F+1,1), fim_uv (1,256,640), wim_uv (1,256,640,3), faces_uv_coord (1,F,3,2),
  obj_tex_img (256,256,3).  Index -1 addresses the LAST table row (python indexing, nmr.py:576,590).
"""
import torch
import torch.nn.functional as F

N_HAND_FACES = 1538          # trainer.py:73,79


def morph_erode(x, ks):
    """util.morph(mode='erode') (utils/util.py:142-153): pad with ones, box-sum, keep where every tap is one."""
    p = ks // 2
    xp = F.pad(x, [p, p, p, p], value=1.0)
    out = F.conv2d(xp, torch.ones(1, 1, ks, ks, dtype=torch.float32))
    return (out == ks * ks).float()


def bary_lookup(table, fim, wim, n_pix):
    """T[p] = sum_k table[fim[p], k, :] * wim[p, k] where a face exists, -2 elsewhere
    (nmr.py:884,898-922 / 975,993-1005 / 1070,1078-1097)."""
    T = -2 * torch.ones((n_pix, 2), dtype=torch.float32)
    idx = fim.long().reshape(-1)
    w = wim.reshape(-1, 3)
    exist = idx != -1
    T[exist] = (table[idx[exist]] * w[exist][:, :, None]).sum(dim=1)
    return T, exist, idx


def texture_backward_warp(im, f2pts, src_fim, tb):
    """get_texture_backward_warp (nmr.py:973-1058) for one sample.  im (1,3,256,256); f2pts (F,3,2); returns (1,3,256,640)."""
    T, exist, idx = bary_lookup(f2pts, tb['fim_uv'][0], tb['wim_uv'][0], 256 * 640)
    from_fim = src_fim.long().reshape(-1)
    t11 = ((T[exist] + 1) / 2.0 * 255.0).long().clamp(0, 255)                   # nmr.py:1012
    vis = torch.zeros(t11.shape[0], dtype=torch.bool)
    for dx in (-1, 0, 1):                                                        # nmr.py:1013-1044: nine neighbours
        for dy in (-1, 0, 1):
            q = (t11 + torch.tensor([dx, dy])).clamp(0, 255)
            vis |= from_fim[q[:, 1] * 256 + q[:, 0]] == idx[exist]
    O = torch.zeros((256 * 640, 1), dtype=torch.float32)
    O[exist, 0] = 1 - vis.float()                                                # nmr.py:1046
    syn = F.grid_sample(im, T.view(1, 256, 640, 2), align_corners=False)         # nmr.py:1048-1050 (default flag)
    O = O.view(1, 256, 640, 1).permute(0, 3, 1, 2)
    O = morph_erode(O, 3)                                                        # nmr.py:1052-1054: open the occlusion mask
    O = 1 - morph_erode(1 - O, 3)
    syn = syn * (1 - O) + 1.0 * torch.ones_like(syn) * O
    syn[:, :, :, 384:] = tb['obj_tex_img'].permute(2, 0, 1)[None]                # nmr.py:1055-1056 (pre_load)
    return syn


def prepare_inputs(src_img, ref_img, src_faces, src_fim, src_wim, ref_fim, ref_wim, tables, bg_both=False, dexycb=False):
    """HandRecoveryFlow.forward (trainer.py:46-145) after the rasteriser.  `tables`: one dict per sample.
    Returns the reference's tuple: (input_G_src_bg, input_G_tsf_bg | None, input_G_src_obj, input_G_tsf_obj,
    input_G_src_hand, input_G_ref_hand, T_hand, src_crop_mask_bg, ref_crop_mask_bg, src_crop_mask_hand,
    ref_crop_mask_hand, None)."""
    B = src_img.shape[0]
    acc = {k: [] for k in ['smh', 'rmh', 'scond', 'rcond', 'sseg', 'rseg', 'rsrc', 'rref', 'T']}
    for i in range(B):
        tb = tables[i]
        nf = tb['map_fn'].shape[0] - 1
        f2pts = src_faces[i, :nf, :, 0:2].clone()                                # trainer.py:67-68
        f2pts[:, :, 1] *= -1
        sfim, rfim = src_fim[i:i + 1], ref_fim[i:i + 1]
        scond = tb['map_fn'][sfim.long()].permute(0, 3, 1, 2)                    # encode_fim, nmr.py:576-579
        rcond = tb['map_fn'][rfim.long()].permute(0, 3, 1, 2)
        sseg = tb['sem_full'][sfim.long()].permute(0, 3, 1, 2)                   # encode_sem, nmr.py:588-593
        rseg = tb['sem_full'][rfim.long()].permute(0, 3, 1, 2)
        sseg = torch.cat([(sseg == j).float() for j in range(1, 16)], dim=1)     # trainer.py:72,78
        rseg = torch.cat([(rseg == j).float() for j in range(1, 16)], dim=1)
        smh = morph_erode(1 - ((sfim != -1) & (sfim < N_HAND_FACES))[:, None].float(), 3)     # trainer.py:73
        rmh = morph_erode(1 - ((rfim != -1) & (rfim < N_HAND_FACES))[:, None].float(), 3)     # trainer.py:79
        T, _, _ = bary_lookup(f2pts, rfim[0], ref_wim[i], 256 * 256)             # cal_bc_transform, nmr.py:874-968 (T only)
        T = T.view(1, 256, 256, 2)
        T_hand = T * (rmh[:, 0][:, :, :, None] == 0) + (-2) * torch.ones_like(T) * (rmh[:, 0][:, :, :, None] == 1)   # :82
        tex = texture_backward_warp(src_img[i:i + 1], f2pts, sfim[0], tb)        # trainer.py:84
        T_ref, _, _ = bary_lookup(tb['faces_uv_coord'][0], rfim[0], ref_wim[i], 256 * 256)    # :85, nmr.py:1068-1100
        T_src, _, _ = bary_lookup(tb['faces_uv_coord'][0], sfim[0], src_wim[i], 256 * 256)    # :87
        rref = F.grid_sample(tex, T_ref.view(1, 256, 256, 2), align_corners=True)             # :86
        rsrc = F.grid_sample(tex, T_src.view(1, 256, 256, 2), align_corners=True)             # :88
        for k, v in zip(['smh', 'rmh', 'scond', 'rcond', 'sseg', 'rseg', 'rsrc', 'rref', 'T'],
                        [smh, rmh, scond, rcond, sseg, rseg, rsrc, rref, T_hand]):
            acc[k].append(v)
    c = {k: torch.cat(v, dim=0) for k, v in acc.items()}
    smb = morph_erode(c['scond'][:, -1:], 3)                                     # trainer.py:110-111
    rmb = morph_erode(c['rcond'][:, -1:], 3)

    def split(cond):                                                             # trainer.py:113-125
        hm = (cond[:, :1] < 1.5).float()
        om = (cond[:, :1] > 1.5).float()
        return (torch.cat([hm * cond[:, :2], cond[:, 2:] + 1 - hm], dim=1),
                torch.cat([om * cond[:, :2], cond[:, 2:] + 1 - om], dim=1))
    s_hand, s_obj = split(c['scond'])
    r_hand, r_obj = split(c['rcond'])
    in_src_obj = torch.cat([c['rsrc'] * (c['smh'] - smb), s_obj, c['sseg'][:, 6:]], dim=1)    # trainer.py:128
    in_src_hand = torch.cat([src_img * (1 - c['smh']), s_hand] + ([c['sseg'][:, :6]] if dexycb else []), dim=1)   # :129
    in_tsf_obj = torch.cat([c['rref'] * (c['rmh'] - rmb), r_obj, c['rseg'][:, 6:]], dim=1)    # :132
    in_ref_hand = torch.cat([c['rref'] * (1 - c['rmh']), r_hand] + ([c['rseg'][:, :6]] if dexycb else []), dim=1)  # :133
    # (dexycb: HOIG_DexYCB/models/trainer.py:131,135 append the six hand-part one-hots; everything else is identical)
    sbg = morph_erode(c['scond'][:, -1:], 15)                                                 # :136
    in_src_bg = torch.cat([src_img * sbg, sbg], dim=1)                                        # :137
    in_tsf_bg = None
    if bg_both:                                                                               # :139-141
        rbg = morph_erode(c['rcond'][:, -1:], 15)
        in_tsf_bg = torch.cat([ref_img * rbg, rbg], dim=1)
    return (in_src_bg, in_tsf_bg, in_src_obj, in_tsf_obj, in_src_hand, in_ref_hand, c['T'], smb, rmb, c['smh'], c['rmh'],
            None)


def to_prepared(out, src_img, ref_img, armask_src=None, armask_tsf=None):
    """The a2 attributes Trainer.set_input derives from that tuple (trainer.py:346-362), keyed as hoig_amd's PREPARED_KEYS."""
    d = dict(input_G_bg=out[0] if out[1] is None else torch.cat([out[0], out[1]], dim=0),
             input_G_src_obj=out[2], input_G_tsf_obj=out[3], input_G_src_hand=out[4], input_G_tsf_hand=out[5], T=out[6],
             bg_mask=torch.cat((out[7], out[8]), dim=0), hand_mask=torch.cat((out[9], out[10]), dim=0),
             real_src=src_img, real_tsf=ref_img)
    if armask_src is not None:
        d['armask_src'], d['armask_tsf'] = armask_src, armask_tsf
    return d
