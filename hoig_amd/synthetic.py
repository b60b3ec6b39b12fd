"""Seeded synthetic hand-object-pose tensors: the a2 input surface of the path.

These are the attributes ``Trainer.set_input`` stages after the (out-of-scope)
``HandRecoveryFlow`` (reference models/trainer.py:346-362); value ranges follow
trainer.py:109-136, utils/nmr.py:325,884 and data/hov3_dataset.py:235-236 as
summarised in SURVEY.md §8d.  Generated with numpy's counter-based Philox so
the same (seed, batch, side) gives bit-identical tensors on every platform.
"""
import numpy as np
import torch

N_OBJECTS = 9          # HOv3 object classes (models/trainer.py:11)


def _rng(seed, stream):
    return np.random.Generator(np.random.Philox(key=[seed & 0xFFFFFFFF, stream]))


def _blob(g, S, lo, hi):
    """A disk of radius in [lo,hi]*S with a random centre near the middle."""
    cy, cx = g.uniform(0.35, 0.65, size=2) * S
    r = g.uniform(lo, hi) * S
    yy, xx = np.mgrid[0:S, 0:S].astype(np.float32)
    return ((yy - cy) ** 2 + (xx - cx) ** 2 <= r * r).astype(np.float32), (cy, cx, r)


def _smooth_noise(g, S, sigma):
    """Low-frequency displacement field: coarse N(0, sigma) grid, bilinearly upsampled."""
    n = max(2, S // 16)
    coarse = g.normal(0.0, sigma, size=(n, n)).astype(np.float32)
    t = torch.from_numpy(coarse)[None, None]
    return torch.nn.functional.interpolate(t, size=(S, S), mode='bilinear', align_corners=True)[0, 0].numpy()


def make_inputs(batch, side, seed=8, dataset='hov3'):
    """Returns a dict of CPU fp32 tensors keyed by the a2 attribute names
    (without the leading underscore)."""
    B, S = batch, side
    hov3 = dataset == 'hov3'
    out = {k: [] for k in ['real_src', 'real_tsf', 'input_G_bg', 'input_G_src_obj', 'input_G_tsf_obj',
                           'input_G_src_hand', 'input_G_tsf_hand', 'T', 'armask_src', 'armask_tsf']}
    bg_masks = {'src': [], 'tsf': []}
    hand_masks = {'src': [], 'tsf': []}
    ident = np.linspace(-1.0, 1.0, S, dtype=np.float32)
    for b in range(B):
        g = _rng(seed, b)
        k = int(g.integers(0, N_OBJECTS))
        per = {}
        for side_name in ('src', 'tsf'):
            rgb = g.uniform(-1.0, 1.0, size=(3, S, S)).astype(np.float32)
            hand, (cy, cx, r) = _blob(g, S, 0.12, 0.2)
            objm, (oy, ox, orad) = _blob(g, S, 0.10, 0.18)
            if not (objm * (1.0 - hand)).any():
                # the object disk fell entirely inside the hand disk: a hand-object pair always shows some of the object, and an
                # all-zero object image makes obj_model's input spatially constant -- its instance norms then normalise pure
                # rounding noise (variance 0 + eps), an ill-conditioned case in the reference too.  Slide the object along the
                # line of centres until it sticks out (no random numbers consumed: every other sample stays bit-identical).
                dy_, dx_ = oy - cy, ox - cx
                nrm = float(np.hypot(dy_, dx_))
                dy_, dx_ = (dy_ / nrm, dx_ / nrm) if nrm > 1e-6 else (0.0, 1.0)
                oy, ox = cy + dy_ * (r + 0.5 * orad), cx + dx_ * (r + 0.5 * orad)
                yy, xx = np.mgrid[0:S, 0:S].astype(np.float32)
                objm = ((yy - oy) ** 2 + (xx - ox) ** 2 <= orad * orad).astype(np.float32)
            fg = np.maximum(hand, objm)
            bgm = 1.0 - fg                           # background = outside hand-object region
            # 15x15 erosion of the background mask (trainer.py:136 feeds the eroded mask)
            t = torch.from_numpy(bgm)[None, None]
            er = -torch.nn.functional.max_pool2d(-t, 15, 1, 7)[0, 0].numpy() if S >= 16 else bgm
            obj_only = objm * (1.0 - hand)
            hand_mask = 1.0 - hand                   # "hand mask" channel: 1 outside the hand (nmr.py:325)
            uv = g.uniform(0.0, 1.0, size=(2, S, S)).astype(np.float32) * hand
            bgflag = (1.0 - hand)[None]
            hand_in = [rgb * hand, uv, bgflag]
            if not hov3:
                part = g.integers(0, 6, size=(S, S))
                seg6 = np.stack([(part == j).astype(np.float32) * hand for j in range(6)])
                hand_in.append(seg6)
            u = (1.5 * (k + 1) + g.uniform(0.0, 1.0, size=(S, S))).astype(np.float32) * obj_only
            v = g.uniform(0.0, 1.0, size=(S, S)).astype(np.float32) * obj_only
            onehot = np.zeros((N_OBJECTS, S, S), np.float32)
            onehot[k] = obj_only
            obj_in = [rgb * obj_only, u[None], v[None], (1.0 - obj_only)[None], onehot]
            arm = (1.99 * hand)[None] * float(g.integers(0, 2))
            per[side_name] = dict(rgb=rgb, hand=hand, bgm=bgm, er=er, hand_mask=hand_mask,
                                  hand_in=np.concatenate(hand_in, 0), obj_in=np.concatenate(obj_in, 0), arm=arm)
            bg_masks[side_name].append(bgm[None])
            hand_masks[side_name].append(hand_mask[None])
        s, t_ = per['src'], per['tsf']
        out['real_src'].append(s['rgb'])
        out['real_tsf'].append(t_['rgb'])
        out['input_G_bg'].append(np.concatenate([s['rgb'] * s['er'], s['er'][None]], 0))
        out['input_G_src_obj'].append(s['obj_in'])
        out['input_G_tsf_obj'].append(t_['obj_in'])
        out['input_G_src_hand'].append(s['hand_in'])
        out['input_G_tsf_hand'].append(t_['hand_in'])
        out['armask_src'].append(s['arm'])
        out['armask_tsf'].append(t_['arm'])
        # flow T: identity grid + smooth displacement inside the tsf hand blob, -2 sentinel elsewhere (nmr.py:884)
        gx = np.broadcast_to(ident[None, :], (S, S)) + _smooth_noise(g, S, 0.05)
        gy = np.broadcast_to(ident[:, None], (S, S)) + _smooth_noise(g, S, 0.05)
        T = np.stack([gx, gy], -1).astype(np.float32)
        T[t_['hand'] == 0] = -2.0
        out['T'].append(T)
    res = {k: torch.from_numpy(np.stack(v).astype(np.float32)) for k, v in out.items()}
    res['bg_mask'] = torch.from_numpy(np.concatenate([np.stack(bg_masks['src']), np.stack(bg_masks['tsf'])], 0))
    res['hand_mask'] = torch.from_numpy(np.concatenate([np.stack(hand_masks['src']), np.stack(hand_masks['tsf'])], 0))
    if not hov3:
        del res['armask_src'], res['armask_tsf']
    return res


# ---------------------------------------------------------------------------------------------------------------------
# Synthetic RASTERISER outputs + per-object tables: the input surface of hoig_amd.input_prep (the stage of
# HandRecoveryFlow.forward, trainer.py:46-145, that follows render_fim_wim).  The reference hard-wires 256 x 256 images and a
# 256 x 640 texture atlas (nmr.py:975,1040-1051,1070), so there is one size.  Scene: 8x8-pixel cells; the cells of a hand disk
# carry hand faces (ids < 1538), the cells of an object box object faces (ids >= 1538); a face present in the SOURCE view
# is a triangle over its source cell, so the visibility test of get_texture_backward_warp (nmr.py:1006-1044) has both outcomes.
N_HAND_FACES = 1538
PREP_SIDE, TEX_W = 256, 640


def _cells(g, kind):
    """8x8-pixel cell masks (32 x 32) of a hand disk and an object box."""
    yy, xx = np.mgrid[0:32, 0:32]
    cy, cx, r = g.uniform(10, 22), g.uniform(10, 22), g.uniform(5, 8)
    hand = (yy - cy) ** 2 + (xx - cx) ** 2 <= r * r
    y0, x0 = int(g.integers(3, 18)), int(g.integers(3, 18))
    obj = np.zeros((32, 32), bool)
    obj[y0:y0 + int(g.integers(6, 11)), x0:x0 + int(g.integers(6, 11))] = True
    obj &= ~hand
    return hand, obj


def make_object_tables(obj_id, seed=8):
    """Per-object tables with the shapes / value conventions of MANORenderer's buffers (nmr.py:295-406)."""
    g = _rng(seed, 1000 + obj_id)
    n_obj = int(g.integers(200, 500))
    F = N_HAND_FACES + n_obj
    map_fn = np.zeros((F + 1, 3), np.float32)                  # (u, v, background flag); last row = background
    map_fn[:N_HAND_FACES, :2] = g.uniform(0.05, 0.95, size=(N_HAND_FACES, 2))
    map_fn[N_HAND_FACES:F, 0] = 1.5 * (obj_id + 1) + g.uniform(0.05, 0.95, size=n_obj)      # nmr.py:325
    map_fn[N_HAND_FACES:F, 1] = g.uniform(0.05, 0.95, size=n_obj)
    map_fn[F] = (0.0, 0.0, 1.0)
    sem = np.zeros((F + 1, 1), np.float32)                     # nmr.py:303-313
    sem[:N_HAND_FACES, 0] = g.integers(0, 7, size=N_HAND_FACES)
    sem[N_HAND_FACES:F, 0] = obj_id + 7
    # texture atlas: hand charts in columns 0..255, nothing in 256..383, object charts in 384..639 (nmr.py:393-394)
    fim_uv = -np.ones((PREP_SIDE, TEX_W), np.int32)
    uv_coord = g.uniform(-1.0, 1.0, size=(F, 3, 2)).astype(np.float32)        # faces absent from the atlas: anywhere
    cells = [(cy, cx) for cy in range(32) for cx in range(32)] + [(cy, cx) for cy in range(32) for cx in range(48, 80)]
    hand_ids = g.permutation(N_HAND_FACES)[:900]
    obj_ids = N_HAND_FACES + g.permutation(n_obj)[:min(n_obj, 700)]
    order = g.permutation(1024)
    for n, f in enumerate(hand_ids):
        cy, cx = cells[order[n]]
        fim_uv[cy * 8:cy * 8 + 8, cx * 8:cx * 8 + 8] = f
    order = g.permutation(1024)
    for n, f in enumerate(obj_ids):
        cy, cx = cells[1024 + order[n]]
        fim_uv[cy * 8:cy * 8 + 8, cx * 8:cx * 8 + 8] = f
    for f in np.concatenate([hand_ids, obj_ids]):
        ys, xs = np.nonzero(fim_uv == f)
        y0, x0 = ys.min(), xs.min()
        px = np.array([[x0, y0], [x0 + 7, y0], [x0, y0 + 7]], np.float32)
        uv_coord[f, :, 0] = px[:, 0] / (TEX_W - 1) * 2 - 1                     # align_corners=True texel centres
        uv_coord[f, :, 1] = px[:, 1] / (PREP_SIDE - 1) * 2 - 1
    wim_uv = g.uniform(0.05, 1.0, size=(PREP_SIDE, TEX_W, 3)).astype(np.float32)
    wim_uv /= wim_uv.sum(-1, keepdims=True)
    wim_uv[fim_uv < 0] = 0.0
    tex = g.uniform(-1.0, 1.0, size=(PREP_SIDE, PREP_SIDE, 3)).astype(np.float32)
    t = torch.from_numpy
    return dict(n_faces=F, map_fn=t(map_fn), sem_full=t(sem), fim_uv=t(fim_uv)[None], wim_uv=t(wim_uv)[None],
                faces_uv_coord=t(uv_coord)[None], obj_tex_img=t(tex))


def make_raster(batch, seed=8):
    """Seeded synthetic outputs of `render_fim_wim` for `batch` source / reference views (+ the two images, the object id of
    each sample).  CPU tensors: faces (B, Fmax, 3, 3) fp32 [only the first n_faces rows of a sample are meaningful], fim
    (B,256,256) int32 with -1 = no face, wim (B,256,256,3) fp32."""
    S = PREP_SIDE
    objs = [int(_rng(seed, 2000 + b).integers(0, N_OBJECTS)) for b in range(batch)]
    tables = {k: make_object_tables(k, seed) for k in sorted(set(objs))}
    fmax = max(tb['n_faces'] for tb in tables.values())
    out = dict(src_img=[], ref_img=[], src_faces=[], src_fim=[], src_wim=[], ref_fim=[], ref_wim=[])
    for b in range(batch):
        g = _rng(seed, 3000 + b)
        F = tables[objs[b]]['n_faces']
        faces = np.zeros((fmax, 3, 3), np.float32)
        faces[:F, :, :2] = g.uniform(-1.1, 1.1, size=(F, 1, 2)) + g.uniform(-0.03, 0.03, size=(F, 3, 2))
        faces[:F, :, 2] = g.uniform(1.0, 3.0, size=(F, 3))
        views = {}
        for name in ('src', 'ref'):
            hand, obj = _cells(g, name)
            fim = -np.ones((S, S), np.int32)
            hid = g.permutation(N_HAND_FACES)
            oid = N_HAND_FACES + g.permutation(F - N_HAND_FACES)
            for n, (cy, cx) in enumerate(zip(*np.nonzero(hand))):
                fim[cy * 8:cy * 8 + 8, cx * 8:cx * 8 + 8] = hid[n]
            for n, (cy, cx) in enumerate(zip(*np.nonzero(obj))):
                fim[cy * 8:cy * 8 + 8, cx * 8:cx * 8 + 8] = oid[n % len(oid)]
            # ragged silhouette: drop random pixels at the cell borders so that the 3x3 erosions see 1-pixel structures
            drop = (g.uniform(size=(S, S)) < 0.02)
            fim[drop] = -1
            wim = g.uniform(0.05, 1.0, size=(S, S, 3)).astype(np.float32)
            wim /= wim.sum(-1, keepdims=True)
            wim[fim < 0] = 0.0
            views[name] = (fim, wim)
        # faces seen in the source view: a triangle over their source cell (image coords in the align_corners=True convention
        # of nmr.py:1012; y is stored NEGATED, trainer.py:69 flips it back)
        sfim = views['src'][0]
        for f in np.unique(sfim[sfim >= 0]):
            ys, xs = np.nonzero(sfim == f)
            y0, x0 = ys.min(), xs.min()
            px = np.array([[x0 + 1, y0 + 1], [x0 + 6, y0 + 1], [x0 + 1, y0 + 6]], np.float32)
            faces[f, :, 0] = px[:, 0] / (S - 1) * 2 - 1
            faces[f, :, 1] = -(px[:, 1] / (S - 1) * 2 - 1)
        out['src_img'].append(g.uniform(-1.0, 1.0, size=(3, S, S)).astype(np.float32))
        out['ref_img'].append(g.uniform(-1.0, 1.0, size=(3, S, S)).astype(np.float32))
        out['src_faces'].append(faces)
        out['src_fim'].append(views['src'][0])
        out['src_wim'].append(views['src'][1])
        out['ref_fim'].append(views['ref'][0])
        out['ref_wim'].append(views['ref'][1])
    res = {k: torch.from_numpy(np.stack(v)) for k, v in out.items()}
    res['obj_ids'] = objs
    res['tables'] = tables
    return res
