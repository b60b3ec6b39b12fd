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
            objm, _ = _blob(g, S, 0.10, 0.18)
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
