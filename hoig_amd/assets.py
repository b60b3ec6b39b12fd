"""``MANORenderer``'s per-object buffers, built from the asset files (HOIG_HOv3/utils/nmr.py:243-406).

The reference's renderer constructs, once, for the hand mesh and each of nine YCB objects: the merged face list over the
[778 hand | object] vertex buffer (nmr.py:283-301), the per-face semantic labels (:303-319), the per-face ``uv_seg`` table (:321-337,
utils/mesh.py:156-212,368-407), and -- by rasterising the two UV layouts -- the 256 x 640 texture atlas's face-index / weight maps
and the faces' atlas coordinates (:364-399), plus the object's own texture image (:401-405).  ``HandRecoveryFlow`` (hoig_amd/
hand_recovery.py) takes exactly these as ``opt.object_assets``; until round 6 the caller had to supply them ready-made.  The asset
FILES are still the caller's (MANO_UV_right.obj, assets/obj/<name>/<name>.obj + texture_map.png, semantics_hand.pkl: not in the
reference checkout, .gitignore:3): this module only does what ``MANORenderer.__init__`` does with them.

Host part (OBJ parsing, the uv_seg tables, labels, atlas coordinates): numpy, float32 where the reference is float32.  Device part:
the two UV rasterisations (hoig_amd.raster: the product's rasteriser, as the reference uses neural_renderer's) and the texture's
resize (hoig_resize_linear_u8: cv2.resize's fixed-point INTER_LINEAR).  No CPU fallback for those two.
"""
import math
import os

import numpy as np
import torch

N_HAND_VERTS, N_HAND_FACES = 778, 1538        # nmr.py:292,304 (MANO's right hand)
HAND_PARTS = ['palm', 'thumb', 'index_finger', 'middle_finger', 'ring_finger', 'little_finger']      # nmr.py:309
ATLAS = 256                                    # nmr.py:384-385,404: the atlas tiles are 256 x 256 whatever image_size is


def load_obj(path):
    """Wavefront OBJ -> dict(vertices (V,3) f32, faces (F,3) i32, vts (T,2) f32, faces_vts (F,3) i32), zero-based.
    utils/mesh.py:28-77 reads `f a/b/c` triples (first three corners); neural_renderer's load_obj (load_obj.py:99-130) fan-triangulates
    polygons over the vertex index alone.  The two agree on triangle meshes, which is what the renderer's assets are: anything else is
    refused here rather than read two ways."""
    verts, vts, faces, faces_vts = [], [], [], []
    with open(path, 'r') as fp:
        for line in fp:
            parts = line.split()
            if not parts or parts[0].startswith('#'):
                continue
            if parts[0] == 'v':
                verts.append([float(parts[1]), float(parts[2]), float(parts[3])])
            elif parts[0] == 'vt':
                vts.append([float(parts[1]), float(parts[2])])
            elif parts[0] == 'f':
                if len(parts) != 4:
                    raise ValueError('%s: the renderer\'s tables are defined for triangle meshes; found a face with %d corners'
                                     % (path, len(parts) - 1))
                corner = [p.split('/') for p in parts[1:4]]
                faces.append([int(c[0]) - 1 for c in corner])
                if all(len(c) > 1 and c[1] for c in corner):
                    faces_vts.append([int(c[1]) - 1 for c in corner])
    if not verts or not faces:
        raise ValueError('%s: no vertices / faces' % path)
    if faces_vts and len(faces_vts) != len(faces):
        raise ValueError('%s: some faces carry texture indices and some do not' % path)
    return dict(vertices=np.asarray(verts, np.float32), faces=np.asarray(faces, np.int32),
                vts=np.asarray(vts, np.float32).reshape(-1, 2), faces_vts=np.asarray(faces_vts, np.int32).reshape(-1, 3))


def uv_seg_mapping(obj):
    """mesh.create_mapping('uv_seg', path, contain_bg=True) (utils/mesh.py:368-407): per face the barycentre of its UV triangle with v
    flipped (get_f2vts :173-194, compute_barycenter :156-170: v2 + 0.5 (v0 - v2) + 0.5 (v1 - v2), float32) and a zero third column;
    last row = the background (0, 0, 1)."""
    if obj['faces_vts'].shape[0] == 0:
        raise ValueError('the mesh has no texture coordinates (vt / f a/b/c)')
    vts = obj['vts'].copy()
    vts[:, 1] = np.float32(1) - vts[:, 1]
    vts = np.concatenate([vts, np.zeros((vts.shape[0], 1), np.float32)], axis=-1)
    f2vts = vts[obj['faces_vts']]                                   # (F,3,3)
    v2 = f2vts[:, 2]
    fbc = v2 + np.float32(0.5) * (f2vts[:, 0] - v2) + np.float32(0.5) * (f2vts[:, 1] - v2)
    return np.concatenate([fbc.astype(np.float32), np.array([[0, 0, 1]], np.float32)], axis=0)


def hand_semantics(sem_hand):
    """nmr.py:306-310: (1538, 1) per-FACE labels 1..6 from semantics_hand.pkl's ['right'][part] index lists (0 elsewhere)."""
    sem = np.zeros((N_HAND_FACES, 1), np.float32)
    right = sem_hand['right'] if 'right' in sem_hand else sem_hand
    for i, key in enumerate(HAND_PARTS):
        sem[np.asarray(right[key], np.int64)] = i + 1
    return sem


def _eye(viewing_angle):
    return np.float32(-(1.0 / math.tan(math.radians(viewing_angle)) + 1.0))


def uv_layout(obj, viewing_angle=30.0):
    """nmr.py:373-380: the UV vertices as the rasteriser sees them -- ((vt - 0.5) * 2, 1) through nr.look_at with the renderer's eye
    (the rotation is the identity for eye on -z, at = origin, up = +y: what remains is the shift by -eye) -> (uv_vert (T,3) f32,
    the same mapped to [0,1]: (uv_vert + 1) / 2)."""
    vts = (obj['vts'] - np.float32(0.5)) * np.float32(2)
    uv = np.concatenate([vts, np.ones((vts.shape[0], 1), np.float32)], axis=1)
    uv = uv - np.array([0, 0, _eye(viewing_angle)], np.float32)
    return uv.astype(np.float32), ((uv + np.float32(1)) / np.float32(2)).astype(np.float32)


def _texture_image(path, device):
    """nmr.py:401-405: cv2.imread(...)[:, :, ::-1] -> cv2.resize(., (256, 256)) -> float32 / 255 * 2 - 1, (256,256,3) RGB."""
    from . import _lib as L
    from .data.hov3_dataset import imread_bgr
    bgr = imread_bgr(path)                                                             # (H,W,3) uint8, cv2's channel order
    rgb = torch.from_numpy(np.ascontiguousarray(bgr[:, :, ::-1])).to(device)
    out = torch.empty((1, ATLAS, ATLAS, 3), dtype=torch.uint8, device=device)
    L.call('hoig_resize_linear_u8', rgb.data_ptr(), 1, int(rgb.shape[0]), int(rgb.shape[1]), 3, out.data_ptr(), ATLAS, ATLAS,
           torch.cuda.current_stream().cuda_stream)
    # (the three float operations on the HOST, as the reference's torch-CPU expression does them: a device division is not IEEE-exact)
    return (out[0].cpu().float() / 255.0 * 2.0 - 1).to(device)


def host_tables(hand_obj_path, objects, sem_hand, viewing_angle=30.0):
    """The part of MANORenderer.__init__ that needs no rasteriser: -> {object id: {'faces', 'map_fn', 'sem_full', 'faces_uv_coord'
    (numpy, the reference's dtypes), 'raster_hand', 'raster_obj' ((1,F,3,3) f32: what the two UV rasterisations receive)}}."""
    hand = load_obj(hand_obj_path)
    if hand['vertices'].shape[0] != N_HAND_VERTS or hand['faces'].shape[0] != N_HAND_FACES:
        raise ValueError('%s: expected MANO\'s %d vertices / %d faces' % (hand_obj_path, N_HAND_VERTS, N_HAND_FACES))
    hand_map = uv_seg_mapping(hand)
    hand_sem = hand_semantics(sem_hand)
    hand_uv, hand_uv01 = uv_layout(hand, viewing_angle)
    out = {}
    for rank, oid in enumerate(sorted(objects)):
        ob = load_obj(objects[oid][0])
        nf = ob['faces'].shape[0]
        faces = np.concatenate([hand['faces'], ob['faces'] + N_HAND_VERTS], axis=0).astype(np.int32)            # nmr.py:286-301
        ob_map = uv_seg_mapping(ob)
        ob_map[:-1, :2] = ob_map[:-1, :2] + np.array([1.5, 0.0], np.float32) * np.float32(rank + 1)              # :330
        map_fn = np.concatenate([hand_map[:-1], ob_map], axis=0)                                               # :332-334
        sem_full = np.concatenate([hand_sem, np.full((nf, 1), rank + 7, np.float32), np.zeros((1, 1), np.float32)], axis=0)   # :317-319
        ob_uv, ob_uv01 = uv_layout(ob, viewing_angle)
        coord = np.concatenate([hand_uv01[hand['faces_vts']], ob_uv01[ob['faces_vts']] + np.array([1.5, 0, 0], np.float32)], axis=0)
        coord = (coord[None, :, :, :2] - np.array([[1.25, 0.5]], np.float32)) * np.array([[0.8, -2]], np.float32)          # :388-391
        out[int(oid)] = dict(faces=faces, map_fn=map_fn, sem_full=sem_full, faces_uv_coord=coord.astype(np.float32),
                             raster_hand=hand_uv[hand['faces_vts']][None], raster_obj=ob_uv[ob['faces_vts']][None])    # nr.vertices_to_faces
    return out


def build_object_assets(hand_obj_path, objects, sem_hand, device=None, viewing_angle=30.0):
    """-> {object id: {'faces', 'map_fn', 'sem_full', 'fim_uv', 'wim_uv', 'faces_uv_coord', 'obj_tex_img'}}: ``opt.object_assets``.

    hand_obj_path: MANO_UV_right.obj (nmr.py:244-245: face list and UV layout of the hand).
    objects: {object id: (obj path, texture path)}; the id is what the batch's ``objName`` holds (trainer.py:11,65), and it also
        fixes the object's atlas column: MANORenderer walks ``sorted(os.listdir('assets/obj'))`` and shifts object number i (0-based)
        by 1.5 * (i + 1) in u (nmr.py:330) and labels its faces i + 7 (:317) -- here i = the rank of the id among the given ids.
    sem_hand: the unpickled semantics_hand.pkl (or its ['right'] dict)."""
    from . import raster
    device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    if device.type != 'cuda':
        raise NotImplementedError('hoig_amd.assets rasterises the UV layouts on the HIP device (no CPU path)')
    host = host_tables(hand_obj_path, objects, sem_hand, viewing_angle)
    rast = lambda f: raster.rasterize_fim_wim(torch.from_numpy(f).to(device), ATLAS)          # nmr.py:382: no anti-aliasing, default near / far
    hand_fim = hand_wim = None
    out = {}
    for oid, t in host.items():
        if hand_fim is None:
            hand_fim, hand_wim = rast(t['raster_hand'])
        ob_fim, ob_wim = rast(t['raster_obj'])
        gap_f = torch.full((1, ATLAS, ATLAS // 2), -1, dtype=torch.int32, device=device)                         # :384-385
        gap_w = torch.zeros((1, ATLAS, ATLAS // 2, 3), dtype=torch.float32, device=device)
        fim_uv = torch.cat([hand_fim, gap_f, ob_fim + (ob_fim != -1).to(torch.int32) * N_HAND_FACES], dim=2)    # :386
        wim_uv = torch.cat([hand_wim, gap_w, ob_wim], dim=2)
        out[oid] = dict(faces=torch.from_numpy(t['faces'].astype(np.int64)), map_fn=torch.from_numpy(t['map_fn']),
                        sem_full=torch.from_numpy(t['sem_full']), fim_uv=fim_uv, wim_uv=wim_uv,
                        faces_uv_coord=torch.from_numpy(t['faces_uv_coord']), obj_tex_img=_texture_image(objects[oid][1], device))
    return out


def object_assets_from_tree(root, obj_ids=None, device=None):
    """The reference's layout under `root` (its working directory): assets/MANO_UV_right.obj, assets/semantics_hand.pkl,
    assets/obj/<name>/<name>.obj + texture_map.png (nmr.py:244,280,303,401).  Object ids = positions in the sorted directory list
    (MANORenderer's order), or the given {id: name}."""
    from .data.dataset_base import read_pickle
    names = sorted(os.listdir(os.path.join(root, 'assets', 'obj')))
    ids = {i: n for i, n in enumerate(names)} if obj_ids is None else dict(obj_ids)
    objects = {i: (os.path.join(root, 'assets', 'obj', n, n + '.obj'), os.path.join(root, 'assets', 'obj', n, 'texture_map.png'))
               for i, n in ids.items()}
    sem = read_pickle(os.path.join(root, 'assets', 'semantics_hand.pkl'), 'hand semantics')
    return build_object_assets(os.path.join(root, 'assets', 'MANO_UV_right.obj'), objects, sem, device=device)
