"""Generates tests/golden/assets_tables.npz: the renderer's per-object tables for the synthetic asset tree of tests/assets_fixture.py,
computed by the REFERENCE's own code -- utils/mesh.py (load_obj, create_mapping), thirdparty/neural_renderer's look_at.py and
vertices_to_faces.py, loaded from /root/reference by file path, driven by the statements of MANORenderer.__init__ (utils/nmr.py:283-391)
that need neither the CUDA rasteriser nor cv2.  Run in the build container only:  python tests/golden/make_golden_assets.py
Stored: faces, map_fn, sem_full, faces_uv_coord per object, and the (1,F,3,3) face tensors the two UV rasterisations receive.
Nothing of the reference's source travels; the fixture is data."""
import importlib.util
import os
import pickle
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import assets_fixture as AF                       # noqa: E402

REF = '/root/reference/HOIG_HOv3'


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def main():
    mesh = _load('ref_mesh', os.path.join(REF, 'utils', 'mesh.py'))
    look_at = _load('ref_look_at', os.path.join(REF, 'thirdparty', 'neural_renderer', 'neural_renderer', 'look_at.py')).look_at
    v2f = _load('ref_v2f', os.path.join(REF, 'thirdparty', 'neural_renderer', 'neural_renderer', 'vertices_to_faces.py')).vertices_to_faces
    out = {}
    with tempfile.TemporaryDirectory() as root:
        AF.build(root)
        a = os.path.join(root, 'assets')
        hand_path = os.path.join(a, 'MANO_UV_right.obj')
        names = sorted(os.listdir(os.path.join(a, 'obj')))
        with open(os.path.join(a, 'semantics_hand.pkl'), 'rb') as f:
            sem_hand = pickle.load(f)
        eye = [0, 0, -(1. / np.tan(np.radians(30)) + 1)]                                     # nmr.py:356-357
        hand_map = torch.tensor(mesh.create_mapping('uv_seg', hand_path, contain_bg=True, fill_back=False)).float()      # :321-322
        for j, name in enumerate(names):
            p = os.path.join(a, 'obj', name, name + '.obj')
            # :283-301 (nr.load_obj's face list = the `f` lines' vertex indices; mesh.load_obj reads the same triangles)
            faces = torch.cat([torch.from_numpy(mesh.load_obj(hand_path)['faces']), torch.from_numpy(mesh.load_obj(p)['faces']) + 778], dim=0).int()
            sem_tensor = torch.zeros(1538, 1)                                                # :306-319
            for i, key in enumerate(['palm', 'thumb', 'index_finger', 'middle_finger', 'ring_finger', 'little_finger']):
                sem_tensor[sem_hand['right'][key]] = i + 1
            obj_map = torch.tensor(mesh.create_mapping('uv_seg', p, contain_bg=True, fill_back=False)).float()
            sem_full = torch.cat([sem_tensor, torch.ones(obj_map.shape[0] - 1, 1) * (j + 7), torch.zeros(1, 1)], dim=0)
            obj_map[:-1, :2] = obj_map[:-1, :2] + torch.tensor([1.5, 0.0]) * (j + 1)         # :330
            map_fn = torch.cat([hand_map[:-1], obj_map], dim=0)
            faces_uv_list, raster_in = [], []
            for k, path in enumerate((hand_path, p)):                                        # :364-383
                info = mesh.load_obj(path)
                vts = (torch.from_numpy(info['vts'])[None] - 0.5) * 2
                uv_vert = torch.cat([vts, torch.ones_like(vts[:, :, 0:1])], dim=2)
                uv_vert = look_at(uv_vert, eye)
                faces_uv = torch.LongTensor(info['faces_vts'])[None]
                uv_new = (uv_vert + 1) / 2
                faces_uv_list.append(uv_new[0, faces_uv] + (torch.Tensor([1.5, 0, 0])[None, None, None] if k else 0))
                raster_in.append(v2f(uv_vert, faces_uv))
            coord = torch.cat(faces_uv_list, dim=1)[:, :, :, :2]                             # :388-391
            coord = (coord - torch.Tensor([[1.25, 0.5]])) * torch.Tensor([[0.8, -2]])
            for key, v in (('faces', faces), ('map_fn', map_fn), ('sem_full', sem_full), ('faces_uv_coord', coord),
                           ('raster_hand', raster_in[0]), ('raster_obj', raster_in[1])):
                out['%d/%s' % (j, key)] = v.numpy()
    path = os.path.join(HERE, 'assets_tables.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, {k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    main()
