"""Generates tests/golden/raster_vertex_stage.npz: seeded vertices / camera / face indices and the (B,F,3,3) face tensor the
REFERENCE's own vertex stage of render_fim_wim produces for them (oracle/ref_harness.py::reference_vertex_stage:
utils/nmr.py:109-140,503-511 + neural_renderer's look_at / vertices_to_faces).  Build container only:
    python tests/golden/make_golden_raster_vertex.py            # HOIG_HOv3 copy
    python tests/golden/make_golden_raster_vertex.py dexycb     # HOIG_DexYCB copy -> raster_vertex_stage_dexycb.npz (its nmr.py:38-48,146-163)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_harness as RH           # noqa: E402


def inputs(seed=2, B=2, V=60, F=90):
    g = np.random.default_rng(seed)
    verts = np.concatenate([g.uniform(-0.1, 0.1, size=(B, V, 2)), g.uniform(-0.7, -0.5, size=(B, V, 1))], -1).astype(np.float32)
    cam = np.array([[600.0, 0, 128, 0, 610.0, 120, 0, 0, 1, 1.1, 0.05, -3, -0.02, 0.95, 4]] * B, np.float32)
    cam[1, 2] = 140
    idx = g.integers(0, V, size=(F, 3)).astype(np.int32)
    return torch.from_numpy(cam), torch.from_numpy(verts), torch.from_numpy(idx)


def inputs_dexycb(seed=3, B=3, V=60, F=90):
    """The HOIG_DexYCB copy: cam = [fx, fy, cx, cy | 2x3 crop transform], vertices in front of the camera at positive depth."""
    g = np.random.default_rng(seed)
    verts = np.concatenate([g.uniform(-0.1, 0.1, size=(B, V, 2)), g.uniform(0.5, 0.9, size=(B, V, 1))], -1).astype(np.float32)
    cam = np.array([[615.0, 614.5, 312.25, 241.5, 1.1, 0.0, -210.0, 0.0, 1.05, -130.0]] * B, np.float32)
    cam[1, :4] = (600.0, 601.0, 320.0, 240.0)
    cam[2, 4:] = (0.9, 0.01, -150.0, -0.02, 0.95, -100.0)
    idx = g.integers(0, V, size=(F, 3)).astype(np.int32)
    return torch.from_numpy(cam), torch.from_numpy(verts), torch.from_numpy(idx)


if __name__ == '__main__':
    here = os.path.dirname(os.path.abspath(__file__))
    if len(sys.argv) > 1 and sys.argv[1] == 'dexycb':          # (the harness imports one reference copy per process)
        # the DexYCB copy's projection only runs one sample at a time (its cam2pixel multiplies (1,V) by a (1,) focal length)
        cam, verts, idx = inputs_dexycb()
        faces = torch.cat([RH.reference_vertex_stage(cam[i:i + 1].clone(), verts[i:i + 1].clone(), idx, copy='dexycb')
                           for i in range(cam.shape[0])])
        path = os.path.join(here, 'raster_vertex_stage_dexycb.npz')
    else:
        cam, verts, idx = inputs()
        faces = RH.reference_vertex_stage(cam.clone(), verts.clone(), idx)
        path = os.path.join(here, 'raster_vertex_stage.npz')
    np.savez_compressed(path, cam=cam.numpy(), vertices=verts.numpy(), faces_idx=idx.numpy(), faces=faces.numpy())
    print('wrote', path, os.path.getsize(path), 'bytes')
