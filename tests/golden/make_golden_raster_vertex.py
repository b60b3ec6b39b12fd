"""Generates tests/golden/raster_vertex_stage.npz: seeded vertices / camera / face indices and the (B,F,3,3) face tensor the
REFERENCE's own vertex stage of render_fim_wim produces for them (oracle/ref_harness.py::reference_vertex_stage:
utils/nmr.py:109-140,503-511 + neural_renderer's look_at / vertices_to_faces).  Build container only:
    python tests/golden/make_golden_raster_vertex.py"""
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


if __name__ == '__main__':
    cam, verts, idx = inputs()
    faces = RH.reference_vertex_stage(cam.clone(), verts.clone(), idx)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'raster_vertex_stage.npz')
    np.savez_compressed(path, cam=cam.numpy(), vertices=verts.numpy(), faces_idx=idx.numpy(), faces=faces.numpy())
    print('wrote', path, os.path.getsize(path), 'bytes')
