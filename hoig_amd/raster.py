"""Face-index / weight-map rasteriser on the device: what ``MANORenderer.render_fim_wim`` (utils/nmr.py:496-513)
gets from ``nr.rasterize_face_index_map_and_weight_map(faces, image_size, False)`` -- neural_renderer's CUDA kernels
(thirdparty/neural_renderer/neural_renderer/cuda/rasterize_cuda_kernel.cu:40-186) plus the wrapper's fill and vertical flip
(neural_renderer/rasterize.py:50-52,334-338) -- as one tile-binned HIP launch pair (hoig_amd/csrc/raster.hip).
``render_fim_wim`` mirrors the reference method: projection (nmr.py:109-140), y flip, look-at translation
(neural_renderer/look_at.py with the renderer's fixed eye) and ``vertices_to_faces`` are a handful of batched torch ops
(plumbing: 3 x 3 matrix products over ~2 k vertices); the rasterisation is the kernel."""
import math

import torch

from . import _lib as L

DEFAULT_NEAR, DEFAULT_FAR = 0.1, 100.0        # neural_renderer/rasterize.py:10-11 (the call at nmr.py:512 passes neither)


def rasterize_fim_wim(faces, image_size=256, near=DEFAULT_NEAR, far=DEFAULT_FAR):
    """faces (B,F,3,3) fp32 on the HIP device -> (fim (B,S,S) int32 with -1 = no face, wim (B,S,S,3))."""
    if not faces.is_cuda:
        raise NotImplementedError('hoig_amd.raster runs on the HIP device only (no CPU path)')
    if faces.dim() != 4 or tuple(faces.shape[2:]) != (3, 3):
        raise ValueError('faces must be (B,F,3,3)')
    faces = faces.float().contiguous()
    B, F = int(faces.shape[0]), int(faces.shape[1])
    S = int(image_size)
    fim = torch.empty(B, S, S, dtype=torch.int32, device=faces.device)
    wim = torch.empty(B, S, S, 3, dtype=torch.float32, device=faces.device)
    ws = torch.empty(int(L.lib.hoig_rasterize_workspace_bytes(B, F)), dtype=torch.uint8, device=faces.device)
    L.call('hoig_rasterize_fim_wim', faces.data_ptr(), B, F, S, float(near), float(far), fim.data_ptr(), wim.data_ptr(),
           ws.data_ptr(), torch.cuda.current_stream().cuda_stream)
    return fim, wim


_consts = {}


def _const(device, key, values):
    """A small constant tensor, uploaded ONCE per device: torch.tensor(..., device=...) on every call is a pageable host-to-device
    copy, i.e. a host wait for everything queued on the current stream."""
    t = _consts.get((device, key))
    if t is None:
        t = _consts[(device, key)] = torch.tensor(values, dtype=torch.float32, device=device)
    return t


def project(vertices, cam):
    """orthographic_proj_withz_idrot (nmr.py:109-140): OpenGL axis change, camera matrix, perspective divide, 2x3 crop
    transform, pixel -> [-1,1]; z = the axis-changed depth.  vertices (B,V,3), cam (B,15).
    cam (B,10) = [fx, fy, cx, cy | 2x3 crop transform] is the HOIG_DexYCB copy's function of the same name (its utils/nmr.py:146-163
    with cam2pixel :38-48): no axis change, x / (z + 1e-8) * f + c, the crop transform, pixel -> [-1,1]; z = the vertex's own depth."""
    bs = cam.shape[0]
    if cam.shape[1] == 10:
        f, c, trans = cam[:, 0:2], cam[:, 2:4], cam[:, 4:].reshape(bs, 2, 3)
        # (the reference multiplies (B,V) by f[:, 0] of shape (B,): it only runs with B = 1, one call per sample, trainer.py:66 --
        # the per-sample scalar is what the batched form below applies)
        x = vertices[:, :, 0] / (vertices[:, :, 2] + 1e-8) * f[:, 0:1] + c[:, 0:1]
        y = vertices[:, :, 1] / (vertices[:, :, 2] + 1e-8) * f[:, 1:2] + c[:, 1:2]
        xy1 = torch.stack([x, y, torch.ones_like(x)], dim=1)
        xy = torch.einsum('ijk,ikm->ijm', trans, xy1).permute(0, 2, 1)
        return torch.cat((xy / 255.0 * 2 - 1, vertices[:, :, 2:3]), dim=2)
    cam_mat, trans = cam[:, 0:9].reshape(bs, 3, 3), cam[:, 9:].reshape(bs, 2, 3)
    change = _const(vertices.device, 'change', [[1., 0., 0.], [0., -1., 0.], [0., 0., -1.]])
    pts = torch.einsum('ijk,mk->ijm', vertices, change)
    proj = torch.einsum('ijk,imk->ijm', pts, cam_mat)
    xy = torch.stack([proj[:, :, 0] / proj[:, :, 2], proj[:, :, 1] / proj[:, :, 2], torch.ones_like(proj[:, :, 0])], dim=2)
    xy = torch.einsum('ijk,imk->ijm', trans, xy).permute(0, 2, 1)
    return torch.cat((xy / 255.0 * 2 - 1, pts[:, :, 2:3]), dim=2)


def project_to_faces(cam, vertices, faces_idx, viewing_angle=30.0):
    """The vertex stage of render_fim_wim (nmr.py:503-511): projection, y flip, nr.look_at, nr.vertices_to_faces -> (B,F,3,3)."""
    v = project(vertices, cam)
    v = torch.stack([v[:, :, 0], -v[:, :, 1], v[:, :, 2]], dim=2)                     # nmr.py:506
    # nr.look_at with eye = (0, 0, -(1/tan(angle) + 1)), at = origin, up = +y (nmr.py:357,508): the rotation is the identity,
    # what remains is the translation by -eye
    eye = _const(v.device, ('eye', viewing_angle), [0.0, 0.0, -(1.0 / math.tan(math.radians(viewing_angle)) + 1.0)])
    v = v - eye
    if faces_idx.dim() == 2:
        faces_idx = faces_idx[None].expand(v.shape[0], -1, -1)
    idx = faces_idx.long()
    return v[torch.arange(v.shape[0], device=v.device)[:, None, None], idx]           # vertices_to_faces


def project_faces_batched(cam, vertices, face_lists, fmax, viewing_angle=30.0, pad_value=-1.0e6):
    """project_to_faces for a batch whose samples hold DIFFERENT objects, as ONE launch (hoig_project_faces): cam (B,15|10), vertices
    (B,V,3), face_lists: B int64 (F_b,3) device tensors over the sample's [hand | object] vertex buffer -> faces (B,fmax,3,3) with the
    rows beyond a sample's own face count at `pad_value` (a point no pixel can see: the rasteriser's empty box)."""
    import ctypes
    if not vertices.is_cuda:
        raise NotImplementedError('hoig_amd.raster runs on the HIP device only (no CPU path)')
    B, V = int(vertices.shape[0]), int(vertices.shape[1])
    cam, vertices = cam.float().contiguous(), vertices.float().contiguous()
    if len(face_lists) != B or cam.shape[0] != B or cam.shape[1] not in (10, 15):
        raise ValueError('one face list per sample; cam (B,15) or (B,10)')
    for fl in face_lists:
        if fl.dtype != torch.int64 or fl.dim() != 2 or fl.shape[1] != 3 or not fl.is_contiguous() or fl.shape[0] > fmax:
            raise ValueError('face lists must be contiguous int64 (F,3) with F <= fmax')
    out = torch.empty(B, int(fmax), 3, 3, dtype=torch.float32, device=vertices.device)
    ptrs = ctypes.cast((ctypes.c_void_p * B)(*[fl.data_ptr() for fl in face_lists]), ctypes.c_void_p)
    nf = ctypes.cast((ctypes.c_int * B)(*[int(fl.shape[0]) for fl in face_lists]), ctypes.c_void_p)
    eye_z = -(1.0 / math.tan(math.radians(viewing_angle)) + 1.0)
    L.call('hoig_project_faces', cam.data_ptr(), int(cam.shape[1]), vertices.data_ptr(), V, ptrs, nf, B, int(fmax), eye_z,
           float(pad_value), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    return out


def render_fim_wim(cam, vertices, faces_idx, image_size=256, viewing_angle=30.0):
    """MANORenderer.render_fim_wim (nmr.py:496-513): returns (faces (B,F,3,3), fim, wim).  faces_idx (F,3) or (B,F,3) int."""
    faces = project_to_faces(cam, vertices, faces_idx, viewing_angle)
    fim, wim = rasterize_fim_wim(faces, image_size)
    return faces, fim, wim
