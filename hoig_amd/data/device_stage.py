"""The device half of the loader: a raw host batch (``HOv3Dataset.__getitem__`` records, stacked by ``collate_raw``
into pinned memory) -> the batch dict the reference's DataLoader yields (HOIG_HOv3/data/hov3_dataset.py:208-213 after default
collation), its tensors on the device.

Per batch and view: one H2D copy of the frames and one of the masks, ``hoig_resize_linear_u8`` (mask -> 640 x 480, :219),
``hoig_warp_affine_u8`` twice (frame -> normalised RGB planes, mask -> last channel / 128; :220-223,209,267) and the posed object
vertices (:246-248) from meshes that were parsed once and live on the device.  Everything is issued on a side stream, one batch ahead
of the training step (``submit`` / ``finish``)."""
import ctypes

import numpy as np
import torch

from . import geometry as G

PATCH = 256
MASK_SIZE = (640, 480)              # cv2.resize(mask, (640, 480)): (width, height)


def collate_raw(items):
    """Stack the per-sample records of a batch key by key (tensors -> one tensor, everything else -> a list)."""
    out = {}
    for side in ('A', 'B'):
        recs = [it[side] for it in items]
        col = {}
        for k in recs[0]:
            vals = [r[k] for r in recs]
            same = torch.is_tensor(vals[0]) and all(v.shape == vals[0].shape for v in vals)
            col[k] = torch.stack(vals) if same else vals
        out[side] = col
    return out


class MeshCache(object):
    """object id -> (n, 3) float64 vertices on the device, read from the dataset's mesh file of that object on first use."""

    def __init__(self, dataset, device):
        self._dataset, self._device, self._verts = dataset, device, {}

    def get(self, obj_id):
        v = self._verts.get(obj_id)
        if v is None:
            path = self._dataset.mesh_path(obj_id)
            v = torch.from_numpy(G.read_obj_vertices(path)).to(self._device)
            if v.shape[0] > self._dataset.max_obj_verts:
                raise ValueError('%s: %d vertices, the batch tensor holds %d' % (path, v.shape[0], self._dataset.max_obj_verts))
            self._verts[obj_id] = v
        return v


class DeviceStage(object):
    def __init__(self, dataset, device=None, prepare=None):
        """prepare (optional): batch dict -> dict of further entries, run on the loader stream right behind the batch's pixel work --
        i.e. ONE BATCH AHEAD of the step that consumes it.  ``CustomDatasetDataLoader`` passes the raw-batch stage of
        ``Trainer.set_input`` here (hand_recovery.HandRecoveryFlow + input_prep.to_prepared) when the options carry the MANO model and
        the object assets: its chain of ~100 small dependent launches then runs beside the previous step instead of in front of the
        next one, and ``set_input`` finds the prepared tensors in the batch (round 6)."""
        from .. import _lib as L
        self._L = L
        self._prepare = prepare
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self._meshes = MeshCache(dataset, self.device)
        self._max_verts = dataset.max_obj_verts
        self._stream = None

    # ---- one view (A or B) of a batch
    def _images(self, col):
        L, dev = self._L, self.device
        frames, masks = col['frame'], col.get('mask')
        if not torch.is_tensor(frames) or not (masks is None or torch.is_tensor(masks)):
            raise ValueError('the frames (and the masks) of a batch must have one size')
        B, Hs, Ws, _ = frames.shape
        st = torch.cuda.current_stream().cuda_stream
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        trans = np.stack([G.patch_transform(b) for b in col['bbox'].numpy()])                   # (B, 2, 3) float32, host
        m_dev = torch.from_numpy(trans.astype(np.float64).reshape(B, 6)).to(dev, non_blocking=True)
        f_dev = frames.to(dev, non_blocking=True)
        image = torch.empty((B, 3, PATCH, PATCH), dtype=torch.float32, device=dev)
        L.call('hoig_warp_affine_u8', p(f_dev), B, Hs, Ws, 3, p(m_dev), PATCH, PATCH, 1, p(image), st)
        if masks is None:                                                   # (the DexYCB copy has no arm mask)
            return image, None, torch.from_numpy(trans)
        k_dev = masks.to(dev, non_blocking=True)
        big = torch.empty((B, MASK_SIZE[1], MASK_SIZE[0], 3), dtype=torch.uint8, device=dev)
        L.call('hoig_resize_linear_u8', p(k_dev), B, masks.shape[1], masks.shape[2], 3, p(big), MASK_SIZE[1], MASK_SIZE[0], st)
        mask = torch.empty((B, 1, PATCH, PATCH), dtype=torch.float32, device=dev)
        L.call('hoig_warp_affine_u8', p(big), B, MASK_SIZE[1], MASK_SIZE[0], 3, p(m_dev), PATCH, PATCH, 2, p(mask), st)
        return image, mask, torch.from_numpy(trans)

    def _object_vertices(self, col):
        """hov3_dataset.py:246-248: zeros((7866, 3), float32); [:n] = v @ Rodrigues(objRot).T + objTrans; ycb_dataset.py:165-169,292-293:
        zeros((8000, 3)); [:n] = (pose_obj @ [v | 1].T)[:3].T -- float64 on the device, rounded to float32 on assignment."""
        from .. import ops
        dev = self.device
        ids = [int(k) for k in col['objName']]
        if 'obj_pose' in col:
            P = col['obj_pose'].to(dev, non_blocking=True)
            R_dev, t_dev = P[:, :3, :3], P[:, :3, 3]
        else:
            rot, f32 = col['obj_rot'].numpy(), col['rot_is_f32']
            R = np.stack([G.rodrigues(r.astype(np.float32) if f else r).astype(np.float64) for r, f in zip(rot, f32)])
            R_dev = torch.from_numpy(R).to(dev, non_blocking=True)
            t_dev = col['obj_trans'].to(dev, non_blocking=True)
        out = torch.zeros((len(ids), self._max_verts, 3), dtype=torch.float32, device=dev)
        for k in sorted(set(ids)):
            rows = [i for i, o in enumerate(ids) if o == k]
            idx = ops.device_index(rows, dev)             # (pinned + non-blocking: no host wait on the loader stream)
            v = self._meshes.get(k)
            now = torch.matmul(v.unsqueeze(0), R_dev[idx].transpose(1, 2)) + t_dev[idx].unsqueeze(1)
            out[idx, :v.shape[0]] = now.float()
        return out

    def _view(self, col):
        image, mask, trans = self._images(col)
        dev = self.device
        mano = {'cam': col['cam'].to(dev, non_blocking=True), 'trans': trans.to(dev, non_blocking=True),
                'pose': col['pose'].to(dev, non_blocking=True), 'shape': col['shape'].to(dev, non_blocking=True)}
        if 'handtrans' in col:
            mano['handtrans'] = col['handtrans'].to(dev, non_blocking=True)
        mano['vertices_obj'] = self._object_vertices(col)
        mano['objName'] = torch.tensor([int(k) for k in col['objName']], dtype=torch.int64)          # (stays on the host: hand_recovery.py)
        return image, mask, mano, list(col['name'])

    # ---- a batch: issue on the side stream, hand over on the caller's
    def submit(self, raw):
        if self._stream is None:
            from .. import ops
            self._stream = ops.new_stream(self.device, 'loader')
        with torch.cuda.stream(self._stream):
            a, b = self._view(raw['A']), self._view(raw['B'])
            batch = {'imageA': a[0], 'maskA': a[1], 'manoA': a[2], 'nameA': a[3],
                     'imageB': b[0], 'maskB': b[1], 'manoB': b[2], 'nameB': b[3]}
            if a[1] is None:                                                # ycb_dataset.py:278-279: no mask keys
                del batch['maskA'], batch['maskB']
            if self._prepare is not None:
                batch.update(self._prepare(batch))
            done = torch.cuda.Event()
            done.record()
        return batch, done, raw                                         # (raw: the pinned source stays alive until the copies ran)

    def finish(self, pending):
        batch, done, _ = pending
        cur = torch.cuda.current_stream()
        cur.wait_event(done)
        for t in self._tensors(batch):
            t.record_stream(cur)
        return batch

    @staticmethod
    def _tensors(batch):
        for v in batch.values():
            if torch.is_tensor(v) and v.is_cuda:
                yield v
            elif isinstance(v, dict):
                for w in v.values():
                    if torch.is_tensor(w) and w.is_cuda:
                        yield w

    def __call__(self, raw):
        return self.finish(self.submit(raw))
