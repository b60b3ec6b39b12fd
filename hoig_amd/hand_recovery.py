"""``HandRecoveryFlow`` -- the step in front of the path, wired together on the device.

The reference's ``HandRecoveryFlow.forward(src_img, ref_img, src_mano, ref_mano)`` (HOIG_HOv3/models/trainer.py:46-145) turns a
raw dataloader batch into the generator's inputs: the MANO hand layer (``HandModelRecovery.get_details``, trainer.py:48-49), per
sample and view the renderer's projection + rasteriser (``MANORenderer.render_fim_wim``, trainer.py:66,74; utils/nmr.py:496-513),
and the tensor stage after it (trainer.py:67-145).  Here the same three stages run as the three device stages of this package --
``hoig_amd.mano`` (one launch per view), ``hoig_amd.raster`` (ONE rasteriser launch pair per view for the whole batch: the
samples' projected faces are padded with faces far outside the image, which no pixel can hit) and ``hoig_amd.input_prep`` -- with no
host synchronisation: the object id of a sample arrives in the batch as a CPU tensor (``manoA['objName']``), everything else
stays on the device.

The licensed / unshipped assets the reference builds its tables from (MANO_RIGHT.pkl, the YCB meshes and UV atlases:
utils/nmr.py:283-406) are the CALLER's: `mano` = a ``hoig_amd.mano.ManoModel`` or the path of MANO_RIGHT.pkl (``opt.mano_model``),
`objects` = {object id (index into OBJNAMES, trainer.py:11): {'faces': (F,3) integer face list over the [hand | object] vertex
buffer, 'map_fn', 'sem_full', 'fim_uv', 'wim_uv', 'faces_uv_coord', 'obj_tex_img'}} -- MANORenderer's per-object buffers
(``opt.object_assets``).  Returns the reference's 12-tuple (NCHW)."""
import torch

from . import input_prep as IP
from . import ops
from . import raster
from .mano import HandModelRecovery, ManoModel
from .options import is_dexycb

FAR_AWAY = -1.0e6        # padding faces: all three vertices at one point far left of the image -> the rasteriser's empty box


class HandRecoveryFlow(object):
    def __init__(self, opt, mano=None, objects=None, device=None):
        self._name = 'HandRecoveryFlow'
        self._opt = opt
        self.device = device if device is not None else torch.device('cuda', torch.cuda.current_device())
        self._dexycb = is_dexycb(opt)
        mano = mano if mano is not None else getattr(opt, 'mano_model', None)
        objects = objects if objects is not None else getattr(opt, 'object_assets', None)
        if not isinstance(mano, ManoModel) and not mano or not objects:
            raise ValueError('HandRecoveryFlow needs the MANO model (opt.mano_model: a hoig_amd.mano.ManoModel or the path of '
                             'MANO_RIGHT.pkl) and the per-object renderer buffers (opt.object_assets): neither ships with the '
                             'reference (.gitignore:3, utils/nmr.py:283-406)')
        self._hmr = HandModelRecovery(mano, variant='dexycb' if self._dexycb else 'hov3', device=self.device)
        self._objects = {}
        for k, ob in objects.items():
            faces = torch.as_tensor(ob['faces']).to(device=self.device, dtype=torch.int64).contiguous()
            tables = ob['tables'] if isinstance(ob.get('tables'), IP.ObjectTables) else IP.ObjectTables(ob, self.device)
            if faces.dim() != 2 or faces.shape[1] != 3 or faces.shape[0] != tables.n_faces:
                raise ValueError('object %r: faces must be (F,3) with F = the face count of its tables (%d)' % (k, tables.n_faces))
            self._objects[int(k)] = dict(faces=faces, tables=tables, length=int(faces.max()) + 1)      # trainer.py:65

    def _render(self, info, obj_ids, fmax):
        """render_fim_wim (nmr.py:496-513) for every sample of one view: ONE launch projects every sample's vertices through its own
        object's face list (round 6: hoig_project_faces; ~20 batched torch ops per object group before), one launch pair rasterises
        the whole batch."""
        B = len(obj_ids)
        if B <= IP.MAX_BATCH:
            need = max(self._objects[k]['length'] for k in obj_ids)
            if info['verts'].shape[1] < need:
                raise ValueError('the batch holds %d vertices per sample; its objects\' face lists address %d' % (info['verts'].shape[1], need))
            faces = raster.project_faces_batched(info['cam'], info['verts'], [self._objects[k]['faces'] for k in obj_ids], fmax,
                                                 pad_value=FAR_AWAY)
        else:
            faces = torch.full((B, fmax, 3, 3), FAR_AWAY, dtype=torch.float32, device=self.device)
            for k in sorted(set(obj_ids)):
                rows = [i for i, o in enumerate(obj_ids) if o == k]
                ob = self._objects[k]
                idx = ops.device_index(rows, self.device)    # (no host wait: pinned + non-blocking, cached)
                f = raster.project_to_faces(info['cam'][idx], info['verts'][idx, :ob['length']], ob['faces'])
                faces[idx, :f.shape[1]] = f
        fim, wim = raster.rasterize_fim_wim(faces, raster_size(self._opt))
        return faces, fim, wim

    def forward(self, src_img, ref_img, src_mano, ref_mano):
        with torch.no_grad():
            src_info = self._hmr.get_details(src_mano)                   # trainer.py:48-49
            ref_info = self._hmr.get_details(ref_mano)
            obj_ids = [int(v) for v in torch.as_tensor(src_info['objName']).reshape(-1).tolist()]      # a CPU tensor of the batch
            for k in obj_ids:
                if k not in self._objects:
                    raise KeyError('no renderer buffers for object id %d (opt.object_assets)' % k)
            fmax = max(self._objects[k]['tables'].n_faces for k in obj_ids)
            src_faces, src_fim, src_wim = self._render(src_info, obj_ids, fmax)
            _, ref_fim, ref_wim = self._render(ref_info, obj_ids, fmax)
            tabs = [self._objects[k]['tables'] for k in obj_ids]
            # the rasteriser's indices are in range by construction (padding faces are never hit): no range check, no host wait
            return IP.prepare_inputs(src_img.to(self.device), ref_img.to(self.device), src_faces, src_fim, src_wim, ref_fim,
                                     ref_wim, tabs, bg_both=bool(getattr(self._opt, 'bg_both', False)), dexycb=self._dexycb,
                                     validate=False)

    __call__ = forward


def raster_size(opt):
    return int(getattr(opt, 'image_size', 256))
