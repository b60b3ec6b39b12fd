"""``HOv3Dataset``: the host half of the HO3D-v3 loader (the reference class is HOIG_HOv3/data/hov3_dataset.py:164-270: same
constructor arguments, ``opt`` fields, directory layout, pair selection and length).  A worker DECODES and reads the annotation,
nothing else: ``__getitem__`` returns, per view, the 8-bit frame and mask as ``cv2.imread`` delivers them (BGR, three channels), the
sequence's bounding box and the annotation's MANO / object values; ``hoig_amd.data.device_stage.DeviceStage`` turns a batch of such
records into the reference's batch dict on the device (mask resize, affine crop, scaling, normalisation, posed object vertices:
:215-250).  Workers never touch the GPU."""
import os

import numpy as np
import torch

from .dataset_base import DatasetBase, PairIndex, read_pickle

OBJNAMES = ['003_cracker_box', '004_sugar_box', '006_mustard_bottle', '010_potted_meat_can', '011_banana', '021_bleach_cleanser',
            '025_mug', '035_power_drill', '037_scissors']                      # object id = position in this list (hov3_dataset.py:13)
MAX_OBJ_VERTS = 7866                                                           # rows of the batch's vertices_obj tensor (:246)


def imread_bgr(path):
    """What ``cv2.imread(path)`` (IMREAD_COLOR) returns: (H, W, 3) uint8 in BGR order.  Decoded with Pillow (cv2 is not a dependency of
    this package).  Like OpenCV's decoders: the EXIF orientation is applied, 16-bit samples keep their high byte (libpng's strip-16),
    alpha is dropped, grey and palette files become three channels.  JPEG goes through libjpeg in both libraries (integer DCT, fancy
    upsampling); that the two builds decode a given file to the same bytes is NOT verified here (no cv2 in this image) --
    ``tests/test_data_cpu.py::test_imread_matches_cv2_when_available`` compares them wherever cv2 can be imported."""
    from PIL import Image, ImageOps
    with Image.open(path) as im:
        im = ImageOps.exif_transpose(im)
        if im.mode in ('I;16', 'I;16B', 'I;16L', 'I'):
            grey = (np.asarray(im).astype(np.uint32) >> 8).clip(0, 255).astype(np.uint8)
            return np.ascontiguousarray(np.repeat(grey[:, :, None], 3, axis=2))
        rgb = np.asarray(im.convert('RGB'))
    return np.ascontiguousarray(rgb[:, :, ::-1])


class HOv3Dataset(DatasetBase):
    _name = 'HOv3Dataset'
    max_obj_verts = MAX_OBJ_VERTS
    objnames = OBJNAMES

    def __init__(self, opt, is_for_train=True):
        super(HOv3Dataset, self).__init__(opt, is_for_train)
        params, self._pics = self._subdir('params_dir', 'param_dir'), self._subdir('images_dir', 'pic_dir')
        # the reference opens 'assets/obj/<name>/<name>.obj' relative to its working directory (:239); opt.obj_dir overrides that
        self._meshes = getattr(opt, 'obj_dir', os.path.join('assets', 'obj'))
        self._bbox_of_video = read_pickle(os.path.join(params, 'HOv3-CR_bbx.pkl'), 'bounding boxes')
        listing = 'HOv3-CR_%s_new.pkl' % ('train' if is_for_train else 'test')
        self._index = PairIndex(read_pickle(os.path.join(params, listing), 'video list'), opt.pairs_dir)

    def mesh_path(self, obj_id):
        return os.path.join(self._meshes, OBJNAMES[obj_id], OBJNAMES[obj_id] + '.obj')

    def _sequence_dir(self, seq, frame):
        """A sequence lives under train/ or test/ whatever split the listing came from: the frame file decides (:216-219)."""
        for split in ('train', 'test'):
            if split == 'test' or os.path.exists(os.path.join(self._pics, split, seq, 'rgb', frame)):
                return os.path.join(self._pics, split, seq)

    def _get_raw_sample(self, video, frame):
        seq, stem = video.split('_')[0], frame.split('.')[0]
        base = self._sequence_dir(seq, frame)
        meta = os.path.join(base, 'meta', stem + '.pkl')
        assert os.path.exists(meta), 'File does not exists: %s' % meta
        anno = read_pickle(meta, 'annotation')
        f32 = lambda key: torch.from_numpy(np.asarray(anno[key]).astype(np.float32))
        f64 = lambda key: torch.from_numpy(np.asarray(anno[key], dtype=np.float64).reshape(3))
        return {
            'frame': torch.from_numpy(imread_bgr(os.path.join(base, 'rgb', frame))),
            'mask': torch.from_numpy(imread_bgr(os.path.join(base, 'mask', '%05d.png' % int(stem)))),
            'bbox': torch.as_tensor(np.asarray(self._bbox_of_video[video], dtype=np.float64)),
            'cam': f32('camMat'), 'pose': f32('handPose'), 'shape': f32('handBeta'), 'handtrans': f32('handTrans'),
            # cv2.Rodrigues answers in its argument's precision: the device stage needs to know which one the file holds (:247)
            'obj_rot': f64('objRot'), 'obj_trans': f64('objTrans'), 'rot_is_f32': bool(np.asarray(anno['objRot']).dtype == np.float32),
            'objName': OBJNAMES.index(anno['objName']),
            'name': os.path.join(video, frame),
        }
