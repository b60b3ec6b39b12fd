"""``HOv3Dataset`` (HOIG_HOv3/data/hov3_dataset.py:164-270): the same constructor arguments, directory layout, pair selection and
length; ``__getitem__`` stops after the DECODE -- it returns the two samples' 8-bit frames and masks (as ``cv2.imread`` would: BGR,
three channels) with their annotations, and ``hoig_amd.data.device_stage.DeviceStage`` turns a batch of them into the reference's
batch dict on the device (resize + warp + scaling + normalisation, posed object vertices).  Workers never touch the GPU."""
import os
import pickle

import numpy as np
import torch

from .dataset_base import DatasetBase

OBJNAMES = ['003_cracker_box', '004_sugar_box', '006_mustard_bottle', '010_potted_meat_can', '011_banana', '021_bleach_cleanser',
            '025_mug', '035_power_drill', '037_scissors']                      # hov3_dataset.py:13
MAX_OBJ_VERTS = 7866                                                           # :246


def load_pickle_data(f_name):                                                  # :97-107
    if not os.path.exists(f_name):
        raise Exception('Unable to find annotations picle file at %s. Aborting.' % (f_name))
    with open(f_name, 'rb') as f:
        try:
            return pickle.load(f, encoding='latin1')
        except Exception:
            return pickle.load(f)


def read_annotation(base_dir, seq_name, file_id, split):                      # :109-113
    meta_filename = os.path.join(base_dir, split, seq_name, 'meta', file_id + '.pkl')
    assert os.path.exists(meta_filename), 'File does not exists: %s' % meta_filename
    return load_pickle_data(meta_filename)


def imread_bgr(path):
    """cv2.imread(path) (IMREAD_COLOR): (H, W, 3) uint8, BGR.  Decoded with Pillow (cv2 is not a dependency of this package): the same
    libjpeg defaults (slow integer DCT, fancy upsampling) for JPEG, lossless PNG either way; grey / palette files become three channels."""
    from PIL import Image
    with Image.open(path) as im:
        rgb = np.asarray(im.convert('RGB'))
    return np.ascontiguousarray(rgb[:, :, ::-1])


class HOv3Dataset(DatasetBase):
    max_obj_verts = MAX_OBJ_VERTS
    objnames = OBJNAMES

    def mesh_path(self, obj_id):
        return os.path.join(self.obj_dir, OBJNAMES[obj_id], OBJNAMES[obj_id] + '.obj')                  # :239

    def __init__(self, opt, is_for_train=True):
        super(HOv3Dataset, self).__init__(opt, is_for_train)
        self._name = 'HOv3Dataset'
        self.data_dir = opt.data_dir
        self.param_dir = os.path.join(opt.data_dir, opt.params_dir)
        self.pic_dir = os.path.join(opt.data_dir, opt.images_dir)
        self.obj_dir = getattr(opt, 'obj_dir', os.path.join('assets', 'obj'))      # (the reference's path is relative to its cwd, :239)
        self.data_split = 'train' if is_for_train else 'test'
        self.pairs_dir = opt.pairs_dir
        if not os.path.exists(self.param_dir):
            raise ValueError("param_dir: %s not exist" % self.param_dir)
        if not os.path.exists(self.pic_dir):
            raise ValueError("pic_dir: %s not exist" % self.pic_dir)
        with open(os.path.join(self.param_dir, 'HOv3-CR_bbx.pkl'), 'rb') as f:
            self.bbx_params = pickle.load(f)
        _vid_list_dir = os.path.join(self.param_dir, 'HOv3-CR_train_new.pkl' if is_for_train else 'HOv3-CR_test_new.pkl')
        with open(_vid_list_dir, 'rb') as f:
            self._vids_dict = pickle.load(f)
        if self.pairs_dir and os.path.exists(self.pairs_dir):
            with open(self.pairs_dir, "rb") as f:
                self._pairs_list = pickle.load(f)
        else:
            self._pairs_list = None
        self._vids_list = list(self._vids_dict)
        self._num_videos = len(self._vids_list) if self._pairs_list is None else len(self._pairs_list)

    def __getitem__(self, index):                                              # :198-213
        if self._pairs_list is None:
            vid_id = self._vids_list[index % self._num_videos]
            frame_list = self._vids_dict[vid_id]
            vid_a, vid_b = vid_id, vid_id
            frame_a, frame_b = np.random.choice(frame_list, size=2, replace=False)
        else:
            path_a, path_b = self._pairs_list[index % self._num_videos]
            vid_a, frame_a = path_a.split('/')
            vid_b, frame_b = path_b.split('/')
        return {'A': self._get_raw_sample(vid_a, frame_a), 'B': self._get_raw_sample(vid_b, frame_b)}

    def _get_raw_sample(self, vid_id, frame_id):                               # :215-257, up to and including the decode
        seq = vid_id.split('_')[0]
        split = 'train' if os.path.exists(os.path.join(self.pic_dir, 'train', seq, 'rgb', frame_id)) else 'test'
        frame = imread_bgr(os.path.join(self.pic_dir, split, seq, 'rgb', frame_id))
        mask = imread_bgr(os.path.join(self.pic_dir, split, seq, 'mask', '%05d.png' % int(frame_id.split('.')[0])))
        anno = read_annotation(self.pic_dir, seq, frame_id.split('.')[0], split)
        return {'frame': torch.from_numpy(frame), 'mask': torch.from_numpy(mask),
                'bbox': torch.as_tensor(np.asarray(self.bbx_params[vid_id], dtype=np.float64)),
                'cam': torch.from_numpy(np.asarray(anno['camMat']).astype(np.float32)),
                'pose': torch.from_numpy(np.asarray(anno['handPose']).astype(np.float32)),
                'shape': torch.from_numpy(np.asarray(anno['handBeta']).astype(np.float32)),
                'handtrans': torch.from_numpy(np.asarray(anno['handTrans']).astype(np.float32)),
                'obj_rot': torch.from_numpy(np.asarray(anno['objRot'], dtype=np.float64).reshape(3)),
                'obj_trans': torch.from_numpy(np.asarray(anno['objTrans'], dtype=np.float64).reshape(3)),
                'rot_is_f32': bool(np.asarray(anno['objRot']).dtype == np.float32),
                'objName': OBJNAMES.index(anno['objName']),
                'name': os.path.join(vid_id, frame_id)}

    def __len__(self):
        return self._num_videos * self._opt.num_repeats
