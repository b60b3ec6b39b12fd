"""``YCBDataset``: the host half of the HOIG_DexYCB copy's loader (reference class: HOIG_DexYCB/data/ycb_dataset.py:230-316, its
``read_annotation`` :132-174; same constructor arguments, files and pair selection).  As in hov3_dataset.py a worker only decodes and
reads the label file; the device stage finishes the batch.  Differences of this copy that the records carry: no mask; the bounding
box file holds corners (x0, y0, x1, y1), the crop wants (x, y, w, h); the camera is four intrinsics of the VIDEO; the grasped object
is posed by a 3x4 matrix of the frame's label file, into 8000 vertex rows."""
import os

import numpy as np
import torch

from .dataset_base import DatasetBase, PairIndex, read_pickle
from .hov3_dataset import imread_bgr

OBJNAMES = ['002_master_chef_can', '003_cracker_box', '004_sugar_box', '005_tomato_soup_can', '006_mustard_bottle',
            '007_tuna_fish_can', '008_pudding_box', '009_gelatin_box', '010_potted_meat_can', '011_banana',
            '019_pitcher_base', '021_bleach_cleanser', '024_bowl', '025_mug', '035_power_drill', '036_wood_block',
            '037_scissors', '040_large_marker', '051_large_clamp', '052_extra_large_clamp', '061_foam_brick']      # YCB class id - 1
MAX_OBJ_VERTS = 8000                                                                                                # ycb_dataset.py:292


class YCBDataset(DatasetBase):
    _name = 'YCBDataset'
    max_obj_verts = MAX_OBJ_VERTS
    objnames = OBJNAMES

    def __init__(self, opt, is_for_train=True):
        super(YCBDataset, self).__init__(opt, is_for_train)
        params, self._pics = self._subdir('params_dir', 'param_dir'), self._subdir('images_dir', 'pic_dir')
        self._corners_of_video = read_pickle(os.path.join(params, 'DexYCB-bbx.pkl'), 'bounding boxes')
        self._info_of_video = read_pickle(os.path.join(params, 'valid_video_info.pkl'), 'video info')
        listing = 'DexYCB_%s.pkl' % ('train' if is_for_train else 'test')
        self._index = PairIndex(read_pickle(os.path.join(params, listing), 'video list'), opt.pairs_dir)

    def mesh_path(self, obj_id):
        return os.path.join(self._opt.data_dir, 'models', OBJNAMES[obj_id], 'textured_pre.obj')                     # :150

    def _get_raw_sample(self, video, frame):
        frame = int(frame)
        info = self._info_of_video[video]
        x0, y0, x1, y1 = self._corners_of_video[video]
        # the label file is looked up under <data_dir>/images whatever opt.images_dir says (:141)
        label = np.load(os.path.join(self._opt.data_dir, 'images', video, 'labels_%06d.npz' % frame))
        if not np.any(label['pose_m']):
            # a frame without a hand: the reference's read_annotation never binds its return values and dies there (:163-174)
            raise UnboundLocalError('%s frame %d: no MANO pose in the label file (the reference fails on this frame too)' % (video, frame))
        # The reference keeps only the objects whose 3x4 pose is not all zero and then indexes THAT list with ycb_grasp_ind (:157-166):
        # an unposed object in front of the grasped one shifts the pick.  Reproduced, not repaired.
        posed = [p for p in label['pose_y'] if np.any(p)]
        grasp = info['ycb_grasp_ind']
        pose = np.eye(4, dtype=np.float64)
        pose[:3] = posed[grasp]                                                  # float32 values, widened: the device works in float64
        k = info['intrinsics']
        return {
            'frame': torch.from_numpy(imread_bgr(os.path.join(self._pics, video, 'color_%06d.jpg' % frame))),
            'bbox': torch.tensor([x0, y0, x1 - x0, y1 - y0], dtype=torch.float64),
            'cam': torch.tensor([k['fx'], k['fy'], k['ppx'], k['ppy']], dtype=torch.float32),
            'pose': torch.from_numpy(label['pose_m'])[0].float(),
            'shape': torch.tensor(info['mano_betas'], dtype=torch.float32),
            'obj_pose': torch.from_numpy(pose),
            'objName': info['ycb_ids'][grasp] - 1,                               # class ids count from 1 (:18-40)
            'name': os.path.join(video, str(frame)),
        }
