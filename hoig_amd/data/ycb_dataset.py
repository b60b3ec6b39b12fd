"""``YCBDataset`` of the HOIG_DexYCB copy (HOIG_DexYCB/data/ycb_dataset.py:230-316, ``read_annotation`` :132-174): the same
constructor arguments, files and pair selection; as in hov3_dataset.py ``__getitem__`` stops after the decode and the device stage
finishes the batch (no mask in this copy; the object's vertices are posed by the 3x4 matrix of the label file, into 8000 rows)."""
import os
import pickle

import numpy as np
import torch

from .dataset_base import DatasetBase
from .hov3_dataset import imread_bgr

OBJNAMES = ['002_master_chef_can', '003_cracker_box', '004_sugar_box', '005_tomato_soup_can', '006_mustard_bottle',
            '007_tuna_fish_can', '008_pudding_box', '009_gelatin_box', '010_potted_meat_can', '011_banana',
            '019_pitcher_base', '021_bleach_cleanser', '024_bowl', '025_mug', '035_power_drill', '036_wood_block',
            '037_scissors', '040_large_marker', '051_large_clamp', '052_extra_large_clamp', '061_foam_brick']      # ycb_dataset.py:13-16
_YCB_CLASSES = dict(enumerate(OBJNAMES, start=1))                                                                   # :18-40
MAX_OBJ_VERTS = 8000                                                                                                # :292


class YCBDataset(DatasetBase):
    max_obj_verts = MAX_OBJ_VERTS
    objnames = OBJNAMES

    def __init__(self, opt, is_for_train=True):
        super(YCBDataset, self).__init__(opt, is_for_train)
        self._name = 'YCBDataset'
        self.data_dir = opt.data_dir
        self.param_dir = os.path.join(opt.data_dir, opt.params_dir)
        self.pic_dir = os.path.join(opt.data_dir, opt.images_dir)
        self.data_split = 'train' if is_for_train else 'test'
        self.pairs_dir = opt.pairs_dir
        if not os.path.exists(self.param_dir):
            raise ValueError("param_dir: %s not exist" % self.param_dir)
        if not os.path.exists(self.pic_dir):
            raise ValueError("pic_dir: %s not exist" % self.pic_dir)
        with open(os.path.join(self.param_dir, 'DexYCB-bbx.pkl'), 'rb') as f:
            self.bbx_params = pickle.load(f)
        with open(os.path.join(self.param_dir, 'valid_video_info.pkl'), 'rb') as f:
            self.cam_params = pickle.load(f)
        with open(os.path.join(self.param_dir, 'DexYCB_train.pkl' if is_for_train else 'DexYCB_test.pkl'), 'rb') as f:
            self._vids_dict = pickle.load(f)
        if self.pairs_dir and os.path.exists(self.pairs_dir):
            with open(self.pairs_dir, "rb") as f:
                self._pairs_list = pickle.load(f)
        else:
            self._pairs_list = None
        self._vids_list = list(self._vids_dict)
        self._num_videos = len(self._vids_list) if self._pairs_list is None else len(self._pairs_list)

    def mesh_path(self, obj_id):
        return os.path.join(self.data_dir, 'models', OBJNAMES[obj_id], 'textured_pre.obj')                          # :150

    def __getitem__(self, index):                                                                                    # :260-279
        if self._pairs_list is None:
            vid_id = self._vids_list[index % self._num_videos]
            frame_list = self._vids_dict[vid_id]
            vid_a, vid_b = vid_id, vid_id
            frame_a, frame_b = np.random.choice(frame_list, size=2, replace=False)
        else:
            path_a, path_b = self._pairs_list[index % self._num_videos]
            vid_a, frame_a = os.path.join(*path_a.split('/')[:-1]), int(path_a.split('/')[-1])
            vid_b, frame_b = os.path.join(*path_b.split('/')[:-1]), int(path_b.split('/')[-1])
        return {'A': self._get_raw_sample(vid_a, frame_a), 'B': self._get_raw_sample(vid_b, frame_b)}

    def _get_raw_sample(self, vid_id, frame_id):                                                                     # :281-305, :132-174
        frame_id = int(frame_id)
        frame = imread_bgr(os.path.join(self.pic_dir, vid_id, "color_{:06d}.jpg".format(frame_id)))
        bbox = self.bbx_params[vid_id]
        bbox = [bbox[0], bbox[1], bbox[2] - bbox[0], bbox[3] - bbox[1]]
        sample = self.cam_params[vid_id]
        intr = sample['intrinsics']
        cam = torch.Tensor([[intr['fx'], intr['fy'], intr['ppx'], intr['ppy']]]).float()
        betas = torch.tensor(sample['mano_betas'], dtype=torch.float32)
        grasp_id = sample['ycb_grasp_ind']
        grasp_name = _YCB_CLASSES[sample['ycb_ids'][grasp_id]]
        label = np.load(os.path.join(self.data_dir, 'images', vid_id, "labels_{:06d}.npz".format(frame_id)))
        pose_y, pose_m = label['pose_y'], label['pose_m']
        pose_obj_list = [np.vstack((pose_y[o], np.array([[0, 0, 0, 1]], dtype=np.float32)))
                         for o in range(len(pose_y)) if not np.all(pose_y[o] == 0.0)]
        if np.all(pose_m == 0.0):
            # the reference's read_annotation leaves obj_mesh / pose unbound for a frame without a hand and fails in its return (:163-174)
            raise UnboundLocalError('%s frame %d: no MANO pose in the label file (the reference fails on this frame too)' % (vid_id, frame_id))
        return {'frame': torch.from_numpy(frame), 'bbox': torch.as_tensor(np.asarray(bbox, dtype=np.float64)),
                'cam': cam[0].float(), 'pose': torch.from_numpy(pose_m)[0].float(), 'shape': betas.float(),
                'obj_pose': torch.from_numpy(np.asarray(pose_obj_list[grasp_id], dtype=np.float64)),       # (4, 4): float32 values, widened
                'objName': OBJNAMES.index(grasp_name), 'name': os.path.join(vid_id, str(frame_id))}

    def __len__(self):
        return self._num_videos * self._opt.num_repeats
