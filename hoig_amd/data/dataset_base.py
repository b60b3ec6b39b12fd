"""What the two dataset classes of this package share: the ``name`` the factory prints (HOIG_HOv3/data/dataset_base.py:19-21 is
the only part of the reference's base class anything on this path calls) and the pair index -- which two frames make sample
``i`` (hov3_dataset.py:198-207, ycb_dataset.py:260-270)."""
import os
import pickle

import numpy as np
import torch.utils.data


def read_pickle(path, what='file'):
    if not os.path.isfile(path):
        raise FileNotFoundError('%s not found: %s' % (what, path))
    with open(path, 'rb') as f:
        try:
            return pickle.load(f, encoding='latin1')       # (HO3D's annotation files were written by Python 2)
        except UnicodeDecodeError:
            f.seek(0)
            return pickle.load(f)


class PairIndex(object):
    """Sample i -> ((video, frame) of view A, (video, frame) of view B).

    With a pairs file (a pickled list of ('<video>/<frame>', '<video>/<frame>') strings: evaluation) the list is walked in order;
    without one (training) sample i belongs to video ``i mod n`` and its two frames are drawn without replacement from that
    video's frame list with numpy's global generator, as the reference draws them -- a seeded run visits the same pairs."""

    def __init__(self, frames_of_video, pairs_file=None):
        self._frames = frames_of_video
        self._videos = list(frames_of_video)
        self._pairs = read_pickle(pairs_file, 'pairs file') if pairs_file and os.path.exists(pairs_file) else None

    def __len__(self):
        return len(self._videos) if self._pairs is None else len(self._pairs)

    def fixed(self):
        return self._pairs is not None

    def __getitem__(self, i):
        i %= len(self)
        if self._pairs is not None:
            a, b = self._pairs[i]
            return a.rsplit('/', 1), b.rsplit('/', 1)
        video = self._videos[i]
        fa, fb = np.random.choice(self._frames[video], size=2, replace=False)
        return (video, fa), (video, fb)


class DatasetBase(torch.utils.data.Dataset):
    """Subclasses set ``_name`` and ``_index`` (a ``PairIndex``) and implement ``_get_raw_sample(video, frame)``."""
    _name = 'BaseDataset'

    def __init__(self, opt, is_for_train):
        super(DatasetBase, self).__init__()
        self._opt, self._is_for_train = opt, is_for_train
        self._index = None

    @property
    def name(self):
        return self._name

    def _subdir(self, field, label):
        """``opt.<field>`` under ``opt.data_dir``; a missing one is a ValueError that names it (as in the reference)."""
        path = os.path.join(self._opt.data_dir, getattr(self._opt, field))
        if not os.path.exists(path):
            raise ValueError('%s: %s not exist' % (label, path))
        return path

    def __len__(self):
        return len(self._index) * self._opt.num_repeats

    def __getitem__(self, i):
        (va, fa), (vb, fb) = self._index[i]
        return {'A': self._get_raw_sample(va, fa), 'B': self._get_raw_sample(vb, fb)}

    def _get_raw_sample(self, video, frame):
        raise NotImplementedError
