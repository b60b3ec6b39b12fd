"""eval.py's output stage for the HIP path.

The reference's evaluation loop (HOIG_HOv3/eval.py:59-79) turns ``get_current_visuals()`` into per-pair PNG crops:

    <out>/source/<srcvid>_<srcframe>_<tsfframe>.png      crop of '16_batch_src_img'
    <out>/imitators/<...>.png                            crop of '15_batch_fake_img'
    <out>/gt/<...>.png                  (opt.sav_gt)     crop of '14_batch_real_img'

where the three visuals are uint8 CHW batch grids (utils/util.py:249-264, ``make_grid(nrow=int(sqrt(B)), padding=0)``)
and crop (r, c) = (i // cols, i % cols) with cols = grid_width // side.  The grids come from the fused
denormalise + tile kernel (``hoig_tensor2im_u8``), one device-to-host copy per grid; encoding and file writes run on a small
thread pool so that the next batch's forward is not held up by zlib (the reference writes synchronously).
"""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np

GRIDS = (('source', '16_batch_src_img'), ('imitators', '15_batch_fake_img'), ('gt', '14_batch_real_img'))


def pair_name(name_a, name_b):
    """eval.py:70-74: '<vid>/<frame>.ext' x '<vid>/<frame>.ext' -> '<srcvid>_<srcframe>_<tsfframe>.png'."""
    src_vid, src_frame = name_a.split('/')
    _, tsf_frame = name_b.split('/')
    return src_vid + '_' + src_frame[:-4] + '_' + tsf_frame[:-4] + '.png'


def crops_of(grid_chw, count, side):
    """The `count` side x side HWC crops of a CHW batch grid, in sample order (eval.py:66-69)."""
    g = np.asarray(grid_chw).transpose(1, 2, 0)
    cols = g.shape[1] // side
    if cols * side != g.shape[1] or g.shape[0] % side:
        raise ValueError('grid %s is not tiled by %d-pixel images' % (g.shape, side))
    if count > cols * (g.shape[0] // side):
        raise ValueError('grid %s holds fewer than %d images' % (g.shape, count))
    return [g[(i // cols) * side:(i // cols + 1) * side, (i % cols) * side:(i % cols + 1) * side] for i in range(count)]


def _save_png(arr, path):
    from PIL import Image                     # utils/util.py:298-301 uses PIL as well
    Image.fromarray(np.ascontiguousarray(arr)).save(path)


class EvalWriter(object):
    """``w = EvalWriter(out_dir, sav_gt=True); w.write(model.get_current_visuals(), batch['nameA'], batch['nameB']); w.close()``"""

    def __init__(self, out_dir, sav_gt=True, side=256, workers=4):
        self.out_dir, self.side = out_dir, side
        self.grids = [g for g in GRIDS if sav_gt or g[0] != 'gt']
        for sub, _ in self.grids:                                  # eval.py:47-53
            os.makedirs(os.path.join(out_dir, sub), exist_ok=True)
        self._pool = ThreadPoolExecutor(max_workers=workers) if workers > 0 else None
        self._pending = []
        self.written = 0

    def write(self, visuals, names_a, names_b):
        if len(names_a) != len(names_b):
            raise ValueError('nameA / nameB length mismatch')
        for sub, key in self.grids:
            for crop, a, b in zip(crops_of(visuals[key], len(names_a), self.side), names_a, names_b):
                path = os.path.join(self.out_dir, sub, pair_name(a, b))
                if self._pool is None:
                    _save_png(crop, path)
                else:
                    self._pending.append(self._pool.submit(_save_png, crop.copy(), path))
                self.written += 1

    def close(self):
        for f in self._pending:
            f.result()                         # re-raise encoder / IO errors
        self._pending = []
        if self._pool is not None:
            self._pool.shutdown()
            self._pool = None
