"""Generates tests/golden/input_prep_256.npz by running the REFERENCE's own HandRecoveryFlow.forward (imported from
/root/reference on CPU, oracle/ref_harness.py::reference_input_prep: only the rasteriser call is replaced by the seeded
synthetic rasteriser outputs of hoig_amd.synthetic.make_raster).  Run in the build container only:
    python tests/golden/make_golden_input_prep.py            (HOIG_HOv3 copy  -> input_prep_256.npz)
    python tests/golden/make_golden_input_prep.py dexycb     (HOIG_DexYCB copy -> input_prep_256_dexycb.npz: 12-channel hand inputs)
Stored per output tensor (inputs are re-derived from the seed): CRC32 of the fp32 bytes (bit-exact pin), float64 sum and
sum of squares, and every 4th pixel of every channel (for diagnostics when a CRC differs).  Nothing of the reference's
source travels."""
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_harness as RH           # noqa: E402
from hoig_amd import synthetic                 # noqa: E402

NAMES = ['input_G_src_bg', 'input_G_tsf_bg', 'input_G_src_obj', 'input_G_tsf_obj', 'input_G_src_hand', 'input_G_ref_hand',
         'T_hand', 'src_crop_mask_bg', 'ref_crop_mask_bg', 'src_crop_mask_hand', 'ref_crop_mask_hand']


def summarise(out, a, key):
    a = np.ascontiguousarray(a, dtype=np.float32)
    out[key + '/crc'] = np.uint32(zlib.crc32(a.tobytes()))
    out[key + '/sum'] = np.float64(a.astype(np.float64).sum())
    out[key + '/sumsq'] = np.float64((a.astype(np.float64) ** 2).sum())
    out[key + '/sub'] = a[:, 1::4, 2::4, :] if key.endswith('T_hand') else a[:, :, 1::4, 2::4]


def main(batch=2, seed=8, copy='hov3'):
    out = dict(batch=batch, seed=seed, copy=copy)
    for bg_both in (False, True):
        r = synthetic.make_raster(batch, seed)
        ref = RH.reference_input_prep(r, bg_both=bg_both, copy=copy)
        for name, v in zip(NAMES, ref):
            if v is not None:
                summarise(out, v.numpy(), 'bg_both%d/%s' % (bg_both, name))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                        'input_prep_256.npz' if copy == 'hov3' else 'input_prep_256_%s.npz' % copy)
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main(copy=sys.argv[1] if len(sys.argv) > 1 else 'hov3')          # one reference copy per process: run once per copy
