"""Throughput of the loader's device stage: a raw batch of decoded 640x480 frames + 320x240 masks in pinned memory
-> the reference's batch dict on the device (H2D, resize, two warps, object vertices), against the CPU oracle's restatement of the
reference's per-sample host code (one core, as one DataLoader worker runs it).
usage: python tools/bench_loader.py [batch] [iters]"""
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import data_fixture as FX                                  # noqa: E402
from hoig_amd.data import CustomDatasetDataLoader          # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
with tempfile.TemporaryDirectory() as root:
    opt = FX.build(root, seed=1, frames=max(4, B), n_obj_verts={2: 7866, 5: 3000})
    opt.batch_size = B
    FX.write_pairs(opt, [('ABF1_0/%04d.png' % (k % 4), 'MC2_0/%04d.png' % ((k + 1) % 4)) for k in range(B)])
    loader = CustomDatasetDataLoader(opt, is_for_train=True)
    raw = next(iter(loader.load_raw_data()))
    loader.load_data()
    stage = loader._stage
    for _ in range(3):
        stage(raw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        stage(raw)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    bytes_in = sum(raw[s]['frame'].numel() + raw[s]['mask'].numel() for s in 'AB')
    print('device stage: batch %d (2 views)  %.3f ms per batch = %.0f samples/s   (%.1f MB of 8-bit pixels over PCIe per batch: %.1f GB/s)'
          % (B, dt * 1e3, 2 * B / dt, bytes_in / 1e6, bytes_in / dt / 1e9))
    t0 = time.perf_counter()
    names = [('ABF1_0/%04d.png' % (k % 4), 'MC2_0/%04d.png' % ((k + 1) % 4)) for k in range(4)]
    FX.oracle_batch(opt, [n[0] for n in names], [n[1] for n in names])
    dt_cpu = (time.perf_counter() - t0) / 8
    print('CPU oracle (decode + resize + warps + vertices, one core, numpy): %.1f ms per sample = %.1f samples/s' % (dt_cpu * 1e3, 1 / dt_cpu))
    t0 = time.perf_counter()
    n = 0
    for _ in range(3):
        for b in loader.load_data():
            n += 2 * len(b['nameA'])
    torch.cuda.synchronize()
    print('end to end, 0 workers (PIL decode + pickles on the main thread): %.1f samples/s' % (n / (time.perf_counter() - t0)))
