"""Generator-only inference (Trainer.forward under no_grad), B=32 at 256x256: for rocprofv3 kernel stats."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import bench
from common import opt_namespace
from hoig_amd import ops
ops.set_precision(os.environ.get('HOIG_PRECISION', 'bf16x3:f16x2'))
print(bench.gen_forward_latency(opt_namespace(), int(os.environ.get('B', 32)), 256, iters=3))
