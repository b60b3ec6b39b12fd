"""ddp_mode='bucket' under HOIG_DDP_CHECK for many steps on one GPU (RCCL, world of one rank, exchange forced): every slice that went on the
wire during the backward must still hold after the step what it held at its launch (the SUM over one rank is the identity).
   python tools/ddp_bucket_stress.py [steps]      -> 'late writes: 0 of N early launches' is the pass line"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29577', HOIG_DDP_FORCE='1', HOIG_DDP_CHECK='1',
                  HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
import torch.distributed as dist          # noqa: E402
from common import product_trainer       # noqa: E402
from hoig_amd import ops                  # noqa: E402

torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
ops.set_precision('bf16x3:f16x2')
m = product_trainer('generator_spade_attn', 8, 256, use_ddp=True, ddp_mode='bucket')
late = early = 0
for s in range(steps):
    m.optimize_parameters()
    torch.cuda.synchronize()
    late += m._G.sync.verify_early_slices()
    early += m._G.sync.early_launches
print('late writes: %d of %d early launches over %d steps (batch 8, 256x256)' % (late, early, steps))
dist.barrier()
dist.destroy_process_group()
