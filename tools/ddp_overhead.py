"""What the data-parallel machinery costs on ONE GPU (no 8-GPU node is available to this build): the step in four forms --
plain captured step; RCCL group of one rank with the exchange forced on (two graphs + eager collectives), fp32 and bf16 payload;
the same eagerly -- and the HOST time per step (how long the Python thread needs to issue a step; under DDP every rank has one
core's worth of it).  Launch under `taskset -c 0` to see the one-core case.   python tools/ddp_overhead.py [steps]"""
import os, sys, time, contextlib, io
import torch
import torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hoig_amd import ops, synthetic
from hoig_amd.models import ModelsFactory
from hoig_amd.options import opt_namespace

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ops.set_precision(os.environ.get('HOIG_PRECISION', 'bf16x3:f16x2'))
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29533')
os.environ['HOIG_DDP_FORCE'] = '1'
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))


def run(tag, use_ddp, graph, payload=None, mode=None):
    opt = opt_namespace(hip_graph=graph, ddp_payload=payload, ddp_mode=mode)
    torch.manual_seed(8)
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        m = ModelsFactory.get_by_name('trainer', opt, use_ddp=use_ddp)
    m.set_train()
    m.set_input(synthetic.make_inputs(8, 256, seed=8))
    for _ in range(5):
        m.optimize_parameters()
    torch.cuda.synchronize()
    host = 0.0
    t0 = time.perf_counter()
    for _ in range(steps):
        h0 = time.perf_counter()
        m.optimize_parameters()
        host += time.perf_counter() - h0
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    early = (', %d of %d slices on the wire before the backward ended' % (m._G.sync.early_launches, len(m._G.sync.slices))
             if use_ddp and mode == 'bucket' else '')
    print('%-50s %7.2f ms/step   host issue %6.2f ms/step   cores available %d%s' % (tag, dt * 1e3, host / steps * 1e3,
                                                                                   len(os.sched_getaffinity(0)), early), flush=True)
    del m
    torch.cuda.empty_cache()


run('plain, captured step (one graph)', False, True)
run('plain, eager', False, False)
run('RCCL world 1 forced, captured (2 graphs), fp32', True, True, 'f32')
run('RCCL world 1 forced, captured (2 graphs), bf16', True, True, 'bf16')
run('RCCL world 1 forced, eager, fp32, mode after', True, False, 'f32', 'after')
run('RCCL world 1 forced, eager, fp32, mode bucket', True, False, 'f32', 'bucket')
run('plain, eager (again)', False, False)
run('RCCL world 1 forced, eager, fp32, mode after', True, False, 'f32', 'after')
run('RCCL world 1 forced, eager, fp32, mode bucket', True, False, 'f32', 'bucket')
dist.destroy_process_group()
