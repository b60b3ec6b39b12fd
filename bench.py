#!/usr/bin/env python
"""bench.py -- HOGAN G+D training-step throughput on synthetic 256x256 hand-object-pose tensors.

  python bench.py --gpus N --steps K --warmup W            (N>1: launched by torch.distributed.run, one rank per GPU)

A "step" is one ``Trainer.optimize_parameters()`` (forward, G loss, backward, Adam(G), D loss, backward, Adam(D);
under DDP also the RCCL gradient exchange) on a per-GPU batch of 8 pairs (BASELINE.json configs[1]; weak scaling:
configs[2] is the same per-GPU batch on 8 GPUs).  Inputs are resident in HBM before the timed region.
Prints ONE JSON line on rank 0.
"""
import argparse
import contextlib
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

GFLOP_PER_PAIR_TRAIN_256 = 2555.2      # BASELINE.md §2 / SURVEY.md §8d (3*G + 3*VGG + 9*D), generator_spade_attn
PEAK = {'f32': 157.3, 'bf16x3': 2500.0, 'bf16': 2500.0}          # TFLOP/s dense MFMA, MI355X_MICROARCH.md


def dominant_kernel_roofline(batch, side, precision, iters=50):
    """The dominant kernel of the step is the 3x3 stride-1 512->512 convolution at side/8 (72 of ~260 conv calls of a
    forward, 44% of G's MACs).  Most of its time is spent in the launches of bg_model and obj_model,
    which process the src and the tsf batch STACKED (2*batch images per launch: profiles/r01_conv_table.txt), so that is
    the launch shape timed here: the kernel alone, HIP events on the launch stream."""
    from hoig_amd import ops
    h = side // 8
    batch = 2 * batch
    x = torch.randn(batch, h, h, 512, device='cuda')
    w = ops.pack_weight(torch.randn(512, 512, 3, 3, device='cuda') * 0.02)
    # a weight outside a ParamTree has no version, so ops would re-split it on every call; give it a constant one so that
    # the timed loop launches the convolution kernel only (in the training step the split happens once per optimiser step)
    import types
    w._hoig_owner = types.SimpleNamespace(version=0, packed_planes=lambda w_, for_dgrad: None)
    for _ in range(10):
        ops.conv2d(x, w, None, 1, 1)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(iters):
        ops.conv2d(x, w, None, 1, 1)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    flops = 2.0 * batch * h * h * 512 * 512 * 9            # algorithmic: 2*M*N*K, M=B*h*h, N=512, K=9*512
    # HBM/fabric traffic of this kernel comes from separate rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE cannot
    # share a pass); the committed summary is attached when it was taken on the same kernel and shape
    traffic = None
    try:
        pmc = json.load(open(os.path.join(ROOT, 'profiles', 'r01_pmc_dominant_conv.json')))
        if precision == 'bf16x3' and batch == 16 and side == 256:
            traffic = pmc['traffic_bytes_per_launch']
    except Exception:
        pass
    achieved = flops / (ms * 1e-3) / 1e12
    kname = 'igemm_f32_kernel' if precision == 'f32' else 'conv_halo3_bf16_kernel<%d,4,2,128,2,true>' % (2 if precision == 'bf16x3' else 1)
    return dict(bound='mfma', kernel='%s (conv3x3 s1 512->512 @%dx%d, %d images = src+tsf stacked)' % (kname, h, h, batch),
                achieved=round(achieved, 2), peak=PEAK[precision], unit='TFLOP/s',
                frac=round(achieved / PEAK[precision], 4), traffic=traffic, avg_launch_ms=round(ms, 4),
                algorithmic_flop_per_launch=flops)


def gen_forward_latency(opt, batch, side, iters=10):
    """BASELINE.json's second headline: generator-only inference (`Trainer.forward` under no_grad in eval mode,
    eval.py:59-65), batch 32 at 256x256 (configs[4]); timed eagerly and as a captured hipGraph replay."""
    from hoig_amd import synthetic
    from hoig_amd.models import ModelsFactory
    opt.is_train = False
    opt.load_path = 'None'
    opt.load_epoch = -1
    from hoig_amd.models.trainer import Trainer
    model = Trainer.__new__(Trainer)          # eval.py would load a checkpoint; here: random-init weights (no files)
    from hoig_amd.models.base_model import BaseModel
    BaseModel.__init__(model, opt)
    model._name = 'Trainer'
    model.device = torch.device('cuda', torch.cuda.current_device())
    model._dexycb, model._world, model._side, model._g_ready = False, 1, None, None
    with contextlib.redirect_stdout(sys.stderr):       # 'Network ... was created' banners: keep stdout to the JSON line
        model._init_create_networks(use_ddp=False)
    model._init_prefetch_inputs()
    model.set_eval()
    model.set_input(synthetic.make_inputs(batch, side, seed=8))
    out = {}
    with torch.no_grad():
        for _ in range(2):
            model.forward()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            model.forward()
        e.record()
        torch.cuda.synchronize()
        out['eager_ms_per_img'] = round(s.elapsed_time(e) / iters / batch, 4)
        try:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                res = model.forward()
            graph.replay()
            torch.cuda.synchronize()
            s.record()
            for _ in range(iters):
                graph.replay()
            e.record()
            torch.cuda.synchronize()
            out['hipgraph_ms_per_img'] = round(s.elapsed_time(e) / iters / batch, 4)
            out['finite'] = bool(torch.isfinite(res[3]).all().item())
        except Exception as ex:          # report, do not hide
            out['hipgraph_error'] = repr(ex)[:200]
    out['batch'] = batch
    out['gflop_per_img'] = 787.2 * (side / 256.0) ** 2
    best = out.get('hipgraph_ms_per_img', out['eager_ms_per_img'])
    out['tflops'] = round(out['gflop_per_img'] / best, 2)
    return out


def usable_cores():
    """Host cores this process may really use: scheduler affinity capped by the cgroup CPU quota (a thread pool
    sized from os.cpu_count() inside a quota-limited container oversubscribes badly)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()
        if q != 'max':
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        try:
            q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return max(1, n)


def cpu_baseline_worker(side):
    """Child process: time ONE full oracle G+D step (the reference's algorithm, fp32 torch-CPU) at `side`, batch 1."""
    from common import oracle_trainer
    cores = usable_cores()
    torch.set_num_threads(cores)
    warm = oracle_trainer('generator_spade_attn', 1, 64)
    warm.optimize_parameters()
    ot = oracle_trainer('generator_spade_attn', 1, side)
    t0 = time.time()
    ot.optimize_parameters()
    print(json.dumps(dict(seconds=time.time() - t0, cores=cores, side=side)))


def cpu_baseline(side):
    """The oracle (kind "port": pinned bit-exact to the reference in the build container) timed on this box's host
    cores on a BOUNDED sample, in a child process under a hard timeout so the bench always finishes in minutes:
    one full G+D step at the benchmark resolution, batch 1; if that does not finish in time, the same step at half
    the side, scaled by the pixel ratio (the network is fully convolutional: work is proportional to pixels)."""
    import subprocess
    for s, budget in ((side, 150), (side // 2, 90)):
        try:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-baseline-worker', str(s)],
                                 stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=budget)
            r = json.loads(out.stdout.strip().splitlines()[-1])
        except Exception:
            continue
        scale = (s / float(side)) ** 2
        return dict(value=round(scale / r['seconds'], 4), unit='images/s', cores=r['cores'], kind='port',
                    sample='1 full G+D step (optimize_parameters) of the oracle at %dx%d, batch 1, fp32, after a 64x64 '
                           'warm-up step: %.1f s%s' % (s, s, r['seconds'],
                                                        '' if s == side else '; scaled by the pixel ratio %.2f' % scale))
    return dict(value=None, unit='images/s', cores=usable_cores(), kind='port', sample='did not finish within the budget')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=8, help='pairs per GPU')
    ap.add_argument('--side', type=int, default=256)
    ap.add_argument('--gen_name', default='generator_spade_attn')
    ap.add_argument('--dataset', default='hov3', choices=['hov3', 'dexycb'], help='channel configuration (config C4 = dexycb at --side 512 --batch 4)')
    ap.add_argument('--precision', default=os.environ.get('HOIG_PRECISION', 'bf16x3'))
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-gen-fwd', action='store_true')
    ap.add_argument('--fwd-batch', type=int, default=32, help='batch of the generator-forward latency leg')
    ap.add_argument('--cpu-baseline-worker', type=int, default=0, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_worker:
        return cpu_baseline_worker(args.cpu_baseline_worker)

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    ddp = world > 1
    torch.cuda.set_device(local_rank)
    if ddp:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend='nccl', device_id=torch.device('cuda', local_rank))

    from hoig_amd import ops, synthetic
    from hoig_amd.models import ModelsFactory
    from common import opt_namespace
    ops.set_precision(args.precision)

    opt = opt_namespace(gen_name=args.gen_name, local_rank=local_rank, image_size=args.side, dataset_mode=args.dataset)
    torch.manual_seed(8)
    with contextlib.redirect_stdout(sys.stderr):       # the reference-style construction banners go to stderr
        model = ModelsFactory.get_by_name('trainer', opt, use_ddp=ddp)
    model.set_train()
    model.set_input(synthetic.make_inputs(args.batch, args.side, seed=8 + rank, dataset=args.dataset))
    torch.cuda.synchronize()

    def barrier():
        if ddp:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        model.optimize_parameters()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        model.optimize_parameters()
    barrier()
    dt = time.perf_counter() - t0
    if ddp:
        t = torch.tensor([dt], device='cuda', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    errors = model.get_current_errors()

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = world * args.batch * args.steps / dt
        roof = dominant_kernel_roofline(args.batch, args.side, args.precision)
        out = {
            'metric': 'HOGAN train images/sec at %dx%d' % (args.side, args.side),
            'value': round(value, 3), 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(ms, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': {'bf16x3': 'f16x3 fwd / bf16x3 bwd (split 16-bit operands, f32 accumulate)'}.get(args.precision, args.precision),
            'data': 'synthetic',
            'config': {'workload': '%dx%d %s-shaped synthetic, batch %d per GPU, G+D full step (%s, VGG19 '
                                   'surrogate weights)' % (args.side, args.side, 'HO3Dv3' if args.dataset == 'hov3' else 'DexYCB',
                                                           args.batch, args.gen_name),
                       'global_batch': world * args.batch, 'parallelism': 'dp%d' % world},
            'step_tflops': round(GFLOP_PER_PAIR_TRAIN_256 * (args.side / 256.0) ** 2 * value / 1e3, 2),
            'roofline': roof,
            'losses_finite': all(v == v and abs(v) != float('inf') for v in errors.values()),
        }
        model = None
        torch.cuda.empty_cache()
        if world == 1 and not args.no_gen_fwd:
            out['gen_fwd'] = gen_forward_latency(opt, args.fwd_batch, args.side)
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args.side)
        print(json.dumps(out))
    if ddp:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
