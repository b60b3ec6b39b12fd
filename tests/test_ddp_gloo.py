"""The N>1 path on CPU: two processes, gloo backend, world_size 2 -- parameter broadcast from rank 0 and the
bucketed gradient exchange of hoig_amd/ddp.py over flat buffers (the same code runs over RCCL on the GPUs)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from hoig_amd.ddp import GradSync, FlatDDP
    from hoig_amd.nn import ParamTree
    from hoig_amd.models.networks.schema import discriminator_schema
    torch.manual_seed(100 + rank)                      # different init per rank, as the reference's unseeded CPU init
    tree = ParamTree(discriminator_schema(19, 8, 2).shapes, torch.device('cpu'))
    tree.init_weights()
    before = tree.flat.clone()
    ddp = FlatDDP(tree, bucket_bytes=4096)             # small buckets -> several slices
    assert len(ddp.sync.slices) > 1
    tree.flat_grad.copy_(torch.arange(tree.flat_grad.numel(), dtype=torch.float32) * (rank + 1))
    scale = ddp.sync.all_reduce_grads()
    sd = ddp.state_dict()
    q.put((rank, before.numpy().copy(), tree.flat.numpy().copy(), tree.flat_grad.numpy().copy(), scale,
           list(sd.keys())[:2]))       # numpy: pickled by value (tensors would travel as shm handles)
    dist.barrier()
    dist.destroy_process_group()


def _worker_bf16(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from hoig_amd.ddp import GradSync
    g = torch.Generator().manual_seed(7 + rank)
    grad = torch.randn(5000, generator=g) * (10.0 ** torch.randint(-6, 2, (5000,), generator=g).float())
    param = torch.zeros(5000)
    sync = GradSync(param, grad, bucket_bytes=4096, payload='bf16')
    assert len(sync.slices) > 1 and sync.active
    mine = grad.clone()
    got = [ab for ab in sync.iter_all_reduce()]                 # the sliced form the optimiser consumes
    assert got == sync.slices
    q.put((rank, mine.numpy().copy(), grad.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_bf16_gradient_payload_world2():
    """HOIG_DDP_PAYLOAD=bf16 / GradSync(payload='bf16'): the gradients travel as bf16 (half the bytes per link); both ranks end
    with the SAME fp32 buffer, equal to the sum of the two gradients to bf16 rounding (8 significant bits: 2^-8 per rounding; one of
    each addend, one of the sum)."""
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_bf16, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    import numpy as np
    (_, a0, s0), (_, a1, s1) = res
    assert np.array_equal(s0, s1)
    want = a0.astype(np.float64) + a1.astype(np.float64)
    tol = 2 * 2.0 ** -8 * (np.abs(a0) + np.abs(a1)) + 1e-30
    assert (np.abs(s0 - want) <= tol).all()
    assert np.abs(s0 - want).max() > 0                          # (it really was rounded)


def test_flat_ddp_broadcast_and_allreduce_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, b0, p0, g0, s0, k0), (_, b1, p1, g1, s1, k1) = res
    import numpy as np
    assert not np.array_equal(b0, b1)                   # ranks started from different weights ...
    assert np.array_equal(p0, b0) and np.array_equal(p1, b0)   # ... and both hold rank 0's after construction
    want = np.arange(g0.size, dtype=np.float32) * 3.0   # SUM over ranks of arange*(rank+1)
    assert np.array_equal(g0, want) and np.array_equal(g1, want)
    assert s0 == 0.5 and s1 == 0.5                      # the mean is applied by the optimiser (grad_scale)
    assert all(k.startswith('module.') for k in k0)


def _worker_bucket(rank, world, port, q):
    """Three "backwards" over a flat buffer of several slices, the gradient writes announced through ops' observer hook exactly as the
    kernels' wrappers announce them (ops._grad_target): mode 'bucket' must learn on the first, launch slices early on the later
    ones, and leave the same sums as mode 'after'."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from hoig_amd import ops
    from hoig_amd.ddp import GradSync
    from hoig_amd.nn import ParamTree
    from hoig_amd.models.networks.schema import discriminator_schema
    out = {}
    for mode in ('after', 'bucket'):
        tree = ParamTree(discriminator_schema(19, 8, 2).shapes, torch.device('cpu'))
        sync = GradSync(tree.flat, tree.flat_grad, bucket_bytes=4096, mode=mode)
        assert len(sync.slices) > 3
        params = list(tree.P.values())[::-1]                  # a backward meets the last layers first
        sums, early = [], []
        for step in range(3):
            g = torch.Generator().manual_seed(1000 * step + rank)
            tree.flat_grad.zero_()
            sync.begin_backward('sig')
            for k in range(0, len(params), 2):
                ops._grad_epoch()                                     # a new backward function begins ...
                targets = []
                for p in params[k:k + 2]:                             # ... announces its writes (weight + bias) FIRST ...
                    grad, through_autograd = ops._grad_target(p)
                    assert not through_autograd
                    targets.append((p, grad))
                for p, grad in targets:                               # ... and launches "the kernels" afterwards
                    grad.add_(torch.randn(p.shape, generator=g))
            sync.end_backward()
            early.append(sync.early_launches)
            got = [ab for ab in sync.iter_all_reduce()]
            assert got == sync.slices
            sums.append(tree.flat_grad.clone())
        # a write that arrives after its slice went on the wire must be refused, not summed twice
        late = None
        if mode == 'bucket':
            sync.begin_backward('sig')
            try:
                for p in params + params[:1]:
                    ops._grad_epoch()
                    ops._grad_target(p)
            except RuntimeError as ex:
                late = str(ex)
            sync.end_backward()
            for _ in sync.iter_all_reduce():                  # (every rank drains the same collectives)
                pass
        out[mode] = ([s.numpy().copy() for s in sums], early, late)
        assert ops._grad_observer is None
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_bucket_mode_matches_after_mode_world2():
    """opt.ddp_mode = 'bucket' (VERDICT r5 item 6): slices are exchanged while the backward is still running -- after a learning pass --
    and the result is bit-identical to the default mode's; a changed write pattern is an error."""
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_bucket, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    import numpy as np
    for rank in range(world):
        after, bucket = res[rank]['after'], res[rank]['bucket']
        for a, b in zip(after[0], bucket[0]):
            assert np.array_equal(a, b)
        assert after[1] == [0, 0, 0]
        assert bucket[1][0] == 0 and bucket[1][1] > 3 and bucket[1][2] == bucket[1][1]      # learned, then early launches
        assert bucket[2] is not None and 'after it had gone on the wire' in bucket[2]
    for a, b in zip(res[0]['bucket'][0], res[1]['bucket'][0]):
        assert np.array_equal(a, b)                           # both ranks hold the same sums
