"""CPU suite: host logic, the C-ABI library's exports, schema agreement, and the rule that the product never touches
the oracle.  No compute kernels are launched (no GPU here)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    from hoig_amd import _lib as L
    syms = L.declared_symbols()
    assert len(syms) >= 30
    lib = ctypes.CDLL(L.LIB_PATH)
    for s in syms:
        assert hasattr(lib, s), 'libhoig_hip.so does not export %s (declared in include/hoig_kernels.h)' % s
    assert set(L._SIGS).issubset(set(syms))
    assert b'gfx950' in L.lib.hoig_version()


def test_c_abi_rejects_bad_arguments_without_a_gpu():
    """Argument validation happens before any launch, so it can be exercised on the CPU box."""
    from hoig_amd import _lib as L
    d = L.ConvDesc(1, 8, 8, 4, 8, 8, 4, 3, 3, 1, 1, 0, 0, 0.0, 0)
    assert L.lib.hoig_conv2d_fwd(ctypes.byref(d), None, None, None, None, None) == L.EINVAL
    bad = L.ConvDesc(1, 8, 8, 4, 9, 9, 4, 3, 3, 1, 1, 0, 0, 0.0, 0)          # inconsistent output size
    assert L.lib.hoig_conv2d_fwd(ctypes.byref(bad), 1, 1, None, 1, None) == L.EINVAL
    assert L.lib.hoig_adam_step(None, None, None, None, 4, 1e-3, 0.9, 0.999, 1e-8, 1, 1.0, None) == L.EINVAL
    assert L.lib.hoig_inorm_workspace_bytes(2, 1024, 512) == ((1 << 18) + 2 * 2 * 512) * 4


def test_product_never_imports_the_oracle():
    pat = re.compile(r'^\s*(from|import)\s+(oracle|tests)\b|oracle[./]', re.M)
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'hoig_amd')):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.sh')):
                text = open(os.path.join(dirpath, f)).read()
                assert not pat.search(text), '%s references the oracle' % os.path.join(dirpath, f)


def test_ops_fail_loudly_on_cpu_tensors():
    from hoig_amd import ops
    with pytest.raises(NotImplementedError):
        ops.conv2d(torch.zeros(1, 4, 4, 4), ops.pack_weight(torch.zeros(4, 4, 3, 3)), None, 1, 1)
    with pytest.raises(NotImplementedError):
        ops.instance_norm(torch.zeros(1, 4, 4, 4))


@pytest.mark.parametrize('gen_name', ['generator_base', 'generator_spade', 'generator_spade_attn',
                                      'generator_spade_attn_tiny'])
@pytest.mark.parametrize('dataset', ['hov3', 'dexycb'])
def test_product_schema_equals_oracle_schema(gen_name, dataset):
    from oracle import hogan_oracle as O
    from hoig_amd.models.networks.schema import GeneratorConfig, generator_schema, discriminator_schema, vgg_schema
    cfg = O.make_cfg(gen_name, dataset)
    pc = GeneratorConfig(gen_name, cfg['bg_dim'], 3, 3, cfg['img_cond_dim'], 12)
    assert list(generator_schema(pc).shapes.items()) == list(O.gen_param_shapes(cfg).items())
    assert list(discriminator_schema(cfg['d_input_nc'], 64, 4).shapes.items()) == list(O.disc_param_shapes(cfg).items())
    assert list(vgg_schema().shapes.items()) == list(O.vgg_param_shapes().items())


def test_schema_matches_reference_golden_names_and_counts():
    from oracle import hogan_oracle as O
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'hov3_spade_attn_64.npz'))
    shp = O.gen_param_shapes(O.make_cfg('generator_spade_attn'))
    assert [str(s) for s in g['param_names_G']] == list(shp.keys())
    assert len(shp) == 425 and sum(int(np.prod(s)) for s in shp.values()) == 183501729
    dshp = O.disc_param_shapes(O.make_cfg('generator_spade_attn'))
    assert len(dshp) == 12 and sum(int(np.prod(s)) for s in dshp.values()) == 6975937


def test_packed_weight_views():
    from hoig_amd.ops import pack_weight, packed_strides
    w = torch.arange(2 * 3 * 2 * 2, dtype=torch.float32).view(2, 3, 2, 2)      # conv (Co,Ci,R,S)
    p = pack_weight(w)
    assert torch.equal(p, w) and p.stride() == packed_strides(w.shape, False)
    flat = p.as_strided((p.numel(),), (1,))
    assert torch.equal(flat.view(2, 2, 2, 3), w.permute(0, 2, 3, 1))           # storage is [Co][R][S][Ci]
    wt = torch.arange(3 * 2 * 2 * 2, dtype=torch.float32).view(3, 2, 2, 2)     # convT (Ci,Co,R,S)
    pt = pack_weight(wt, transposed=True)
    assert torch.equal(pt, wt)
    assert torch.equal(pt.as_strided((pt.numel(),), (1,)).view(2, 2, 2, 3), wt.permute(1, 2, 3, 0))


def test_param_tree_state_dict_roundtrip_cpu():
    from hoig_amd.nn import ParamTree
    from hoig_amd.models.networks.schema import discriminator_schema
    sch = discriminator_schema(19, 8, 2)
    tree = ParamTree(sch.shapes, torch.device('cpu'))
    g = torch.Generator().manual_seed(0)
    sd = {k: torch.randn(v, generator=g) for k, v in sch.shapes.items()}
    tree.load_state_dict(sd)
    out = tree.state_dict()
    assert list(out.keys()) == list(sch.shapes.keys())
    for k in sd:
        assert out[k].is_contiguous() and torch.equal(out[k], sd[k])
    assert [n for n, _ in tree.named_parameters()] == list(sch.shapes.keys())
    assert tree.flat.numel() >= sum(v.numel() for v in sd.values())
    for p in tree.parameters():
        assert p.grad is not None and p.grad.data_ptr() >= tree.flat_grad.data_ptr()
    with pytest.raises(RuntimeError):
        tree.load_state_dict({'nope': torch.zeros(1)})
    tree.init_weights(torch.Generator().manual_seed(1))
    assert abs(float(tree.P['model.0.weight'].std()) - 0.02) < 5e-3 and float(tree.P['model.0.bias'].abs().max()) == 0


def test_param_tree_adjacent_groups_and_fused_views_cpu():
    """ParamTree(adjacent=...) moves parameters next to each other in the flat store (the object branch's three 7x7 heads that
    read one feature map: generator.py:219-235,311-315) without changing what a checkpoint sees: names, order, shapes, values."""
    from hoig_amd.nn import ParamTree
    shapes = {'a.img.weight': (3, 8, 7, 7), 'a.norm.weight': (8,), 'b.mask.weight': (1, 16, 7, 7), 'b.other.weight': (4, 8, 3, 3),
              'c.mask.weight': (1, 16, 7, 7)}
    split = ['b.mask.weight', 'c.mask.weight']                      # stored as two halves of their input channels (#t | #s)
    group = ['a.img.weight', 'b.mask.weight#s', 'c.mask.weight#s']
    plain = ParamTree(shapes, torch.device('cpu'), split_names=split)
    tree = ParamTree(shapes, torch.device('cpu'), split_names=split, adjacent=[group])
    assert not plain.fuse_conv_weights('heads', group)              # not adjacent in the reference order: nothing registered
    assert 'heads' not in plain.F
    assert tree.fuse_conv_weights('heads', group)
    offs = [tree._offsets[n] for n in group]
    assert offs[1] == offs[0] + 3 * 8 * 49 and offs[2] == offs[1] + 8 * 49
    g = torch.Generator().manual_seed(3)
    sd = {k: torch.randn(v, generator=g) for k, v in shapes.items()}
    tree.load_state_dict(sd)
    plain.load_state_dict(sd)
    out = tree.state_dict()
    assert list(out.keys()) == list(shapes.keys()) == list(plain.state_dict().keys())
    for k in sd:
        assert torch.equal(out[k], sd[k])
    fused = tree.F['heads']                                          # (5, 8, 7, 7) in the packed [Co][R][S][Ci] layout
    assert tuple(fused.shape) == (5, 8, 7, 7)
    want = torch.cat([sd['a.img.weight'], sd['b.mask.weight'][:, 8:], sd['c.mask.weight'][:, 8:]], dim=0)
    assert torch.equal(fused.detach(), want)
    assert fused.grad.data_ptr() == tree.flat_grad.data_ptr() + 4 * offs[0]
    assert tree.flat.numel() == plain.flat.numel()


def test_synthetic_inputs_deterministic_and_in_range():
    from hoig_amd import synthetic
    a = synthetic.make_inputs(2, 32, seed=8)
    b = synthetic.make_inputs(2, 32, seed=8)
    for k in a:
        assert torch.equal(a[k], b[k])
    assert tuple(a['input_G_bg'].shape) == (2, 4, 32, 32) and tuple(a['input_G_src_obj'].shape) == (2, 15, 32, 32)
    assert tuple(a['input_G_src_hand'].shape) == (2, 6, 32, 32) and tuple(a['T'].shape) == (2, 32, 32, 2)
    assert tuple(a['bg_mask'].shape) == (4, 1, 32, 32)
    assert set(a['bg_mask'].unique().tolist()) <= {0.0, 1.0}
    assert float(a['T'].min()) == -2.0 and float(a['real_src'].abs().max()) <= 1.0
    d = synthetic.make_inputs(1, 32, seed=8, dataset='dexycb')
    assert tuple(d['input_G_src_hand'].shape) == (1, 12, 32, 32) and 'armask_src' not in d
    c = synthetic.make_inputs(2, 32, seed=9)
    assert not torch.equal(a['real_src'], c['real_src'])
    # every pair shows some of its object in both views (an all-zero object image is an ill-conditioned input: the instance
    # norms of obj_model would normalise rounding noise)
    big = synthetic.make_inputs(32, 64, seed=8)
    for k in ('input_G_src_obj', 'input_G_tsf_obj'):
        assert bool((big[k][:, 6:].flatten(1).sum(1) > 0).all()), k


def test_eval_writer_matches_reference_layout(tmp_path):
    """eval.py:59-79: crops (r, c) = (i // cols, i % cols) of the three batch grids, named <srcvid>_<srcframe>_<tsfframe>.png."""
    import numpy as np
    from PIL import Image
    from hoig_amd.eval_output import EvalWriter, crops_of, pair_name
    side, B = 8, 5
    nrow = int(np.sqrt(B))                                   # tensor2im: make_grid(img, nrows=int(sqrt(B)), padding=0)
    rows = (B + nrow - 1) // nrow
    rng = np.random.default_rng(0)
    imgs = {k: rng.integers(0, 256, size=(B, 3, side, side), dtype=np.uint8) for k in ('src', 'fake', 'real')}

    def grid(x):
        g = np.zeros((3, rows * side, nrow * side), np.uint8)
        for i in range(B):
            r, c = i // nrow, i % nrow
            g[:, r * side:(r + 1) * side, c * side:(c + 1) * side] = x[i]
        return g

    vis = {'16_batch_src_img': grid(imgs['src']), '15_batch_fake_img': grid(imgs['fake']), '14_batch_real_img': grid(imgs['real'])}
    a = ['vidA/%04d.jpg' % i for i in range(B)]
    b = ['vidB/%04d.jpg' % (i + 7) for i in range(B)]
    assert pair_name(a[1], b[1]) == 'vidA_0001_0008.png'
    w = EvalWriter(str(tmp_path), sav_gt=True, side=side, workers=2)
    w.write(vis, a, b)
    w.close()
    assert w.written == 3 * B
    for sub, key in (('source', 'src'), ('imitators', 'fake'), ('gt', 'real')):
        for i in range(B):
            got = np.asarray(Image.open(str(tmp_path / sub / pair_name(a[i], b[i]))))
            assert np.array_equal(got, imgs[key][i].transpose(1, 2, 0))
    assert len(crops_of(vis['14_batch_real_img'], B, side)) == B
    with pytest.raises(ValueError):
        crops_of(vis['14_batch_real_img'], rows * nrow + 1, side)


def test_bench_refuses_mislabelled_multi_gpu_runs():
    """bench.py --gpus N must never report a smaller job under the N-GPU label (`--gpus 8` ran one rank):
    with fewer visible GPUs than N it exits 2 before any GPU call; under a launcher whose WORLD_SIZE differs from N it errors."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and 'refusing' in r.stderr + r.stdout
    assert not (r.stdout.strip().startswith('{'))                       # no JSON line for a job that did not run
    env.update(WORLD_SIZE='4', RANK='0', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and 'WORLD_SIZE=4' in r.stderr + r.stdout


def test_bench_launcher_child_forwards_rank0_json_and_the_exit_code():
    """`python bench.py --gpus 2` started plainly launches torch.distributed.run as a CHILD and forwards what rank 0 prints and the
    child's exit code.  Exercised without a GPU through --dry-run-cpu: two ranks over gloo run the real run's
    protocol (warm-up, barrier, K timed steps, barrier, MAX over ranks, one JSON line on rank 0) with a sleep as the step; rank 1
    sleeps 2 ms per step, rank 0 one -- the line must carry the SLOWER rank's time."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '20', '--warmup', '2', '--dry-run-cpu'],
                       env=env, capture_output=True, text=True, timeout=300, cwd='/tmp')
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout                                     # ONE line, from rank 0 only
    out = json.loads(lines[0])
    assert out['dry_run'] is True and out['value'] is None and out['n_gpus'] == 2 and out['steps'] == 20 and out['warmup'] == 2
    assert out['config']['parallelism'] == 'dp2' and out['ms_per_step'] >= 2.0, out       # max over ranks, not rank 0's own 1 ms
    # a rank that fails takes the job down and the parent reports it: WORLD_SIZE / --gpus mismatch inside the child
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', '29671', os.path.join(root, 'bench.py'), '--gpus', '3', '--dry-run-cpu'],
                       env=env, capture_output=True, text=True, timeout=300, cwd='/tmp')
    assert r.returncode != 0 and 'WORLD_SIZE=2' in r.stderr + r.stdout


def test_batched_input_preparation_limit_matches_the_header():
    """hoig_amd.input_prep loops over the per-sample entry points beyond the batched ones' limit: the Python constant must be the
    header's HOIG_PREP_MAX_BATCH (the argument blocks carry that many per-sample pointers by value)."""
    import re
    from hoig_amd import _lib as L, input_prep as IP
    m = re.search(r'#define\s+HOIG_PREP_MAX_BATCH\s+(\d+)', open(L.HEADER_PATH).read())
    assert m and int(m.group(1)) == IP.MAX_BATCH
