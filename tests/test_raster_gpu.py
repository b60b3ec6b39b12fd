"""GPU: the tile-binned HIP rasteriser (hoig_rasterize_fim_wim through the C ABI) against the plain-C oracle of the
reference's brute-force kernels (oracle/raster.c): the face-index map must be IDENTICAL (integer output) and the weight map
bit-identical (same fp32 operations, no contraction on either side) on meshes with back faces, depth ties, degenerate,
clipped, huge and out-of-range triangles."""
import numpy as np
import pytest
import torch

from common import oracle_rasterize, synthetic_mesh_faces

pytestmark = pytest.mark.gpu


def check(faces, S, **kw):
    from hoig_amd import raster
    fim, wim = raster.rasterize_fim_wim(faces.cuda(), S, **kw)
    ofim, owim = oracle_rasterize(faces, S, **kw)
    fim, wim = fim.cpu(), wim.cpu()
    bad = (fim != ofim)
    assert not bad.any(), '%d of %d pixels differ' % (int(bad.sum()), fim.numel())
    assert torch.equal(wim, owim), float((wim - owim).abs().max())
    return fim


@pytest.mark.parametrize('batch,n_random,S', [(2, 1500, 256), (1, 0, 256), (3, 4000, 128), (1, 300, 64)])
def test_hip_matches_oracle_exactly(batch, n_random, S):
    fim = check(synthetic_mesh_faces(batch, n_random), S)
    assert (fim >= 0).float().mean() > 0.05


def test_non_multiple_of_tile_and_near_far():
    f = synthetic_mesh_faces(1, 800, seed=9)
    check(f, 100)                                  # image side not a multiple of the 16-pixel tile
    check(f, 256, near=1.0, far=3.0)


def test_many_faces_in_one_tile():
    """More faces over one tile than the LDS list holds (512): the list is consumed in several rounds."""
    g = np.random.default_rng(3)
    n = 3000
    xy = g.uniform(-0.05, 0.05, size=(n, 1, 2)) + g.uniform(-0.04, 0.04, size=(n, 3, 2))
    z = g.uniform(1.0, 4.0, size=(n, 3, 1))
    f = torch.from_numpy(np.concatenate([xy, z], -1).astype(np.float32))[None]
    fim = check(f, 256)
    assert len(np.unique(fim.numpy())) > 50


def test_empty_and_offscreen():
    f = torch.tensor([[[[2.0, 2.0, 2.0], [3.0, 2.0, 2.0], [2.0, 3.0, 2.0]]]])           # entirely off screen
    fim = check(f, 64)
    assert (fim == -1).all()
    f = torch.tensor([[[[-5.0, -5.0, 2.0], [5.0, -5.0, 2.0], [0.0, 5.0, 2.0]]]])         # covers the whole image
    fim = check(f, 64)
    assert (fim == 0).all()


def test_render_fim_wim_feeds_input_prep():
    """render_fim_wim (projection + look-at + rasterisation) produces maps the input-preparation stage accepts."""
    from hoig_amd import raster
    g = np.random.default_rng(1)
    V = 600
    verts = torch.from_numpy(np.concatenate([g.uniform(-0.1, 0.1, size=(1, V, 2)), g.uniform(-0.7, -0.5, size=(1, V, 1))],
                                            -1).astype(np.float32)).cuda()
    cam = torch.tensor([[600.0, 0, 128, 0, 600.0, 128, 0, 0, 1, 1, 0, 0, 0, 1, 0]], dtype=torch.float32).cuda()
    idx = torch.from_numpy(g.integers(0, V, size=(1200, 3)).astype(np.int32)).cuda()
    faces, fim, wim = raster.render_fim_wim(cam, verts, idx, 256)
    assert faces.shape == (1, 1200, 3, 3) and fim.shape == (1, 256, 256) and wim.shape == (1, 256, 256, 3)
    ofim, owim = oracle_rasterize(faces.cpu(), 256)
    assert torch.equal(fim.cpu(), ofim) and torch.equal(wim.cpu(), owim)
    assert (fim >= 0).any() and (fim == -1).any()
    covered = fim[0] >= 0
    assert torch.allclose(wim[0][covered].sum(-1), torch.ones_like(wim[0][covered].sum(-1)), atol=1e-5)


def test_vertex_stage_on_device_matches_reference_fixture():
    import os
    from hoig_amd import raster
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'raster_vertex_stage.npz'))
    faces = raster.project_to_faces(torch.from_numpy(g['cam']).cuda(), torch.from_numpy(g['vertices']).cuda(),
                                    torch.from_numpy(g['faces_idx']).cuda())
    np.testing.assert_allclose(faces.cpu().numpy(), g['faces'], rtol=0, atol=2e-5)


def test_vertex_stage_on_device_matches_reference_fixture_dexycb():
    import os
    from hoig_amd import raster
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'raster_vertex_stage_dexycb.npz'))
    faces = raster.project_to_faces(torch.from_numpy(g['cam']).cuda(), torch.from_numpy(g['vertices']).cuda(),
                                    torch.from_numpy(g['faces_idx']).cuda())
    np.testing.assert_allclose(faces.cpu().numpy(), g['faces'], rtol=0, atol=2e-5)


@pytest.mark.parametrize('fixture', ['raster_vertex_stage.npz', 'raster_vertex_stage_dexycb.npz'])
def test_one_launch_vertex_stage_matches_the_reference_fixture_and_the_torch_form(fixture):
    """hoig_project_faces (round 6: projection + flip + look-at + vertices_to_faces for a batch whose samples hold different objects, one
    launch) against the fixture made by the reference's own projection, and against raster.project_to_faces on a batch with two
    face lists of different lengths and a padded tail."""
    import os
    from hoig_amd import raster
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', fixture))
    cam, verts = torch.from_numpy(g['cam']).cuda(), torch.from_numpy(g['vertices']).cuda()
    idx = torch.from_numpy(g['faces_idx']).cuda().long().contiguous()
    B = cam.shape[0]
    lists = [idx if idx.dim() == 2 else idx[b].contiguous() for b in range(B)]
    faces = raster.project_faces_batched(cam, verts, lists, lists[0].shape[0])
    np.testing.assert_allclose(faces.cpu().numpy(), g['faces'], rtol=0, atol=2e-5)
    # two different lists in one batch: sample 0 keeps the full list, the others a prefix; rows beyond a list are the far-away point
    short = lists[0][: lists[0].shape[0] // 3].contiguous()
    mixed = [lists[0]] + [short] * (B - 1)
    got = raster.project_faces_batched(cam, verts, mixed, lists[0].shape[0], pad_value=-1.0e6)
    full = raster.project_to_faces(cam, verts, lists[0])
    np.testing.assert_allclose(got[0].cpu().numpy(), full[0].cpu().numpy(), rtol=0, atol=2e-5)
    for b in range(1, B):
        np.testing.assert_allclose(got[b, :short.shape[0]].cpu().numpy(), full[b, :short.shape[0]].cpu().numpy(), rtol=0, atol=2e-5)
        assert bool((got[b, short.shape[0]:] == -1.0e6).all())


def test_vertices_to_generator_inputs_end_to_end():
    """The whole HandRecoveryFlow.forward chain on the device -- render_fim_wim for the source and the reference pose, then
    the input preparation -- against the oracle chain (C rasteriser + torch restatement) on the same vertices."""
    from hoig_amd import raster, input_prep as IP, synthetic
    from oracle import input_prep_oracle as P
    tb = synthetic.make_object_tables(3, seed=21)
    F, NH = tb['n_faces'], synthetic.N_HAND_FACES
    g = np.random.default_rng(4)
    hand_v = np.concatenate([g.uniform(-0.06, 0.02, size=(300, 2)), g.uniform(-0.62, -0.56, size=(300, 1))], -1)
    obj_v = np.concatenate([g.uniform(0.0, 0.08, size=(200, 2)), g.uniform(-0.66, -0.58, size=(200, 1))], -1)
    verts = torch.from_numpy(np.concatenate([hand_v, obj_v])[None].astype(np.float32))
    near = lambda n, lo, hi: g.integers(lo, hi, size=(n, 3))
    idx = torch.from_numpy(np.concatenate([near(NH, 0, 300), near(F - NH, 300, 500)]).astype(np.int32))
    cams = [torch.tensor([[600.0, 0, 128, 0, 600.0, 128, 0, 0, 1, 1, 0, 0, 0, 1, 0]], dtype=torch.float32),
            torch.tensor([[560.0, 0, 120, 0, 580.0, 140, 0, 0, 1, 1, 0.02, 3, -0.02, 1, -2]], dtype=torch.float32)]
    src_img = torch.from_numpy(g.uniform(-1, 1, size=(1, 3, 256, 256)).astype(np.float32))
    ref_img = torch.from_numpy(g.uniform(-1, 1, size=(1, 3, 256, 256)).astype(np.float32))
    dev = torch.device('cuda', 0)
    s_faces, s_fim, s_wim = raster.render_fim_wim(cams[0].to(dev), verts.to(dev), idx.to(dev))
    _, r_fim, r_wim = raster.render_fim_wim(cams[1].to(dev), verts.to(dev), idx.to(dev))
    assert ((s_fim >= 0) & (s_fim < NH)).any() and (s_fim >= NH).any() and (s_fim == -1).any()
    out = IP.prepare_inputs(src_img.to(dev), ref_img.to(dev), s_faces, s_fim, s_wim, r_fim, r_wim, [IP.ObjectTables(tb, dev)])
    # oracle chain from the same face tensors (the vertex stage is pinned separately)
    of, ow = oracle_rasterize(s_faces.cpu(), 256)
    rf, rw = oracle_rasterize(raster.project_to_faces(cams[1].to(dev), verts.to(dev), idx.to(dev)).cpu(), 256)
    assert torch.equal(s_fim.cpu(), of) and torch.equal(r_fim.cpu(), rf)
    want = P.prepare_inputs(src_img, ref_img, s_faces.cpu(), of, ow, rf, rw, [tb])
    for a, b in zip(out, want):
        if a is not None:
            assert (a.cpu() - b).abs().max() <= 2e-6
    for i in (7, 8, 9, 10):
        assert torch.equal(out[i].cpu(), want[i])
