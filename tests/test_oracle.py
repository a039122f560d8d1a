"""Pins the CPU oracle (oracle/) against the reference: golden vectors generated from the reference
(tests/golden/make_golden.py), the reference's own compiled C++ (oracle/_ref) when present, and the
brute-force construction of the reference's own DynamicScatter test."""
import numpy as np
import pytest
import torch

from conftest import DROP_TEST, DROP_TRAIN, load_golden
from oracle import build_ref, sst_oracle, voxel_oracle


# ------------------------------------------------------------------------------------------------
# dynamic_voxelize
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('case', ['sst', 'fsd', 'fsdv2'])
def test_voxelize_oracle_matches_reference_golden(case):
    g = load_golden('voxelize.npz')
    coors = voxel_oracle.dynamic_voxelize(g[f'{case}::points'], g[f'{case}::voxel_size'], g[f'{case}::range'])
    np.testing.assert_array_equal(coors, g[f'{case}::coors'])


def test_voxelize_oracle_matches_compiled_reference_live():
    mod = build_ref.load()
    if mod is None:
        pytest.skip('oracle/_ref not built (reference tree absent)')
    g = torch.Generator().manual_seed(123)
    pts = torch.rand(20000, 4, generator=g) * torch.tensor([160.0, 160.0, 8.0, 1.0]) + torch.tensor(
        [-80.0, -80.0, -3.0, 0.0])
    for vs, rng in (([0.32, 0.32, 6], [-74.88, -74.88, -2, 74.88, 74.88, 4]),
                    ([0.1, 0.1, 0.1], [-74.88, -74.88, -2, 74.88, 74.88, 4])):
        ref = torch.zeros((pts.size(0), 3), dtype=torch.int32)
        mod.dynamic_voxelize(pts, ref, vs, rng, 3)
        np.testing.assert_array_equal(voxel_oracle.dynamic_voxelize(pts.numpy(), vs, rng), ref.numpy())


# ------------------------------------------------------------------------------------------------
# DynamicScatter: the construction of tests/test_models/test_voxel_encoder/test_dynamic_scatter.py:8-93
# ------------------------------------------------------------------------------------------------
def _bruteforce(feats, coors, mode):
    ref_coors = coors[coors.min(dim=-1).values >= 0]
    ref_coors = ref_coors.unique(dim=0, sorted=True)
    out = []
    for rc in ref_coors:
        sel = feats[(coors == rc).all(dim=-1)]
        out.append(sel.mean(0) if mode == 'mean' else sel.max(0).values)
    return torch.stack(out), ref_coors


@pytest.mark.parametrize('mode', ['mean', 'max'])
def test_scatter_oracle_against_reference_test_construction(mode):
    g = torch.Generator().manual_seed(0)
    feats = torch.rand(3000, 3, generator=g) * 100 - 50
    coors = torch.randint(-1, 8, (3000, 3), dtype=torch.int32, generator=g)
    red, out_coors, cmap, cnt = voxel_oracle.dynamic_point_to_voxel_forward(feats, coors, mode)
    ref_feats, ref_coors = _bruteforce(feats, coors, mode)
    assert torch.equal(out_coors, ref_coors)
    assert torch.allclose(red, ref_feats, atol=1e-5)
    # map / count consistency
    valid = cmap >= 0
    assert torch.equal(out_coors[cmap[valid].long()], coors[valid])
    assert int(cnt.sum()) == int(valid.sum())


def test_scatter_oracle_empty_and_all_invalid():
    feats = torch.rand(0, 3)
    coors = torch.zeros((0, 3), dtype=torch.int32)
    red, out_coors, cmap, cnt = voxel_oracle.dynamic_point_to_voxel_forward(feats, coors, 'max')
    assert red.shape == (0, 3) and out_coors.shape == (0, 3) and cmap.numel() == 0 and cnt.numel() == 0
    feats = torch.rand(50, 3)
    coors = -torch.ones((50, 3), dtype=torch.int32)
    red, out_coors, cmap, cnt = voxel_oracle.dynamic_point_to_voxel_forward(feats, coors, 'mean')
    assert red.shape == (0, 3) and out_coors.shape == (0, 3)
    assert (cmap == -1).all()


def test_scatter_oracle_first_row_quirk():
    """With no invalid point the reference still discards sorted row 0 (scatter_points_cuda.cu:207-210)."""
    feats = torch.arange(12, dtype=torch.float32).view(6, 2)
    coors = torch.tensor([[0, 1, 1], [0, 0, 5], [0, 1, 1], [0, 2, 0], [0, 0, 5], [0, 0, 2]], dtype=torch.int32)
    red, out_coors, cmap, cnt = voxel_oracle.dynamic_point_to_voxel_forward(feats, coors, 'max')
    assert out_coors.tolist() == [[0, 0, 5], [0, 1, 1], [0, 2, 0]]       # (0,0,2) silently dropped
    assert cmap.tolist() == [1, 0, 1, 2, 0, -1]
    red2, out2, cmap2, _ = voxel_oracle.dynamic_point_to_voxel_forward(feats, coors, 'max', reference_compat=False)
    assert out2.tolist() == [[0, 0, 2], [0, 0, 5], [0, 1, 1], [0, 2, 0]]
    assert cmap2.tolist() == [2, 1, 2, 3, 1, 0]


@pytest.mark.parametrize('mode', ['max', 'mean', 'sum'])
def test_scatter_oracle_backward_numeric(mode):
    g = torch.Generator().manual_seed(1)
    feats = (torch.rand(40, 4, generator=g, dtype=torch.float64) * 100 - 50).float()
    coors = torch.randint(-1, 3, (40, 3), dtype=torch.int32, generator=g)
    f = feats.clone().requires_grad_(True)
    red, _ = voxel_oracle.dynamic_scatter(f, coors, mode)
    gout = torch.rand(red.shape, generator=g)
    (red * gout).sum().backward()
    # independent autograd reference built from plain torch ops
    f2 = feats.clone().requires_grad_(True)
    _, _, cmap, cnt = voxel_oracle.dynamic_point_to_voxel_forward(feats, coors, mode)
    rows = []
    for v in range(red.size(0)):
        sel = f2[cmap == v]
        rows.append(sel.max(0).values if mode == 'max' else (sel.mean(0) if mode == 'mean' else sel.sum(0)))
    (torch.stack(rows) * gout).sum().backward()
    assert torch.allclose(f.grad, f2.grad, atol=1e-5)


# ------------------------------------------------------------------------------------------------
# SST input layer restatement vs golden from the reference Python
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('tag', ['eval', 'train'])
def test_region_batching_oracle_matches_reference_golden(tag):
    g = load_golden(f'input_layer_{tag}.npz')
    coors = g['in::voxel_coors'].astype(np.int64)
    drop = DROP_TRAIN if int(g['in::training']) else DROP_TEST
    w0, c0 = sst_oracle.window_coors(coors, (468, 468, 1), (12, 12, 1), False)
    w1, c1 = sst_oracle.window_coors(coors, (468, 468, 1), (12, 12, 1), True)
    rb = sst_oracle.region_batching(w0, w1, drop)
    keep = rb['keep_idx']
    np.testing.assert_array_equal(keep, g['out::voxel_keep_inds'])
    np.testing.assert_array_equal(coors[keep], g['out::voxel_coors'])
    for s, (w, c) in enumerate(((w0, c0), (w1, c1))):
        np.testing.assert_array_equal(w[keep], g[f'out::batch_win_inds_shift{s}'])
        np.testing.assert_array_equal(c[keep], g[f'out::coors_in_win_shift{s}'])
        np.testing.assert_array_equal(rb[f'level{s}'], g[f'out::voxel_drop_level_shift{s}'])
        np.testing.assert_array_equal(rb[f'flat2win{s}'], g[f'out::flat2win_shift{s}'])
        pos = sst_oracle.pos_embed(c[keep], (12, 12, 1), 128)
        np.testing.assert_allclose(pos, g[f'out::pos_flat_shift{s}'], atol=2e-6, rtol=0)
        # the CSR is a partition of the survivors and every window respects its cap
        tok, off = rb[f'tok{s}'], rb[f'winoff{s}']
        assert sorted(tok.tolist()) == list(range(len(keep)))
        assert off[-1] == len(keep) and (np.diff(off) > 0).all()
    if tag == 'train':
        assert len(keep) < coors.shape[0], 'the train fixture must exercise voxel drop'


def _layer_params(g, prefix):
    return {k[len('w::' + prefix):]: v for k, v in g.items() if k.startswith('w::' + prefix)}


@pytest.mark.parametrize('tag', ['std', 'prenorm'])
def test_encoder_layer_oracle_matches_reference_golden(tag):
    g = load_golden(f'sst_block_{tag}.npz')
    coors = g['in::voxel_coors'].astype(np.int64)
    d, h = int(g['cfg::d_model']), int(g['cfg::nhead'])
    w0, c0 = sst_oracle.window_coors(coors, (468, 468, 1), (12, 12, 1), False)
    w1, c1 = sst_oracle.window_coors(coors, (468, 468, 1), (12, 12, 1), True)
    rb = sst_oracle.region_batching(w0, w1, DROP_TEST)
    assert len(rb['keep_idx']) == coors.shape[0]
    x = g['in::voxel_feats'].astype(np.float64)
    for s, c in enumerate((c0, c1)):
        pos = sst_oracle.pos_embed(c, (12, 12, 1), d)
        params = _layer_params(g, f'block_list.0.encoder_list.{s}.')
        x = sst_oracle.encoder_layer(x, pos, rb[f'tok{s}'], rb[f'winoff{s}'], params, h,
                                     post_norm=(tag != 'prenorm'))
    np.testing.assert_allclose(x, g['out::voxel_feats'], atol=2e-4, rtol=1e-4)


def test_sra_core_backward_oracle_numeric():
    rng = np.random.default_rng(0)
    m, h = 23, 2
    q, k, v, do = (rng.standard_normal((m, h * 16)) for _ in range(4))
    tok = rng.permutation(m)
    off = np.array([0, 5, 6, 23])
    dq, dk, dv = sst_oracle.sra_core_backward(q, k, v, do, tok, off, h)
    eps = 1e-6
    for arr, grad in ((q, dq), (k, dk), (v, dv)):
        for (i, j) in ((0, 0), (7, 17), (22, 31)):
            a = arr.copy()
            a[i, j] += eps
            b = arr.copy()
            b[i, j] -= eps
            args_a = [a if x is arr else x for x in (q, k, v)]
            args_b = [b if x is arr else x for x in (q, k, v)]
            fa = (sst_oracle.sra_core(*args_a, tok, off, h) * do).sum()
            fb = (sst_oracle.sra_core(*args_b, tok, off, h) * do).sum()
            assert abs((fa - fb) / (2 * eps) - grad[i, j]) < 1e-5


@pytest.mark.parametrize('tag', ['car', 'cyclist', 'pedestrian'])
def test_cluster_oracle_matches_reference_golden(tag):
    """connected-components restatement against labels produced by the reference's own functions
    (tests/golden/make_golden.py::gen_cluster)."""
    from oracle import cluster_oracle
    g = load_golden('cluster.npz')
    pts, batch, dist = g[f'in::{tag}::points'], g[f'in::{tag}::batch'], float(g[f'in::{tag}::dist'])
    np.testing.assert_array_equal(cluster_oracle.find_connected_components(pts, batch, dist), g[f'out::{tag}::train'])
    np.testing.assert_array_equal(cluster_oracle.find_connected_components_single_batch(pts, dist),
                                  g[f'out::{tag}::test'])


def test_cluster_oracle_matches_reference_function_live():
    from oracle import cluster_oracle, ref_loader
    if not ref_loader.available():
        pytest.skip('reference tree not present')
    from scipy.sparse.csgraph import connected_components
    f = ref_loader.load_reference_function('mmdet3d/models/detectors/single_stage_fsd.py', 'find_connected_componets',
                                           {'connected_components': connected_components})
    gen = torch.Generator().manual_seed(3)
    pts = torch.rand(700, 3, generator=gen) * torch.tensor([30.0, 30.0, 2.0])
    batch = torch.sort(torch.randint(0, 4, (700,), generator=gen))[0].int()
    for dist in (0.3, 1.0, 2.5):
        ref = f(pts, batch, dist).numpy()
        np.testing.assert_array_equal(cluster_oracle.find_connected_components(pts.numpy(), batch.numpy(), dist), ref)
