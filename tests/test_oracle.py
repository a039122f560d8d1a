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


def _pairs_set(pairs):
    return set(map(tuple, np.asarray(pairs).tolist()))


@pytest.mark.parametrize('tag', ['veh', 'ped'])
def test_point_pool_oracle_membership_matches_reference_golden(tag):
    """box convention / membership of the dynamic-point-pool restatement against pairs produced by the reference's own
    points_in_boxes_cpu (tests/golden/make_golden.py::gen_point_pool): exact for the boxes themselves; for the
    enlarged boxes up to pairs that sit within 1e-5 of a face (the enlarged centre is formed differently)."""
    from oracle import point_pool_oracle as O
    g = load_golden('point_pool.npz')
    rois, pts, extra = g[f'in::{tag}::rois'], g[f'in::{tag}::pts'], g[f'in::{tag}::extra_wlh']
    lx, ly, lz = O.local_coords(rois, pts)
    small = O.inside(lx, ly, lz, rois[:, 3], rois[:, 4], rois[:, 5])
    assert _pairs_set(np.stack(np.nonzero(small), 1)) == _pairs_set(g[f'out::{tag}::pairs_in_box'])
    pts_idx, roi_idx, feats = O.dynamic_point_pool(rois, pts, extra, 1 << 30, 1 << 30)
    got = _pairs_set(np.stack([roi_idx, pts_idx], 1))
    want = _pairs_set(g[f'out::{tag}::pairs_in_enlarged_box'])
    clear = O.face_clearance(rois, pts, extra)
    assert all(clear[r, p] < 1e-5 for r, p in got ^ want)
    assert len(got & want) > 500
    # is_in_margin <=> in the enlarged box but not in the box
    in_box = _pairs_set(g[f'out::{tag}::pairs_in_box'])
    for (r, p), m in zip(zip(roi_idx.tolist(), pts_idx.tolist()), feats[:, 12].tolist()):
        assert (m == 0.0) == ((r, p) in in_box)
    O.check_invariants(rois, pts, extra, pts_idx, roi_idx, feats)
    assert (np.diff(roi_idx * (len(pts) + 1) + pts_idx) > 0).all()  # sorted by (roi, point)


def test_point_pool_oracle_caps_and_batches():
    from oracle import point_pool_oracle as O
    g = load_golden('point_pool.npz')
    rois, pts, extra = g['in::veh::rois'], g['in::veh::pts'], g['in::veh::extra_wlh']
    full_p, full_r, full_f = O.dynamic_point_pool(rois, pts, extra, 1 << 30, 1 << 30)
    p8, r8, f8 = O.dynamic_point_pool(rois, pts, extra, 8, 1 << 30)
    assert np.bincount(r8).max() == 8
    for r in np.unique(full_r):
        np.testing.assert_array_equal(p8[r8 == r], full_p[full_r == r][:8])
    pc, rc, fc = O.dynamic_point_pool(rois, pts, extra, 8, 100)
    np.testing.assert_array_equal(pc, p8[:100])
    np.testing.assert_array_equal(fc, f8[:100])
    rb = (np.arange(len(rois)) % 2).astype(np.int32)
    pb = (np.arange(len(pts)) % 2).astype(np.int32)
    pm, rm, fm = O.dynamic_point_pool(rois, pts, extra, 1 << 30, 1 << 30, rb, pb)
    sel = rb[full_r] == pb[full_p]
    np.testing.assert_array_equal(pm, full_p[sel])
    np.testing.assert_array_equal(fm, full_f[sel])


def test_point_pool_oracle_matches_compiled_reference_live():
    from oracle import point_pool_oracle as O
    mod = build_ref.load_points_in_boxes()
    if mod is None:
        pytest.skip('oracle/_ref/points_in_boxes_ref.so not built (reference tree absent)')
    rng = np.random.default_rng(5)
    n_rois, n_pts = 120, 9000
    rois = np.concatenate([rng.uniform(-30, 30, (n_rois, 2)), rng.uniform(-2, 1, (n_rois, 1)),
                           rng.uniform(0.5, 6, (n_rois, 3)), rng.uniform(-7, 7, (n_rois, 1))], 1).astype(np.float32)
    k = rng.integers(0, n_rois, n_pts)
    pts = (rois[k, :3] + rng.normal(0, 1.5, (n_pts, 3))).astype(np.float32)
    flags = torch.zeros(n_rois, n_pts, dtype=torch.int32)
    mod.points_in_boxes_cpu(torch.from_numpy(rois), torch.from_numpy(pts), flags)
    lx, ly, lz = O.local_coords(rois, pts)
    mine = O.inside(lx, ly, lz, rois[:, 3], rois[:, 4], rois[:, 5])
    assert flags.sum() > 1000
    np.testing.assert_array_equal(mine, flags.numpy().astype(bool))


def spconv_case(g, tag):
    """(indices, batch, shape, ksize, stride, padding, dilation, subm, transpose) of a tests/golden/spconv.npz case"""
    p = g[f'in::{tag}::params'].tolist()
    return (g[f'in::{tag}::indices'], p[0], p[1:4], p[4:7], p[7:10], p[10:13], p[13:16], bool(p[16]), bool(p[17]))


def _rulebook_equal_by_coordinates(ref_outids, ref_pairs, ref_num, outids, pairs, num):
    assert (np.asarray(ref_num) == np.asarray(num)).all()
    assert set(map(tuple, ref_outids.tolist())) == set(map(tuple, outids.tolist())) and len(ref_outids) == len(outids)
    for k in range(len(num)):
        a = {(j, tuple(ref_outids[i])) for j, i in zip(ref_pairs[k, 0, :ref_num[k]].tolist(),
                                                       ref_pairs[k, 1, :ref_num[k]].tolist())}
        b = {(j, tuple(outids[i])) for j, i in zip(pairs[k, 0, :num[k]].tolist(), pairs[k, 1, :num[k]].tolist())}
        assert a == b, k
        assert (pairs[k, :, num[k]:] == -1).all()


@pytest.mark.parametrize('tag', ['subm3', 'down3s2', 'down_k313', 'down2s2', 'transposed', 'subm_dil2'])
def test_spconv_rulebook_oracle_matches_reference_golden(tag):
    """sparse-convolution rulebook restatement against the output of the reference's own CPU templates
    (tests/golden/make_golden.py::gen_spconv): same output voxels, same (input row, output voxel) pairs per kernel
    offset, same counts; outputs sorted by (b, z, y, x), pairs by input row."""
    from oracle import spconv_oracle as O
    g = load_golden('spconv.npz')
    ind, batch, shape, ks, st, pd, dl, subm, tr = spconv_case(g, tag)
    outids, pairs, num, out_shape = O.indice_pairs(ind, batch, shape, ks, st, pd, dl, (0, 0, 0), subm, tr)
    assert list(out_shape) == g[f'out::{tag}::out_shape'].tolist()
    _rulebook_equal_by_coordinates(g[f'out::{tag}::outids'], g[f'out::{tag}::pairs'], g[f'out::{tag}::num'], outids,
                                   pairs, num)
    if subm:
        np.testing.assert_array_equal(outids, ind)
    else:
        vol = int(np.prod(out_shape))
        lin = ((outids[:, 0].astype(np.int64) * out_shape[0] + outids[:, 1]) * out_shape[1] + outids[:, 2]) \
            * out_shape[2] + outids[:, 3]
        assert (np.diff(lin) > 0).all() and lin.max() < batch * vol
    for k in range(len(num)):
        assert (np.diff(pairs[k, 0, :num[k]]) > 0).all()
    in2out, out2in = O.maps_from_pairs(pairs, num, len(ind), len(outids))
    assert ((in2out >= 0).sum(1) == num).all() and ((out2in >= 0).sum(1) == num).all()


def test_spconv_rulebook_oracle_matches_compiled_reference_live():
    from oracle import spconv_oracle as O
    mod = build_ref.load_spconv_rulebook()
    if mod is None:
        pytest.skip('oracle/_ref/spconv_rulebook_ref.so not built (reference tree absent)')
    rng = np.random.default_rng(77)
    for n, batch, shape, ks, st, pd, dl, subm, tr in [(2500, 3, [10, 36, 28], [3, 3, 3], [1, 1, 1], [1, 1, 1], [1, 1, 1], True, False),
                                                      (2500, 3, [10, 36, 28], [3, 3, 3], [2, 2, 2], [1, 1, 1], [1, 1, 1], False, False),
                                                      (900, 1, [5, 14, 15], [3, 3, 3], [2, 2, 2], [1, 1, 1], [1, 1, 1], False, True),
                                                      (1800, 2, [9, 25, 25], [3, 3, 3], [1, 1, 1], [2, 2, 2], [2, 2, 2], False, False)]:
        vol = int(np.prod(shape))
        lin = rng.choice(batch * vol, n, replace=False)
        b, r = lin // vol, lin % vol
        ind = np.stack([b, r // (shape[1] * shape[2]), (r // shape[2]) % shape[1], r % shape[2]], 1).astype(np.int32)
        outids, pairs, num, out_shape = O.indice_pairs(ind, batch, shape, ks, st, pd, dl, (0, 0, 0), subm, tr)
        st_ref, pd_ref = ([1, 1, 1], [k // 2 for k in ks]) if subm else (st, pd)
        ro, rp, rn = mod.get_indice_pairs_3d(torch.from_numpy(ind), batch, out_shape, ks, st_ref, pd_ref, dl, subm, tr)
        _rulebook_equal_by_coordinates(ro.numpy(), rp.numpy(), rn.numpy(), outids, pairs, num)


def test_spconv_conv_oracle_forward_backward_consistency():
    """indice_conv restatement: against a dense float64 convolution (scipy-free: explicit loops over offsets on a
    dense canvas) and its backward against finite differences."""
    from oracle import spconv_oracle as O
    rng = np.random.default_rng(3)
    shape, batch, n, cin, cout = [6, 9, 8], 2, 300, 5, 4
    vol = int(np.prod(shape))
    lin = rng.choice(batch * vol, n, replace=False)
    b, r = lin // vol, lin % vol
    ind = np.stack([b, r // (shape[1] * shape[2]), (r // shape[2]) % shape[1], r % shape[2]], 1).astype(np.int32)
    x = rng.normal(size=(n, cin))
    w = rng.normal(size=(3, 3, 3, cin, cout))
    outids, pairs, num, out_shape = O.indice_pairs(ind, batch, shape, [3, 3, 3], [2, 2, 2], [1, 1, 1], [1, 1, 1])
    y = O.indice_conv(x, w, pairs, num, len(outids))
    dense = np.zeros([batch] + [s + 2 for s in shape] + [cin])
    dense[ind[:, 0], ind[:, 1] + 1, ind[:, 2] + 1, ind[:, 3] + 1] = x
    for i, (bb, z, yy, xx) in enumerate(outids.tolist()):
        patch = dense[bb, 2 * z:2 * z + 3, 2 * yy:2 * yy + 3, 2 * xx:2 * xx + 3]   # in = out * 2 - 1 + k
        np.testing.assert_allclose(y[i], np.einsum('zyxc,zyxco->o', patch, w), atol=1e-10)
    g = rng.normal(size=y.shape)
    dx, dw = O.indice_conv_backward(x, w, g, pairs, num)
    eps = 1e-6
    for idx in [(0, 0), (17, 3), (299, 4)]:
        xp = x.copy()
        xp[idx] += eps
        num_grad = ((O.indice_conv(xp, w, pairs, num, len(outids)) - y) * g).sum() / eps
        assert abs(num_grad - dx[idx]) < 1e-4
    wp = w.copy()
    wp[1, 2, 0, 3, 1] += eps
    num_grad = ((O.indice_conv(x, wp, pairs, num, len(outids)) - y) * g).sum() / eps
    assert abs(num_grad - dw[1, 2, 0, 3, 1]) < 1e-4


@pytest.mark.parametrize('tag', ['down3s2', 'down2s2', 'subm3'])
def test_spconv_maxpool_oracle_matches_reference_golden(tag):
    """sparse max pooling restatement against the reference's own CPU functors (src/maxpool.cc, golden): zero-filled
    start (negative maxima clip to 0), ties all receive the gradient.  Golden outputs are indexed by the CPU path's
    output numbering: compared through the output coordinates."""
    from oracle import spconv_oracle as O
    g = load_golden('spconv.npz')
    ind, batch, shape, ks, st, pd, dl, subm, tr = spconv_case(g, tag)
    outids, pairs, num, _ = O.indice_pairs(ind, batch, shape, ks, st, pd, dl, (0, 0, 0), subm, tr)
    x, gout_ref = g[f'in::{tag}::pool_features'], g[f'in::{tag}::pool_grad_out']
    ref_ids = g[f'out::{tag}::outids']
    pos = {tuple(c): i for i, c in enumerate(ref_ids.tolist())}
    perm = np.array([pos[tuple(c)] for c in outids.tolist()])      # my output row -> reference output row
    y = O.indice_maxpool(x, pairs, num, len(outids))
    np.testing.assert_array_equal(y, g[f'out::{tag}::pooled'][perm])
    assert (y >= 0).all() and (y == 0).any()
    dx = O.indice_maxpool_backward(x, y, gout_ref[perm], pairs, num)
    np.testing.assert_allclose(dx, g[f'out::{tag}::pool_grad_in'], atol=1e-5)


@pytest.mark.parametrize('tag', ['dense', 'all_kept'])
def test_hard_voxelize_oracle_matches_reference_golden(tag):
    """hard voxelization restated without the sequential loop against the reference's compiled C++ (golden):
    voxel order by first appearance, max_voxels cut, first max_points points per voxel."""
    g = load_golden('hard_voxelize.npz')
    prm = g[f'in::{tag}::params']
    vs, rng, mp, mv = prm[:3].tolist(), prm[3:9].tolist(), int(prm[9]), int(prm[10])
    voxels, coors, num = voxel_oracle.hard_voxelize(g[f'in::{tag}::points'], vs, rng, mp, mv)
    np.testing.assert_array_equal(coors, g[f'out::{tag}::coors'])
    np.testing.assert_array_equal(num, g[f'out::{tag}::num_points'])
    np.testing.assert_array_equal(voxels, g[f'out::{tag}::voxels'])
