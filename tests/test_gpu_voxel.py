"""GPU: dynamic voxelization and DynamicScatter parity (bit-exact indices, fp32 features) vs the oracle,
the golden vectors from the reference's C++, and size-independent properties at the full bench size."""
import numpy as np
import pytest
import torch

from conftest import PC_RANGE, VOXEL_SIZE, load_golden

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.mark.parametrize('case', ['sst', 'fsd', 'fsdv2'])
def test_voxelize_matches_reference_golden(case):
    import sst_amd
    g = load_golden('voxelize.npz')
    pts = torch.from_numpy(g[f'{case}::points']).to(DEV)
    coors = sst_amd.voxelization(pts, g[f'{case}::voxel_size'].tolist(), g[f'{case}::range'].tolist(), -1, -1)
    assert coors.dtype == torch.int32 and coors.shape == (pts.size(0), 3)
    np.testing.assert_array_equal(coors.cpu().numpy(), g[f'{case}::coors'])


def test_voxelize_module_and_empty_and_strided():
    import sst_amd
    from oracle import voxel_oracle
    vox = sst_amd.Voxelization(VOXEL_SIZE, PC_RANGE, -1, (-1, -1))
    empty = vox(torch.zeros((0, 5), device=DEV))
    assert empty.shape == (0, 3) and empty.dtype == torch.int32
    g = torch.Generator().manual_seed(5)
    pts = (torch.rand(1001, 7, generator=g) * 200 - 100)
    out = vox(pts.to(DEV))
    np.testing.assert_array_equal(out.cpu().numpy(), voxel_oracle.dynamic_voxelize(pts.numpy(), VOXEL_SIZE, PC_RANGE))
    # batched helper writes (b,z,y,x) rows in place
    pts2 = (torch.rand(77, 7, generator=g) * 100 - 50)
    cat, coors = vox.voxelize_batch([pts.to(DEV), pts2.to(DEV)])
    assert cat.shape == (1078, 7)
    ref = np.concatenate([np.pad(voxel_oracle.dynamic_voxelize(p.numpy(), VOXEL_SIZE, PC_RANGE), ((0, 0), (1, 0)),
                                 constant_values=b) for b, p in enumerate((pts, pts2))])
    np.testing.assert_array_equal(coors.cpu().numpy(), ref)


@pytest.mark.parametrize('ncol', [3, 4, 5, 6, 7, 8, 9])
@pytest.mark.parametrize('n', [4096, 4097, 20000, 116001])
def test_voxelize_streaming_rows_kernel_bit_exact(n, ncol):
    """the LDS-staged (b, z, y, x)-row kernel taken by voxelize_batch for rows of up to 8 floats and n >= 4096, the
    plain kernel beyond (9 columns) and for the 3-column output of Voxelization.forward: all bit-exact vs the oracle,
    including out-of-range points (clamped) and a second sample (batch column, unaligned sample offset)."""
    import sst_amd
    from oracle import voxel_oracle
    g = torch.Generator().manual_seed(n * 16 + ncol)
    pts = torch.rand(n, ncol, generator=g) * 200 - 100
    pts[:, 2] = torch.rand(n, generator=g) * 10 - 4
    pts2 = torch.rand(4099, ncol, generator=g) * 160 - 80
    vox = sst_amd.Voxelization(VOXEL_SIZE, PC_RANGE, -1, (-1, -1))
    _, coors = vox.voxelize_batch([pts.to(DEV), pts2.to(DEV)])
    ref = np.concatenate([np.pad(voxel_oracle.dynamic_voxelize(p.numpy(), VOXEL_SIZE, PC_RANGE), ((0, 0), (1, 0)),
                                 constant_values=b) for b, p in enumerate((pts, pts2))])
    np.testing.assert_array_equal(coors.cpu().numpy(), ref)
    np.testing.assert_array_equal(vox(pts.to(DEV)).cpu().numpy(), ref[:n, 1:])


def test_voxelize_full_size_bit_exact_and_idempotent():
    import sst_amd
    from oracle import voxel_oracle
    g = torch.Generator().manual_seed(0)
    n = 116000
    pts = torch.rand(n, 3, generator=g) * torch.tensor([149.76, 149.76, 6.0]) + torch.tensor([-74.88, -74.88, -2.0])
    out = sst_amd.voxelization(pts.to(DEV), VOXEL_SIZE, PC_RANGE, -1, -1).cpu().numpy()
    np.testing.assert_array_equal(out, voxel_oracle.dynamic_voxelize(pts.numpy(), VOXEL_SIZE, PC_RANGE))
    assert out.min() >= 0 and out[:, 1].max() <= 467 and out[:, 2].max() <= 467 and out[:, 0].max() == 0
    # voxel centres map back to their own voxel (idempotence of the grid)
    centers = np.stack([(out[:, 2] + 0.5) * 0.32 - 74.88, (out[:, 1] + 0.5) * 0.32 - 74.88,
                        np.ones(n)], 1).astype(np.float32)
    again = sst_amd.voxelization(torch.from_numpy(centers).to(DEV), VOXEL_SIZE, PC_RANGE, -1, -1).cpu().numpy()
    np.testing.assert_array_equal(again, out)


# ------------------------------------------------------------------------------------------------
def _rand_scatter_inputs(n, c, lo, hi, seed):
    g = torch.Generator().manual_seed(seed)
    feats = torch.rand(n, c, generator=g) * 100 - 50
    coors = torch.randint(lo, hi, (n, 3), dtype=torch.int32, generator=g)
    return feats, coors


@pytest.mark.parametrize('mode', ['max', 'mean', 'sum'])
@pytest.mark.parametrize('compat', [True, False])
def test_dynamic_scatter_forward_matches_oracle(mode, compat):
    import sst_amd
    from oracle import voxel_oracle
    feats, coors = _rand_scatter_inputs(20000, 5, -1, 20, 3)
    red, oc, cmap, cnt = sst_amd.dynamic_point_to_voxel_forward(feats.to(DEV), coors.to(DEV), mode,
                                                                reference_compat=compat)
    r_red, r_oc, r_map, r_cnt = voxel_oracle.dynamic_point_to_voxel_forward(feats, coors, mode, compat)
    np.testing.assert_array_equal(oc.cpu().numpy(), r_oc.numpy())
    np.testing.assert_array_equal(cmap.cpu().numpy(), r_map.numpy())
    np.testing.assert_array_equal(cnt.cpu().numpy(), r_cnt.numpy())
    assert oc.dtype == torch.int32 and cmap.dtype == torch.int32 and cnt.dtype == torch.int32
    if mode == 'max':
        np.testing.assert_array_equal(red.cpu().numpy(), r_red.numpy())     # max is exact
    else:
        np.testing.assert_allclose(red.cpu().numpy(), r_red.numpy(), rtol=1e-5, atol=1e-4)


def test_dynamic_scatter_reference_test_construction():
    """tests/test_models/test_voxel_encoder/test_dynamic_scatter.py:8-93 (200k points, coords in [-1,20))."""
    import sst_amd
    feats, coors = _rand_scatter_inputs(200000, 3, -1, 20, 0)
    dsmean = sst_amd.DynamicScatter([0.32, 0.32, 6], [-74.88, -74.88, -2, 74.88, 74.88, 4], True)
    dsmax = sst_amd.DynamicScatter([0.32, 0.32, 6], [-74.88, -74.88, -2, 74.88, 74.88, 4], False)
    # empty input and all-invalid input (:22-53)
    ef, ec = dsmean(torch.rand(0, 3, device=DEV), torch.randint(0, 1, (0, 3), dtype=torch.int32, device=DEV))
    assert ef.shape == (0, 3) and ec.shape == (0, 3)
    inv_c = -torch.ones((200, 3), dtype=torch.int32, device=DEV)
    ef, ec = dsmax(torch.rand(200, 3, device=DEV), inv_c)
    assert ef.shape == (0, 3) and ec.shape == (0, 3)
    # the reference test's brute-force construction (:56-65) for EVERY voxel, vectorised: per-voxel mean in float64
    # and per-voxel max over the points whose coordinate row equals the voxel's.  Bar: max exact,
    # mean within 1e-5 relative (to the summands' magnitude, 50, where the mean itself is close to zero)
    fm, cm = dsmean(feats.to(DEV), coors.to(DEV))
    fx, cx = dsmax(feats.to(DEV), coors.to(DEV))
    valid = coors.min(dim=-1).values >= 0
    ref_coors, inv = coors[valid].unique(dim=0, sorted=True, return_inverse=True)
    np.testing.assert_array_equal(cm.cpu().numpy(), ref_coors.numpy())
    np.testing.assert_array_equal(cx.cpu().numpy(), ref_coors.numpy())
    m = ref_coors.size(0)
    fv = feats[valid]
    cnt = torch.zeros(m, dtype=torch.float64).index_add_(0, inv, torch.ones(inv.numel(), dtype=torch.float64))
    ref_mean = torch.zeros((m, 3), dtype=torch.float64).index_add_(0, inv, fv.double()) / cnt[:, None]
    ref_max = torch.full((m, 3), -float('inf')).scatter_reduce_(0, inv[:, None].expand(-1, 3), fv, 'amax')
    assert torch.equal(fx.cpu(), ref_max)
    np.testing.assert_allclose(fm.cpu().double().numpy(), ref_mean.numpy(), rtol=1e-5, atol=1e-5 * 50)
    # and the literal per-voxel loop of the reference test on a subsample
    fm, fx = fm.cpu(), fx.cpu()
    for vi in range(0, m, 397):
        sel = feats[(coors == ref_coors[vi]).all(dim=-1)]
        assert torch.allclose(fm[vi], sel.mean(0), rtol=1e-5, atol=5e-4)
        assert torch.equal(fx[vi], sel.max(0).values)


@pytest.mark.parametrize('mode', ['max', 'mean', 'sum'])
def test_dynamic_scatter_backward_matches_oracle(mode):
    import sst_amd
    from oracle import voxel_oracle
    feats, coors = _rand_scatter_inputs(5000, 4, -1, 6, 11)
    feats = torch.round(feats)  # ties on purpose: exercises the smallest-index rule of the max backward
    f_gpu = feats.to(DEV).requires_grad_(True)
    red, _ = sst_amd.dynamic_scatter(f_gpu, coors.to(DEV), mode)
    g = torch.Generator().manual_seed(12)
    gout = torch.rand(red.shape, generator=g)
    (red * gout.to(DEV)).sum().backward()
    f_cpu = feats.clone().requires_grad_(True)
    r_red, _ = voxel_oracle.dynamic_scatter(f_cpu, coors, mode)
    (r_red * gout).sum().backward()
    np.testing.assert_allclose(f_gpu.grad.cpu().numpy(), f_cpu.grad.numpy(), rtol=1e-6, atol=1e-6)


def test_dynamic_scatter_batched_equals_per_sample_loop():
    import sst_amd
    from oracle import voxel_oracle
    g = torch.Generator().manual_seed(21)
    feats = torch.rand(9000, 6, generator=g)
    coors = torch.cat([torch.sort(torch.randint(0, 3, (9000, 1), generator=g), 0)[0],
                       torch.randint(0, 9, (9000, 3), generator=g)], 1).int()
    for avg in (True, False):
        ours = sst_amd.DynamicScatter(VOXEL_SIZE, PC_RANGE, avg)
        ref = voxel_oracle.DynamicScatterOracle(VOXEL_SIZE, PC_RANGE, avg)
        vf, vc = ours(feats.to(DEV), coors.to(DEV))
        rf, rc = ref(feats, coors)
        np.testing.assert_array_equal(vc.cpu().numpy(), rc.numpy())
        np.testing.assert_allclose(vf.cpu().numpy(), rf.numpy(), rtol=1e-5, atol=1e-5)


def test_dynamic_vfe_matches_reference_golden():
    import sst_amd
    g = load_golden('dynamic_vfe.npz')
    vfe = sst_amd.build_voxel_encoder(dict(
        type='DynamicVFE', in_channels=3, feat_channels=[64, 128], with_distance=False, voxel_size=VOXEL_SIZE,
        with_cluster_center=True, with_voxel_center=True, point_cloud_range=PC_RANGE,
        norm_cfg=dict(type='naiveSyncBN1d', eps=1e-3, momentum=0.01)))
    sd = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith('w::')}
    missing = vfe.load_state_dict(sd, strict=True)
    vfe.to(DEV).train()
    pts = torch.from_numpy(g['in::points']).to(DEV).requires_grad_(True)
    coors = torch.from_numpy(g['in::coors']).to(DEV)
    vf, vc = vfe(pts, coors)
    np.testing.assert_array_equal(vc.cpu().numpy(), g['out::voxel_coors'])
    np.testing.assert_allclose(vf.detach().cpu().numpy(), g['out::voxel_feats'], rtol=1e-3, atol=1e-3)
    (vf * torch.from_numpy(g['in::grad_out']).to(DEV)).sum().backward()
    np.testing.assert_allclose(pts.grad.cpu().numpy(), g['out::grad_points'], rtol=1e-3, atol=2e-3)


def test_scatter_full_size_properties():
    """N=116k bench cloud: sortedness, inverse consistency, checksum of sums (linearity of 'sum')."""
    import sst_amd
    g = torch.Generator().manual_seed(0)
    n = 116000
    pts = torch.rand(n, 3, generator=g) * torch.tensor([149.76, 149.76, 6.0]) + torch.tensor([-74.88, -74.88, -2.0])
    pts = pts.to(DEV)
    coors = sst_amd.voxelization(pts, VOXEL_SIZE, PC_RANGE, -1, -1)
    plan = sst_amd.build_scatter_plan(coors, reference_compat=False)
    vc = plan.voxel_coors.long()
    key = (vc[:, 0] * 468 + vc[:, 1]) * 468 + vc[:, 2]
    assert bool((key[1:] > key[:-1]).all()), 'voxel coordinates must be strictly increasing (sorted unique)'
    assert torch.equal(vc[plan.coors_map.long()].int(), coors)
    assert int(plan.reduce_count.sum()) == n
    feats = torch.rand(n, 16, generator=g).to(DEV)
    s = plan.reduce(feats, 'sum')
    assert torch.allclose(s.sum(0), feats.sum(0), rtol=1e-4)
    mx = plan.reduce(feats, 'max')
    assert bool((mx[plan.coors_map.long()] >= feats).all())
    assert 85000 < plan.num_voxels < 95000


@pytest.mark.parametrize('tag', ['dense', 'all_kept'])
def test_hard_voxelize_matches_reference_golden(tag):
    """Voxelization with max_num_points / max_voxels set (voxelize.py:47-58): bit-exact against the reference's
    compiled C++ (tests/golden/hard_voxelize.npz)"""
    import sst_amd
    g = load_golden('hard_voxelize.npz')
    prm = g[f'in::{tag}::params']
    vs, rng, mp, mv = prm[:3].tolist(), prm[3:9].tolist(), int(prm[9]), int(prm[10])
    pts = torch.from_numpy(g[f'in::{tag}::points']).to(DEV)
    voxels, coors, num = sst_amd.voxelization(pts, vs, rng, mp, mv)
    assert coors.dtype == torch.int32 and num.dtype == torch.int32 and voxels.shape[1:] == (mp, 4)
    np.testing.assert_array_equal(coors.cpu().numpy(), g[f'out::{tag}::coors'])
    np.testing.assert_array_equal(num.cpu().numpy(), g[f'out::{tag}::num_points'])
    np.testing.assert_array_equal(voxels.cpu().numpy(), g[f'out::{tag}::voxels'])
    layer = sst_amd.Voxelization(vs, rng, mp, (mv, mv)).eval()
    v2, c2, n2 = layer(pts)
    assert torch.equal(v2, voxels) and torch.equal(c2, coors) and torch.equal(n2, num)
