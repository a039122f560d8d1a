"""GPU: DynamicVFE's two-layer stack as one node over fused passes (sst_amd/vfe_fused.py) against the
same module run layer by layer (fused_stack = False: Linear -> batch_norm_act -> segment_reduce -> concat_gather, the path the
reference-produced golden and the CPU port pin), on uniform and crowded clouds, one and two samples, both drop conventions,
training and evaluation mode; the pieces of the fused node against float64."""
import copy

import numpy as np
import pytest
import torch

from conftest import PC_RANGE, VOXEL_SIZE, load_golden

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _vfe(reference_compat=True):
    import sst_amd
    torch.manual_seed(3)
    vfe = sst_amd.build_voxel_encoder(dict(
        type='DynamicVFE', in_channels=3, feat_channels=[64, 128], with_distance=False, voxel_size=VOXEL_SIZE,
        with_cluster_center=True, with_voxel_center=True, point_cloud_range=PC_RANGE,
        norm_cfg=dict(type='naiveSyncBN1d', eps=1e-3, momentum=0.01), reference_compat=reference_compat))
    with torch.no_grad():
        for l in vfe.vfe_layers:
            l.norm.weight.uniform_(0.5, 1.5)
            l.norm.bias.uniform_(-0.3, 0.3)
    return vfe.to(DEV)


def _clouds(kind, n):
    import bench
    if kind == 'uniform':
        return [bench.make_cloud(n, 3, DEV)]
    if kind == 'two_samples':
        return [bench.make_cloud(n, 3, DEV), bench.make_cloud(n // 3, 4, DEV)]
    return [bench.make_lidar_cloud(5, DEV, beams=32, azimuth_steps=max(64, n // 32))]


@pytest.mark.parametrize('train', [True, False])
@pytest.mark.parametrize('reference_compat', [True, False])
@pytest.mark.parametrize('kind,n', [('uniform', 30000), ('two_samples', 20000), ('lidar', 40000), ('uniform', 300)])
def test_fused_stack_equals_the_layerwise_modules(kind, n, reference_compat, train):
    import sst_amd
    fused = _vfe(reference_compat)
    plain = copy.deepcopy(fused)
    plain.fused_stack = False
    fused.train(train), plain.train(train)
    vox = sst_amd.Voxelization(VOXEL_SIZE, PC_RANGE, -1, (-1, -1))
    pts, coors = vox.voxelize_batch(_clouds(kind, n))
    plan = fused.scatter_plan(coors)
    out_f, vc_f = fused(pts, coors, scatter_plan=plan)
    out_p, vc_p = plain(pts, coors, scatter_plan=plan)
    assert torch.equal(vc_f, vc_p)
    assert out_f.shape == out_p.shape and out_f.size(0) == plan.num_voxels
    scale = float(out_p.detach().abs().max())
    assert float((out_f - out_p).detach().abs().max()) <= 2e-5 * max(scale, 1.0)
    for a, b in zip(fused.vfe_layers, plain.vfe_layers):
        assert torch.allclose(a.norm.running_mean, b.norm.running_mean, rtol=1e-5, atol=1e-6)
        assert torch.allclose(a.norm.running_var, b.norm.running_var, rtol=1e-5, atol=1e-6)
        assert int(a.norm.num_batches_tracked) == int(b.norm.num_batches_tracked)
    gen = torch.Generator(device=DEV).manual_seed(1)
    g = torch.randn(out_p.shape, device=DEV, generator=gen)
    (out_f * g).sum().backward()
    (out_p * g).sum().backward()
    for (name, pa), (_, pb) in zip(fused.named_parameters(), plain.named_parameters()):
        assert pa.grad is not None and pb.grad is not None, name
        ref = float(pb.grad.abs().max())
        err = float((pa.grad - pb.grad).abs().max())
        # The two paths round the products differently (FMA chain / split-weight sum against the library GEMM over the
        # concatenation), so a pre-activation within rounding of zero, or two points of a voxel within rounding of each other,
        # may be decided differently: ONE such decision moves a gradient by one term of its sum (measured: 1 decision in
        # 3.8 M values, 0.7 of a bias gradient of 315).  The bound is tests/test_gpu_end_to_end.py's for such rows; that the
        # routing itself is exact is test_fused_stack_routes_gradients_exactly (integer data, no rounding anywhere).
        assert err <= 1e-2 * max(ref, 1e-6), (name, err, ref)
    # same answer every time (no float atomics anywhere in the node)
    for p in fused.parameters():
        p.grad = None
    if train:
        return
    out2, _ = fused(pts, coors, scatter_plan=plan)
    assert torch.equal(out2, out_f)


@pytest.mark.parametrize('kind,n', [('uniform', 20000), ('two_samples', 9000), ('lidar', 30000)])
def test_fused_stack_routes_gradients_exactly(kind, n):
    """Integer-valued inputs, weights, norm scale / shift (evaluation mode: invstd = 1) and upstream gradient: every product and
    sum of both paths is exact in fp32 (|values| < 2^24), so the fused node and the layer-wise modules must agree BIT FOR BIT -
    pooled values, arg-max routing (ties included: thousands of equal values per voxel), the hand-back of the pooled feature to
    the points of discarded voxels, the split weight - in the output and in every parameter gradient."""
    import sst_amd
    from sst_amd import voxel_encoder as VE
    from sst_amd.vfe_fused import fused_vfe2
    fused = _vfe(True).eval()
    gen = torch.Generator().manual_seed(n)
    with torch.no_grad():
        for l in fused.vfe_layers:
            l.linear.weight.copy_(torch.randint(-2, 3, l.linear.weight.shape, generator=gen).float())
            l.norm.weight.copy_(torch.randint(1, 3, l.norm.weight.shape, generator=gen).float())
            l.norm.bias.copy_(torch.randint(-3, 4, l.norm.bias.shape, generator=gen).float())
            l.norm.running_mean.copy_(torch.randint(-2, 3, l.norm.running_mean.shape, generator=gen).float())
            l.norm.running_var.fill_(1.0)
            l.norm.eps = 0.0                                      # invstd = rsqrt(1) = 1 exactly
    plain = copy.deepcopy(fused)
    vox = sst_amd.Voxelization(VOXEL_SIZE, PC_RANGE, -1, (-1, -1))
    pts, coors = vox.voxelize_batch(_clouds(kind, n))
    plan = fused.scatter_plan(coors)
    x = torch.randint(-3, 4, (pts.size(0), fused.vfe_layers[0].linear.in_features), generator=gen).float().to(DEV)
    grouping = VE._VoxelGrouping(plan)
    out_f = fused_vfe2(fused, x, plan)
    _, pooled = plain._encode(x, grouping, 'max')
    out_p = pooled[-1]
    assert float(out_p.detach().abs().max()) < 2 ** 22
    assert torch.equal(out_f, out_p)
    g = torch.randint(-2, 3, out_p.shape, generator=gen).float().to(DEV)
    out_f.backward(g)
    out_p.backward(g)
    for (name, pa), (_, pb) in zip(fused.named_parameters(), plain.named_parameters()):
        assert float(pb.grad.abs().max()) < 2 ** 24, name
        assert torch.equal(pa.grad, pb.grad), (name, float((pa.grad - pb.grad).abs().max()))


def test_fused_stack_is_bit_reproducible_in_training():
    import sst_amd
    fused = _vfe().train()
    vox = sst_amd.Voxelization(VOXEL_SIZE, PC_RANGE, -1, (-1, -1))
    pts, coors = vox.voxelize_batch(_clouds('lidar', 60000))
    plan = fused.scatter_plan(coors)
    runs = []
    for _ in range(3):
        for p in fused.parameters():
            p.grad = None
        out, _ = fused(pts, coors, scatter_plan=plan)
        out.backward(torch.ones_like(out) * torch.arange(1, 129, device=DEV))
        runs.append((out.detach().clone(), [p.grad.clone() for p in fused.parameters()]))
    for out, grads in runs[1:]:
        assert torch.equal(out, runs[0][0])
        for a, b in zip(grads, runs[0][1]):
            assert torch.equal(a, b)


def test_fused_stack_matches_the_reference_golden():
    """the forward of tests/golden/dynamic_vfe.npz (the reference's own DynamicVFE) through the fused node (points without
    a gradient: the golden's gradient check keeps running through the layer-wise path, tests/test_gpu_voxel.py)"""
    import sst_amd
    g = load_golden('dynamic_vfe.npz')
    vfe = sst_amd.build_voxel_encoder(dict(
        type='DynamicVFE', in_channels=3, feat_channels=[64, 128], with_distance=False, voxel_size=VOXEL_SIZE,
        with_cluster_center=True, with_voxel_center=True, point_cloud_range=PC_RANGE,
        norm_cfg=dict(type='naiveSyncBN1d', eps=1e-3, momentum=0.01)))
    vfe.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith('w::')}, strict=True)
    vfe.to(DEV).train()
    pts = torch.from_numpy(g['in::points']).to(DEV)
    coors = torch.from_numpy(g['in::coors']).to(DEV)
    vf, vc = vfe(pts, coors)
    assert vf.grad_fn is not None and 'FusedVFE2' in type(vf.grad_fn).__name__
    np.testing.assert_array_equal(vc.cpu().numpy(), g['out::voxel_coors'])
    np.testing.assert_allclose(vf.detach().cpu().numpy(), g['out::voxel_feats'], rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize('m', [5000, 116000])
def test_small_k_linear_and_its_moments(m):
    """y = x W^T for K = 10 with the batch-norm partials from the same pass, against float64"""
    from sst_amd import _lib
    from sst_amd.norm import bn_prepare
    torch.manual_seed(m)
    x = torch.randn(m, 10, device=DEV) * torch.tensor([50.0, 50, 3, 1, 1, 1, 0.2, 0.2, 0.2, 1.0], device=DEV)
    w = torch.randn(64, 10, device=DEV) * 0.3
    lib = _lib.load()
    y = torch.empty(m, 64, device=DEV)
    ws = _lib.workspace(lib.sst_bn_workspace_bytes(m, 64), x.device)
    rc = lib.sst_vfe_linear_moments_f32(_lib.ptr(x), 10, m, 10, _lib.ptr(w), 10, 64, _lib.ptr(y), 64, _lib.ptr(ws), _lib.stream_ptr())
    assert rc == 0
    ref = x.double() @ w.double().t()
    assert float((y.double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
    bn = torch.nn.BatchNorm1d(64, eps=1e-3, momentum=0.01).to(DEV).train()
    prep, batch_stats, count, sync = bn_prepare(bn, y, partials=ws)
    assert batch_stats and count == float(m) and not sync
    mean, var = y.double().mean(0), y.double().var(0, unbiased=False)
    assert torch.allclose(prep[0].double(), mean, rtol=1e-5, atol=1e-5)
    assert torch.allclose(prep[1].double(), (var + 1e-3).rsqrt(), rtol=1e-5)
    assert torch.allclose(bn.running_mean.double(), 0.01 * mean, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize('m,rows', [(40000, 9000), (116000, 90107), (33, 7)])
def test_split_weight_second_layer(m, rows):
    """pf W[:, :64]^T + (pooled W[:, 64:]^T)[index] == cat([pf, pooled[index]]) W^T (negative index: row 0)"""
    import ctypes
    from sst_amd import _lib
    torch.manual_seed(rows)
    pf = torch.randn(m, 64, device=DEV)
    pooled = torch.randn(rows, 64, device=DEV)
    w = torch.randn(128, 128, device=DEV) * 0.2
    idx = torch.randint(-1, rows, (m,), device=DEV, dtype=torch.int32)
    lib = _lib.load()
    t = torch.empty(rows, 128, device=DEV)
    rc = lib.sst_tall_linear_epi_f32x6(_lib.ptr(pooled), 64, ctypes.c_void_p(w.data_ptr() + 256), 128, 0, None, rows, 64, 128, 0,
                                       None, None, 0, _lib.ptr(t), 128, _lib.stream_ptr())
    assert rc == 0
    y = torch.empty(m, 128, device=DEV)
    rc = lib.sst_tall_linear_add_rows_f32x6(_lib.ptr(pf), 64, _lib.ptr(w), 128, m, 64, 128, _lib.ptr(t), 128, _lib.ptr(idx),
                                            _lib.ptr(y), 128, _lib.stream_ptr())
    assert rc == 0
    ref = torch.cat([pf, pooled[idx.long().clamp(min=0)]], 1).double() @ w.double().t()
    f32 = torch.cat([pf, pooled[idx.long().clamp(min=0)]], 1) @ w.t()
    err = float((y.double() - ref).abs().max())
    assert err <= 2.0 * max(float((f32.double() - ref).abs().max()), 1e-6), err
    # the data gradient shapes: (128 -> 64) with the transposed weight halves
    dy = torch.randn(m, 128, device=DEV)
    for off in (0, 64):
        dx = torch.empty(m, 64, device=DEV)
        rc = lib.sst_tall_linear_epi_f32x6(_lib.ptr(dy), 128, ctypes.c_void_p(w.data_ptr() + 4 * off), 128, 1, None, m, 128, 64, 0,
                                           None, None, 0, _lib.ptr(dx), 64, _lib.stream_ptr())
        assert rc == 0
        ref = dy.double() @ w[:, off:off + 64].double()
        f32 = dy @ w[:, off:off + 64]
        err = float((dx.double() - ref).abs().max())
        assert err <= 2.0 * max(float((f32.double() - ref).abs().max()), 1e-6), (off, err)


@pytest.mark.parametrize('feat_channels', [[64, 64], [64, 128]])
@pytest.mark.parametrize('train', [True, False])
def test_scatter_vfe_fused_equals_layerwise(feat_channels, train):
    """DynamicScatterVFE (FSD's voxel encoder: plain sorted-unique grouping, every group kept; configs/fsd: [64, 64],
    FSDv2's virtual-voxel encoder: [64, 128]) through the fused node against its layer-wise modules"""
    import sst_amd
    torch.manual_seed(5)
    vs, rng = (0.25, 0.25, 0.2), [-40.0, -40.0, -2.0, 40.0, 40.0, 4.0]
    fused = sst_amd.build_voxel_encoder(dict(
        type='DynamicScatterVFE', in_channels=5, feat_channels=feat_channels, voxel_size=vs, with_cluster_center=True,
        with_voxel_center=True, point_cloud_range=rng, norm_cfg=dict(type='naiveSyncBN1d', eps=1e-3, momentum=0.01),
        unique_once=True)).to(DEV)
    plain = copy.deepcopy(fused)
    plain.fused_stack = False
    fused.train(train), plain.train(train)
    n = 50000
    pts = torch.rand(n, 5, device=DEV) * torch.tensor([80.0, 80.0, 6.0, 1.0, 1.0], device=DEV) \
        + torch.tensor([-40.0, -40.0, -2.0, 0.0, 0.0], device=DEV)
    pts[:8000, :3] = pts[:8000, :3] * 0.02 + 3.0                                  # a crowded corner: long groups
    coors = torch.cat([torch.zeros(n, 1, device=DEV), ((pts[:, [2, 1, 0]] - torch.tensor([-2.0, -40.0, -40.0], device=DEV))
                                                         / torch.tensor([0.2, 0.25, 0.25], device=DEV)).floor()], 1).long()
    out_f, vc_f, inv_f = fused(pts, coors, return_inv=True)
    out_p, vc_p, inv_p = plain(pts, coors, return_inv=True)
    assert 'FusedVFE2' in type(out_f.grad_fn).__name__ and 'FusedVFE2' not in type(out_p.grad_fn).__name__
    assert torch.equal(vc_f, vc_p) and torch.equal(inv_f, inv_p) and out_f.shape == out_p.shape
    scale = float(out_p.detach().abs().max())
    assert float((out_f - out_p).detach().abs().max()) <= 2e-5 * max(scale, 1.0)
    g = torch.randn(out_p.shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(2))
    (out_f * g).sum().backward()
    (out_p * g).sum().backward()
    for (name, pa), (_, pb) in zip(fused.named_parameters(), plain.named_parameters()):
        ref = float(pb.grad.abs().max())
        assert float((pa.grad - pb.grad).abs().max()) <= 1e-2 * max(ref, 1e-6), name    # see the DynamicVFE case above


@pytest.mark.parametrize('c1', [64, 128])
def test_scatter_vfe_fused_routes_gradients_exactly(c1):
    """the bit-for-bit routing check (integer data, evaluation-mode norms with unit variance) on the sorted-unique grouping"""
    import sst_amd
    from sst_amd import voxel_encoder as VE
    from sst_amd.vfe_fused import UniquePlanAdapter, fused_vfe2
    torch.manual_seed(c1)
    fused = sst_amd.build_voxel_encoder(dict(
        type='DynamicScatterVFE', in_channels=5, feat_channels=[64, c1], voxel_size=(0.25, 0.25, 0.2), with_cluster_center=True,
        with_voxel_center=True, point_cloud_range=[-40.0, -40.0, -2.0, 40.0, 40.0, 4.0],
        norm_cfg=dict(type='naiveSyncBN1d', eps=1e-3, momentum=0.01), unique_once=True)).to(DEV).eval()
    gen = torch.Generator().manual_seed(c1)
    with torch.no_grad():
        for l in fused.vfe_layers:
            l.linear.weight.copy_(torch.randint(-2, 3, l.linear.weight.shape, generator=gen).float())
            l.norm.weight.copy_(torch.randint(1, 3, l.norm.weight.shape, generator=gen).float())
            l.norm.bias.copy_(torch.randint(-3, 4, l.norm.bias.shape, generator=gen).float())
            l.norm.running_mean.copy_(torch.randint(-2, 3, l.norm.running_mean.shape, generator=gen).float())
            l.norm.running_var.fill_(1.0)
            l.norm.eps = 0.0
    plain = copy.deepcopy(fused)
    n = 30000
    ids = torch.cat([torch.randint(0, 9000, (n - 6000,), generator=gen), torch.full((6000,), 17)])      # one group of 6 000
    coors = torch.stack([torch.zeros(n, dtype=torch.long), ids // 100, ids % 100, torch.zeros(n, dtype=torch.long)], 1).to(DEV)
    grouping = VE._UniqueGrouping(coors)
    x = torch.randint(-3, 4, (n, fused.vfe_layers[0].linear.in_features), generator=gen).float().to(DEV)
    out_f = fused_vfe2(fused, x, UniquePlanAdapter(grouping.plan))
    _, pooled = plain._encode(x, grouping, 'max')
    assert torch.equal(out_f, pooled[-1])
    g = torch.randint(-2, 3, out_f.shape, generator=gen).float().to(DEV)
    out_f.backward(g)
    pooled[-1].backward(g)
    for (name, pa), (_, pb) in zip(fused.named_parameters(), plain.named_parameters()):
        assert float(pb.grad.abs().max()) < 2 ** 24, name
        assert torch.equal(pa.grad, pb.grad), (name, float((pa.grad - pb.grad).abs().max()))
