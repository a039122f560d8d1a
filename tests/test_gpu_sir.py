"""GPU: scatter_v2 / SIRLayer / SIR (FSD point-group MLP + scatter-max) vs the oracle and golden tensors
from the reference's own Python."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.mark.parametrize('mode', ['max', 'mean', 'sum', 'avg'])
def test_scatter_v2_matches_oracle(mode):
    import sst_amd
    from oracle import sst_oracle
    g = torch.Generator().manual_seed(3)
    n = 30000
    coors = torch.stack([torch.randint(0, 3, (n,), generator=g), torch.randint(0, 2, (n,), generator=g),
                         torch.randint(0, 600, (n,), generator=g)], 1)
    feat = torch.randn(n, 67, generator=g)
    fg = feat.to(DEV).requires_grad_(True)
    new_feat, new_coors, inv = sst_amd.scatter_v2(fg, coors.to(DEV), mode)
    uniq, rinv, cnt = sst_oracle.unique_rows(coors.numpy())
    np.testing.assert_array_equal(new_coors.cpu().numpy(), uniq)
    np.testing.assert_array_equal(inv.cpu().numpy(), rinv)
    assert inv.dtype == torch.int64 and new_coors.dtype == torch.int64
    ref = sst_oracle.segment_reduce(feat.numpy(), rinv, len(uniq), mode)
    np.testing.assert_allclose(new_feat.detach().cpu().numpy(), ref, rtol=1e-5, atol=1e-5)
    # reuse of the grouping (unique_once) gives the same result, and gradients match autograd on the oracle form
    again, _, _ = sst_amd.scatter_v2(fg, coors.to(DEV), mode, unq_inv=inv, new_coors=new_coors)
    assert torch.equal(again, new_feat)
    gout = torch.randn(new_feat.shape, generator=g)
    (new_feat * gout.to(DEV)).sum().backward()
    f2 = feat.clone().double().requires_grad_(True)
    idx = torch.from_numpy(rinv).view(-1, 1).expand(-1, 67)
    red = {'max': 'amax', 'mean': 'mean', 'avg': 'mean', 'sum': 'sum'}[mode]
    base = torch.zeros(len(uniq), 67, dtype=torch.float64)
    out2 = base.scatter_reduce(0, idx, f2, reduce=red, include_self=False)
    (out2 * gout.double()).sum().backward()
    if mode == 'max':   # random floats: no ties, so amax's tie rule is irrelevant
        np.testing.assert_allclose(fg.grad.cpu().numpy(), f2.grad.numpy(), rtol=1e-5, atol=1e-6)
    else:
        np.testing.assert_allclose(fg.grad.cpu().numpy(), f2.grad.numpy(), rtol=1e-5, atol=1e-6)


def test_scatter_v2_min_points_and_foreign_inverse():
    import sst_amd
    g = torch.Generator().manual_seed(4)
    n = 5000
    coors = torch.randint(0, 40, (n, 3), generator=g)
    feat = torch.randn(n, 8, generator=g)
    nf, nc, inv = sst_amd.scatter_v2(feat.to(DEV), coors.to(DEV), 'avg', min_points=3)
    uniq, rinv, cnt = torch.unique(coors, dim=0, return_inverse=True, return_counts=True)
    valid = cnt[rinv] >= 3
    u2, i2 = torch.unique(coors[valid], dim=0, return_inverse=True)
    assert torch.equal(nc.cpu(), u2) and torch.equal(inv.cpu(), i2)
    # an inverse that did not come from this library (plain torch.unique) still works
    u3, i3 = torch.unique(coors, dim=0, return_inverse=True)
    nf3, _, _ = sst_amd.scatter_v2(feat.to(DEV), coors.to(DEV), 'max', unq_inv=i3.to(DEV), new_coors=u3.to(DEV))
    ref = torch.full((u3.size(0), 8), float('-inf')).scatter_reduce(0, i3.view(-1, 1).expand(-1, 8), feat, 'amax')
    assert torch.equal(nf3.cpu(), ref)
    # ... also when ids are missing from it (a slice of the points): torch_scatter sizes the output by the number of
    # groups and leaves the rows of absent ids at 0 (sst_ops.py:172-177)
    part = (i3 % 3) != 0
    i4 = i3[part].clone()
    for mode, red in (('max', 'amax'), ('sum', 'sum')):
        nf4, nc4, _ = sst_amd.scatter_v2(feat[part].to(DEV), coors[part].to(DEV), mode, unq_inv=i4.to(DEV),
                                         new_coors=u3.to(DEV))
        assert nf4.shape == (u3.size(0), 8) and nc4.size(0) == u3.size(0)
        ref4 = torch.zeros((u3.size(0), 8)).scatter_reduce(0, i4.view(-1, 1).expand(-1, 8), feat[part], red,
                                                           include_self=False)
        assert torch.allclose(nf4.cpu(), ref4, atol=1e-5)


def test_sir_matches_reference_golden():
    import sst_amd
    g = load_golden('sir.npz')
    hidden = [[16, 32]] * 3   # one shared list on purpose: the layer must not mutate its argument
    sir = sst_amd.build_backbone(dict(type='SIR', num_blocks=3, in_channels=[84, 133, 133],
                                      feat_channels=[[128, 128]] * 3, rel_mlp_hidden_dims=hidden,
                                      norm_cfg=dict(type='LN', eps=1e-3), mode='max', xyz_normalizer=[20, 20, 4],
                                      act='gelu', unique_once=True))
    assert hidden == [[16, 32]] * 3
    sd = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith('w::')}
    sir.load_state_dict(sd, strict=True)
    sir.to(DEV).train()
    feats = torch.from_numpy(g['in::features']).to(DEV).requires_grad_(True)
    pts_feats, cluster_feats, cluster_coors = sir(torch.from_numpy(g['in::points']).to(DEV), feats,
                                                  torch.from_numpy(g['in::coors']).to(DEV),
                                                  torch.from_numpy(g['in::f_cluster']).to(DEV))
    np.testing.assert_array_equal(cluster_coors.cpu().numpy(), g['out::cluster_coors'])
    np.testing.assert_allclose(pts_feats.detach().cpu().numpy(), g['out::pts_feats'], rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(cluster_feats.detach().cpu().numpy(), g['out::cluster_feats'], rtol=1e-3, atol=1e-3)
    ((pts_feats * torch.from_numpy(g['in::g_pts']).to(DEV)).sum()
     + (cluster_feats * torch.from_numpy(g['in::g_cluster']).to(DEV)).sum()).backward()
    np.testing.assert_allclose(feats.grad.cpu().numpy(), g['out::grad_features'], rtol=2e-3, atol=2e-3)


def test_dynamic_scatter_vfe_matches_reference_golden():
    import sst_amd
    g = load_golden('scatter_vfe.npz')
    vfe = sst_amd.build_voxel_encoder(dict(
        type='DynamicScatterVFE', in_channels=5, feat_channels=[64, 64], voxel_size=(0.25, 0.25, 0.2),
        with_cluster_center=True, with_voxel_center=True, point_cloud_range=[-80, -80, -2, 80, 80, 4],
        norm_cfg=dict(type='naiveSyncBN1d', eps=1e-3, momentum=0.01), unique_once=True))
    vfe.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith('w::')}, strict=True)
    vfe.to(DEV).train()
    pts = torch.from_numpy(g['in::points']).to(DEV).requires_grad_(True)
    vf, vc, inv = vfe(pts, torch.from_numpy(g['in::coors']).to(DEV), return_inv=True)
    np.testing.assert_array_equal(vc.cpu().numpy(), g['out::voxel_coors'])
    np.testing.assert_array_equal(inv.cpu().numpy(), g['out::inv'])
    np.testing.assert_allclose(vf.detach().cpu().numpy(), g['out::voxel_feats'], rtol=1e-3, atol=1e-3)
    (vf * torch.from_numpy(g['in::grad_out']).to(DEV)).sum().backward()
    np.testing.assert_allclose(pts.grad.cpu().numpy(), g['out::grad_points'], rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize('c', [128, 132, 64])
def test_segment_reduce_long_groups_one_workgroup_per_group(c):
    """FSD-like clusters: a few groups with thousands of points, many with a handful (average >= 4 points: the
    one-workgroup-per-group kernel of csrc/scatter.hip); max with its arg-max (ties -> smallest row), sum, mean and the
    gradients, against float64 torch"""
    from sst_amd import kernels as K
    from sst_amd.sst_ops import plan_of_inverse
    rng = np.random.default_rng(c)
    sizes = np.concatenate([[3000, 1700, 512, 65, 64, 63], rng.integers(1, 40, size=700)])
    m = len(sizes)
    inv = np.repeat(np.arange(m), sizes)
    rng.shuffle(inv)
    n = inv.size
    assert n >= 8 * m
    feat = torch.from_numpy(rng.standard_normal((n, c)).astype(np.float32))
    feat[:, :8] = torch.round(feat[:, :8])            # ties of the maximum in the first channels
    inv_t = torch.from_numpy(inv).to(DEV)
    plan = plan_of_inverse(inv_t, m)
    fg = feat.to(DEV).requires_grad_(True)
    idx = torch.from_numpy(inv).view(-1, 1).expand(-1, c)
    f64 = feat.double()
    for mode, red in (('max', 'amax'), ('sum', 'sum'), ('mean', 'mean')):
        got = K.segment_reduce(fg, plan, mode)
        ref = torch.zeros(m, c, dtype=torch.float64).scatter_reduce(0, idx, f64, reduce=red, include_self=False)
        assert float((got.detach().cpu().double() - ref).abs().max()) < 1e-4, mode
    mx, arg = K.segment_argmax(feat.to(DEV), plan)
    ref = torch.full((m, c), float('-inf'), dtype=torch.float64).scatter_reduce(0, idx, f64, reduce='amax')
    assert torch.equal(mx.cpu().double(), ref)
    # smallest row index among the rows attaining the maximum
    hit = (f64 == ref[inv])
    cand = torch.where(hit, torch.arange(n).view(-1, 1).expand(-1, c), torch.full((n, c), n))
    first = torch.full((m, c), n, dtype=torch.long).scatter_reduce(0, idx, cand, reduce='amin')
    assert torch.equal(arg.cpu().long(), first)
    out = K.segment_reduce(fg, plan, 'max')
    gout = torch.from_numpy(rng.standard_normal((m, c)).astype(np.float32))
    (out * gout.to(DEV)).sum().backward()
    want = torch.zeros(n, c, dtype=torch.float64)
    want.scatter_(0, first, gout.double())
    assert float((fg.grad.cpu().double() - want).abs().max()) < 1e-6
