"""GPU: degenerate inputs through the path — empty clouds, a batch with an empty sample, all points outside the
range, a single voxel, windows exactly at / one past every token cap, duplicate points.  The reference's own test
covers the empty and the all-invalid scatter (tests/test_models/test_voxel_encoder/test_dynamic_scatter.py:60-76);
the rest follows its in-code invariants (sst_input_layer_v2.py debug asserts)."""
import numpy as np
import pytest
import torch

from conftest import DROP_TEST, DROP_TRAIN

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
VS = (0.32, 0.32, 6.0)
PCR = (-74.88, -74.88, -2.0, 74.88, 74.88, 4.0)


def test_sra_core_with_no_windows_and_no_tokens():
    from sst_amd import kernels as K
    plan = K.WindowPlan(torch.zeros(0, dtype=torch.int32, device=DEV), torch.zeros(1, dtype=torch.int32, device=DEV), 0, 0, 0)
    q = torch.zeros(0, 128, device=DEV, requires_grad=True)
    o = K.sra_attention(q, q, q, plan, 8)
    assert o.shape == (0, 128)
    o.sum().backward()
    assert q.grad.shape == (0, 128)
    # tokens that belong to no window (dropped voxels): output rows are zero, gradients are zero
    plan = K.WindowPlan(torch.tensor([3, 1], dtype=torch.int32, device=DEV),
                        torch.tensor([0, 2], dtype=torch.int32, device=DEV), 1, 2, 2)
    x = torch.randn(5, 128, device=DEV, requires_grad=True)
    o = K.sra_attention(x, x, x, plan, 8)
    assert torch.count_nonzero(o[[0, 2, 4]]).item() == 0 and torch.count_nonzero(o[[1, 3]]).item() > 0
    o.sum().backward()
    assert torch.count_nonzero(x.grad[[0, 2, 4]]).item() == 0


@pytest.mark.parametrize('cap', [30, 60, 100, 144])
def test_window_sizes_around_every_token_cap(cap):
    """windows of cap - 1, cap, cap + 1 tokens: tile-class boundaries of the register-resident kernels (and the
    generic kernel above 144) against the float64 oracle."""
    from sst_amd import kernels as K
    from oracle import sst_oracle
    sizes = [cap - 1, cap, cap + 1, 1]
    rng = np.random.default_rng(cap)
    m = sum(sizes)
    tok = rng.permutation(m).astype(np.int32)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    plan = K.WindowPlan(torch.from_numpy(tok).to(DEV), torch.from_numpy(off).to(DEV), len(sizes), m, max(sizes))
    g = torch.Generator().manual_seed(cap)
    q, k, v = (torch.randn(m, 128, generator=g) for _ in range(3))
    o = K.sra_attention(q.to(DEV), k.to(DEV), v.to(DEV), plan, 8)
    ref = sst_oracle.sra_core(q.numpy(), k.numpy(), v.numpy(), tok, off, 8)
    assert np.abs(o.cpu().numpy() - ref).max() < 1e-3


def test_voxelize_and_scatter_of_empty_and_out_of_range_clouds():
    import sst_amd
    vox = sst_amd.Voxelization(VS, PCR, -1, (-1, -1))
    empty = torch.zeros(0, 5, device=DEV)
    assert vox(empty).shape == (0, 3)
    outside = torch.tensor([[500.0, 0, 0, 1, 1], [0, -500.0, 0, 1, 1], [0, 0, 50.0, 1, 1]], device=DEV)
    c = vox(outside)
    assert c.shape == (3, 3) and c.dtype == torch.int32
    scatter = sst_amd.DynamicScatter(VS, PCR, True)
    f, vc = scatter(empty[:, :4], vox(empty))
    assert f.shape[0] == 0 and vc.shape[0] == 0
    # every point invalid (-1 coordinates): no voxel comes out (reference test, :68-76)
    feats = torch.rand(64, 4, device=DEV)
    coors = torch.full((64, 3), -1, dtype=torch.int32, device=DEV)
    f, vc = scatter(feats, coors)
    assert f.shape[0] == 0 and vc.shape[0] == 0


def test_batch_with_an_empty_sample_and_duplicate_points():
    import sst_amd
    vox = sst_amd.Voxelization(VS, PCR, -1, (-1, -1))
    g = torch.Generator().manual_seed(0)
    p0 = torch.rand(500, 5, generator=g) * torch.tensor([140.0, 140.0, 5.0, 1, 1]) + torch.tensor([-70.0, -70.0, -1.5, 0, 0])
    p0 = torch.cat([p0, p0[:100]])  # exact duplicates land in the same voxel
    frames = [p0.to(DEV), torch.zeros(0, 5), p0[:50].to(DEV)]
    frames[1] = frames[1].to(DEV)
    points, coors = vox.voxelize_batch(frames)
    assert points.shape[0] == 650 and set(coors[:, 0].unique().tolist()) == {0, 2}
    plan = sst_amd.build_scatter_plan(coors, grid_zyx=[1, 468, 468], reference_compat=False)
    cnt = plan.reduce(torch.ones(650, 4, device=DEV), 'sum')[:, 0]
    assert int(cnt.sum().item()) == 650
    # the same cloud without its duplicates gives the same voxels, with smaller counts
    pts2, coors2 = vox.voxelize_batch([frames[0][:500], frames[1], frames[2]])
    plan2 = sst_amd.build_scatter_plan(coors2, grid_zyx=[1, 468, 468], reference_compat=False)
    assert torch.equal(plan.voxel_coors, plan2.voxel_coors)
    mx = plan.reduce(points[:, :4].contiguous(), 'max')
    mx2 = plan2.reduce(pts2[:, :4].contiguous(), 'max')
    assert torch.equal(mx, mx2)  # duplicates do not change a max


@pytest.mark.parametrize('training', [True, False])
def test_input_layer_and_block_on_a_single_voxel_and_on_one_full_window(training):
    import sst_amd
    layer = sst_amd.SSTInputLayerV2((DROP_TRAIN, DROP_TEST), (12, 12, 1), (468, 468, 1), shuffle_voxels=False, mute=True,
                                    debug=True)
    layer.train(training)
    backbone = sst_amd.SSTv2(d_model=[128] * 2, nhead=[8] * 2, num_blocks=2, dim_feedforward=[256] * 2,
                             output_shape=[468, 468], num_attached_conv=0, debug=True, to_bev=False,
                             layer_cfg=dict(use_bn=False, cosine=False, tau_min=0.01), checkpoint_blocks=[]).to(DEV)
    backbone.train(training)
    # one voxel
    coors = torch.tensor([[0, 0, 17, 250]], dtype=torch.int64, device=DEV)
    feats = torch.randn(1, 128, device=DEV)
    info = layer(feats, coors, 1)
    out = backbone(info)[0]
    assert out['voxel_feats'].shape == (1, 128) and torch.isfinite(out['voxel_feats']).all()
    # one completely full window (144 voxels): in training the cap (100) drops 44 of them, in eval none
    yy, xx = torch.meshgrid(torch.arange(12), torch.arange(12), indexing='ij')
    coors = torch.stack([torch.zeros(144, dtype=torch.int64), torch.zeros(144, dtype=torch.int64), yy.reshape(-1) + 24,
                         xx.reshape(-1) + 36], 1).to(DEV)
    feats = torch.randn(144, 128, device=DEV)
    info = layer(feats, coors, 1)
    kept = info['voxel_feats'].size(0)
    assert kept <= 144 and (kept == 144 or training)
    out = backbone(info)[0]
    assert out['voxel_feats'].shape == (kept, 128) and torch.isfinite(out['voxel_feats']).all()
