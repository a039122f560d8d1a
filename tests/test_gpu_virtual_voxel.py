"""FSDv2's virtual-voxel stage (sst_amd/virtual_voxel.py) against tensors produced by the reference's own
SingleStageFSDV2.extract_feat (single_stage_fsd_v2.py:159-271) executed on CPU with the reference's submodules
(tests/golden/make_golden.py::gen_virtual_voxel -> tests/golden/virtual_voxel.npz)."""
import numpy as np
import pytest
import torch

from conftest import load_golden

DEV = 'cuda:0'
MIXER_CFG = dict(in_channels=8, sparse_shape=[16, 40, 40], order=('conv', 'norm', 'act'),
                 norm_cfg=dict(type='BN1d', eps=1e-3, momentum=0.01), base_channels=16, output_channels=24,
                 encoder_channels=((16, ), (16, 16), (16, 16)), encoder_paddings=((1, ), (1, 1), (1, 1)),
                 decoder_channels=((16, 16, 16), (16, 16, 16), (16, 16, 16)),
                 decoder_paddings=((1, 1), (1, 1), (1, 1)))
CFG = dict(
    virtual_point_projector=dict(in_channels=16 + 3 + 4 + 2, hidden_dims=[16, 16], norm_cfg=dict(type='naiveSyncBN1d'),
                                 ori_in_channels=16, ori_hidden_dims=[16, 16]),
    voxel_encoder=dict(type='DynamicScatterVFE', in_channels=3 + 16, feat_channels=[16, 8], voxel_size=(0.4, 0.4, 0.4),
                       with_cluster_center=True, with_voxel_center=True, point_cloud_range=[-8, -8, -3.2, 8, 8, 3.2],
                       norm_cfg=dict(type='naiveSyncBN1d', eps=1e-3, momentum=0.01), unique_once=True),
    backbone=dict(type='VirtualVoxelMixer', **MIXER_CFG))


def _weights(g):
    return {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith('w::')}


def test_virtual_voxel_state_dict_matches_the_reference_submodules():
    import sst_amd
    g = load_golden('virtual_voxel.npz')
    net = sst_amd.VirtualVoxelExtractor(**CFG)
    mine, ref = net.state_dict(), _weights(g)
    assert set(mine) == set(ref) and all(tuple(mine[k].shape) == tuple(ref[k].shape) for k in ref)


@pytest.mark.gpu
def test_virtual_voxel_stage_matches_reference_golden():
    import sst_amd
    g = load_golden('virtual_voxel.npz')
    net = sst_amd.VirtualVoxelExtractor(**CFG)
    net.load_state_dict(_weights(g), strict=True)
    net = net.to(DEV).train()
    t = lambda k: torch.from_numpy(g[k]).to(DEV)
    leaves = {k: t(k).requires_grad_(True) for k in ('in::ori_feats', 'in::smp_feats', 'in::smp_logits')}
    origin = dict(seg_points=t('in::ori_points'), seg_feats=leaves['in::ori_feats'], batch_idx=t('in::ori_batch_idx'))
    sampled = dict(seg_points=t('in::smp_points'), center_preds=t('in::smp_centers'), seg_logits=leaves['in::smp_logits'],
                   seg_feats=leaves['in::smp_feats'], batch_idx=t('in::smp_batch_idx'))
    raw_centers = sampled['center_preds'].clone()
    out = net(sampled, origin)
    # the reference clips the predicted centres IN PLACE (single_stage_fsd_v2.py:124-129): the caller's dictionary holds the
    # clipped values afterwards
    rng_lo = torch.tensor(CFG['voxel_encoder']['point_cloud_range'][:3], device=DEV)
    rng_hi = torch.tensor(CFG['voxel_encoder']['point_cloud_range'][3:], device=DEV)
    assert torch.equal(sampled['center_preds'], torch.max(torch.min(raw_centers, rng_hi - 1e-5), rng_lo + 1e-5))
    # the index part is exact: the same voxels, in the same (sorted-unique) order, flagged virtual
    assert torch.equal(out['virtual_coors'].cpu(), torch.from_numpy(g['out::virtual_coors']).to(out['virtual_coors'].dtype))
    assert list(out['sparse_shape']) == list(g['out::sparse_shape'])
    for key in ('virtual_centers', 'virtual_centroid', 'virtual_feats'):
        want = g['out::' + key]
        err = np.abs(out[key].detach().cpu().numpy() - want).max()
        assert err < 1e-3 * max(1.0, np.abs(want).max()), (key, err)
    (out['virtual_feats'] * t('in::grad_out')).sum().backward()
    params = dict(net.named_parameters())
    checks = [(leaves['in::ori_feats'].grad, 'out::grad_ori_feats'), (leaves['in::smp_feats'].grad, 'out::grad_smp_feats'),
              (leaves['in::smp_logits'].grad, 'out::grad_smp_logits'),
              (params['virtual_proj.0.0.weight'].grad, 'out::grad::virtual_proj.0.0.weight'),
              (params['ori_proj.1.0.weight'].grad, 'out::grad::ori_proj.1.0.weight')]
    for got, key in checks:
        want = g[key]
        err = np.abs(got.cpu().numpy() - want).max()
        assert err < 2e-3 * max(1.0, np.abs(want).max()), (key, err)


@pytest.mark.gpu
def test_virtual_voxel_only_virtual_keeps_the_virtual_voxels_only():
    import sst_amd
    g = load_golden('virtual_voxel.npz')
    cfg = dict(CFG, virtual_point_projector=dict(CFG['virtual_point_projector'], only_virtual=True))
    net = sst_amd.VirtualVoxelExtractor(**cfg).to(DEV).eval()
    t = lambda k: torch.from_numpy(g[k]).to(DEV)
    origin = dict(seg_points=t('in::ori_points'), seg_feats=t('in::ori_feats'), batch_idx=t('in::ori_batch_idx'))
    sampled = dict(seg_points=t('in::smp_points'), center_preds=t('in::smp_centers'), seg_logits=t('in::smp_logits'),
                   seg_feats=t('in::smp_feats'), batch_idx=t('in::smp_batch_idx'))
    with torch.no_grad():
        out = net(sampled, origin)
    assert torch.equal(out['virtual_coors'].cpu(), torch.from_numpy(g['out::virtual_coors']).to(out['virtual_coors'].dtype))
    assert out['virtual_feats'].shape == (g['out::virtual_coors'].shape[0], MIXER_CFG['output_channels'])
    assert 'virtual_centroid' not in out


# ------------------------------------------------------------------------------------------------------------------------
# multi-scale fusion + as_rpn outputs (single_stage_fsd_v2.py:131-155, 208-221, 263-270, 375-433): what every shipped
# configs/fsdv2/*.py turns on.  Golden: the reference's own extract_feat / multiscale_fusion / ms_coors_proj /
# recover_point_features executed from their source (tests/golden/make_golden.py::gen_virtual_voxel(multiscale=True)).
# ------------------------------------------------------------------------------------------------------------------------
MS_CFG = dict(
    CFG,
    virtual_point_projector=dict(CFG['virtual_point_projector'], recover_in_channels=24 + 3, recover_hidden_dims=[16, 16]),
    multiscale_cfg=dict(multiscale_levels=[0, 1], projector_hiddens=[[12, 8], [8, 16, 8]], fusion_mode='avg',
                        target_sparse_shape=[16, 40, 40], norm_cfg=dict(type='naiveSyncBN1d')),
    bbox_head=dict(type='FSDV2Head', as_rpn=True))


def _ms_inputs(g, dev, grad=True):
    import types
    t = lambda k: torch.from_numpy(g[k]).to(dev)
    leaves = {k: t(k).requires_grad_(grad) for k in ('in::ori_feats', 'in::smp_feats', 'in::smp_logits')}
    origin = dict(seg_points=t('in::ori_points'), seg_feats=leaves['in::ori_feats'], batch_idx=t('in::ori_batch_idx'))
    sampled = dict(seg_points=t('in::smp_points'), center_preds=t('in::smp_centers'), seg_logits=leaves['in::smp_logits'],
                   seg_feats=leaves['in::smp_feats'], batch_idx=t('in::smp_batch_idx'))
    ms = [types.SimpleNamespace(features=t(f'in::ms{lvl}::features').requires_grad_(grad), indices=t(f'in::ms{lvl}::indices'),
                                spatial_shape=[int(v) for v in g[f'in::ms{lvl}::spatial_shape']]) for lvl in range(2)]
    return leaves, origin, sampled, ms


def _check_ms(net, g, dev, tol):
    leaves, origin, sampled, ms = _ms_inputs(g, dev)
    out = net(sampled, origin, multiscale_features=ms)
    assert torch.equal(out['virtual_coors'].cpu().long(), torch.from_numpy(g['out::virtual_coors']).long())
    assert torch.equal(out['pts_batch_inds'].cpu().long(), torch.from_numpy(g['out::pts_batch_inds']).long())
    assert torch.equal(out['pts_indicators'].cpu(), torch.from_numpy(g['out::pts_indicators']))
    for key in ('virtual_centers', 'virtual_centroid', 'virtual_feats', 'pts_feats', 'pts_xyz'):
        want = g['out::' + key]
        err = np.abs(out[key].detach().cpu().numpy() - want).max()
        assert err < tol * max(1.0, np.abs(want).max()), (key, err)
    t = lambda k: torch.from_numpy(g[k]).to(dev)
    ((out['virtual_feats'] * t('in::grad_out')).sum() + (out['pts_feats'] * t('in::grad_pts')).sum()).backward()
    params = dict(net.named_parameters())
    checks = [(leaves['in::ori_feats'].grad, 'out::grad_ori_feats'), (leaves['in::smp_feats'].grad, 'out::grad_smp_feats'),
              (leaves['in::smp_logits'].grad, 'out::grad_smp_logits'), (ms[0].features.grad, 'out::grad_ms0'),
              (ms[1].features.grad, 'out::grad_ms1')]
    checks += [(params[k[11:]].grad, k) for k in g if k.startswith('out::grad::')]
    assert len(checks) == 10
    for got, key in checks:
        want = g[key]
        err = np.abs(got.cpu().numpy() - want).max()
        assert err < 2 * tol * max(1.0, np.abs(want).max()), (key, err)


def test_virtual_voxel_multiscale_state_dict_matches_the_reference_submodules():
    import sst_amd
    g = load_golden('virtual_voxel_ms.npz')
    net = sst_amd.VirtualVoxelExtractor(**MS_CFG)
    assert net.as_rpn
    mine, ref = net.state_dict(), _weights(g)
    assert set(mine) == set(ref) and all(tuple(mine[k].shape) == tuple(ref[k].shape) for k in ref)
    assert any(k.startswith('ms_projectors.1.1.') for k in mine) and any(k.startswith('recover_proj.') for k in mine)


def test_cpu_port_multiscale_stage_matches_reference_golden():
    """pins oracle/fsd_cpu.VirtualVoxelExtractor (multiscale_fusion, ms_coors_proj, recover_point_features) - the bench's
    cpu_baseline / parity checker - to the reference-produced fixture, on any box"""
    from oracle import fsd_cpu
    g = load_golden('virtual_voxel_ms.npz')
    net = fsd_cpu.VirtualVoxelExtractor(**MS_CFG)
    net.load_state_dict(_weights(g), strict=True)
    _check_ms(net.train(), g, 'cpu', 1e-5)


@pytest.mark.gpu
def test_virtual_voxel_multiscale_as_rpn_matches_reference_golden():
    import sst_amd
    g = load_golden('virtual_voxel_ms.npz')
    net = sst_amd.VirtualVoxelExtractor(**MS_CFG)
    net.load_state_dict(_weights(g), strict=True)
    _check_ms(net.to(DEV).train(), g, DEV, 1e-3)


@pytest.mark.gpu
def test_virtual_voxel_multiscale_fusion_mask_and_order():
    """the fused voxel set is the sorted-unique of [virtual voxels, projected multi-scale voxels]; the single-scale mask
    marks exactly the virtual-voxel rows, in their order (what the reference's second scatter_v2(max) computes)"""
    import sst_amd
    g = load_golden('virtual_voxel_ms.npz')
    net = sst_amd.VirtualVoxelExtractor(**MS_CFG).to(DEV).eval()
    _, _, _, ms = _ms_inputs(g, DEV, grad=False)
    gen = torch.Generator().manual_seed(3)
    flat = torch.randperm(2 * 16 * 40 * 40, generator=gen)[:400].sort()[0]
    coors = torch.stack([flat // 25600, flat // 1600 % 16, flat // 40 % 40, flat % 40], 1).to(DEV)
    feats = torch.randn(400, 8, generator=gen).to(DEV)
    with torch.no_grad():
        of, oc, mask = net.multiscale_fusion(ms, feats, coors)
    proj = [net.ms_coors_proj(d.indices, d.spatial_shape).long() for d in ms]
    want, inv = torch.unique(torch.cat([coors] + proj), dim=0, return_inverse=True)
    assert torch.equal(oc, want) and int(mask.sum()) == 400 and torch.equal(oc[mask], coors)
    # projected coordinates: stride 2 levels land on the odd cells' centres (c * 2 + 1), stride 1 levels are unchanged
    assert torch.equal(proj[0][:, 1:], ms[0].indices[:, 1:].long() * 2 + 1) and torch.equal(proj[1], ms[1].indices.long())
