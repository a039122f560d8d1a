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
