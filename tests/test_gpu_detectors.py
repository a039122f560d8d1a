"""sst_amd.detectors on the GPU: the segmentor mirror (VoteSegmentor) against the same modules composed by hand, and the
FSDV2 mirror end to end (segmentor with multi-scale decoder features -> sampling stand-in -> virtual-voxel stage with
multiscale_cfg and as_rpn), forward + backward, at fixture size in the shape of configs/fsdv2/fsdv2_waymo_1x.py."""
import pytest
import torch

import bench_workloads as BW

DEV = 'cuda:0'
BN = dict(type='naiveSyncBN1d', eps=1e-3, momentum=0.01)
RANGE = [-12.8, -12.8, -2, 12.8, 12.8, 1.2]


def _model_cfg():
    small = BW.FSDV2_SMALL_CFG
    segmentor = dict(
        type='VoteSegmentor', tanh_dims=[],
        voxel_layer=dict(voxel_size=small['seg_voxel'], max_num_points=-1, point_cloud_range=RANGE, max_voxels=(-1, -1)),
        voxel_encoder=dict(type='DynamicScatterVFE', voxel_size=small['seg_voxel'], with_cluster_center=True,
                           with_voxel_center=True, point_cloud_range=RANGE, norm_cfg=BN, unique_once=True, **small['vfe']),
        middle_encoder=dict(type='PseudoMiddleEncoderForSpconvFSD'),
        backbone=dict(type='SimpleSparseUNet', order=('conv', 'norm', 'act'), norm_cfg=BN, return_multiscale_features=True,
                      **small['unet']),
        decode_neck=dict(type='Voxel2PointScatterNeck', voxel_size=small['seg_voxel'], point_cloud_range=RANGE),
        segmentation_head=dict(type='VoteSegHead', in_channel=16 + 3, hidden_dims=[16, 16], num_classes=3, dropout_ratio=0.0,
                               norm_cfg=dict(type='naiveSyncBN1d'), act_cfg=dict(type='ReLU'),
                               loss_decode=dict(type='FocalLoss', use_sigmoid=True), loss_vote=dict(type='L1Loss')),
        train_cfg=dict(point_loss=True, score_thresh=[0.5, 0.5, 0.5]))
    return dict(
        type='FSDV2', segmentor=segmentor,
        virtual_point_projector=dict(in_channels=(16 + 3) + 3 + 3 + 2, hidden_dims=[16, 16], norm_cfg=dict(type='naiveSyncBN1d'),
                                     ori_in_channels=16 + 3, ori_hidden_dims=[16, 16], recover_in_channels=16 + 3,
                                     recover_hidden_dims=[16, 16]),
        multiscale_cfg=small['multiscale'],
        voxel_encoder=dict(type='DynamicScatterVFE', in_channels=3 + 16, voxel_size=small['virtual_voxel'], with_cluster_center=True,
                           with_voxel_center=True, point_cloud_range=RANGE, norm_cfg=BN, unique_once=True, **small['virtual_vfe']),
        backbone=dict(type='VirtualVoxelMixer', order=('conv', 'norm', 'act'), norm_cfg=BN, **small['mixer']),
        bbox_head=dict(type='FSDV2Head', num_classes=3, as_rpn=True),
        roi_head=dict(type='GroupCorrectionHead', num_classes=3,
                      roi_extractor=dict(type='DynamicPointROIExtractor', extra_wlh=[0.5, 0.5, 0.5], max_inbox_point=256,
                                         max_all_pts=100000, debug=False, with_virtual=False)),
        train_cfg=dict(score_thresh=[0.5, 0.5, 0.5]), test_cfg=dict(score_thresh=[0.5, 0.5, 0.5]))


@pytest.mark.gpu
def test_vote_segmentor_is_its_modules_composed():
    import sst_amd
    torch.manual_seed(0)
    seg = sst_amd.build_detector(_model_cfg()['segmentor']).to(DEV).eval()
    clouds = [BW.chain_cloud(3000, 5, half_extent=12.0).to(DEV), BW.chain_cloud(2500, 6, half_extent=12.0).to(DEV)]
    with torch.no_grad():
        out = seg([c.clone() for c in clouds])
        pts, coors = seg.voxel_layer.voxelize_batch(clouds)
        vf, vc, inv = seg.voxel_encoder(pts, coors.long(), return_inv=True)
        info = seg.middle_encoder(vf, vc)
        info['batch_size'] = 2
        x = seg.backbone(info)[0]
        pf = x['voxel_feats'][inv]
        vs = torch.tensor(BW.FSDV2_SMALL_CFG['seg_voxel'], device=DEV)
        centre = (coors[:, [3, 2, 1]].float() + 0.5) * vs + torch.tensor(RANGE[:3], device=DEV)
        feats = torch.cat([pf, pts[:, :3] - centre], 1)
        logits, votes = seg.segmentation_head(feats)
    assert torch.equal(out['seg_points'], pts) and torch.equal(out['batch_idx'], coors[:, 0].long())
    assert torch.allclose(out['seg_feats'], feats, atol=1e-6) and torch.allclose(out['seg_logits'], logits, atol=1e-6)
    assert torch.allclose(out['offsets'], votes * votes.abs(), atol=1e-6)
    assert out['seg_logits'].shape == (5500, 3) and out['seg_vote_preds'].shape == (5500, 9)
    assert len(out['decoder_features']) == 4 and out['decoder_features'][0].features.size(1) == 32
    assert (feats[:, -3:].abs() <= vs / 2 + 1e-3).all()                          # the neck's own training-time assertion


@pytest.mark.gpu
def test_fsdv2_detector_hot_path_forward_backward():
    import sst_amd
    torch.manual_seed(1)
    det = sst_amd.build_detector(_model_cfg()).to(DEV).train()
    assert det.as_rpn and isinstance(det.roi_extractor, sst_amd.DynamicPointROIExtractor)
    clouds = [BW.chain_cloud(3000, 7, half_extent=12.0).to(DEV), BW.chain_cloud(2500, 8, half_extent=12.0).to(DEV)]
    seg = det.segmentor(clouds)
    sel = torch.nonzero(seg['seg_points'][:, 2] > -1.4).squeeze(1)              # sampling stand-in (detector glue)
    cls = seg['seg_logits'][sel].argmax(1)
    centers = seg['seg_points'][sel, :3] + seg['offsets'][sel].view(-1, 3, 3)[torch.arange(sel.numel(), device=DEV), cls]
    sampled = dict(seg_points=seg['seg_points'][sel], center_preds=centers.detach(), seg_logits=seg['seg_logits'][sel].detach(),
                   seg_feats=seg['seg_feats'][sel], batch_idx=seg['batch_idx'][sel])
    origin = dict(seg_points=seg['seg_points'], seg_feats=seg['seg_feats'], batch_idx=seg['batch_idx'], batch_size=2)
    out = det.extract_feat(sampled, origin, multiscale_features=seg['decoder_features'])
    n_pts = seg['seg_points'].size(0) + sel.numel()
    assert out['pts_feats'].shape == (n_pts, 16) and out['pts_xyz'].shape == (n_pts, 3)
    assert out['virtual_feats'].shape[1] == 16 and out['virtual_feats'].size(0) == out['virtual_centroid'].size(0) > 0
    assert int(out['pts_indicators'].sum()) == sel.numel()
    # without the fusion the voxel set is the same (the mask brings the rows back), the features are not
    plain = det.extract_feat(dict(sampled, center_preds=sampled['center_preds'].clone()), origin)
    assert torch.equal(plain['virtual_coors'], out['virtual_coors'])
    assert not torch.allclose(plain['virtual_feats'], out['virtual_feats'])
    (out['virtual_feats'].sum() + out['pts_feats'].sum() * 0.1).backward()
    grads = {n: p.grad for n, p in det.named_parameters()}
    for name in ('ms_projectors.0.0.0.weight', 'ms_projectors.1.0.0.weight', 'recover_proj.1.0.weight',
                 'segmentor.backbone.upsample_layer4.0.weight', 'segmentor.voxel_encoder.vfe_layers.0.linear.weight',
                 'backbone.conv_out.0.weight'):
        assert grads[name] is not None and torch.isfinite(grads[name]).all() and float(grads[name].abs().max()) > 0, name


@pytest.mark.gpu
def test_reference_detector_with_the_fused_extract_feat_hook():
    """the reference's own DynamicVoxelNet class (its voxelize loop, its neck) over this library's
    sub-modules, with `install_fused_extract_feat`: same features as sst_amd.DynamicVoxelNet built from the same shipped config,
    and as the piecewise (module by module) path"""
    import bench
    import sst_amd
    from test_config_fixtures import _ReferenceLikeDetector
    cfg = bench.load_config_fixture('sst_waymoD5_1x_3class_8heads_v2')
    cfg['middle_encoder'] = dict(cfg['middle_encoder'], shuffle_voxels=False, window_major=True)   # deterministic voxel order
    cfg['backbone'] = dict(cfg['backbone'], to_bev=False, num_attached_conv=0)
    frames = [bench.make_cloud(30000, 3, 'cuda:0'), bench.make_cloud(20000, 4, 'cuda:0')]
    torch.manual_seed(0)
    a = sst_amd.build_detector(cfg).to('cuda:0').train()
    cls = sst_amd.install_fused_extract_feat(type('Det', (_ReferenceLikeDetector,), {}))
    torch.manual_seed(0)
    b = cls(cfg).to('cuda:0').train()
    b.load_state_dict(a.state_dict(), strict=True)
    a.middle_encoder.mute = b.middle_encoder.mute = True
    outs = []
    for det, fused in ((a, True), (b, True), (b, False)):
        det.fused_index = fused
        with torch.no_grad():
            x = det.extract_feat(frames, None)[0]
        outs.append((x['voxel_coors'], x['voxel_feats']))
    assert b.__dict__['_planner'] is not None and b.__dict__['_planner'] is not False
    for coors, feats in outs[1:]:
        assert torch.equal(coors, outs[0][0])
        assert float((feats - outs[0][1]).abs().max()) <= 1e-5
    assert torch.equal(outs[1][1], outs[0][1])
