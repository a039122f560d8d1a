"""FSDv2's virtual-voxel stage: foreground points vote for object centres, the voted centres ("virtual points") are
voxelised together with the original points, the segmentor's multi-scale decoder features are fused into the same voxel
set, and a sparse U-Net mixes real, virtual and multi-scale voxels.

Reference: SingleStageFSDV2.extract_feat, mmdet3d/models/detectors/single_stage_fsd_v2.py:159-271 (non-baseline mode) with
its helpers voxelize_with_batch_idx (:107-122), clip_points (:124-129), recover_point_features (:131-155, the ``as_rpn``
outputs of :263-270), multiscale_fusion (:375-397) and ms_coors_proj (:399-433); constructor arguments as in
SingleStageFSDV2.__init__ (:38-105) for the part that belongs to this stage (``backbone``, ``voxel_encoder``,
``virtual_point_projector``, ``multiscale_cfg``, ``bbox_head['as_rpn']``), submodule names as there (``virtual_proj``,
``ori_proj``, ``recover_proj``, ``ms_projectors``, ``voxel_encoder``, ``backbone``) so that a detector checkpoint's keys for
them load unchanged.  Every shipped configs/fsdv2/*.py sets ``multiscale_cfg``; the Waymo ones also ``as_rpn=True``.
The detector around it (segmentation / box heads, losses, box decoding) is out of scope;
``sst_amd.detectors`` wires this stage behind the segmentor the way the detector does.

Differences in execution, not in results:
  * the reference groups the same coordinates three times (voxel encoder, the virtual-indicator average, the training-time
    centroid); here the voxel encoder's grouping (one radix sort of the packed coordinates) is reused by the two segmented
    averages;
  * multiscale_fusion sorts the concatenated coordinates twice (feature average, indicator maximum); here ONE sorted-unique
    serves both, and the indicator maximum is read off the inverse map (a fused voxel is "single-scale" iff one of the first
    n rows - the virtual-voxel rows, which are distinct - maps to it): no second reduction;
  * the bounds assertions of ms_coors_proj (nine host read-backs per call) hold by construction for indices inside their
    grid ((s - 1) * stride + stride // 2 < s * stride <= target) and are checked only under SST_AMD_DEBUG.
"""
import os

import torch
from torch import nn

from .registry import MODELS, build_backbone, build_voxel_encoder
from . import kernels as K
from .sst_ops import build_mlp, scatter_v2

_DEBUG = bool(int(os.environ.get('SST_AMD_DEBUG', '0')))


@MODELS.register_module()
class VirtualVoxelExtractor(nn.Module):

    def __init__(self, backbone, voxel_encoder, virtual_point_projector, train_cfg=None, test_cfg=None,
                 multiscale_cfg=None, bbox_head=None, as_rpn=None):
        super().__init__()
        self.voxel_encoder = build_voxel_encoder(voxel_encoder)
        self.virtual_voxel_size = voxel_encoder['voxel_size']
        self.point_cloud_range = voxel_encoder['point_cloud_range']
        self.backbone = build_backbone(backbone)
        vpp = virtual_point_projector
        self.virtual_proj = build_mlp(vpp['in_channels'], vpp['hidden_dims'], vpp['norm_cfg'])
        self.ori_proj = build_mlp(vpp['ori_in_channels'], vpp['ori_hidden_dims'], vpp['norm_cfg'])
        self.zero_virtual_feature = vpp.get('zero_virtual_feature', False)
        self.only_virtual = vpp.get('only_virtual', False)
        self.train_cfg = train_cfg or {}
        self.test_cfg = test_cfg or {}
        # the detector reads as_rpn off its bbox_head config (single_stage_fsd_v2.py:83); a bare flag is accepted too
        if as_rpn is None:
            as_rpn = bool((bbox_head or {}).get('as_rpn', False))
        self.as_rpn = bool(as_rpn)
        if self.as_rpn:
            self.recover_proj = build_mlp(vpp['recover_in_channels'], vpp['recover_hidden_dims'], vpp['norm_cfg'])
        if (self.train_cfg or self.test_cfg).get('baseline_mode', False):
            raise NotImplementedError('baseline_mode (extract_feat_baseline, no shipped config sets it) is not part of this stage')
        self.multiscale_cfg = multiscale_cfg
        if multiscale_cfg is not None:
            self.ms_projectors = nn.ModuleList([build_mlp(proj[0], proj[1:], multiscale_cfg['norm_cfg'])
                                                for proj in multiscale_cfg['projector_hiddens']])
            if multiscale_cfg['fusion_mode'] not in ('avg', 'mean', 'max', 'sum'):
                raise NotImplementedError(multiscale_cfg['fusion_mode'])
        self.print_info = {}

    @torch.no_grad()
    def voxelize_with_batch_idx(self, points, batch_idx):
        """floor((xyz - range_min) / voxel) in zyx order behind the sample index; no clamping (the centres were clipped,
        the original points lie inside the range by construction of the segmentor's voxelisation)"""
        xyz = points[:, :3]
        vs = K.const_tensor(self.virtual_voxel_size, xyz.device, xyz.dtype)
        lo = K.const_tensor(self.point_cloud_range[:3], xyz.device, xyz.dtype)
        cells = torch.div(xyz - lo[None], vs[None], rounding_mode='floor').long()
        return torch.cat([batch_idx[:, None], cells[:, [2, 1, 0]]], dim=1)

    def clip_points(self, points, pc_range):
        """IN PLACE, as the reference does (single_stage_fsd_v2.py:124-129 assigns into the columns of
        sampled_dict['center_preds']): callers that read the dictionary after extract_feat see the clipped centres."""
        eps = 1e-5
        lo = K.const_tensor(pc_range[:3], points.device, points.dtype) + eps
        hi = K.const_tensor(pc_range[3:], points.device, points.dtype) - eps
        return points.clamp_(min=lo, max=hi)

    # ------------------------------------------------------------------------------------------------ multi-scale fusion
    def ms_coors_proj(self, coors, sparse_shape):
        """coordinates of a coarser decoder level -> the cell of the target grid that holds the coarse cell's centre
        (single_stage_fsd_v2.py:399-433: integer strides, stride // 2 offset)"""
        tgt = self.multiscale_cfg['target_sparse_shape']
        bev_stride = tgt[1] // sparse_shape[1]
        z_stride = tgt[0] // sparse_shape[0]
        assert bev_stride == tgt[2] / sparse_shape[2]
        assert z_stride >= 1 and bev_stride >= 1
        stride = K.const_tensor([1, z_stride, bev_stride, bev_stride], coors.device, coors.dtype)
        shift = K.const_tensor([0, z_stride // 2, bev_stride // 2, bev_stride // 2], coors.device, coors.dtype)
        out = coors * stride[None] + shift[None]
        if _DEBUG:
            top = out.max(0)[0].tolist()
            assert top[1] < tgt[0] and top[2] < tgt[1] and top[3] < tgt[2], (top, tgt)
        return out

    def multiscale_fusion(self, ms_data, voxel_feats, coors):
        """-> (fused features, fused coordinates (sorted-unique of voxels + projected multi-scale voxels), mask of the rows
        that are the input voxels); single_stage_fsd_v2.py:375-397"""
        cfg = self.multiscale_cfg
        levels = [ms_data[lvl] for lvl in cfg['multiscale_levels']]
        ms_feats = [proj(data.features) for proj, data in zip(self.ms_projectors, levels)]
        ms_coors = [self.ms_coors_proj(data.indices, data.spatial_shape).to(coors.dtype) for data in levels]
        n = voxel_feats.size(0)
        cat_feats = torch.cat([voxel_feats] + ms_feats, 0)
        cat_coors = torch.cat([coors] + ms_coors, 0)
        out_feats, out_coors, inv = scatter_v2(cat_feats, cat_coors, mode=cfg['fusion_mode'], return_inv=True)
        singlescale_mask = torch.zeros(out_coors.size(0), dtype=torch.bool, device=coors.device)
        singlescale_mask[inv[:n]] = True
        if _DEBUG:
            assert int(singlescale_mask.sum()) == n
        return out_feats, out_coors, singlescale_mask

    # --------------------------------------------------------------------------------------------------- as_rpn outputs
    def recover_point_features(self, out_voxel_feats, out_coors, out_sparse_shape, cat_pts, cat_batch_idx,
                               voxel_encoder_coors, voxel_encoder_inv):
        """per-point features for the second stage: the point's voxel feature + its offset to the voxel centre in half-voxel
        units through recover_proj (single_stage_fsd_v2.py:131-155).  The submanifold mixer keeps the voxel set, so the
        reference's ``is_same`` branch is the only one it implements; checked under SST_AMD_DEBUG only (a read-back)."""
        if _DEBUG:
            assert bool((out_coors == voxel_encoder_coors).all())
        vs = K.const_tensor(self.virtual_voxel_size, out_voxel_feats.device, out_voxel_feats.dtype)
        lo = K.const_tensor(self.point_cloud_range[:3], out_voxel_feats.device, out_voxel_feats.dtype)
        coors_per_pts = out_coors[voxel_encoder_inv]
        feat_per_pts = out_voxel_feats[voxel_encoder_inv]
        center_per_pts = (coors_per_pts[:, [3, 2, 1]].to(vs.dtype) + 0.5) * vs[None] + lo[None]
        offset = (center_per_pts - cat_pts) / vs[None] * 2
        return self.recover_proj(torch.cat([feat_per_pts, offset], 1))

    # ------------------------------------------------------------------------------------------------------------ stage
    def extract_feat(self, sampled_dict, origin_dict, gt_bboxes_3d=None, multiscale_features=None):
        fg_pts, fg_batch = sampled_dict['seg_points'], sampled_dict['batch_idx']
        centers = self.clip_points(sampled_dict['center_preds'], self.point_cloud_range)   # votes may leave the range
        offset = (centers - fg_pts[:, :3]) / 10                                             # the reference's normaliser
        vir_feat = self.virtual_proj(torch.cat([sampled_dict['seg_feats'], offset, sampled_dict['seg_logits'],
                                                fg_pts[:, 3:]], 1))
        if self.zero_virtual_feature:
            vir_feat = vir_feat * 0
        ori_pts = origin_dict['seg_points']
        ori_feat = self.ori_proj(origin_dict['seg_feats'])

        n_ori, n_vir = ori_pts.size(0), centers.size(0)
        cat_pts = torch.cat([ori_pts[:, :3], centers], 0)
        cat_batch = torch.cat([origin_dict['batch_idx'], fg_batch], 0)
        coors = self.voxelize_with_batch_idx(cat_pts, cat_batch)
        voxel_feats, voxel_coors, unq_inv = self.voxel_encoder(torch.cat([cat_pts, torch.cat([ori_feat, vir_feat], 0)], 1),
                                                               coors, return_inv=True)
        # a voxel is "virtual" if any voted centre fell into it: mean of the 0 / 1 point indicators > 0
        indicators = cat_pts.new_zeros((n_ori + n_vir, 1))
        indicators[n_ori:] = 1.0
        grouped_coors = voxel_coors
        voxel_ind, _ = scatter_v2(indicators, coors, mode='avg', return_inv=False, unq_inv=unq_inv, new_coors=grouped_coors)
        virtual_mask = voxel_ind[:, 0] > 0
        batch_size = int(cat_batch.max().item()) + 1 if 'batch_size' not in origin_dict else int(origin_dict['batch_size'])

        singlescale_mask = None
        if multiscale_features is not None:
            if self.multiscale_cfg is None:
                raise ValueError('multiscale_features were passed to a stage built without multiscale_cfg')
            voxel_feats, voxel_coors, singlescale_mask = self.multiscale_fusion(multiscale_features, voxel_feats, voxel_coors)
        if self.only_virtual:
            assert multiscale_features is None
            voxel_feats, voxel_coors = voxel_feats[virtual_mask], voxel_coors[virtual_mask]
        out_feats, out_coors, sparse_shape = self.backbone(voxel_feats, voxel_coors, batch_size)
        if singlescale_mask is not None:       # back to the rows of the voxel encoder (same order: both are sorted-unique)
            out_feats, out_coors = out_feats[singlescale_mask], out_coors[singlescale_mask]

        vs = K.const_tensor(self.virtual_voxel_size, out_feats.device, out_feats.dtype)
        lo = K.const_tensor(self.point_cloud_range[:3], out_feats.device, out_feats.dtype)
        voxel_centers = (out_coors[:, [3, 2, 1]].to(out_feats.dtype) + 0.5) * vs[None] + lo[None]
        if self.only_virtual:
            out = dict(virtual_feats=out_feats, virtual_coors=out_coors, virtual_centers=voxel_centers)
        else:
            out = dict(virtual_feats=out_feats[virtual_mask], virtual_coors=out_coors[virtual_mask],
                       virtual_centers=voxel_centers[virtual_mask])
        out['sparse_shape'] = sparse_shape
        if self.training:
            self.print_info['num_virtual'] = out_feats.new_ones(1) * out['virtual_feats'].size(0)
            if self.train_cfg.get('centroid_alpha', None) is not None:
                raise NotImplementedError('centroid_alpha needs the ground-truth boxes (detector side)')
            centroid, _ = scatter_v2(cat_pts, coors, mode='avg', return_inv=False, unq_inv=unq_inv, new_coors=grouped_coors)
            out['virtual_centroid'] = centroid[virtual_mask]
        if self.as_rpn:
            out['pts_feats'] = self.recover_point_features(out_feats, out_coors, sparse_shape, cat_pts, cat_batch,
                                                           grouped_coors, unq_inv)
            out['pts_xyz'] = cat_pts
            out['pts_indicators'] = indicators[:, 0]
            out['pts_batch_inds'] = cat_batch
        return out

    forward = extract_feat
