"""FSDv2's virtual-voxel stage: foreground points vote for object centres, the voted centres ("virtual points") are
voxelised together with the original points, and a sparse U-Net mixes real and virtual voxels.

Reference: SingleStageFSDV2.extract_feat, mmdet3d/models/detectors/single_stage_fsd_v2.py:159-271 (non-baseline mode) with
its helpers voxelize_with_batch_idx (:107-122) and clip_points (:124-129); constructor arguments as in
SingleStageFSDV2.__init__ (:38-105) for the part that belongs to this stage (``backbone``, ``voxel_encoder``,
``virtual_point_projector``), submodule names as there (``virtual_proj``, ``ori_proj``, ``voxel_encoder``, ``backbone``)
so that a detector checkpoint's keys for them load unchanged.  The detector around it (segmentor, heads, losses, box
decoding) is out of scope.

Differences in execution, not in results: the reference groups the same coordinates three times (voxel encoder, the
virtual-indicator average, the training-time centroid); here the voxel encoder's grouping (one radix sort of the packed
coordinates) is reused by the two segmented averages.
"""
import torch
from torch import nn

from .registry import MODELS, build_backbone, build_voxel_encoder
from . import kernels as K
from .sst_ops import build_mlp, scatter_v2


@MODELS.register_module()
class VirtualVoxelExtractor(nn.Module):

    def __init__(self, backbone, voxel_encoder, virtual_point_projector, train_cfg=None, test_cfg=None,
                 multiscale_cfg=None):
        super().__init__()
        if multiscale_cfg is not None:
            raise NotImplementedError('multiscale fusion (single_stage_fsd_v2.py:273-340) is not part of this stage')
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
        if (self.train_cfg or {}).get('as_rpn', False) or (self.test_cfg or {}).get('as_rpn', False):
            # the detector takes as_rpn from its bbox_head config (single_stage_fsd_v2.py:83) and then returns pts_feats /
            # pts_xyz / ... for a second stage (:263-270): not produced by this stage
            raise NotImplementedError('as_rpn outputs (single_stage_fsd_v2.py:263-270) are not part of this stage')
        if (self.train_cfg or self.test_cfg).get('baseline_mode', False):
            raise NotImplementedError('baseline_mode (extract_feat_baseline) is not part of this stage')
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

    def extract_feat(self, sampled_dict, origin_dict):
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

        if self.only_virtual:
            voxel_feats, voxel_coors = voxel_feats[virtual_mask], voxel_coors[virtual_mask]
        out_feats, out_coors, sparse_shape = self.backbone(voxel_feats, voxel_coors, batch_size)

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
        return out

    forward = extract_feat
