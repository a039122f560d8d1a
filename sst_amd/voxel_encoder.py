"""Voxel / point-group encoders on the hot path: DynamicVFE, DynamicScatterVFE, SIRLayer.

Mirrors mmdet3d/models/voxel_encoders/voxel_encoder.py:92-298 (DynamicVFE), :502-612
(DynamicScatterVFE), :617-764 (SIRLayer) and voxel_encoders/utils.py:107-189 (DynamicVFELayer,
DynamicVFELayerV2): same registry names, constructor kwargs, forward signatures and parameter names
(``vfe_layers.{i}.linear.weight``, ``vfe_layers.{i}.norm.*``, ``rel_mlp.*``).

Reference cost being removed: every DynamicScatter call re-runs at::unique_dim on the same coordinates
(3 sorts per frame in DynamicVFE) and reduces with float atomics; map_voxel_center_to_point builds a
dense B*z*y*x long canvas.  Here the point->voxel grouping is computed once per forward
(voxel.build_scatter_plan), every reduce is a CSR segmented reduce, and the point<-voxel lookup is a
gather by the inverse map.
"""
import torch
from torch import nn
from torch.nn import functional as F

from . import kernels as K
from .dense import add_layer_norm, tall_linear
from .norm import batch_norm_act, build_norm_layer
from .registry import VOXEL_ENCODERS
from .sst_ops import build_mlp, get_activation_layer, scatter_v2, unique_with_plan
from .voxel import DynamicScatter, build_scatter_plan


class DynamicVFELayer(nn.Module):
    """Linear(no bias) -> norm -> ReLU (utils.py:107-144)."""

    def __init__(self, in_channels, out_channels, norm_cfg=dict(type='BN1d', eps=1e-3, momentum=0.01)):
        super(DynamicVFELayer, self).__init__()
        self.fp16_enabled = False
        self.norm = build_norm_layer(norm_cfg, out_channels)[1]
        self.linear = nn.Linear(in_channels, out_channels, bias=False)

    def forward(self, inputs):
        x = tall_linear(inputs, self.linear.weight, None)
        if isinstance(self.norm, nn.BatchNorm1d):
            return batch_norm_act(self.norm, x, relu=True)  # fused norm + ReLU (csrc/bn.hip)
        return F.relu(self.norm(x))


class DynamicVFELayerV2(nn.Module):
    """[dropout] -> Linear(no bias) -> norm -> act (utils.py:147-189)."""

    def __init__(self, in_channels, out_channels, norm_cfg=dict(type='BN1d', eps=1e-3, momentum=0.01), act='relu',
                 dropout=0.0):
        super(DynamicVFELayerV2, self).__init__()
        self.fp16_enabled = False
        self.norm = build_norm_layer(norm_cfg, out_channels)[1]
        self.linear = nn.Linear(in_channels, out_channels, bias=False)
        self.act = get_activation_layer(act, out_channels)
        self.dropout = nn.Dropout(p=dropout) if dropout > 0 else None

    def forward(self, inputs):
        if self.dropout is not None:
            inputs = self.dropout(inputs)
        x = tall_linear(inputs, self.linear.weight, None)
        if isinstance(self.norm, nn.BatchNorm1d):
            if isinstance(self.act, nn.ReLU):
                return batch_norm_act(self.norm, x, relu=True)
            return self.act(batch_norm_act(self.norm, x, relu=False))
        if isinstance(self.norm, nn.LayerNorm):  # FSD's SIR layers: norm_cfg = LN (row kernel of csrc/dense.hip)
            return self.act(add_layer_norm(x, None, self.norm))
        return self.act(self.norm(x))


@VOXEL_ENCODERS.register_module()
class DynamicVFE(nn.Module):
    """Dynamic voxel feature encoder (voxel_encoder.py:92-298)."""

    def __init__(self,
                 in_channels=4,
                 feat_channels=[],
                 with_distance=False,
                 with_cluster_center=False,
                 with_voxel_center=False,
                 voxel_size=(0.2, 0.2, 4),
                 point_cloud_range=(0, -40, -3, 70.4, 40, 1),
                 norm_cfg=dict(type='BN1d', eps=1e-3, momentum=0.01),
                 mode='max',
                 fusion_layer=None,
                 return_point_feats=False,
                 reference_compat=True,
                 ):
        super(DynamicVFE, self).__init__()
        assert len(feat_channels) > 0
        if with_cluster_center:
            in_channels += 3
        if with_voxel_center:
            in_channels += 3
        if with_distance:
            in_channels += 3
        self.in_channels = in_channels
        self._with_distance = with_distance
        self._with_cluster_center = with_cluster_center
        self._with_voxel_center = with_voxel_center
        self.return_point_feats = return_point_feats
        self.fp16_enabled = False
        self.reference_compat = reference_compat

        self.vx = voxel_size[0]
        self.vy = voxel_size[1]
        self.vz = voxel_size[2]
        self.x_offset = self.vx / 2 + point_cloud_range[0]
        self.y_offset = self.vy / 2 + point_cloud_range[1]
        self.z_offset = self.vz / 2 + point_cloud_range[2]
        self.point_cloud_range = point_cloud_range
        self.scatter = DynamicScatter(voxel_size, point_cloud_range, True, reference_compat)

        feat_channels = [self.in_channels] + list(feat_channels)
        vfe_layers = []
        for i in range(len(feat_channels) - 1):
            in_filters = feat_channels[i]
            out_filters = feat_channels[i + 1]
            if i > 0:
                in_filters *= 2
            vfe_layers.append(DynamicVFELayer(in_filters, out_filters, norm_cfg))
        self.vfe_layers = nn.ModuleList(vfe_layers)
        self.num_vfe = len(vfe_layers)
        self.mode = mode
        self.vfe_scatter = DynamicScatter(voxel_size, point_cloud_range, (mode != 'max'), reference_compat)
        self.cluster_scatter = DynamicScatter(voxel_size, point_cloud_range, average_points=True,
                                              reference_compat=reference_compat)
        self.fusion_layer = None
        if fusion_layer is not None:
            raise NotImplementedError('image fusion layers are outside the LiDAR hot path')

    def _grid_zyx(self):
        """Extents of the voxel key space: the grid the voxelizer itself clamps to (fp32 ceil, voxelize.hip), which for
        ranges that are no multiple of the voxel size is one larger than the round() of the reference's canvas
        (voxel_encoder.py:198-203); the larger of the two, so that no clamped coordinate can alias another key."""
        r = self.point_cloud_range
        gx, gy, gz = K.voxel_grid((self.vx, self.vy, self.vz), list(r))
        return [max(gz, round((r[5] - r[2]) / self.vz)), max(gy, round((r[4] - r[1]) / self.vy)),
                max(gx, round((r[3] - r[0]) / self.vx))]

    def map_voxel_center_to_point(self, pts_coors, voxel_mean, voxel_coors, plan=None):
        """voxel feature of every point.  The reference scatters voxel ids into a dense zero-initialised
        canvas (voxel_encoder.py:185-225): a point whose voxel is not in ``voxel_coors`` reads canvas
        value 0, i.e. voxel 0 — reproduced by clamping the inverse map at 0."""
        if plan is None:
            plan = build_scatter_plan(pts_coors, reference_compat=self.reference_compat)
        idx = plan.coors_map.long().clamp(min=0)
        return voxel_mean[idx, ...]

    def scatter_plan(self, coors):
        """The point -> voxel grouping this encoder uses for ``coors`` (index work only, no parameters): callers
        that pipeline frames can build it ahead of time and hand it to forward(..., scatter_plan=...)."""
        coors = coors.contiguous()
        if coors.size(1) == 4:
            return build_scatter_plan(coors, grid_zyx=self._grid_zyx(), reference_compat=self.reference_compat)
        return build_scatter_plan(coors, reference_compat=self.reference_compat)

    def forward(self, features, coors, points=None, img_feats=None, img_metas=None, scatter_plan=None):
        features = features.float()  # @force_fp32 (voxel_encoder.py:229)
        coors = coors.contiguous()
        plan = scatter_plan if scatter_plan is not None else self.scatter_plan(coors)
        inv = plan.coors_map.long().clamp(min=0)

        # decorate: [features | xyz - voxel mean | xyz - voxel centre] in one launch (bit-identical to the composed
        # subtractions / cat of voxel_encoder.py:252-271)
        if features.is_cuda and features.stride(1) == 1 and coors.size(1) == 4 and not features.requires_grad:
            voxel_mean = plan.reduce(features, 'mean') if self._with_cluster_center else None
            decorated = K.vfe_decorate(features, plan.coors_map, voxel_mean, 1.0, coors, (self.vx, self.vy, self.vz),
                                       (self.x_offset, self.y_offset, self.z_offset), self._with_cluster_center,
                                       self._with_voxel_center)
            features_ls = [decorated]
        else:
            features_ls = [features]
            if self._with_cluster_center:
                voxel_mean = plan.reduce(features, 'mean')
                points_mean = voxel_mean[inv]
                f_cluster = features[:, :3] - points_mean[:, :3]
                features_ls.append(f_cluster)

            if self._with_voxel_center:
                f_center = features.new_zeros(size=(features.size(0), 3))
                f_center[:, 0] = features[:, 0] - (coors[:, 3].type_as(features) * self.vx + self.x_offset)
                f_center[:, 1] = features[:, 1] - (coors[:, 2].type_as(features) * self.vy + self.y_offset)
                f_center[:, 2] = features[:, 2] - (coors[:, 1].type_as(features) * self.vz + self.z_offset)
                features_ls.append(f_center)

        if self._with_distance:
            points_dist = torch.norm(features[:, :3], 2, 1, keepdim=True)
            features_ls.append(points_dist)

        features = torch.cat(features_ls, dim=-1) if len(features_ls) > 1 else features_ls[0]
        reduce_mode = 'max' if self.mode == 'max' else 'mean'
        for i, vfe in enumerate(self.vfe_layers):
            point_feats = vfe(features)
            voxel_feats = plan.reduce(point_feats, reduce_mode)
            if i != len(self.vfe_layers) - 1:
                feat_per_point = voxel_feats[inv]
                features = torch.cat([point_feats, feat_per_point], dim=1)
        if self.return_point_feats:
            return point_feats
        return voxel_feats, plan.voxel_coors


@VOXEL_ENCODERS.register_module()
class DynamicScatterVFE(DynamicVFE):
    """DynamicVFE on scatter_v2 (voxel_encoder.py:502-612): no "first row" quirk, int64 coordinates."""

    def __init__(self,
                 in_channels=4,
                 feat_channels=[],
                 with_distance=False,
                 with_cluster_center=False,
                 with_voxel_center=False,
                 voxel_size=(0.2, 0.2, 4),
                 point_cloud_range=(0, -40, -3, 70.4, 40, 1),
                 norm_cfg=dict(type='BN1d', eps=1e-3, momentum=0.01),
                 mode='max',
                 fusion_layer=None,
                 return_point_feats=False,
                 return_inv=True,
                 rel_dist_scaler=1.0,
                 unique_once=False,
                 ):
        super(DynamicScatterVFE, self).__init__(in_channels, feat_channels, with_distance, with_cluster_center,
                                                with_voxel_center, voxel_size, point_cloud_range, norm_cfg, mode,
                                                fusion_layer, return_point_feats)
        self.scatter = None
        self.vfe_scatter = None
        self.cluster_scatter = None
        self.rel_dist_scaler = rel_dist_scaler
        self.mode = mode
        self.unique_once = unique_once

    def map_voxel_center_to_point(self, voxel_mean, voxel2point_inds):
        return voxel_mean[voxel2point_inds]

    def forward(self, features, coors, points=None, img_feats=None, img_metas=None, return_inv=False):
        features = features.float()
        if self.unique_once:
            new_coors, unq_inv_once = unique_with_plan(coors)
        else:
            new_coors = unq_inv_once = None

        if features.is_cuda and features.stride(1) == 1 and coors.size(1) == 4 and not features.requires_grad:
            # decorate in one launch (bit-identical to the composed ops of voxel_encoder.py:569-589)
            voxel_mean = unq_inv = None
            if self._with_cluster_center:
                voxel_mean, _, unq_inv = scatter_v2(features[:, :3], coors, mode='avg', new_coors=new_coors,
                                                    unq_inv=unq_inv_once)
            features_ls = [K.vfe_decorate(features, unq_inv, voxel_mean, self.rel_dist_scaler, coors,
                                          (self.vx, self.vy, self.vz), (self.x_offset, self.y_offset, self.z_offset),
                                          self._with_cluster_center, self._with_voxel_center)]
        else:
            features_ls = [features]
            if self._with_cluster_center:
                voxel_mean, _, unq_inv = scatter_v2(features[:, :3], coors, mode='avg', new_coors=new_coors,
                                                    unq_inv=unq_inv_once)
                points_mean = self.map_voxel_center_to_point(voxel_mean, unq_inv)
                f_cluster = features[:, :3] - points_mean[:, :3]
                features_ls.append(f_cluster / self.rel_dist_scaler)

            if self._with_voxel_center:
                f_center = features.new_zeros(size=(features.size(0), 3))
                f_center[:, 0] = features[:, 0] - (coors[:, 3].type_as(features) * self.vx + self.x_offset)
                f_center[:, 1] = features[:, 1] - (coors[:, 2].type_as(features) * self.vy + self.y_offset)
                f_center[:, 2] = features[:, 2] - (coors[:, 1].type_as(features) * self.vz + self.z_offset)
                features_ls.append(f_center)

        if self._with_distance:
            points_dist = torch.norm(features[:, :3], 2, 1, keepdim=True)
            features_ls.append(points_dist)

        features = torch.cat(features_ls, dim=-1) if len(features_ls) > 1 else features_ls[0]
        for i, vfe in enumerate(self.vfe_layers):
            point_feats = vfe(features)
            voxel_feats, voxel_coors, unq_inv = scatter_v2(point_feats, coors, mode=self.mode, new_coors=new_coors,
                                                           unq_inv=unq_inv_once)
            if i != len(self.vfe_layers) - 1:
                feat_per_point = self.map_voxel_center_to_point(voxel_feats, unq_inv)
                features = torch.cat([point_feats, feat_per_point], dim=1)
        if self.return_point_feats:
            return point_feats
        if return_inv:
            return voxel_feats, voxel_coors, unq_inv
        return voxel_feats, voxel_coors


@VOXEL_ENCODERS.register_module()
class SIRLayer(DynamicVFE):
    """FSD point-group MLP + scatter-max pooling (voxel_encoder.py:617-764)."""

    def __init__(self,
                 in_channels=4,
                 feat_channels=[],
                 with_distance=False,
                 with_cluster_center=False,
                 with_rel_mlp=True,
                 rel_mlp_hidden_dims=[16, ],
                 rel_mlp_in_channel=3,
                 with_voxel_center=False,
                 voxel_size=(0.2, 0.2, 4),
                 point_cloud_range=(0, -40, -3, 70.4, 40, 1),
                 norm_cfg=dict(type='BN1d', eps=1e-3, momentum=0.01),
                 mode='max',
                 fusion_layer=None,
                 return_point_feats=False,
                 return_inv=True,
                 rel_dist_scaler=1.0,
                 with_shortcut=True,
                 xyz_normalizer=[1.0, 1.0, 1.0],
                 act='relu',
                 dropout=0.0,
                 ):
        super().__init__(in_channels, feat_channels, with_distance, with_cluster_center, with_voxel_center,
                         voxel_size, point_cloud_range, norm_cfg, mode, fusion_layer, return_point_feats)
        self.scatter = None
        self.vfe_scatter = None
        self.cluster_scatter = None
        self.rel_dist_scaler = rel_dist_scaler
        self.mode = mode
        self.with_shortcut = with_shortcut
        self._with_rel_mlp = with_rel_mlp
        self.xyz_normalizer = xyz_normalizer
        if with_rel_mlp:
            # 3 -> hidden... -> in_channels (the reference appends to its argument in place,
            # voxel_encoder.py:665; a copy is used here so a shared config list is not mutated)
            dims = list(rel_mlp_hidden_dims) + [in_channels]  # not self.in_channels
            self.rel_mlp = build_mlp(rel_mlp_in_channel, dims, norm_cfg, act=act)

        if act != 'relu' or dropout > 0:  # do not double in_filter
            feat_channels = [self.in_channels] + list(feat_channels)
            vfe_layers = []
            for i in range(len(feat_channels) - 1):
                in_filters = feat_channels[i]
                out_filters = feat_channels[i + 1]
                if i > 0:
                    in_filters *= 2
                vfe_layers.append(DynamicVFELayerV2(in_filters, out_filters, norm_cfg, act=act, dropout=dropout))
            self.vfe_layers = nn.ModuleList(vfe_layers)
            self.num_vfe = len(vfe_layers)

    def map_voxel_center_to_point(self, voxel_mean, voxel2point_inds):
        return voxel_mean[voxel2point_inds]

    def forward(self,
                features,
                coors,
                f_cluster=None,
                points=None,
                img_feats=None,
                img_metas=None,
                return_inv=False,
                return_both=False,
                unq_inv_once=None,
                new_coors_once=None,
                ):
        features = features.float()
        xyz_normalizer = torch.tensor(self.xyz_normalizer, device=features.device, dtype=features.dtype)
        features_ls = [torch.cat([features[:, :3] / xyz_normalizer[None, :], features[:, 3:]], dim=1)]
        if self.with_shortcut:
            shortcut = features[:, 3:]
        if f_cluster is None:
            voxel_mean, mean_coors, unq_inv = scatter_v2(features[:, :3], coors, mode='avg', unq_inv=unq_inv_once,
                                                         new_coors=new_coors_once)
            points_mean = self.map_voxel_center_to_point(voxel_mean, unq_inv)
            f_cluster = (features[:, :3] - points_mean[:, :3]) / self.rel_dist_scaler
        else:
            f_cluster = f_cluster / self.rel_dist_scaler

        if self._with_cluster_center:
            features_ls.append(f_cluster / 10.0)

        if self._with_rel_mlp:
            features_ls[0] = features_ls[0] * self.rel_mlp(f_cluster)

        if self._with_distance:
            points_dist = torch.norm(features[:, :3], 2, 1, keepdim=True)
            features_ls.append(points_dist)

        features = torch.cat(features_ls, dim=-1)

        voxel_feats_list = []
        for i, vfe in enumerate(self.vfe_layers):
            point_feats = vfe(features)
            voxel_feats, voxel_coors, unq_inv = scatter_v2(point_feats, coors, mode=self.mode, unq_inv=unq_inv_once,
                                                           new_coors=new_coors_once)
            voxel_feats_list.append(voxel_feats)
            if i != len(self.vfe_layers) - 1:
                feat_per_point = self.map_voxel_center_to_point(voxel_feats, unq_inv)
                features = torch.cat([point_feats, feat_per_point], dim=1)

        voxel_feats = torch.cat(voxel_feats_list, dim=1)

        if return_both:
            if self.with_shortcut and point_feats.shape == shortcut.shape:
                point_feats = point_feats + shortcut
            return point_feats, voxel_feats, voxel_coors

        if self.return_point_feats:
            if self.with_shortcut and point_feats.shape == shortcut.shape:
                point_feats = point_feats + shortcut
            return point_feats, voxel_feats

        if return_inv:
            return voxel_feats, voxel_coors, unq_inv
        return voxel_feats, voxel_coors
