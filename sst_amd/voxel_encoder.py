"""Point-group encoders of the hot path: DynamicVFE, DynamicScatterVFE, SIRLayer.

Drop-in for the reference's registry entries (mmdet3d/models/voxel_encoders/voxel_encoder.py:92-298 DynamicVFE,
:502-612 DynamicScatterVFE, :617-764 SIRLayer; layers voxel_encoders/utils.py:107-189): same constructor keywords,
forward signatures and parameter names (``vfe_layers.{i}.linear.weight``, ``vfe_layers.{i}.norm.*``, ``rel_mlp.*``).

All three are the same computation on different groupings of the points - per layer
``Linear -> norm -> activation`` on the points, a pooled feature per group, and the pooled feature handed back to the
group's points next to their own - so they share one implementation here (`_PointGroupEncoder`), parameterised by a
*grouping* object:

    grouping.index      [N] point -> row of the pooled tensor (negative: row 0, DynamicVFE's zero-initialised canvas)
    grouping.reduce     segmented max / mean over the CSR of the grouping (csrc/scatter.hip; no atomics)
    grouping.coors      coordinates of the groups
    grouping.group_sum  the CSR sum used as the gradient of the hand-back (None: the index has rows outside the CSR)

The reference recomputes the grouping inside every pooling call (``at::unique_dim`` three times per DynamicVFE
forward, ``torch.unique`` per scatter_v2) and builds a dense B*z*y*x canvas to find a point's voxel; here it is
computed once per forward (or handed in: ``scatter_plan=``, ``unq_inv_once=``), and the hand-back
``cat([point_feats, pooled[index]])`` is one kernel (``kernels.concat_gather``).
"""
import torch
from torch import nn
from torch.nn import functional as F

from . import _lib
from . import kernels as K
from .dense import add_layer_norm, tall_linear
from .norm import batch_norm_act, build_norm_layer
from .registry import VOXEL_ENCODERS
from .sst_ops import build_mlp, get_activation_layer, plan_of_inverse, unique_with_plan
from .vfe_fused import UniquePlanAdapter, fused_vfe2, fused_vfe2_ok
from .voxel import DynamicScatter, build_scatter_plan


# ------------------------------------------------------------------------------------------------------------------
# per-point layers
# ------------------------------------------------------------------------------------------------------------------
class DynamicVFELayer(nn.Module):
    """Linear(no bias) -> norm -> ReLU (utils.py:107-144); norm + ReLU run as one kernel pair (csrc/bn.hip)."""

    def __init__(self, in_channels, out_channels, norm_cfg=dict(type='BN1d', eps=1e-3, momentum=0.01)):
        super(DynamicVFELayer, self).__init__()
        self.fp16_enabled = False
        self.norm = build_norm_layer(norm_cfg, out_channels)[1]
        self.linear = nn.Linear(in_channels, out_channels, bias=False)

    def forward(self, inputs):
        y = tall_linear(inputs, self.linear.weight, None)
        if isinstance(self.norm, nn.BatchNorm1d):
            return batch_norm_act(self.norm, y, relu=True)
        return F.relu(self.norm(y))


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
        x = inputs if self.dropout is None else self.dropout(inputs)
        y = tall_linear(x, self.linear.weight, None)
        if isinstance(self.norm, nn.BatchNorm1d):
            relu = isinstance(self.act, nn.ReLU)
            y = batch_norm_act(self.norm, y, relu=relu)
            return y if relu else self.act(y)
        if isinstance(self.norm, nn.LayerNorm):  # FSD's SIR layers: norm_cfg = LN (row kernel of csrc/dense.hip)
            return add_layer_norm(y, None, self.norm, act=self.act)   # norm + activation in one pass
        return self.act(self.norm(y))


# ------------------------------------------------------------------------------------------------------------------
# groupings
# ------------------------------------------------------------------------------------------------------------------
class _VoxelGrouping(object):
    """points grouped by voxel through a voxel.ScatterPlan or a frame_plan.FramePlan (DynamicVFE; the "first voxel of
    every sample" quirk lives in the plan: its points carry index -1 and read row 0)."""

    def __init__(self, plan):
        self.plan = plan
        self.index = plan.coors_map
        self.coors = plan.voxel_coors
        # points of a discarded voxel hand their gradient to row 0 (not a CSR member of it): the plan adds their rows in
        self.group_sum = getattr(plan, 'group_sum', None)

    def reduce(self, feats, mode):
        return self.plan.reduce(feats, mode)


class _UniqueGrouping(object):
    """points grouped by their integer coordinate rows: the grouping of scatter_v2 (ops/sst/sst_ops.py:151-182)"""

    def __init__(self, coors, new_coors=None, unq_inv=None):
        if unq_inv is None:
            new_coors, unq_inv = unique_with_plan(coors)
        self.plan = plan_of_inverse(unq_inv, new_coors.size(0))
        self.index = self.plan.inverse
        self.unq_inv = unq_inv
        self.coors = new_coors
        self.group_sum = lambda part: K.segment_reduce(part, self.plan, 'sum')

    def reduce(self, feats, mode):
        return K.segment_reduce(feats.contiguous(), self.plan, 'mean' if mode == 'avg' else mode)


# ------------------------------------------------------------------------------------------------------------------
# shared implementation
# ------------------------------------------------------------------------------------------------------------------
class _PointGroupEncoder(_lib.Fp32Master, nn.Module):

    def _init_common(self, in_channels, feat_channels, with_distance, with_cluster_center, with_voxel_center,
                     voxel_size, point_cloud_range, mode, return_point_feats, fusion_layer):
        assert len(feat_channels) > 0
        if fusion_layer is not None:
            raise NotImplementedError('image fusion layers are outside the LiDAR hot path')
        # decorated input width (voxel_encoder.py:143-149): + cluster offset, + voxel-centre offset, + (sic) 3 for the norm
        self.in_channels = in_channels + 3 * (bool(with_cluster_center) + bool(with_voxel_center) + bool(with_distance))
        self._with_distance = with_distance
        self._with_cluster_center = with_cluster_center
        self._with_voxel_center = with_voxel_center
        self.return_point_feats = return_point_feats
        self.fp16_enabled = False
        self.mode = mode
        self.vx, self.vy, self.vz = voxel_size[0], voxel_size[1], voxel_size[2]
        self.point_cloud_range = point_cloud_range
        self.x_offset = self.vx / 2 + point_cloud_range[0]
        self.y_offset = self.vy / 2 + point_cloud_range[1]
        self.z_offset = self.vz / 2 + point_cloud_range[2]
        self.fusion_layer = None

    def _layer_widths(self, feat_channels):
        """(in, out) of every layer: from the second layer on the input is [own feature | pooled feature]"""
        widths, prev = [], self.in_channels
        for i, out in enumerate(feat_channels):
            widths.append((prev if i == 0 else 2 * prev, out))
            prev = out
        return widths

    def _voxel_centre_offsets(self, features, coors):
        """xyz minus the centre of the point's voxel, composed as the reference does (voxel_encoder.py:264-272):
        coordinate * voxel size + (half a voxel + range minimum), one rounding per operation"""
        centre = torch.stack([coors[:, 3].type_as(features) * self.vx + self.x_offset,
                              coors[:, 2].type_as(features) * self.vy + self.y_offset,
                              coors[:, 1].type_as(features) * self.vz + self.z_offset], dim=1)
        return features[:, :3] - centre

    def _decorate(self, features, coors, grouping, cluster_div=1.0, mean_of_xyz_only=False):
        """[features | xyz - group mean (/ cluster_div) | xyz - voxel centre | norm]: one kernel for plain CUDA inputs
        (bit-identical to the composition), the composition itself when a gradient has to flow into ``features``."""
        mean = None
        if self._with_cluster_center:
            mean = grouping.reduce(features[:, :3] if mean_of_xyz_only else features, 'mean')
        fast = (features.is_cuda and features.stride(1) == 1 and coors.size(1) == 4 and not features.requires_grad)
        if fast:
            cols = [K.vfe_decorate(features, grouping.index, mean, cluster_div, coors, (self.vx, self.vy, self.vz),
                                   (self.x_offset, self.y_offset, self.z_offset), self._with_cluster_center,
                                   self._with_voxel_center)]
        else:
            cols = [features]
            if self._with_cluster_center:
                rel = features[:, :3] - mean[grouping.index.long().clamp(min=0)][:, :3]
                cols.append(rel if cluster_div == 1.0 else rel / cluster_div)
            if self._with_voxel_center:
                cols.append(self._voxel_centre_offsets(features, coors))
        if self._with_distance:
            cols.append(torch.norm(features[:, :3], 2, 1, keepdim=True))
        return cols[0] if len(cols) == 1 else torch.cat(cols, dim=-1)

    def _encode(self, x, grouping, mode):
        """the layer stack: returns (point features of the last layer, pooled features of every layer)"""
        pooled, last = [], len(self.vfe_layers) - 1
        for li, layer in enumerate(self.vfe_layers):
            x = layer(x)
            pooled.append(grouping.reduce(x, mode))
            if li != last:
                x = K.concat_gather(x, pooled[-1], grouping.index, grouping.group_sum)
        return x, pooled


@VOXEL_ENCODERS.register_module()
class DynamicVFE(_PointGroupEncoder):
    """Dynamic voxel feature encoder (voxel_encoder.py:92-298) on DynamicScatter's grouping."""

    def __init__(self, in_channels=4, feat_channels=[], with_distance=False, with_cluster_center=False,
                 with_voxel_center=False, voxel_size=(0.2, 0.2, 4), point_cloud_range=(0, -40, -3, 70.4, 40, 1),
                 norm_cfg=dict(type='BN1d', eps=1e-3, momentum=0.01), mode='max', fusion_layer=None,
                 return_point_feats=False, reference_compat=True):
        super(DynamicVFE, self).__init__()
        self._init_common(in_channels, feat_channels, with_distance, with_cluster_center, with_voxel_center, voxel_size,
                          point_cloud_range, mode, return_point_feats, fusion_layer)
        self.reference_compat = reference_compat
        self.fused_stack = True     # False: layer by layer through the modules (what the fused node is tested against)
        self.vfe_layers = nn.ModuleList([DynamicVFELayer(cin, cout, norm_cfg)
                                         for cin, cout in self._layer_widths(feat_channels)])
        self.num_vfe = len(self.vfe_layers)
        # parameter-free members of the reference module, kept for code that reaches for them
        self.scatter = DynamicScatter(voxel_size, point_cloud_range, True, reference_compat)
        self.vfe_scatter = DynamicScatter(voxel_size, point_cloud_range, (mode != 'max'), reference_compat)
        self.cluster_scatter = DynamicScatter(voxel_size, point_cloud_range, True, reference_compat)

    def _grid_zyx(self):
        """Extents of the voxel key space: the grid the voxelizer itself clamps to (fp32 ceil, voxelize.hip), which for
        ranges that are no multiple of the voxel size is one larger than the round() of the reference's canvas
        (voxel_encoder.py:198-203); the larger of the two, so that no clamped coordinate can alias another key."""
        r = self.point_cloud_range
        gx, gy, gz = K.voxel_grid((self.vx, self.vy, self.vz), list(r))
        return [max(gz, round((r[5] - r[2]) / self.vz)), max(gy, round((r[4] - r[1]) / self.vy)),
                max(gx, round((r[3] - r[0]) / self.vx))]

    def scatter_plan(self, coors):
        """The point -> voxel grouping this encoder uses for ``coors`` (index work only, no parameters): callers
        that pipeline frames can build it ahead of time and hand it to forward(..., scatter_plan=...)."""
        coors = coors.contiguous()
        grid = self._grid_zyx() if coors.size(1) == 4 else None
        return build_scatter_plan(coors, grid_zyx=grid, reference_compat=self.reference_compat)

    def map_voxel_center_to_point(self, pts_coors, voxel_mean, voxel_coors, plan=None):
        """voxel feature of every point.  The reference looks the voxel up in a dense zero-initialised canvas
        (voxel_encoder.py:185-225): a point whose voxel is not in ``voxel_coors`` reads voxel 0."""
        plan = plan if plan is not None else self.scatter_plan(pts_coors)
        return voxel_mean[plan.coors_map.long().clamp(min=0), ...]

    def forward(self, features, coors, points=None, img_feats=None, img_metas=None, scatter_plan=None):
        features = features.float()   # @force_fp32 (voxel_encoder.py:229)
        coors = coors.contiguous()
        grouping = _VoxelGrouping(scatter_plan if scatter_plan is not None else self.scatter_plan(coors))
        x = self._decorate(features, coors, grouping)
        if self.fused_stack and fused_vfe2_ok(self, x, grouping.plan):
            # the two-layer stack as one node over fused passes (vfe_fused.py): split-weight second layer, norm + activation
            # applied while pooling, the pooling's gradient routed inside the batch-norm backward
            return fused_vfe2(self, x, grouping.plan), grouping.coors
        point_feats, pooled = self._encode(x, grouping, 'max' if self.mode == 'max' else 'mean')
        if self.return_point_feats:
            return point_feats
        return pooled[-1], grouping.coors


@VOXEL_ENCODERS.register_module()
class DynamicScatterVFE(DynamicVFE):
    """DynamicVFE on scatter_v2's grouping (voxel_encoder.py:502-612): plain sorted-unique of the int64 coordinates,
    no "first row" quirk; the cluster offsets are divided by ``rel_dist_scaler``."""

    def __init__(self, in_channels=4, feat_channels=[], with_distance=False, with_cluster_center=False,
                 with_voxel_center=False, voxel_size=(0.2, 0.2, 4), point_cloud_range=(0, -40, -3, 70.4, 40, 1),
                 norm_cfg=dict(type='BN1d', eps=1e-3, momentum=0.01), mode='max', fusion_layer=None,
                 return_point_feats=False, return_inv=True, rel_dist_scaler=1.0, unique_once=False):
        super(DynamicScatterVFE, self).__init__(in_channels, feat_channels, with_distance, with_cluster_center,
                                                with_voxel_center, voxel_size, point_cloud_range, norm_cfg, mode,
                                                fusion_layer, return_point_feats)
        self.scatter = self.vfe_scatter = self.cluster_scatter = None
        self.rel_dist_scaler = rel_dist_scaler
        self.unique_once = unique_once

    def map_voxel_center_to_point(self, voxel_mean, voxel2point_inds):
        return voxel_mean[voxel2point_inds]

    def grouping_of(self, coors):
        """the point -> voxel grouping this encoder reduces over (index work only: no parameters, no features): callers that
        pipeline batches build it ahead and hand it to forward(..., grouping=...)"""
        return _UniqueGrouping(coors)

    def forward(self, features, coors, points=None, img_feats=None, img_metas=None, return_inv=False, grouping=None):
        features = features.float()
        if grouping is None:
            grouping = _UniqueGrouping(coors)   # one sorted-unique whether or not unique_once is set: it is never redone
        x = self._decorate(features, coors, grouping, cluster_div=self.rel_dist_scaler, mean_of_xyz_only=True)
        fused_plan = UniquePlanAdapter(grouping.plan) if self.fused_stack else None
        if fused_plan is not None and fused_vfe2_ok(self, x, fused_plan):
            voxel_feats = fused_vfe2(self, x, fused_plan)      # the layer stack as one node (vfe_fused.py)
            if return_inv:
                return voxel_feats, grouping.coors, grouping.unq_inv
            return voxel_feats, grouping.coors
        point_feats, pooled = self._encode(x, grouping, self.mode)
        if self.return_point_feats:
            return point_feats
        if return_inv:
            return pooled[-1], grouping.coors, grouping.unq_inv
        return pooled[-1], grouping.coors


@VOXEL_ENCODERS.register_module()
class SIRLayer(_PointGroupEncoder):
    """FSD's sparse-instance-recognition block (voxel_encoder.py:617-764): point-group MLP + pooling over the points of
    an instance (cluster), with the relative-position gate ``rel_mlp``."""

    def __init__(self, in_channels=4, feat_channels=[], with_distance=False, with_cluster_center=False,
                 with_rel_mlp=True, rel_mlp_hidden_dims=[16, ], rel_mlp_in_channel=3, with_voxel_center=False,
                 voxel_size=(0.2, 0.2, 4), point_cloud_range=(0, -40, -3, 70.4, 40, 1),
                 norm_cfg=dict(type='BN1d', eps=1e-3, momentum=0.01), mode='max', fusion_layer=None,
                 return_point_feats=False, return_inv=True, rel_dist_scaler=1.0, with_shortcut=True,
                 xyz_normalizer=[1.0, 1.0, 1.0], act='relu', dropout=0.0):
        super().__init__()
        self._init_common(in_channels, feat_channels, with_distance, with_cluster_center, with_voxel_center, voxel_size,
                          point_cloud_range, mode, return_point_feats, fusion_layer)
        self.scatter = self.vfe_scatter = self.cluster_scatter = None
        self.rel_dist_scaler = rel_dist_scaler
        self.with_shortcut = with_shortcut
        self._with_rel_mlp = with_rel_mlp
        self.xyz_normalizer = xyz_normalizer
        if with_rel_mlp:
            # 3 -> hidden... -> in_channels (the UNdecorated width; the reference appends to its argument in place,
            # voxel_encoder.py:665 - a copy is used here so that a shared config list is not mutated)
            self.rel_mlp = build_mlp(rel_mlp_in_channel, list(rel_mlp_hidden_dims) + [in_channels], norm_cfg, act=act)
        plain = act == 'relu' and not dropout > 0
        self.vfe_layers = nn.ModuleList([
            DynamicVFELayer(cin, cout, norm_cfg) if plain else DynamicVFELayerV2(cin, cout, norm_cfg, act=act, dropout=dropout)
            for cin, cout in self._layer_widths(feat_channels)])
        self.num_vfe = len(self.vfe_layers)

    def map_voxel_center_to_point(self, voxel_mean, voxel2point_inds):
        return voxel_mean[voxel2point_inds]

    def forward(self, features, coors, f_cluster=None, points=None, img_feats=None, img_metas=None, return_inv=False,
                return_both=False, unq_inv_once=None, new_coors_once=None):
        features = features.float()
        grouping = _UniqueGrouping(coors, new_coors_once, unq_inv_once)
        xyz, rest = features[:, :3], features[:, 3:]
        scale = K.const_tensor(self.xyz_normalizer, features.device, features.dtype)
        base = torch.cat([xyz / scale[None, :], rest], dim=1)
        if f_cluster is None:   # offsets from the instance centre
            centre = grouping.reduce(xyz, 'mean')
            f_cluster = xyz - centre[grouping.index.long()][:, :3]
        f_cluster = f_cluster / self.rel_dist_scaler
        if self._with_rel_mlp:
            base = base * self.rel_mlp(f_cluster)
        cols = [base]
        if self._with_cluster_center:
            cols.append(f_cluster / 10.0)
        if self._with_distance:
            cols.append(torch.norm(xyz, 2, 1, keepdim=True))
        x = cols[0] if len(cols) == 1 else torch.cat(cols, dim=-1)

        point_feats, pooled = self._encode(x, grouping, self.mode)
        group_feats = torch.cat(pooled, dim=1)
        if return_both or self.return_point_feats:
            if self.with_shortcut and point_feats.shape == rest.shape:
                point_feats = point_feats + rest
            if return_both:
                return point_feats, group_feats, grouping.coors
            return point_feats, group_feats
        if return_inv:
            return group_feats, grouping.coors, grouping.unq_inv
        return group_feats, grouping.coors
