"""Hot-path mirrors of the detectors that CALL the path (SURVEY.md section 8 "callers either side"): what a config's
``model = dict(type='FSD' | 'SingleStageFSD' | 'FSDV2' | 'SingleStageFSDV2', segmentor=dict(type='VoteSegmentor', ...))``
constructs on the way from points to the features the heads consume, with the reference's constructor arguments,
sub-module names (-> ``state_dict`` keys) and data flow.  Every sub-config whose ``type`` this library implements is BUILT
(Voxelization, DynamicScatterVFE, PseudoMiddleEncoderForSpconvFSD / SSTInputLayerV2, SimpleSparseUNet / SSTv2,
Voxel2PointScatterNeck, VoteSegHead, SIR, ClusterAssigner, the virtual-voxel stage with multiscale_cfg / as_rpn,
DynamicPointROIExtractor); the rest - box heads, losses, target assignment, box decoding, NMS: out of scope - is kept as
its config under ``self.unbuilt`` and never run.  Nothing here computes a loss.

Reference:
  VoteSegmentor            mmdet3d/models/detectors/single_stage_fsd.py:155-384 (__init__, voxelize, extract_feat, reorder,
                           the prediction part of simple_test)
  Voxel2PointScatterNeck   mmdet3d/models/necks/voxel2point_neck.py:9-62
  VoteSegHead              mmdet3d/models/decode_heads/segmentation_head.py:16-105, 327-331 (forward, decode_vote_targets)
  SingleStageFSD / FSD     single_stage_fsd.py:389-483 (__init__, extract_feat), two_stage_fsd.py (roi_head: config only)
  SingleStageFSDV2 / FSDV2 single_stage_fsd_v2.py:38-271, 375-433 (the stage itself: sst_amd/virtual_voxel.py),
                           two_stage_fsd_v2.py:11-60
  DynamicVoxelNet /        mmdet3d/models/detectors/dynamic_voxelnet.py:10-71, 73-110 (__init__, extract_feat, voxelize): the
  DynamicCenterPoint       detectors of configs/sst_refactor/*.py - the caller of the whole SST hot path
"""
import torch
from torch import nn

from . import kernels as K
from .cluster import ClusterAssigner
from .registry import BACKBONES, MODELS, ROI_EXTRACTORS, Registry, build_backbone, build_middle_encoder, build_voxel_encoder
from .sst_ops import build_mlp, scatter_v2
from .virtual_voxel import VirtualVoxelExtractor
from .voxel import Voxelization

DETECTORS = Registry('detector')
NECKS = Registry('neck')
HEADS = Registry('head')


def build_detector(cfg, train_cfg=None, test_cfg=None):
    """mmdet3d.models.builder.build_detector (builder.py:62-76): train_cfg / test_cfg may come from the outer level"""
    args = dict(cfg)
    if train_cfg is not None:
        args.setdefault('train_cfg', train_cfg)
    if test_cfg is not None:
        args.setdefault('test_cfg', test_cfg)
    return DETECTORS.build(args)


build_model = build_detector


def build_neck(cfg):
    return NECKS.build(cfg)


def build_head(cfg):
    return HEADS.build(cfg)


def _get(cfg, key, default=None):
    """mmcv's ConfigDict answers both cfg['a'] and cfg.a; plain dicts (what the tests exec from the config text) only the first"""
    if cfg is None:
        return default
    return cfg.get(key, default) if hasattr(cfg, 'get') else getattr(cfg, key, default)


@NECKS.register_module()
class Voxel2PointScatterNeck(nn.Module):
    """voxel features back onto their points + the point's offset from its voxel centre (necks/voxel2point_neck.py:9-62).

    ``voxel_padding``: rows of dropped voxels (SST-based segmentor with region batching) are filled with it and their points
    masked out; the sparse-convolution segmentor drops nothing, which the caller says with ``all_valid=True``: no mask
    compaction, no host synchronisation."""

    def __init__(self, point_cloud_range=None, voxel_size=None, with_xyz=True, normalize_local_xyz=False):
        super().__init__()
        self.point_cloud_range, self.voxel_size = point_cloud_range, voxel_size
        self.with_xyz, self.normalize_local_xyz = with_xyz, normalize_local_xyz

    def forward(self, points, pts_coors, voxel_feats, voxel2point_inds, voxel_padding=-1, all_valid=False):
        assert points.size(0) == pts_coors.size(0) == voxel2point_inds.size(-1)
        pts_feats = voxel_feats[voxel2point_inds]
        if all_valid:
            pts_mask = torch.ones(points.size(0), dtype=torch.bool, device=points.device)
        else:
            pts_mask = ~((pts_feats == voxel_padding).all(1))
            pts_feats, pts_coors, points = pts_feats[pts_mask], pts_coors[pts_mask], points[pts_mask]
        if not self.with_xyz:
            return pts_feats, pts_mask
        vs = K.const_tensor(self.voxel_size, pts_feats.device, pts_feats.dtype).reshape(1, 3)
        lo = K.const_tensor(self.point_cloud_range[:3], pts_feats.device, pts_feats.dtype).reshape(1, 3)
        center = (pts_coors[:, [3, 2, 1]].to(pts_feats.dtype) + 0.5) * vs + lo
        local_xyz = points[:, :3] - center
        if self.normalize_local_xyz:
            local_xyz = local_xyz / (vs / 2)
        return torch.cat([pts_feats, local_xyz], 1), pts_mask


@HEADS.register_module()
class VoteSegHead(nn.Module):
    """per-point class logits and class-wise centre votes (decode_heads/segmentation_head.py:16-105): the forward pass and
    the vote decoding.  Parameters: ``pre_seg_conv`` (build_mlp), ``conv_seg``, ``voting`` - the reference's names.  Losses
    and target assignment (points in boxes) are the detector's training side: not here."""

    def __init__(self, in_channel, num_classes, hidden_dims=(), dropout_ratio=0.5, conv_cfg=dict(type='Conv1d'),
                 norm_cfg=dict(type='naiveSyncBN1d'), act_cfg=dict(type='ReLU'), loss_decode=None, loss_vote=None,
                 loss_aux=None, ignore_index=255, logit_scale=1, checkpointing=False, init_bias=None, init_cfg=None):
        super().__init__()
        hidden_dims = list(hidden_dims)
        end_channel = hidden_dims[-1] if hidden_dims else in_channel
        self.pre_seg_conv = build_mlp(in_channel, hidden_dims, norm_cfg, act=act_cfg['type']) if hidden_dims else None
        self.use_sigmoid = bool((loss_decode or {}).get('use_sigmoid', False))
        self.bg_label = num_classes
        self.num_classes = num_classes if self.use_sigmoid else num_classes + 1       # softmax heads carry a background class
        self.logit_scale = logit_scale
        self.dropout = nn.Dropout(dropout_ratio) if dropout_ratio > 0 else None
        self.conv_seg = nn.Linear(end_channel, self.num_classes)
        self.voting = nn.Linear(end_channel, self.num_classes * 3)
        self.loss_cfg = dict(loss_decode=loss_decode, loss_vote=loss_vote, loss_aux=loss_aux)   # not built: out of scope
        self.train_cfg = self.test_cfg = None

    def forward(self, voxel_feat):
        x = voxel_feat if self.pre_seg_conv is None else self.pre_seg_conv(voxel_feat)
        logits = self.conv_seg(x if self.dropout is None else self.dropout(x))
        return logits, self.voting(x)

    forward_test = forward

    @staticmethod
    def decode_vote_targets(preds):
        return preds * preds.abs()

    def losses(self, *args, **kwargs):
        raise NotImplementedError('losses / target assignment are outside the hot path (SURVEY.md section 8)')

    forward_train = get_targets = losses


@DETECTORS.register_module()
class VoteSegmentor(nn.Module):
    """points -> dynamic voxelisation -> DynamicScatterVFE -> middle encoder -> sparse backbone -> point features -> class
    logits + votes (single_stage_fsd.py:155-384 without the loss side)."""

    def __init__(self, voxel_layer, voxel_encoder, middle_encoder, backbone, segmentation_head, decode_neck=None,
                 auxiliary_head=None, voxel_downsampling_size=None, train_cfg=None, test_cfg=None, init_cfg=None,
                 pretrained=None, tanh_dims=None, **extra_kwargs):
        super().__init__()
        assert voxel_encoder['type'] == 'DynamicScatterVFE'
        self.voxel_layer = Voxelization(**voxel_layer)
        self.voxel_encoder = build_voxel_encoder(voxel_encoder)
        self.middle_encoder = build_middle_encoder(middle_encoder)
        self.backbone = build_backbone(backbone)
        self.segmentation_head = build_head(segmentation_head)
        self.segmentation_head.train_cfg, self.segmentation_head.test_cfg = train_cfg, test_cfg
        self.decode_neck = build_neck(decode_neck)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.cfg = train_cfg if train_cfg is not None else test_cfg
        self.num_classes = segmentation_head['num_classes']
        self.point_cloud_range, self.voxel_size = voxel_layer['point_cloud_range'], voxel_layer['voxel_size']
        self.voxel_downsampling_size = voxel_downsampling_size
        self.tanh_dims = tanh_dims
        self.use_multiscale_features = backbone.get('return_multiscale_features', False)
        self.print_info = {}
        self.unbuilt = {k: v for k, v in dict(auxiliary_head=auxiliary_head).items() if v is not None}
        self._dropped = False

    @torch.no_grad()
    def voxelize(self, points):
        """list of per-sample clouds -> (all points, (b, z, y, x) per point); one launch per sample, no device sync"""
        return self.voxel_layer.voxelize_batch(points)

    def reorder(self, data, shuffle_inds, keep_inds, padding=-1):
        """dropped voxels padded, voxels back in the voxel encoder's order (single_stage_fsd.py:253-266)"""
        temp = data.new_full((len(shuffle_inds), data.size(1)), padding)
        out = data.new_full((len(shuffle_inds), data.size(1)), padding)
        temp[keep_inds] = data
        out[shuffle_inds] = temp
        return out

    def voxel_downsample(self, points_list):
        vs = K.const_tensor(self.voxel_downsampling_size, points_list[0].device)
        lo = K.const_tensor(self.point_cloud_range[:3], points_list[0].device)
        out = []
        for points in points_list:
            coors = torch.div(points[:, :3] - lo[None], vs[None], rounding_mode='floor').long()
            out.append(scatter_v2(points, coors, mode='avg', return_inv=False)[0])
        return out

    def preprocess(self, points):
        """the intensity / elongation squashing and the optional down-sampling both forward_train and simple_test start with
        (single_stage_fsd.py:286-297)"""
        if self.tanh_dims is not None:
            if len(self.tanh_dims) > 0:
                for p in points:
                    p[:, self.tanh_dims] = torch.tanh(p[:, self.tanh_dims])
        elif points[0].size(1) in (4, 5):
            points = [torch.cat([p[:, :3], torch.tanh(p[:, 3:])], dim=1) for p in points]
        if self.voxel_downsampling_size is not None:
            points = self.voxel_downsample(points)
        return points

    def extract_feat(self, points, img_metas=None):
        batch_points, coors = self.voxelize(points)
        coors = coors.long()
        voxel_features, voxel_coors, voxel2point_inds = self.voxel_encoder(batch_points, coors, return_inv=True)
        voxel_info = self.middle_encoder(voxel_features, voxel_coors)
        if isinstance(voxel_info, dict):
            voxel_info.setdefault('batch_size', len(points))
        x = self.backbone(voxel_info)[0]
        padding = -1
        dropped = isinstance(voxel_info, dict) and 'shuffle_inds' in voxel_info
        feats = x['voxel_feats'] if not dropped else \
            self.reorder(x['voxel_feats'], voxel_info['shuffle_inds'], voxel_info['voxel_keep_inds'], padding)
        self._dropped = dropped
        out = self.decode_neck(batch_points, coors, feats, voxel2point_inds, padding, all_valid=not dropped)
        if self.use_multiscale_features:
            return out, coors, batch_points, x['decoder_features']
        return out, coors, batch_points

    def forward(self, points, img_metas=None):
        """the prediction half of simple_test / forward_train(as_subsegmentor=True): -> dict(seg_points, seg_logits,
        seg_vote_preds, offsets, seg_feats, batch_idx, decoder_features)"""
        points = self.preprocess(points)
        res = self.extract_feat(points, img_metas)
        (feats, valid), pts_coors, batch_points = res[0], res[1], res[2]
        decoder_features = res[3] if self.use_multiscale_features else None
        if self._dropped:           # SST-based segmentor: points of dropped voxels leave; the sparse-convolution one keeps all
            batch_points, pts_coors = batch_points[valid], pts_coors[valid]
        seg_logits, vote_preds = self.segmentation_head(feats)
        return dict(seg_points=batch_points, seg_logits=seg_logits, seg_vote_preds=vote_preds,
                    offsets=self.segmentation_head.decode_vote_targets(vote_preds), seg_feats=feats,
                    batch_idx=pts_coors[:, 0], decoder_features=decoder_features)

    simple_test = forward


@DETECTORS.register_module()
class DynamicVoxelNet(nn.Module):
    """points -> dynamic voxelisation -> DynamicVFE -> SSTInputLayerV2 -> SSTv2 (dynamic_voxelnet.py:10-71): what
    ``model = dict(type='DynamicVoxelNet', voxel_layer=..., voxel_encoder=..., middle_encoder=..., backbone=..., neck=...,
    bbox_head=...)`` of configs/sst_refactor/*.py constructs up to the backbone's output, with the reference's constructor
    arguments and sub-module names (-> ``state_dict`` keys ``voxel_encoder.*``, ``backbone.*``).  The dense BEV neck (SECOND
    FPN) and the box head are SURVEY.md section 2 OUT-OF-SCOPE: their configs are kept under ``self.unbuilt``, never run.

    ``extract_feat`` has the reference's semantics (:38-47: voxelize -> voxel_encoder -> middle_encoder(.., batch_size) ->
    backbone).  When the three index stages are the ones the fused frame plan covers (csrc/frame_plan.hip: DynamicVFE +
    SSTInputLayerV2 on the same grid, voxels allowed to leave in window-major order - automatic with ``shuffle_voxels=True``)
    the same results come from ONE device-side plan per batch: no host round trip until the voxel encoder's kernels are
    queued (the reference reads ``coors[-1, 0].item()`` and three sizes back; the piecewise path of this package one size per
    module boundary).  ``fused_index = False`` forces the piecewise path; ``prepare()`` builds the plan of a batch ahead of
    time (it depends on the point clouds only - what a data-loader prefetch would do) for ``extract_feat(.., prepared=)``."""

    def __init__(self, voxel_layer, voxel_encoder, middle_encoder, backbone, neck=None, bbox_head=None, train_cfg=None,
                 test_cfg=None, pretrained=None, init_cfg=None):
        super().__init__()
        self.voxel_layer = Voxelization(**voxel_layer)
        self.voxel_encoder = build_voxel_encoder(voxel_encoder)
        self.middle_encoder = build_middle_encoder(middle_encoder)
        self.backbone = build_backbone(backbone)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.unbuilt = {k: v for k, v in dict(neck=neck, bbox_head=bbox_head).items() if v is not None}
        self.fused_index = True
        self._planner = None

    with_neck = False          # the dense neck is not built (out of scope): extract_feat ends at the backbone's output

    @torch.no_grad()
    def voxelize(self, points):
        """list of per-sample clouds -> (all points, (b, z, y, x) int32 per point) (dynamic_voxelnet.py:49-71)"""
        return self.voxel_layer.voxelize_batch(points)

    def _frame_planner(self, batch_size):
        # (written against `self` = ANY detector object with voxel_layer / voxel_encoder / middle_encoder sub-modules: the
        # reference's own DynamicVoxelNet instance after install_fused_extract_feat(), whose class never set these attributes)
        from .frame_plan import FramePlanner
        from .sst_input_layer import SSTInputLayerV2
        from .voxel_encoder import DynamicVFE
        if not getattr(self, 'fused_index', True):
            return None
        planner = self.__dict__.get('_planner')
        if planner is None:
            ok = (type(self.voxel_encoder) is DynamicVFE and type(self.middle_encoder) is SSTInputLayerV2
                  and isinstance(self.voxel_layer, Voxelization) and hasattr(self.voxel_encoder, '_grid_zyx'))
            planner = FramePlanner(self.voxel_layer, self.voxel_encoder, self.middle_encoder) if ok else False
            self.__dict__['_planner'] = planner
        return planner if (planner and planner.supported(batch_size)) else None

    def prepare(self, points, overlap=False, ready_event=None):
        """the index work of one batch (voxelize, point -> voxel grouping, window bucketing / drop / window CSR, positional
        rows): a FramePlan, or None when the fused plan does not cover this configuration / batch (extract_feat then runs the
        piecewise path itself).

        ``overlap=True`` (a data-loader hook's mode): the plan is built on a stream of its own, concurrently with the step the
        caller's stream is still working on; the point clouds must be complete, or complete once ``ready_event`` has happened
        (FramePlanner.build_overlapped).  Host tensors (the data loader's pinned batch) are copied to the device on that
        stream first - then nothing has to be promised."""
        if len(points) == 0:
            return None
        planner = DynamicVoxelNet._frame_planner(self, len(points))
        if planner is None:
            return None
        if overlap and not any(p.is_cuda for p in points):
            return planner.build_overlapped_from_host(points, next(self.parameters()).device)
        if not all(p.is_cuda for p in points):
            return None
        return planner.build_overlapped(points, ready_event) if overlap else planner.build(points)

    def voxel_info(self, points, prepared=None):
        """everything in front of the backbone: -> the ``voxel_info`` dictionary SSTInputLayerV2 returns"""
        plan = prepared if prepared is not None else DynamicVoxelNet.prepare(self, points)
        if plan is not None:     # the voxel encoder is queued before the plan's sizes are read (FramePlan.finalize)
            voxel_features, _ = self.voxel_encoder(plan.points, plan.coors, scatter_plan=plan)
            return plan.finalize(voxel_features, self.middle_encoder)
        voxels, coors = self.voxelize(points)          # the detector's own voxelize(): the reference's per-sample loop works too
        voxel_features, feature_coors = self.voxel_encoder(voxels, coors)
        return self.middle_encoder(voxel_features, feature_coors, len(points))

    def extract_feat(self, points, img_metas=None, prepared=None):
        x = self.backbone(self.voxel_info(points, prepared))
        return x

    def extract_voxel_feats(self, points, img_metas=None, prepared=None):
        """the path up to the shift blocks' output, whatever the backbone's ``to_bev`` says: (features [M', C] of the kept
        voxels, their (b, z, y, x)) - the part of ``extract_feat`` SURVEY.md section 8(d) defines the frames/s metric on"""
        info = self.voxel_info(points, prepared)
        return self.backbone.forward_voxels(info), info['voxel_coors']

    forward = extract_feat

    def losses(self, *args, **kwargs):
        raise NotImplementedError('the box head, its losses and target assignment are outside the hot path (SURVEY.md section 8)')

    forward_train = simple_test = aug_test = losses


@DETECTORS.register_module()
class DynamicCenterPoint(DynamicVoxelNet):
    """dynamic_voxelnet.py:73-110: same feature path, CenterHead on top (not built)"""


def install_fused_extract_feat(detector_cls):
    """Reference-side hook: give the REFERENCE's own detector class - mmdet3d's DynamicVoxelNet /
    DynamicCenterPoint, which keeps its neck, heads, losses and test-time code - the ``extract_feat`` of this module: same
    semantics as detectors/dynamic_voxelnet.py:38-47 (voxelize -> voxel_encoder -> middle_encoder(.., batch_size) -> backbone
    -> neck), with the three index stages as ONE device-side plan when the sub-modules are this library's.

        from mmdet3d.models.detectors import DynamicVoxelNet, DynamicCenterPoint
        sst_amd.detectors.install_fused_extract_feat(DynamicVoxelNet)        # DynamicCenterPoint inherits it
    """
    def extract_feat(self, points, img_metas=None):
        x = self.backbone(DynamicVoxelNet.voxel_info(self, points))
        if getattr(self, 'with_neck', False):
            x = self.neck(x)
        return x

    extract_feat.__doc__ = 'Extract features from points (sst_amd: fused index plan in front of the backbone).'
    detector_cls.extract_feat = extract_feat
    return detector_cls


class _HotPathDetector(nn.Module):
    """shared: the segmentor, the record of what was not built"""

    def _init_common(self, segmentor, bbox_head, train_cfg, test_cfg, **others):
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.cfg = train_cfg if train_cfg else test_cfg
        self.segmentor = build_detector(segmentor)
        self.head_type = bbox_head['type']
        self.num_classes = bbox_head['num_classes']
        self.as_rpn = bool(bbox_head.get('as_rpn', False))
        self.unbuilt = {'bbox_head': bbox_head}
        self.unbuilt.update({k: v for k, v in others.items() if v is not None})
        self.print_info, self.runtime_info = {}, {}

    def combine_classes(self, data_dict, name_list):
        return {name: torch.cat(data_dict[name], 0) for name in data_dict if name in name_list}


@DETECTORS.register_module()
class SingleStageFSD(_HotPathDetector):
    """segmentor -> (sampling, clustering: ``cluster_assigner``) -> SIR backbone over the clusters
    (single_stage_fsd.py:389-483)."""

    def __init__(self, backbone, segmentor, voxel_layer=None, voxel_encoder=None, middle_encoder=None, neck=None,
                 bbox_head=None, train_cfg=None, test_cfg=None, cluster_assigner=None, pretrained=None, init_cfg=None):
        super().__init__()
        self._init_common(segmentor, bbox_head, train_cfg, test_cfg, neck=neck)
        self.backbone = build_backbone(backbone)
        if voxel_layer is not None:
            self.voxel_layer = Voxelization(**voxel_layer)
        if voxel_encoder is not None:
            self.voxel_encoder = build_voxel_encoder(voxel_encoder)
        if middle_encoder is not None:
            self.middle_encoder = build_middle_encoder(middle_encoder)
        if 'radius' in cluster_assigner or 'hybrid' in cluster_assigner:
            raise NotImplementedError('SSGAssigner / HybridAssigner (furthest point sampling) are not on the path')
        self.cluster_assigner = ClusterAssigner(**cluster_assigner)
        self.cluster_assigner.num_classes = self.num_classes

    def extract_feat(self, points, pts_feats, pts_cluster_inds, img_metas, center_preds):
        cluster_xyz, _, inv_inds = scatter_v2(center_preds, pts_cluster_inds, mode='avg', return_inv=True)
        f_cluster = points[:, :3] - cluster_xyz[inv_inds]
        out_pts_feats, cluster_feats, out_coors = self.backbone(points, pts_feats, pts_cluster_inds, f_cluster)
        out = dict(cluster_feats=cluster_feats, cluster_xyz=cluster_xyz, cluster_inds=out_coors)
        if self.as_rpn:
            out['cluster_pts_feats'], out['cluster_pts_xyz'] = out_pts_feats, points
        return out


@DETECTORS.register_module()
class FSD(SingleStageFSD):
    """two_stage_fsd.py: + ``roi_head`` (GroupCorrectionHead); its DynamicPointROIExtractor is built, the box head not"""

    def __init__(self, backbone, segmentor, roi_head=None, **kw):
        super().__init__(backbone, segmentor, **kw)
        self.unbuilt['roi_head'] = roi_head
        ext = dict(_get(roi_head, 'roi_extractor', None) or {})
        self.with_virtual = ext.pop('with_virtual', False)
        self.roi_extractor = ROI_EXTRACTORS.build(ext) if ext else None


@DETECTORS.register_module()
class SingleStageFSDV2(_HotPathDetector, VirtualVoxelExtractor):
    """segmentor (with multi-scale decoder features) -> (sampling) -> the virtual-voxel stage
    (single_stage_fsd_v2.py:38-271, 375-433; the stage: sst_amd/virtual_voxel.py - ``extract_feat`` is its forward)."""

    def __init__(self, backbone, segmentor, voxel_layer=None, voxel_encoder=None, middle_encoder=None, neck=None,
                 virtual_point_projector=None, pre_voxel_encoder=None, bbox_head=None, train_cfg=None, test_cfg=None,
                 pretrained=None, init_cfg=None, multiscale_cfg=None):
        VirtualVoxelExtractor.__init__(self, backbone, voxel_encoder, virtual_point_projector, train_cfg=train_cfg,
                                       test_cfg=test_cfg, multiscale_cfg=multiscale_cfg, bbox_head=bbox_head)
        self._init_common(segmentor, bbox_head, train_cfg or {}, test_cfg or {}, neck=neck, pre_voxel_encoder=pre_voxel_encoder)
        if voxel_layer is not None:
            self.voxel_layer = Voxelization(**voxel_layer)
        if middle_encoder is not None:
            self.middle_encoder = build_middle_encoder(middle_encoder)
        self.use_multiscale_features = self.segmentor.use_multiscale_features

    forward = VirtualVoxelExtractor.extract_feat


@DETECTORS.register_module()
class FSDV2(SingleStageFSDV2):
    """two_stage_fsd_v2.py:11-60: + ``roi_head``; the point RoI extractor is built (the SIR layers of its box head are
    sst_amd.SIRLayer, constructed by whoever builds that head)"""

    def __init__(self, backbone, segmentor, roi_head=None, **kw):
        super().__init__(backbone, segmentor, **kw)
        self.unbuilt['roi_head'] = roi_head
        ext = dict(_get(roi_head, 'roi_extractor', None) or {})
        self.with_virtual = ext.pop('with_virtual', False)
        self.roi_extractor = ROI_EXTRACTORS.build(ext) if ext else None
