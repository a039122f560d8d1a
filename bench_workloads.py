"""bench.py workloads beyond the SST headline: BASELINE.json configs[3] (FSD) and configs[4] (FSDv2) as hot-path
pipelines on synthetic clouds.  The modules are this repository's, wired as the reference's detectors wire them; the
detector glue that is out of scope (heads, losses, box decoding, target assignment) is replaced by small linear layers,
and the foreground selection by a geometric rule - said in each class.  `python bench.py --workload fsd|fsdv2`.

  FSDPath    VoteSegmentor.extract_feat (single_stage_fsd.py:228-250) -> Voxel2PointScatterNeck
             (necks/voxel2point_neck.py:28-63) -> ClusterAssigner (single_stage_fsd.py:922-999) -> SingleStageFSD.extract_feat
             (:467-483, SIR x 3) -> DynamicPointROIExtractor -> SIR x 2 (fsd_bbox_head.py:69-97); configs/fsd/fsd_waymoD1_1x.py
  FSDv2Path  the same segmentor at nuScenes geometry (configs/fsdv2/fsdv2_nusc_1x.py:7-10, 36-107: 0.2 m voxels,
             [40, 512, 512] grid, ~300 k points of 10 sweeps) -> SingleStageFSDV2.extract_feat (single_stage_fsd_v2.py:159-271:
             virtual points, 0.4 m virtual voxels, VirtualVoxelMixer), sst_amd/virtual_voxel.py
"""
import json
import os
import time

import numpy as np
import torch
import torch.distributed as dist
from torch import nn

import sst_amd
from sst_amd.sst_ops import scatter_v2

HBM_PEAK_GBS = 8000.0
FP32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / 16x16x4_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 (the headline figures with 2:1 sparsity are not used)
BN = dict(type='naiveSyncBN1d', eps=1e-3, momentum=0.01)


def lidar_like_cloud(n, seed, dev, half_extent=75.0, z_ground=-1.8, extra_channels=2):
    """ground plane + boxes: points on surfaces, so that voxels have LiDAR-like neighbourhoods"""
    g = torch.Generator().manual_seed(seed)
    n_obj = n // 4
    ext = 2 * half_extent
    ground = torch.rand(n - n_obj, 3, generator=g) * torch.tensor([ext, ext, 0.15]) + torch.tensor([-half_extent, -half_extent, z_ground])
    centres = torch.rand(60, 3, generator=g) * torch.tensor([ext * 0.8, ext * 0.8, 0.0]) + torch.tensor([-half_extent * 0.8, -half_extent * 0.8, z_ground + 0.9])
    obj = centres[torch.randint(0, 60, (n_obj,), generator=g)] + (torch.rand(n_obj, 3, generator=g) - 0.5) * torch.tensor([4.0, 2.0, 1.6])
    xyz = torch.cat([ground, obj])
    return torch.cat([xyz, torch.rand(n, extra_channels, generator=g)], 1).to(dev), centres


def chain_cloud(n, seed, half_extent=18.0, n_objects=14, z_ground=-1.8):
    """fixture-size cloud for the chain goldens: a ground plane and a few DENSE objects (so that clusters survive the
    min_points filter of the cluster voxels, down to 5 cm for pedestrians), two extra channels"""
    g = torch.Generator().manual_seed(seed)
    n_obj = n // 2
    ext = 2 * half_extent
    ground = torch.rand(n - n_obj, 3, generator=g) * torch.tensor([ext, ext, 0.15]) + torch.tensor([-half_extent, -half_extent, z_ground])
    centres = torch.rand(n_objects, 3, generator=g) * torch.tensor([ext * 0.8, ext * 0.8, 0.0]) \
        + torch.tensor([-half_extent * 0.8, -half_extent * 0.8, z_ground + 0.9])
    size = torch.rand(n_objects, 3, generator=g) * torch.tensor([1.2, 0.6, 0.6]) + torch.tensor([0.3, 0.3, 0.8])
    which = torch.randint(0, n_objects, (n_obj,), generator=g)
    obj = centres[which] + (torch.rand(n_obj, 3, generator=g) - 0.5) * size[which]
    return torch.cat([torch.cat([ground, obj]), torch.rand(n, 2, generator=g)], 1)


class GpuOps(object):
    """the module provider of the product: sst_amd (the reference's registry names, csrc kernels underneath)"""
    name = 'sst_amd'
    DynamicScatterVFE = sst_amd.DynamicScatterVFE
    PseudoMiddleEncoderForSpconvFSD = sst_amd.PseudoMiddleEncoderForSpconvFSD
    SimpleSparseUNet = sst_amd.SimpleSparseUNet
    ClusterAssigner = sst_amd.ClusterAssigner
    SIR = sst_amd.SIR
    DynamicPointROIExtractor = sst_amd.DynamicPointROIExtractor
    VirtualVoxelExtractor = sst_amd.VirtualVoxelExtractor
    scatter_v2 = staticmethod(scatter_v2)

    @staticmethod
    def voxelize(points_list, voxel_size, point_cloud_range):
        layer = sst_amd.Voxelization(voxel_size, point_cloud_range, -1, (-1, -1))
        points, coors = layer.voxelize_batch(points_list)
        return points, coors.long()


SIR_NORM = dict(type='LN', eps=1e-3)
# configs/fsd/fsd_waymoD1_1x.py at hot-path level
FSD_CFG = dict(
    seg_voxel=(0.25, 0.25, 0.2), pc_range=[-80, -80, -2, 80, 80, 4], classes=['Car', 'Pedestrian', 'Cyclist'],
    vfe=dict(in_channels=5, feat_channels=[64, 64]),
    unet=dict(in_channels=64, sparse_shape=[32, 640, 640], base_channels=64, output_channels=128,
              encoder_channels=((64, ), (64, 64, 64), (64, 64, 64), (128, 128, 128), (256, 256, 256)),
              encoder_paddings=((1, ), (1, 1, 1), (1, 1, 1), ((0, 1, 1), 1, 1), (1, 1, 1)),
              decoder_channels=((256, 256, 128), (128, 128, 64), (64, 64, 64), (64, 64, 64), (64, 64, 64)),
              decoder_paddings=((1, 1), (1, 0), (1, 0), (0, 0), (0, 1))),
    cluster=dict(cluster_voxel_size=dict(Car=(0.3, 0.3, 6), Cyclist=(0.2, 0.2, 6), Pedestrian=(0.05, 0.05, 6)), min_points=2,
                 connected_dist=dict(Car=0.6, Cyclist=0.4, Pedestrian=0.1)),
    sir=dict(num_blocks=3, feat=128, rel_hidden=[16, 32]), roi_sir=dict(num_blocks=2, feat=128, rel_hidden=[16, 32]),
    roi=dict(extra_wlh=[0.5, 0.5, 0.5], max_inbox_point=256))
# the same chain at fixture size (tests/golden/fsd_chain.npz: narrow U-Net on a [16, 160, 160] grid, 32-wide SIR)
FSD_SMALL_CFG = dict(
    seg_voxel=(0.25, 0.25, 0.2), pc_range=[-20, -20, -2, 20, 20, 1.2], classes=['Car', 'Pedestrian', 'Cyclist'],
    vfe=dict(in_channels=5, feat_channels=[16, 16]),
    unet=dict(in_channels=16, sparse_shape=[16, 160, 160], base_channels=16, output_channels=16,
              encoder_channels=((16, ), (16, 16), (32, 32)), encoder_paddings=((1, ), (1, 1), (1, 1)),
              decoder_channels=((32, 32, 16), (16, 16, 16), (16, 16, 16)), decoder_paddings=((1, 1), (1, 0), (0, 1))),
    cluster=dict(cluster_voxel_size=dict(Car=(0.3, 0.3, 6), Cyclist=(0.2, 0.2, 6), Pedestrian=(0.05, 0.05, 6)), min_points=2,
                 connected_dist=dict(Car=0.6, Cyclist=0.4, Pedestrian=0.1)),
    sir=dict(num_blocks=3, feat=32, rel_hidden=[8, 16]), roi_sir=dict(num_blocks=2, feat=32, rel_hidden=[8, 16]),
    roi=dict(extra_wlh=[0.5, 0.5, 0.5], max_inbox_point=256))


_CONSTS = {}


def _const(values, dev):
    """small fp32 constant on `dev`, uploaded once (torch.tensor(list, device=cuda) synchronises the host every call)"""
    key = (tuple(float(v) for v in values), str(dev))
    if key not in _CONSTS:
        _CONSTS[key] = torch.tensor([float(v) for v in values], dtype=torch.float32).to(dev)
    return _CONSTS[key]


def mixed_sum(t, scale=1.0):
    """the scalar the chains backpropagate from: sum(t * w) with a FIXED weight w[n, c] = a[n] * b[c] built from integer
    arithmetic (identical on every device, no RNG): a changes sign from row to row (period 13, zero mean), b in [0.25, 1.25].
    A plain t.sum() - or any weight with a column-constant part - is the wrong probe for these networks: the outputs sit
    behind batch / layer normalisation, whose backward pass REMOVES the column mean of the upstream gradient
    (d sum(LN(x)) / dx is exactly 0, d sum(BN(x)) / dx cancels over the batch), so the parameter gradients of such a probe
    are small residues of large cancelling terms and two correct fp32 evaluations of them differ by percents
    (tests/adjudicate_fsd_grads.py: the CPU port in fp32 against itself in float64).  bench.py's SST step starts from a fixed
    random upstream gradient for the same reason."""
    n, c = t.shape
    a = ((torch.arange(n, device=t.device) % 13) - 6).to(t.dtype) / 6
    b = (torch.arange(c, device=t.device) * 7 % 11).to(t.dtype) / 10 + 0.25
    return (t * (a[:, None] * b[None, :])).sum() * scale


def fsd_foreground_stand_in(batch_points, votes, z_cut=-1.4):
    """STAND-IN for the segmentation decision and the vote decoding of VoteSegHead (out of scope): foreground = points
    above the ground plane, class by a position hash, voted centre = point + 0.05 tanh(vote of its class).  Shared by the
    GPU path, the CPU port and the reference chain that produces the golden (tests/golden/make_golden.py)."""
    fg = batch_points[:, 2] > z_cut
    cls = (batch_points[:, 0].abs() * 7).long() % 3
    masks = [fg & (cls == c) for c in range(3)]
    counts = torch.stack([m.sum() for m in masks]).tolist()          # ONE read-back for the three classes
    sel_l, pts_l = [], []
    for c in range(3):
        sel = torch.nonzero_static(masks[c], size=int(counts[c])).squeeze(1)
        sel_l.append(sel)
        pts_l.append((batch_points[sel, :3] + 0.05 * torch.tanh(votes[sel, c])).detach())
    return sel_l, pts_l


def prepare_index_phase(path, points_list):
    """The part of a chain's forward pass that depends on the point clouds alone - voxelisation, the sorted-unique grouping of the
    voxel encoder, every rulebook of the segmentor's sparse U-Net (each with a size read-back) - built AHEAD: a training loop
    calls this for its next batch between the forward and the backward pass of the current one (what a data loader's worker
    does with collation), so that those read-backs wait for the short forward kernels instead of sitting, at the head of the
    next step, behind the whole backward pass.  -> what forward(..., prepared=...) takes; None for providers without the
    split (the CPU port, the reference)."""
    if path.ops is not GpuOps:
        return None
    with torch.no_grad():
        batch_points, coors = path.ops.voxelize(points_list, path.SEG_VOXEL, path.PC_RANGE)
        grouping = path.voxel_encoder.grouping_of(coors)
        uc = grouping.coors
        keep = getattr(path.seg_backbone, 'keep_coors_dims', None)
        rules = path.seg_backbone.build_rulebooks(uc if keep is None else uc[:, keep], len(points_list))
    return dict(batch_points=batch_points, coors=coors, grouping=grouping, indice_dict=rules)


class FSDPath(nn.Module):
    """configs/fsd/fsd_waymoD1_1x.py at hot-path level, over a module provider `ops` (GpuOps = sst_amd; oracle.fsd_cpu = the
    CPU port; oracle.ref_fsd = the reference's own Python in the build container): the SAME wiring for all three.
    Stand-ins: `seg_head` (VoteSegHead), `box_head` (the FSD head's regression branch); foreground / class / vote decoding =
    fsd_foreground_stand_in.  `roi_stage=False` stops after SingleStageFSD.extract_feat (the chain the goldens pin: the
    point-pool features behind it are TorchEx's and unpinned)."""

    def __init__(self, ops=GpuOps, cfg=None, roi_stage=True):
        super().__init__()
        cfg = self.cfg = FSD_CFG if cfg is None else cfg
        self.ops, self.roi_stage = ops, roi_stage
        self.SEG_VOXEL, self.PC_RANGE, self.CLASSES = cfg['seg_voxel'], cfg['pc_range'], cfg['classes']
        c_vox = cfg['vfe']['feat_channels'][-1]
        c_seg = cfg['unet']['decoder_channels'][-1][-1]
        self.voxel_encoder = ops.DynamicScatterVFE(voxel_size=self.SEG_VOXEL, with_cluster_center=True, with_voxel_center=True,
                                                   point_cloud_range=self.PC_RANGE, norm_cfg=BN, unique_once=True, **cfg['vfe'])
        assert cfg['unet']['in_channels'] == c_vox
        self.middle_encoder = ops.PseudoMiddleEncoderForSpconvFSD()
        self.seg_backbone = ops.SimpleSparseUNet(order=('conv', 'norm', 'act'), norm_cfg=BN, **cfg['unet'])
        self.seg_head = nn.Linear(c_seg + 3, 3 + 9)       # stand-in: 3 class logits + 3 x 3 centre votes
        self.cluster_assigner = ops.ClusterAssigner(point_cloud_range=self.PC_RANGE, class_names=self.CLASSES, **cfg['cluster'])
        self.cluster_assigner.num_classes = len(self.CLASSES)      # the detector sets it (single_stage_fsd.py:418)
        f, nb = cfg['sir']['feat'], cfg['sir']['num_blocks']
        c_pts = c_seg + 3 + 3 + 9                          # point features handed to SIR: seg feats + logits + votes
        self.backbone = ops.SIR(num_blocks=nb, in_channels=[5 + c_pts] + [5 + f] * (nb - 1), feat_channels=[[f, f]] * nb,
                                rel_mlp_hidden_dims=[list(cfg['sir']['rel_hidden']) for _ in range(nb)], norm_cfg=SIR_NORM,
                                mode='max', xyz_normalizer=[20, 20, 4], act='gelu', unique_once=True)
        self.box_head = nn.Linear(f * nb * 2, 7)           # stand-in: centre offset, log sizes, yaw
        if roi_stage:
            f2, nb2 = cfg['roi_sir']['feat'], cfg['roi_sir']['num_blocks']
            self.roi_extractor = ops.DynamicPointROIExtractor(debug=False, **cfg['roi'])
            self.roi_backbone = ops.SIR(num_blocks=nb2, in_channels=[13 + f + 13] + [13 + f2] * (nb2 - 1),
                                        feat_channels=[[f2, f2]] * nb2,
                                        rel_mlp_hidden_dims=[list(cfg['roi_sir']['rel_hidden']) for _ in range(nb2)],
                                        norm_cfg=SIR_NORM, mode='max', xyz_normalizer=[20, 20, 4], act='gelu', unique_once=True)

    def make_cloud(self, n, seed, dev):
        return lidar_like_cloud(n, seed, dev)[0]

    def prepare(self, points_list):
        return prepare_index_phase(self, points_list)

    def forward(self, points_list, return_tensors=False, prepared=None):
        ops = self.ops
        dev = points_list[0].device
        if prepared is not None:     # the index phase of this batch was built ahead (prepare_index_phase)
            batch_points, coors = prepared['batch_points'], prepared['coors']
            voxel_feats, voxel_coors, v2p = self.voxel_encoder(batch_points, coors, return_inv=True, grouping=prepared['grouping'])
        else:
            batch_points, coors = ops.voxelize(points_list, self.SEG_VOXEL, self.PC_RANGE)
            voxel_feats, voxel_coors, v2p = self.voxel_encoder(batch_points, coors, return_inv=True)
        info = self.middle_encoder(voxel_feats, voxel_coors)
        if prepared is not None:
            info['indice_dict'] = prepared['indice_dict']
        info.setdefault('batch_size', len(points_list))       # known here: spares the U-Net its read-back of coors[:, 0].max()
        x = self.seg_backbone(info)[0]
        # Voxel2PointScatterNeck (necks/voxel2point_neck.py:28-63)
        pts_feats = x['voxel_feats'][v2p]
        vs = _const(self.SEG_VOXEL, dev).reshape(1, 3)
        centre = (coors[:, [3, 2, 1]].float() + 0.5) * vs + _const(self.PC_RANGE[:3], dev).reshape(1, 3)
        seg_feats = torch.cat([pts_feats, batch_points[:, :3] - centre], 1)            # [N, C + 3]
        head = self.seg_head(seg_feats)
        logits, votes = head[:, :3], head[:, 3:].reshape(-1, 3, 3)
        batch_idx = coors[:, 0]
        sel_l, pts_l = fsd_foreground_stand_in(batch_points, votes)
        # called as SingleStageFSD.forward_train does (single_stage_fsd.py:521): per-class lists, origin_points = the points
        cluster_inds_l, valid_l = self.cluster_assigner(pts_l, [batch_idx[s] for s in sel_l], None, None,
                                                        origin_points=[batch_points[s] for s in sel_l])
        # the GPU assigner hands the indices of the surviving points over with the mask (no second search / read-back)
        keep_l = [getattr(v, '_sst_keep', None) for v in valid_l]
        keep_l = [torch.nonzero(v).squeeze(1) if k is None else k for v, k in zip(valid_l, keep_l)]
        sel = torch.cat([s[k] for s, k in zip(sel_l, keep_l)])
        cluster_inds = torch.cat(cluster_inds_l).long()                               # [P, 3] (class, sample, cluster)
        centres = torch.cat([p[k] for p, k in zip(pts_l, keep_l)])
        points = batch_points[sel]
        feats = torch.cat([seg_feats[sel], logits[sel], votes[sel].reshape(-1, 9)], 1)
        # SingleStageFSD.extract_feat (single_stage_fsd.py:467-483)
        cluster_xyz, _, inv = ops.scatter_v2(centres, cluster_inds, mode='avg', return_inv=True)
        f_cluster = points[:, :3] - cluster_xyz[inv]
        pts_out, cluster_feats, cluster_coors = self.backbone(points, feats, cluster_inds, f_cluster)
        stats = dict(points=batch_points.size(0), voxels=voxel_feats.size(0), fg_points=points.size(0),
                     clusters=cluster_feats.size(0))
        loss = mixed_sum(cluster_feats, 1e-3) + mixed_sum(logits, 1e-3)
        tensors = dict(voxel_coors=voxel_coors, voxel_feats=voxel_feats, unet_feats=x['voxel_feats'], seg_feats=seg_feats,
                       head=head, sel=sel, cluster_inds=cluster_inds, pts_out=pts_out, cluster_feats=cluster_feats,
                       cluster_coors=cluster_coors, cluster_xyz=cluster_xyz)
        if not self.roi_stage:
            loss = loss + cluster_feats.square().mean() + pts_out.square().mean()
            return (loss, stats, tensors) if return_tensors else (loss, stats)
        box = self.box_head(cluster_feats)
        rois = torch.cat([cluster_coors[:, 1:2].float(),
                          cluster_xyz + 0.1 * torch.tanh(box[:, :3]) - _const([0, 0, 0.9], dev),
                          _const([2.0, 4.4, 1.8], dev) * torch.exp(0.1 * torch.tanh(box[:, 3:6])),
                          box[:, 6:7]], 1).detach()
        order = torch.argsort(rois[:, 0], stable=True)                                 # RoIs sample after sample
        rois = rois[order]
        p_order = torch.argsort(cluster_inds[:, 1], stable=True)                       # points sample after sample
        ext_inds, roi_inds, info = self.roi_extractor(points[p_order, :3].contiguous(), cluster_inds[p_order, 1], rois)
        keep = ext_inds >= 0                                                           # drops the "fake" row of an empty pool
        pooled_xyz = points[p_order][ext_inds.clamp(min=0), :3]
        geo = torch.cat([info['local_xyz'], info['boundary_offset'], info['is_in_margin'][:, None], pooled_xyz], 1)[keep]
        ext_inds, roi_inds = ext_inds[keep], roi_inds[keep]
        roi_feats = torch.cat([pts_out[p_order][ext_inds], geo], 1)                     # [Q, F + 13]
        roi_coors = torch.stack([torch.zeros_like(roi_inds), rois[roi_inds, 0].long(), roi_inds], 1)
        _, roi_cluster_feats, _ = self.roi_backbone(geo, roi_feats, roi_coors, geo[:, :3].contiguous())
        stats['pooled_pairs'] = roi_feats.size(0)
        tensors['roi_cluster_feats'] = roi_cluster_feats
        loss = loss + mixed_sum(roi_cluster_feats, 1e-2)
        return (loss, stats, tensors) if return_tensors else (loss, stats)


# configs/fsdv2/fsdv2_nusc_1x.py at hot-path level
FSDV2_CFG = dict(
    seg_voxel=(0.2, 0.2, 0.2), virtual_voxel=(0.4, 0.4, 0.4), pc_range=[-51.2, -51.2, -5, 51.2, 51.2, 3], n_logits=11,
    vfe=dict(in_channels=5, feat_channels=[64, 64]),
    unet=dict(in_channels=64, sparse_shape=[40, 512, 512], base_channels=64, output_channels=128,
              encoder_channels=((128, ), (128, 128), (128, 128), (128, 128, 128), (256, 256, 256), (256, 256, 256)),
              encoder_paddings=((1, ), (1, 1), (1, 1), (1, 1, 1), (1, 1, 1), (1, 1, 1)),
              decoder_channels=((256, 256, 256), (256, 256, 128), (128, 128, 128), (128, 128, 128), (128, 128, 128),
                                (128, 128, 128)),
              decoder_paddings=((1, 1), (1, 0), (1, 0), (0, 0), (0, 1), (1, 1))),
    mixer=dict(in_channels=128, sparse_shape=[20, 256, 256], base_channels=64, output_channels=128,
               encoder_channels=((64, ), (64, 64), (64, 64)), encoder_paddings=((1, ), (1, 1), (1, 1)),
               decoder_channels=((64, 64, 64), (64, 64, 64), (64, 64, 64)), decoder_paddings=((1, 1), (1, 1), (1, 1))),
    virtual_vfe=dict(feat_channels=[64, 128]), proj_hidden=[64, 64],
    # configs/fsdv2/fsdv2_nusc_1x.py:122-128 (norm_cfg is the projector's: naiveSyncBN1d with torch's defaults)
    multiscale=dict(multiscale_levels=[0, 1, 2], projector_hiddens=[[256, 128], [128, 128], [128, 128]], fusion_mode='avg',
                    target_sparse_shape=[20, 256, 256], norm_cfg=dict(type='naiveSyncBN1d')),
    as_rpn=False)
# the same chain at fixture size: a 4-level segmentor U-Net on [16, 128, 128] whose two coarsest decoder levels (strides 2 and 1
# against the [8, 64, 64] virtual-voxel grid) are fused in; as_rpn on, as in configs/fsdv2/fsdv2_waymo_1x.py:103-104, 175
FSDV2_SMALL_CFG = dict(
    seg_voxel=(0.2, 0.2, 0.2), virtual_voxel=(0.4, 0.4, 0.4), pc_range=[-12.8, -12.8, -2, 12.8, 12.8, 1.2], n_logits=5,
    vfe=dict(in_channels=5, feat_channels=[16, 16]),
    unet=dict(in_channels=16, sparse_shape=[16, 128, 128], base_channels=16, output_channels=16,
              encoder_channels=((16, ), (16, 16), (32, 32), (32, 32)), encoder_paddings=((1, ), (1, 1), (1, 1), (1, 1)),
              decoder_channels=((32, 32, 32), (32, 32, 16), (16, 16, 16), (16, 16, 16)),
              decoder_paddings=((1, 1), (1, 0), (0, 0), (0, 1))),
    mixer=dict(in_channels=16, sparse_shape=[8, 64, 64], base_channels=16, output_channels=16,
               encoder_channels=((16, ), (16, 16)), encoder_paddings=((1, ), (1, 1)),
               decoder_channels=((16, 16, 16), (16, 16, 16)), decoder_paddings=((1, 1), (1, 1))),
    virtual_vfe=dict(feat_channels=[16, 16]), proj_hidden=[16, 16],
    multiscale=dict(multiscale_levels=[0, 1], projector_hiddens=[[32, 16], [16, 16]], fusion_mode='avg',
                    target_sparse_shape=[8, 64, 64], norm_cfg=dict(type='naiveSyncBN1d')),
    as_rpn=True, recover_hidden=[16, 16])


class FSDv2Path(nn.Module):
    """configs/fsdv2/fsdv2_nusc_1x.py at hot-path level over a module provider: segmentor (DynamicScatterVFE +
    SimpleSparseUNet with return_multiscale_features) -> point features -> virtual-voxel stage (SingleStageFSDV2.extract_feat
    WITH the config's multiscale_cfg: the three coarsest decoder levels through their projectors into the virtual-voxel
    grid, single_stage_fsd_v2.py:208-221, 375-433; with `as_rpn` also the per-point outputs of :263-270).  Stand-ins:
    `seg_head` (VoteSegHead: class logits + one 3-vector vote), foreground = points above the ground plane."""

    def __init__(self, ops=GpuOps, cfg=None):
        super().__init__()
        cfg = self.cfg = FSDV2_CFG if cfg is None else cfg
        self.ops = ops
        self.SEG_VOXEL, self.VIRTUAL_VOXEL, self.PC_RANGE = cfg['seg_voxel'], cfg['virtual_voxel'], cfg['pc_range']
        self.n_logits = cfg['n_logits']
        c_seg = cfg['unet']['decoder_channels'][-1][-1]
        self.voxel_encoder = ops.DynamicScatterVFE(voxel_size=self.SEG_VOXEL, with_cluster_center=True, with_voxel_center=True,
                                                   point_cloud_range=self.PC_RANGE, norm_cfg=BN, unique_once=True, **cfg['vfe'])
        self.middle_encoder = ops.PseudoMiddleEncoderForSpconvFSD()
        self.multiscale = cfg.get('multiscale')
        self.as_rpn = bool(cfg.get('as_rpn', False))
        self.seg_backbone = ops.SimpleSparseUNet(order=('conv', 'norm', 'act'), norm_cfg=BN,
                                                 return_multiscale_features=self.multiscale is not None, **cfg['unet'])
        self.seg_head = nn.Linear(c_seg + 3, self.n_logits + 3)      # stand-in: classes + background, one centre vote
        hid = cfg['proj_hidden']
        extra = {}
        if self.as_rpn:
            extra = dict(recover_in_channels=cfg['mixer']['output_channels'] + 3, recover_hidden_dims=cfg['recover_hidden'])
        self.virtual_stage = ops.VirtualVoxelExtractor(
            multiscale_cfg=self.multiscale, as_rpn=self.as_rpn,
            backbone=dict(type='VirtualVoxelMixer', order=('conv', 'norm', 'act'), norm_cfg=BN, **cfg['mixer']),
            voxel_encoder=dict(type='DynamicScatterVFE', in_channels=3 + hid[-1], voxel_size=self.VIRTUAL_VOXEL,
                               with_cluster_center=True, with_voxel_center=True, point_cloud_range=self.PC_RANGE, norm_cfg=BN,
                               unique_once=True, **cfg['virtual_vfe']),
            virtual_point_projector=dict(in_channels=(c_seg + 3) + 3 + self.n_logits + 2, hidden_dims=hid,
                                         norm_cfg=dict(type='naiveSyncBN1d'), ori_in_channels=c_seg + 3,
                                         ori_hidden_dims=hid, **extra))
        assert cfg['mixer']['in_channels'] == cfg['virtual_vfe']['feat_channels'][-1]

    def make_cloud(self, n, seed, dev):
        return lidar_like_cloud(n, seed, dev, half_extent=50.0, z_ground=-1.8)[0]

    def prepare(self, points_list):
        return prepare_index_phase(self, points_list)

    def forward(self, points_list, return_tensors=False, prepared=None):
        dev = points_list[0].device
        if prepared is not None:     # the index phase of this batch was built ahead (prepare_index_phase)
            batch_points, coors = prepared['batch_points'], prepared['coors']
            voxel_feats, voxel_coors, v2p = self.voxel_encoder(batch_points, coors, return_inv=True, grouping=prepared['grouping'])
        else:
            batch_points, coors = self.ops.voxelize(points_list, self.SEG_VOXEL, self.PC_RANGE)
            voxel_feats, voxel_coors, v2p = self.voxel_encoder(batch_points, coors, return_inv=True)
        info = self.middle_encoder(voxel_feats, voxel_coors)
        if prepared is not None:
            info['indice_dict'] = prepared['indice_dict']
        info.setdefault('batch_size', len(points_list))
        x = self.seg_backbone(info)[0]
        pts_feats = x['voxel_feats'][v2p]                                               # Voxel2PointScatterNeck
        vs = _const(self.SEG_VOXEL, dev).reshape(1, 3)
        centre = (coors[:, [3, 2, 1]].float() + 0.5) * vs + _const(self.PC_RANGE[:3], dev).reshape(1, 3)
        seg_feats = torch.cat([pts_feats, batch_points[:, :3] - centre], 1)            # [N, C + 3]
        head = self.seg_head(seg_feats)
        logits, vote = head[:, :self.n_logits], head[:, self.n_logits:]
        sel = torch.nonzero(batch_points[:, 2] > -1.4).squeeze(1)                        # foreground stand-in
        sampled = dict(seg_points=batch_points[sel], center_preds=(batch_points[sel, :3] + torch.tanh(vote[sel])).detach(),
                       seg_logits=logits[sel], seg_feats=seg_feats[sel], batch_idx=coors[sel, 0])
        origin = dict(seg_points=batch_points, seg_feats=seg_feats, batch_idx=coors[:, 0], batch_size=len(points_list))
        ms = x['decoder_features'] if self.multiscale is not None else None
        out = self.virtual_stage(sampled, origin, multiscale_features=ms)
        stats = dict(points=batch_points.size(0), voxels=voxel_feats.size(0), fg_points=int(sel.numel()),
                     virtual_voxels=out['virtual_feats'].size(0))
        if ms is not None:
            stats['multiscale_voxels'] = sum(int(ms[lvl].features.size(0)) for lvl in self.multiscale['multiscale_levels'])
        loss = mixed_sum(out['virtual_feats'], 1e-2) + mixed_sum(logits, 1e-3)
        if self.as_rpn:
            loss = loss + mixed_sum(out['pts_feats'], 1e-3)
        if return_tensors:
            return loss, stats, dict(voxel_coors=voxel_coors, voxel_feats=voxel_feats, unet_feats=x['voxel_feats'],
                                     seg_feats=seg_feats, head=head, virtual_feats=out['virtual_feats'],
                                     virtual_coors=out['virtual_coors'], virtual_centers=out['virtual_centers'],
                                     virtual_centroid=out.get('virtual_centroid'), pts_feats=out.get('pts_feats'))
        return loss, stats


WORKLOADS = {
    'fsd': dict(cls=FSDPath, points=160000, metric='LiDAR frames/sec (FSD hot path fwd+bwd), Waymo geometry',
                name='FSD Waymo hot path: DynamicScatterVFE + SimpleSparseUNet segmentor (0.25 m) -> clustering -> SIR x 3 '
                     '-> point RoI pooling -> SIR x 2'),
    'fsdv2': dict(cls=FSDv2Path, points=300000, metric='LiDAR frames/sec (FSDv2 hot path fwd+bwd), nuScenes 10-sweep geometry',
                  name='FSDv2 nuScenes 10-sweep hot path, the wiring of configs/fsdv2/fsdv2_nusc_1x.py: segmentor U-Net at 0.2 m '
                       '[40,512,512] (return_multiscale_features) -> virtual points -> DynamicScatterVFE at 0.4 m -> '
                       'multiscale fusion of decoder levels 0-2 (projectors [256,128],[128,128],[128,128], target '
                       '[20,256,256], avg) -> VirtualVoxelMixer [20,256,256]'),
}


def _conv_roofline(model, clouds):
    """One instrumented forward pass (outside the timed region): HIP events around every sparse convolution's forward on
    torch's current stream - the stream every kernel of this library is launched on - and the FLOPs of its rulebook
    (2 x pairs x C_in x C_out).  The other instrumented kernel family: the segmented reductions (SIR / VFE pooling)."""
    from sst_amd import spconv as SP
    from sst_amd import kernels as K
    from sst_amd import _lib
    records, handles = [], []

    def pre(mod, inp):
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
        mod._bench_e0 = e0

    def post(mod, inp, out):
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        pairs = out.indice_dict.get(mod.indice_key) if mod.indice_key is not None else None
        rb = getattr(pairs[2] if isinstance(pairs, (tuple, list)) else pairs, '_sst_rulebook', None) if pairs is not None else None
        if rb is None:                                   # find the rulebook on whatever tensor of the entry carries it
            for t in (pairs if isinstance(pairs, (tuple, list)) else [pairs]):
                rb = rb or getattr(t, '_sst_rulebook', None)
        if rb is not None:
            records.append((mod._bench_e0, e1, 2.0 * rb.total_pairs * mod.in_channels * mod.out_channels))

    for m in model.modules():
        if isinstance(m, SP.SparseConvolution):
            handles.append(m.register_forward_pre_hook(pre))
            handles.append(m.register_forward_hook(post))
    seg_records = []
    orig_reduce = K.segment_reduce

    lib = _lib.load()

    def timed_reduce(feat, plan, mode, *a, **kw):
        ke = K._KernelEvents(lib)                      # HIP events bound to the kernel launch itself (start / stop)
        lib.sst_segment_reduce_profile_next(ke.start, ke.stop)
        out = orig_reduce(feat, plan, mode, *a, **kw)
        lib.sst_segment_reduce_profile_next(None, None)
        seg_records.append((ke, feat.numel() * 4 + out.numel() * 4 + feat.size(0) * 4))
        return out

    K.segment_reduce = timed_reduce
    try:
        with torch.no_grad():
            model(clouds)
        torch.cuda.synchronize()
    finally:
        K.segment_reduce = orig_reduce
        for h in handles:
            h.remove()
    conv = None
    if records:
        ms = sum(e0.elapsed_time(e1) for e0, e1, _ in records)
        fl = sum(f for _, _, f in records)
        conv = {'bound': 'mfma', 'kernel': 'sparse convolutions of one forward pass (sp_conv_* / spconv_gather_gemm kernels, '
                                           f'{len(records)} layers; events around each layer: rulebook reuse included, '
                                           'rulebook construction of the first layer of a key included)',
                'achieved': round(fl / (ms * 1e-3) / 1e12, 2), 'peak': FP32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                'frac': round(fl / (ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4), 'traffic': None,
                'algorithmic_flops': fl, 'ms': round(ms, 3)}
        from sst_amd import spconv as _sp
        if _sp.conv_precision() == 'f32x6':
            # the timed mode issues SIX bf16 products per algorithmic fp32 product on the bf16 pipe (exact three-way split): the
            # honest denominator for the pipe it runs on is the dense bf16 peak with that factor stated
            issued = 6.0 * fl / (ms * 1e-3) / 1e12
            conv['peak_note'] = ('achieved / peak / frac count ALGORITHMIC fp32 flops against the fp32 matrix pipe (157.3 TFLOP/s); the '
                                 'kernels run on the bf16 pipe and issue 6 x those flops: frac_of_bf16_peak = 6 x achieved / 2500')
            conv['issued_bf16_tflops'] = round(issued, 1)
            conv['frac_of_bf16_peak'] = round(issued / BF16_MFMA_PEAK_TFLOPS, 4)
    seg = None
    timed = [(t, b) for t, b in ((ke.elapsed_time(ke), b) for ke, b in seg_records) if t > 0]
    if timed:
        ms = sum(t for t, _ in timed)
        by = sum(b for _, b in timed)
        seg = {'bound': 'hbm', 'kernel': f'seg_reduce_fwd_* ({len(timed)} segmented reductions of one forward pass; HIP events bound '
                                       'to each launch)',
               'achieved': round(by / (ms * 1e-3) / 1e9, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
               'frac': round(by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), 'algorithmic_bytes': by, 'ms': round(ms, 3),
               'launches_mb_us': [[round(b / 1e6, 2), round(t * 1e3, 1)] for t, b in timed]}
    return conv, seg


def _chain_parity(port, model, cloud):
    """one forward + backward of the CPU port and of the GPU chain on the same frame with the same weights -> (cpu seconds, parity)"""
    frame = [cloud.cpu()]
    t0 = time.perf_counter()
    loss_c, _, tc = port(frame, return_tensors=True)
    loss_c.backward()
    cpu_s = time.perf_counter() - t0
    model.zero_grad(set_to_none=True)
    loss_g, stats, tg = model([cloud], return_tensors=True)
    loss_g.backward()
    ints, feats = {}, {}
    for key, a in tc.items():
        b = tg.get(key)
        if a is None or b is None:
            continue
        b = b.detach().cpu()
        same_shape = tuple(a.shape) == tuple(b.shape)
        if a.is_floating_point():
            feats[key] = float((a.detach() - b).abs().max()) if same_shape else None
        else:
            ints[key] = bool(same_shape and torch.equal(a.long(), b.long()))
    pg, pc = dict(model.named_parameters()), dict(port.named_parameters())
    grad_names = [n for n in ('seg_backbone.conv_input.0.weight', 'seg_head.weight', 'backbone.block_list.0.vfe_layers.0.linear.weight',
                              'virtual_stage.backbone.conv_out.0.weight') if n in pc and pc[n].grad is not None]
    grads = {n: float((pg[n].grad.cpu() - pc[n].grad).abs().max() / pc[n].grad.abs().max().clamp(min=1e-12))
             for n in grad_names}
    finite = [v for v in feats.values() if v is not None]
    parity = {'integer_outputs_equal': ints, 'max_abs_err': {k: (round(v, 9) if v is not None else None) for k, v in feats.items()},
              'max_abs_err_overall': max(finite) if finite else None, 'max_rel_grad_err': grads, 'feature_tolerance': 1e-3,
              'what': 'GPU chain (fp32) vs the CPU port of the reference chain, same weights, first bench frame, training '
                      'mode; a feature entry is null when an integer stage upstream of it differs (its rows are then not '
                      'comparable); roi_stage: the second stage rides on the point pool, whose 13 features and cap survivors '
                      'are UNPINNED (TorchEx source absent: SURVEY.md section 8 f3) - its membership is pinned to the '
                      'reference\'s points_in_boxes_cpu.cpp; gradient entries = max |difference| / max |gradient| of a parameter '
                      'between two fp32 evaluations.  These gradients are ill-conditioned (training-mode batch norm over few '
                      'rows + ReLU / max decisions): the float64 evaluation of the port puts the fp32 PORT ITSELF 1e-3..3e-2 '
                      'from the exact gradient on these frames and the GPU path no further (profiles/r04/'
                      '*_grad_adjudication.json, tests/adjudicate_fsd_grads.py, tests/test_fsd_chain.py::'
                      'test_gpu_gradients_within_fp32_noise_at_40k); the 1e-3 bar is stated for features'}
    return cpu_s, parity


def cpu_chain_leg(spec, model, clouds, parity_points=None):
    """The CPU port of the same chain (oracle/fsd_cpu.py: the reference's algorithm module by module, pinned to the
    reference by tests/test_fsd_chain.py) on the host cores, rank 0 at N = 1 - the ONLY place this file touches oracle/:
      * `cpu_baseline`: frames/s of forward + backward of the first bench frame: one untimed warm-up pass on a 20 000-point
        cloud (thread pools, allocator), then ONE timed pass at full size (tens of seconds);
      * `parity`: the GPU chain with the SAME weights on the SAME frame against that CPU pass - integer side (voxel set,
        foreground selection, cluster assignment / virtual voxels) and the features along the chain; gradients of three
        parameters at the ends and the middle of the chain."""
    from oracle import fsd_cpu
    all_threads = torch.get_num_threads()
    port = spec['cls'](fsd_cpu).train()
    port.load_state_dict({k: v.cpu() for k, v in model.state_dict().items()}, strict=True)
    if parity_points:
        # bench.py --compact (a leg inside the default line): parity only, on a bounded frame of the same generator - one CPU
        # pass of a few seconds at 16 threads instead of the thread sweep + the full-size timed pass
        torch.set_num_threads(min(16, all_threads))
        small = model.make_cloud(int(parity_points), 1000, clouds[0].device)
        try:
            _, parity = _chain_parity(port, model, small)
        finally:
            torch.set_num_threads(all_threads)
        parity['frame_points'] = int(parity_points)
        return None, parity
    # thread sweep on a 20 000-point cloud (the SST leg showed 128 threads to be 3.5 x slower than 16-32 for this kind of
    # work: many small ops): one untimed warm-up pass, then one timed forward + backward per candidate thread count
    probe = [model.make_cloud(20000, 7, 'cpu')]
    port(probe)[0].backward()
    port.zero_grad(set_to_none=True)
    sweep = {}
    for t in sorted({t for t in (8, 16, 32, 64, all_threads) if t <= all_threads}):
        torch.set_num_threads(t)
        t0 = time.perf_counter()
        port(probe)[0].backward()
        sweep[t] = round(time.perf_counter() - t0, 2)
        port.zero_grad(set_to_none=True)
    threads = min(sweep, key=sweep.get)
    torch.set_num_threads(threads)
    torch.set_num_threads(threads)
    cpu_s, parity = _chain_parity(port, model, clouds[0])
    torch.set_num_threads(all_threads)
    frame = [clouds[0]]
    base = {'value': round(1.0 / cpu_s, 5), 'unit': 'frames/s', 'cores': threads, 'kind': 'port',
            'thread_sweep_20k_points_s': sweep,
            'sample': f'1 timed pass of 1 frame ({frame[0].size(0)} points), forward + backward, {cpu_s:.1f} s at {threads} threads '
                      '(the fastest of a sweep on a 20 000-point cloud, after one untimed warm-up pass); CPU port of the '
                      'reference chain (oracle/fsd_cpu.py: '
                      'torch.unique + scatter_reduce, per-offset gather / mm / index_add sparse convolutions, dense-adjacency '
                      'scipy connected components, numpy point pool)'}
    return base, parity


_CONV_MODE_WHAT = {
    'f32x3': 'same step, sparse convolutions (forward + data gradient) as three bf16 MFMA products of two-way split fp32 operands '
             'with fp32 accumulation (csrc/spconv_os_x3.hip; NOT exact: ~1e-5 of the output scale per layer - a leg, never `value`); '
             'filter gradients, everything else and all tensors fp32',
    'f32': 'same step with the sparse convolutions on the fp32 matrix pipe (v_mfma_f32_16x16x4_f32, csrc/spconv_os.hip) - the '
           'arithmetic of rounds 1-3; the timed mode evaluates the same products from the exact three-way bf16 split',
    'f32x6': 'same step, sparse convolutions from the exact three-way bf16 split (csrc/spconv_os_x6.hip)'}


def _conv_precision_leg(mode, timed_mode, model, clouds, step, sync, args, world, dev):
    """the same step with the sparse convolutions (forward contraction and data gradient) in another multiply mode, reported
    BESIDE `value`, with the forward deviation from the timed mode on the same frame"""
    from sst_amd import spconv
    with torch.no_grad():                               # training mode as in the step: batch statistics in both passes
        _, _, ref = model(clouds, return_tensors=True)
        spconv.set_conv_precision(mode)
        try:
            _, _, got = model(clouds, return_tensors=True)
        finally:
            spconv.set_conv_precision(timed_mode)
    errs = {}
    for key in ('unet_feats', 'seg_feats', 'cluster_feats', 'virtual_feats'):
        if key in ref and key in got and ref[key].shape == got[key].shape and ref[key].is_floating_point():
            scale = max(1.0, float(ref[key].abs().max()))
            errs[key] = round(float((ref[key] - got[key]).abs().max()) / scale, 8)
    spconv.set_conv_precision(mode)
    try:
        for _ in range(2):
            step()
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync()
        elapsed = time.perf_counter() - t0
    finally:
        spconv.set_conv_precision(timed_mode)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return {'value': round(world * args.frames_per_gpu * args.steps / elapsed, 3), 'unit': 'frames/s',
            'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'steps': args.steps, 'what': _CONV_MODE_WHAT[mode],
            'max_err_vs_timed_mode_forward_rel_to_output_scale': errs}


def run(args, rank, world, dev, make_reducer, line_out=None, step_times_cls=None):
    """bench.py's contract for --workload fsd | fsdv2: W warm-up steps, K timed steps between barriers, max over ranks,
    one JSON line from rank 0 (to `line_out`: bench.py keeps everything else off its stdout)."""
    import sys
    line_out = line_out if line_out is not None else sys.stdout
    spec = WORKLOADS[args.workload]
    if not getattr(args, 'no_gemm_tuning', False):
        # the point-wise linears (VFE / SIR layers, the stand-in heads: tall and very narrow products) are library GEMMs:
        # TunableOp picks the fastest hipBLASLt / rocBLAS solution per shape during warm-up, as bench.py does for the headline
        # (seeded from the committed results file of the workload, so that a fresh box does not spend ~40 s searching)
        import shutil
        import torch.cuda.tunable as tunable
        seed_file = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'sst_amd', f'tunableop_gfx950_{args.workload}.csv')
        work_file = f'/tmp/sst_amd_tunableop_{args.workload}_rank{rank}.csv'
        if os.path.exists(seed_file) and not os.path.exists(work_file):
            shutil.copyfile(seed_file, work_file)
        tunable.enable(True)
        tunable.tuning_enable(True)
        tunable.set_filename(work_file)
    conv_prec = getattr(args, 'conv_precision', 'f32x6')
    from sst_amd import spconv
    spconv.set_conv_precision(conv_prec)
    if conv_prec == 'f32x3':               # profiling the two-way split: not a headline, no parity legs
        args.no_f32x3_leg = args.no_cpu_baseline = True
    if getattr(args, 'compact', False):
        args.no_f32x3_leg = True
    torch.manual_seed(0)
    model = spec['cls']().to(dev).train()
    params = [p for p in model.parameters() if p.requires_grad]
    n_pts = args.points if args.points_given else spec['points']
    clouds = [model.make_cloud(n_pts, 1000 * rank + i, dev) for i in range(args.frames_per_gpu)]
    stats = {}
    reducer = make_reducer(params, world, args)

    prefetch = not getattr(args, 'no_plan_prefetch', False)
    ahead = []      # the index phase of the NEXT batch (depends on the point clouds only: no parameters, no features)

    def step():
        for p in params:
            p.grad = None
        loss, st = model(clouds, prepared=ahead.pop() if ahead else None)
        stats.update(st)
        if prefetch:
            # between the forward and the backward pass: its read-backs wait for the (short, host-bound) forward kernels, and the
            # next step's forward is then issued without a stop while the device still works through this backward pass
            ahead.append(model.prepare(clouds))
        loss.backward()
        if reducer is not None:
            reducer.finish()        # parameters without a gradient this step (an empty stage) travel as zeros

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    import gc
    gc.collect()
    gc.freeze()      # see bench.py: a full collection of a torch process costs 60-70 ms; frozen objects are not walked
    sync()
    times = step_times_cls() if step_times_cls is not None else None
    if times is not None:
        times.mark()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        if times is not None:
            times.mark()
    sync()
    elapsed = time.perf_counter() - t0
    step_ms = times.close().stats() if times is not None else None
    # interpreter + launch time of a step: forward and backward timed on the host with the device idle at their start
    host_ms = None
    if world == 1:
        tot = 0.0
        for _ in range(2):
            for p in params:
                p.grad = None
            sync()
            ta = time.perf_counter()
            loss, _st = model(clouds, prepared=None)
            tb = time.perf_counter()
            sync()
            tc = time.perf_counter()
            loss.backward()
            td = time.perf_counter()
            sync()
            tot += (tb - ta) + (td - tc)
        host_ms = round(tot / 2 * 1e3, 3)
        ahead.clear()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    conv, seg = _conv_roofline(model, clouds)
    x3 = mfma_leg = None
    if not getattr(args, 'no_f32x3_leg', False):
        x3 = _conv_precision_leg('f32x3', conv_prec, model, clouds, step, sync, args, world, dev)
        if conv_prec == 'f32x6':
            mfma_leg = _conv_precision_leg('f32', conv_prec, model, clouds, step, sync, args, world, dev)
    cpu_base = parity = None
    if rank == 0 and world == 1 and not getattr(args, 'no_cpu_baseline', False):
        cpu_base, parity = cpu_chain_leg(spec, model, clouds,
                                         parity_points=getattr(args, 'parity_points', None) if getattr(args, 'compact', False) else None)
    if rank == 0:
        frames = world * args.frames_per_gpu * args.steps
        res = {'metric': spec['metric'], 'value': round(frames / elapsed, 3), 'unit': 'frames/s', 'n_gpus': world,
               'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 3),
               'step_ms': step_ms, 'host_ms_per_step': host_ms,
               'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
               'dtype': {'f32': 'f32',
                         'f32x6': 'f32 (storage, accumulation, norms, reductions, filter gradients: fp32; the sparse convolutions\' '
                                  'forward and data-gradient products from the EXACT three-way bf16 split of both fp32 operands, six '
                                  'bf16 MFMA products with fp32 accumulation - error vs float64 <= 2 x the fp32 matrix pipe\'s: '
                                  'tests/test_gpu_spconv.py::test_exact_split_convolution_kernel)',
                         'f32x3': 'f32 storage, sparse convolutions as 3 bf16 products (NOT the headline mode)'}[conv_prec],
               'data': 'synthetic',
               'config': {'workload': spec['name'] + f'; {n_pts} points/frame (ground plane + boxes), fwd+bwd',
                          'frames_per_gpu': args.frames_per_gpu, 'points_per_frame': n_pts, 'parallelism': f'dp{world}',
                          'index_plan': ('voxelisation, voxel grouping and the rulebooks of the segmentor U-Net of the next batch built '
                                         'between the forward and the backward pass of the current one (they depend on the point '
                                         'cloud only; the cluster / virtual-voxel stages, which depend on the network, stay inside '
                                         'the forward pass)') if prefetch else 'built at the head of its step',
                          'sizes': {k: int(v) for k, v in stats.items()},
                          'stand_ins': 'segmentation / box heads = linear layers, foreground = geometric rule '
                                       '(detector glue is out of scope)'},
               'roofline': conv if conv is not None else seg, 'roofline_seg_reduce': seg, 'cpu_baseline': cpu_base,
               'parity': parity}
        if x3 is not None:
            res['precision_f32x3'] = x3
        if mfma_leg is not None:
            res['precision_f32_mfma'] = mfma_leg
        print(json.dumps(res), file=line_out)
        line_out.flush()
