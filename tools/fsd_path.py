"""FSD's hot path end to end on synthetic data, with the modules of this repository wired as the reference wires them
(detector glue -- heads, losses, box decoding -- replaced by small linear layers; detectors are out of scope):

  points -> Voxelization -> DynamicScatterVFE -> SimpleSparseUNet          VoteSegmentor.extract_feat, single_stage_fsd.py:228-250
         -> voxel-to-point gather + local xyz                             Voxel2PointScatterNeck, necks/voxel2point_neck.py:28-63
         -> (linear stand-in for the vote / segmentation head)
         -> foreground points, voted centres -> ClusterAssigner           single_stage_fsd.py:922-999 (connected components)
         -> scatter_v2 cluster centres, SIR x 3                            SingleStageFSD.extract_feat, :467-483
         -> (linear stand-in for the box head) -> RoIs
         -> DynamicPointROIExtractor -> SIRLayer x 2 on the pooled points  roi_heads/.../dynamic_point_roi_extractor.py, fsd_bbox_head.py:69-97

Prints forward and forward + backward time per frame.  `python tools/fsd_path.py [points_per_frame] [frames]`."""
import os
import sys

import numpy as np
import torch
from torch import nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sst_amd  # noqa: E402
from sst_amd.sst_ops import scatter_v2  # noqa: E402

DEV = 'cuda:0'
SEG_VOXEL = (0.25, 0.25, 0.2)
PC_RANGE = [-80, -80, -2, 80, 80, 4]
CLASSES = ['Car', 'Pedestrian', 'Cyclist']
BN = dict(type='naiveSyncBN1d', eps=1e-3, momentum=0.01)


def lidar_like_cloud(n, seed):
    """ground plane + boxes: points on surfaces, so that voxels have LiDAR-like neighbourhoods"""
    g = torch.Generator().manual_seed(seed)
    n_obj = n // 4
    ground = torch.rand(n - n_obj, 3, generator=g) * torch.tensor([150.0, 150.0, 0.15]) + torch.tensor([-75.0, -75.0, -1.8])
    centres = torch.rand(60, 3, generator=g) * torch.tensor([120.0, 120.0, 0.0]) + torch.tensor([-60.0, -60.0, -0.9])
    obj = centres[torch.randint(0, 60, (n_obj,), generator=g)] + (torch.rand(n_obj, 3, generator=g) - 0.5) * torch.tensor([4.0, 2.0, 1.6])
    xyz = torch.cat([ground, obj])
    return torch.cat([xyz, torch.rand(n, 2, generator=g)], 1).to(DEV), centres


class FSDPath(nn.Module):

    def __init__(self):
        super().__init__()
        self.voxel_layer = sst_amd.Voxelization(SEG_VOXEL, PC_RANGE, -1, (-1, -1))
        self.voxel_encoder = sst_amd.DynamicScatterVFE(in_channels=5, feat_channels=[64, 64], voxel_size=SEG_VOXEL,
                                                       with_cluster_center=True, with_voxel_center=True,
                                                       point_cloud_range=PC_RANGE, norm_cfg=BN, unique_once=True)
        self.middle_encoder = sst_amd.PseudoMiddleEncoderForSpconvFSD()
        self.seg_backbone = sst_amd.SimpleSparseUNet(
            in_channels=64, sparse_shape=[32, 640, 640], order=('conv', 'norm', 'act'), norm_cfg=BN, base_channels=64,
            output_channels=128, encoder_channels=((64, ), (64, 64, 64), (64, 64, 64), (128, 128, 128), (256, 256, 256)),
            encoder_paddings=((1, ), (1, 1, 1), (1, 1, 1), ((0, 1, 1), 1, 1), (1, 1, 1)),
            decoder_channels=((256, 256, 128), (128, 128, 64), (64, 64, 64), (64, 64, 64), (64, 64, 64)),
            decoder_paddings=((1, 1), (1, 0), (1, 0), (0, 0), (0, 1)))
        self.seg_head = nn.Linear(67, 3 + 9)       # stand-in: 3 class logits + 3 x 3 centre votes
        self.cluster_assigner = sst_amd.ClusterAssigner(
            cluster_voxel_size=dict(Car=(0.3, 0.3, 6), Cyclist=(0.2, 0.2, 6), Pedestrian=(0.05, 0.05, 6)), min_points=2,
            point_cloud_range=PC_RANGE, connected_dist=dict(Car=0.6, Cyclist=0.4, Pedestrian=0.1), class_names=CLASSES)
        self.backbone = sst_amd.SIR(num_blocks=3, in_channels=[84, 133, 133], feat_channels=[[128, 128]] * 3,
                                    rel_mlp_hidden_dims=[[16, 32]] * 3, norm_cfg=dict(type='LN', eps=1e-3), mode='max',
                                    xyz_normalizer=[20, 20, 4], act='gelu', unique_once=True)
        self.box_head = nn.Linear(128 * 3 * 2, 7)  # stand-in: centre offset, log sizes, yaw
        self.roi_extractor = sst_amd.DynamicPointROIExtractor(extra_wlh=[0.5, 0.5, 0.5], max_inbox_point=256, debug=False)
        self.roi_backbone = sst_amd.SIR(num_blocks=2, in_channels=[13 + 128 + 13, 13 + 128], feat_channels=[[128, 128]] * 2,
                                        rel_mlp_hidden_dims=[[16, 32]] * 2, norm_cfg=dict(type='LN', eps=1e-3),
                                        mode='max', xyz_normalizer=[20, 20, 4], act='gelu', unique_once=True)

    def forward(self, points_list):
        batch_points, coors = self.voxel_layer.voxelize_batch(points_list)
        coors = coors.long()
        voxel_feats, voxel_coors, v2p = self.voxel_encoder(batch_points, coors, return_inv=True)
        x = self.seg_backbone(self.middle_encoder(voxel_feats, voxel_coors))[0]
        # Voxel2PointScatterNeck
        pts_feats = x['voxel_feats'][v2p]
        vs = torch.tensor(SEG_VOXEL, device=DEV).reshape(1, 3)
        centre = (coors[:, [3, 2, 1]].float() + 0.5) * vs + torch.tensor(PC_RANGE[:3], device=DEV).reshape(1, 3)
        seg_feats = torch.cat([pts_feats, batch_points[:, :3] - centre], 1)            # [N, 67]
        head = self.seg_head(seg_feats)
        logits, votes = head[:, :3], head[:, 3:].reshape(-1, 3, 3)
        # foreground: points above the ground (the synthetic objects), class by position hash (no labels here)
        fg = batch_points[:, 2] > -1.4
        cls = (batch_points[:, 0].abs() * 7).long() % 3
        batch_idx = coors[:, 0]
        pts_l, bidx_l, sel_l = [], [], []
        for c in range(3):
            sel = torch.nonzero(fg & (cls == c)).squeeze(1)
            sel_l.append(sel)
            pts_l.append((batch_points[sel, :3] + 0.05 * torch.tanh(votes[sel, c])).detach())
            bidx_l.append(batch_idx[sel])
        cluster_inds_l, valid_l = self.cluster_assigner(pts_l, bidx_l)
        sel = torch.cat([s[v] for s, v in zip(sel_l, valid_l)])
        cluster_inds = torch.cat(cluster_inds_l)                                      # [P, 3] (class, sample, cluster)
        centres = torch.cat([p[v] for p, v in zip(pts_l, valid_l)])
        points = batch_points[sel]
        feats = torch.cat([seg_feats[sel], logits[sel], votes[sel].reshape(-1, 9)], 1)  # [P, 79]
        # SingleStageFSD.extract_feat
        cluster_xyz, _, inv = scatter_v2(centres, cluster_inds, mode='avg', return_inv=True)
        f_cluster = points[:, :3] - cluster_xyz[inv]
        pts_out, cluster_feats, cluster_coors = self.backbone(points, feats, cluster_inds, f_cluster)
        box = self.box_head(cluster_feats)
        rois = torch.cat([cluster_coors[:, 1:2].float(), cluster_xyz + 0.1 * torch.tanh(box[:, :3]) - torch.tensor([0, 0, 0.9], device=DEV),
                          torch.tensor([2.0, 4.4, 1.8], device=DEV) * torch.exp(0.1 * torch.tanh(box[:, 3:6])),
                          box[:, 6:7]], 1).detach()
        order = torch.argsort(rois[:, 0], stable=True)                                 # RoIs sample after sample
        rois = rois[order]
        p_order = torch.argsort(cluster_inds[:, 1], stable=True)                       # points sample after sample
        ext_inds, roi_inds, info = self.roi_extractor(points[p_order, :3].contiguous(), cluster_inds[p_order, 1], rois)
        keep = ext_inds >= 0                                                           # drops the "fake" row of an empty pool
        pooled_xyz = points[p_order][ext_inds.clamp(min=0), :3]
        geo = torch.cat([info['local_xyz'], info['boundary_offset'], info['is_in_margin'][:, None], pooled_xyz], 1)[keep]
        ext_inds, roi_inds = ext_inds[keep], roi_inds[keep]
        roi_feats = torch.cat([pts_out[p_order][ext_inds], geo], 1)                     # [Q, 128 + 13]
        roi_coors = torch.stack([torch.zeros_like(roi_inds), rois[roi_inds, 0].long(), roi_inds], 1)
        _, roi_cluster_feats, _ = self.roi_backbone(geo, roi_feats, roi_coors, geo[:, :3].contiguous())
        stats = dict(points=batch_points.size(0), voxels=voxel_feats.size(0), fg_points=points.size(0),
                     clusters=cluster_feats.size(0), pooled_pairs=roi_feats.size(0))
        return roi_cluster_feats.sum() + cluster_feats.sum() * 1e-3 + logits.sum() * 1e-3, stats


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 160000
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    torch.manual_seed(0)
    net = FSDPath().to(DEV).train()
    clouds = [lidar_like_cloud(n, i)[0] for i in range(frames)]

    def step(backward):
        for p in net.parameters():
            p.grad = None
        loss, stats = net(clouds)
        if backward:
            loss.backward()
        return stats

    for _ in range(3):
        stats = step(True)
    times = {}
    for name, bw in (('forward', False), ('forward + backward', True)):
        ts = []
        for _ in range(8):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if bw:
                step(True)
            else:
                with torch.no_grad():
                    step(False)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        times[name] = float(np.median(ts))
    print(f'FSD path, {frames} frame(s) x {n} points: {stats}; forward {times["forward"]:.1f} ms, '
          f'forward + backward {times["forward + backward"]:.1f} ms '
          f'({frames * 1000.0 / times["forward + backward"]:.1f} frames/s fwd+bwd, {frames * 1000.0 / times["forward"]:.1f} fwd)')


if __name__ == '__main__':
    main()
