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
import time

import numpy as np
import torch
import torch.distributed as dist
from torch import nn

import sst_amd
from sst_amd.sst_ops import scatter_v2

HBM_PEAK_GBS = 8000.0
FP32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / 16x16x4_f32, dense
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


class FSDPath(nn.Module):
    """configs/fsd/fsd_waymoD1_1x.py at hot-path level.  Stand-ins: `seg_head` (VoteSegHead), `box_head` (the FSD head's
    regression branch); foreground = points above the ground plane, class by a position hash (no labels here)."""
    SEG_VOXEL = (0.25, 0.25, 0.2)
    PC_RANGE = [-80, -80, -2, 80, 80, 4]
    CLASSES = ['Car', 'Pedestrian', 'Cyclist']

    def __init__(self):
        super().__init__()
        self.voxel_layer = sst_amd.Voxelization(self.SEG_VOXEL, self.PC_RANGE, -1, (-1, -1))
        self.voxel_encoder = sst_amd.DynamicScatterVFE(in_channels=5, feat_channels=[64, 64], voxel_size=self.SEG_VOXEL,
                                                       with_cluster_center=True, with_voxel_center=True,
                                                       point_cloud_range=self.PC_RANGE, norm_cfg=BN, unique_once=True)
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
            point_cloud_range=self.PC_RANGE, connected_dist=dict(Car=0.6, Cyclist=0.4, Pedestrian=0.1),
            class_names=self.CLASSES)
        self.backbone = sst_amd.SIR(num_blocks=3, in_channels=[84, 133, 133], feat_channels=[[128, 128]] * 3,
                                    rel_mlp_hidden_dims=[[16, 32]] * 3, norm_cfg=dict(type='LN', eps=1e-3), mode='max',
                                    xyz_normalizer=[20, 20, 4], act='gelu', unique_once=True)
        self.box_head = nn.Linear(128 * 3 * 2, 7)  # stand-in: centre offset, log sizes, yaw
        self.roi_extractor = sst_amd.DynamicPointROIExtractor(extra_wlh=[0.5, 0.5, 0.5], max_inbox_point=256, debug=False)
        self.roi_backbone = sst_amd.SIR(num_blocks=2, in_channels=[13 + 128 + 13, 13 + 128], feat_channels=[[128, 128]] * 2,
                                        rel_mlp_hidden_dims=[[16, 32]] * 2, norm_cfg=dict(type='LN', eps=1e-3),
                                        mode='max', xyz_normalizer=[20, 20, 4], act='gelu', unique_once=True)

    def make_cloud(self, n, seed, dev):
        return lidar_like_cloud(n, seed, dev)[0]

    def forward(self, points_list):
        dev = points_list[0].device
        batch_points, coors = self.voxel_layer.voxelize_batch(points_list)
        coors = coors.long()
        voxel_feats, voxel_coors, v2p = self.voxel_encoder(batch_points, coors, return_inv=True)
        x = self.seg_backbone(self.middle_encoder(voxel_feats, voxel_coors))[0]
        # Voxel2PointScatterNeck
        pts_feats = x['voxel_feats'][v2p]
        vs = torch.tensor(self.SEG_VOXEL, device=dev).reshape(1, 3)
        centre = (coors[:, [3, 2, 1]].float() + 0.5) * vs + torch.tensor(self.PC_RANGE[:3], device=dev).reshape(1, 3)
        seg_feats = torch.cat([pts_feats, batch_points[:, :3] - centre], 1)            # [N, 67]
        head = self.seg_head(seg_feats)
        logits, votes = head[:, :3], head[:, 3:].reshape(-1, 3, 3)
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
        rois = torch.cat([cluster_coors[:, 1:2].float(),
                          cluster_xyz + 0.1 * torch.tanh(box[:, :3]) - torch.tensor([0, 0, 0.9], device=dev),
                          torch.tensor([2.0, 4.4, 1.8], device=dev) * torch.exp(0.1 * torch.tanh(box[:, 3:6])),
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


class FSDv2Path(nn.Module):
    """configs/fsdv2/fsdv2_nusc_1x.py at hot-path level: segmentor (DynamicScatterVFE + SimpleSparseUNet, 0.2 m,
    [40, 512, 512]) -> point features -> virtual-voxel stage (0.4 m, [20, 256, 256]).  Stand-ins: `seg_head` (VoteSegHead:
    11 logits + one 3-vector vote), foreground = points above the ground plane; the multi-scale fusion of the config
    (multiscale_cfg) is not part of the stage (sst_amd/virtual_voxel.py)."""
    SEG_VOXEL = (0.2, 0.2, 0.2)
    VIRTUAL_VOXEL = (0.4, 0.4, 0.4)
    PC_RANGE = [-51.2, -51.2, -5, 51.2, 51.2, 3]

    def __init__(self):
        super().__init__()
        self.voxel_layer = sst_amd.Voxelization(self.SEG_VOXEL, self.PC_RANGE, -1, (-1, -1))
        self.voxel_encoder = sst_amd.DynamicScatterVFE(in_channels=5, feat_channels=[64, 64], voxel_size=self.SEG_VOXEL,
                                                       with_cluster_center=True, with_voxel_center=True,
                                                       point_cloud_range=self.PC_RANGE, norm_cfg=BN, unique_once=True)
        self.middle_encoder = sst_amd.PseudoMiddleEncoderForSpconvFSD()
        self.seg_backbone = sst_amd.SimpleSparseUNet(
            in_channels=64, sparse_shape=[40, 512, 512], order=('conv', 'norm', 'act'), norm_cfg=BN, base_channels=64,
            output_channels=128,
            encoder_channels=((128, ), (128, 128), (128, 128), (128, 128, 128), (256, 256, 256), (256, 256, 256)),
            encoder_paddings=((1, ), (1, 1), (1, 1), (1, 1, 1), (1, 1, 1), (1, 1, 1)),
            decoder_channels=((256, 256, 256), (256, 256, 128), (128, 128, 128), (128, 128, 128), (128, 128, 128), (128, 128, 128)),
            decoder_paddings=((1, 1), (1, 0), (1, 0), (0, 0), (0, 1), (1, 1)))
        self.seg_head = nn.Linear(131, 11 + 3)      # stand-in: 10 classes + background, one centre vote
        self.virtual_stage = sst_amd.VirtualVoxelExtractor(
            backbone=dict(type='VirtualVoxelMixer', in_channels=128, sparse_shape=[20, 256, 256],
                          order=('conv', 'norm', 'act'), norm_cfg=BN, base_channels=64, output_channels=128,
                          encoder_channels=((64, ), (64, 64), (64, 64)), encoder_paddings=((1, ), (1, 1), (1, 1)),
                          decoder_channels=((64, 64, 64), (64, 64, 64), (64, 64, 64)),
                          decoder_paddings=((1, 1), (1, 1), (1, 1))),
            voxel_encoder=dict(type='DynamicScatterVFE', in_channels=67, feat_channels=[64, 128],
                               voxel_size=self.VIRTUAL_VOXEL, with_cluster_center=True, with_voxel_center=True,
                               point_cloud_range=self.PC_RANGE, norm_cfg=BN, unique_once=True),
            virtual_point_projector=dict(in_channels=83 + 64, hidden_dims=[64, 64], norm_cfg=dict(type='naiveSyncBN1d'),
                                         ori_in_channels=67 + 64, ori_hidden_dims=[64, 64]))

    def make_cloud(self, n, seed, dev):
        return lidar_like_cloud(n, seed, dev, half_extent=50.0, z_ground=-1.8)[0]

    def forward(self, points_list):
        dev = points_list[0].device
        batch_points, coors = self.voxel_layer.voxelize_batch(points_list)
        coors = coors.long()
        voxel_feats, voxel_coors, v2p = self.voxel_encoder(batch_points, coors, return_inv=True)
        x = self.seg_backbone(self.middle_encoder(voxel_feats, voxel_coors))[0]
        pts_feats = x['voxel_feats'][v2p]                                               # Voxel2PointScatterNeck
        vs = torch.tensor(self.SEG_VOXEL, device=dev).reshape(1, 3)
        centre = (coors[:, [3, 2, 1]].float() + 0.5) * vs + torch.tensor(self.PC_RANGE[:3], device=dev).reshape(1, 3)
        seg_feats = torch.cat([pts_feats, batch_points[:, :3] - centre], 1)            # [N, 131]
        head = self.seg_head(seg_feats)
        logits, vote = head[:, :11], head[:, 11:]
        sel = torch.nonzero(batch_points[:, 2] > -1.4).squeeze(1)                        # foreground stand-in
        sampled = dict(seg_points=batch_points[sel], center_preds=(batch_points[sel, :3] + torch.tanh(vote[sel])).detach(),
                       seg_logits=logits[sel], seg_feats=seg_feats[sel], batch_idx=coors[sel, 0])
        origin = dict(seg_points=batch_points, seg_feats=seg_feats, batch_idx=coors[:, 0], batch_size=len(points_list))
        out = self.virtual_stage(sampled, origin)
        stats = dict(points=batch_points.size(0), voxels=voxel_feats.size(0), fg_points=int(sel.numel()),
                     virtual_voxels=out['virtual_feats'].size(0))
        return out['virtual_feats'].sum() + logits.sum() * 1e-3, stats


WORKLOADS = {
    'fsd': dict(cls=FSDPath, points=160000, metric='LiDAR frames/sec (FSD hot path fwd+bwd), Waymo geometry',
                name='FSD Waymo hot path: DynamicScatterVFE + SimpleSparseUNet segmentor (0.25 m) -> clustering -> SIR x 3 '
                     '-> point RoI pooling -> SIR x 2'),
    'fsdv2': dict(cls=FSDv2Path, points=300000, metric='LiDAR frames/sec (FSDv2 hot path fwd+bwd), nuScenes 10-sweep geometry',
                  name='FSDv2 nuScenes 10-sweep hot path: segmentor U-Net at 0.2 m [40,512,512] -> virtual points -> '
                       'DynamicScatterVFE at 0.4 m -> VirtualVoxelMixer [20,256,256]'),
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
    seg = None
    timed = [(t, b) for t, b in ((ke.elapsed_time(ke), b) for ke, b in seg_records) if t > 0]
    if timed:
        ms = sum(t for t, _ in timed)
        by = sum(b for _, b in timed)
        seg = {'bound': 'hbm', 'kernel': f'seg_reduce_fwd_* ({len(timed)} segmented reductions of one forward pass; HIP events bound '
                                       'to each launch)',
               'achieved': round(by / (ms * 1e-3) / 1e9, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
               'frac': round(by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), 'algorithmic_bytes': by, 'ms': round(ms, 3)}
    return conv, seg


def run(args, rank, world, dev, allreduce_grads):
    """bench.py's contract for --workload fsd | fsdv2: W warm-up steps, K timed steps between barriers, max over ranks,
    one JSON line from rank 0."""
    spec = WORKLOADS[args.workload]
    torch.manual_seed(0)
    model = spec['cls']().to(dev).train()
    params = [p for p in model.parameters() if p.requires_grad]
    n_pts = args.points if args.points_given else spec['points']
    clouds = [model.make_cloud(n_pts, 1000 * rank + i, dev) for i in range(args.frames_per_gpu)]
    stats = {}

    def step():
        for p in params:
            p.grad = None
        loss, st = model(clouds)
        stats.update(st)
        loss.backward()
        if world > 1:
            allreduce_grads([p for p in params if p.grad is not None], world)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    conv, seg = _conv_roofline(model, clouds)
    if rank == 0:
        frames = world * args.frames_per_gpu * args.steps
        res = {'metric': spec['metric'], 'value': round(frames / elapsed, 3), 'unit': 'frames/s', 'n_gpus': world,
               'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 3),
               'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
               'config': {'workload': spec['name'] + f'; {n_pts} points/frame (ground plane + boxes), fwd+bwd',
                          'frames_per_gpu': args.frames_per_gpu, 'points_per_frame': n_pts, 'parallelism': f'dp{world}',
                          'sizes': {k: int(v) for k, v in stats.items()},
                          'stand_ins': 'segmentation / box heads = linear layers, foreground = geometric rule '
                                       '(detector glue is out of scope)'},
               'roofline': conv if conv is not None else seg, 'roofline_seg_reduce': seg, 'cpu_baseline': None}
        print(json.dumps(res))
