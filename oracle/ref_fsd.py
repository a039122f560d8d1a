"""TEST INFRASTRUCTURE ONLY, build container only — the REFERENCE'S OWN modules as a provider namespace for
`bench_workloads.FSDPath / FSDv2Path`: the chain the reference wires in VoteSegmentor.extract_feat ->
Voxel2PointScatterNeck -> ClusterAssigner -> SingleStageFSD.extract_feat (detectors/single_stage_fsd.py:228-250,
467-483, 922-999) and SingleStageFSDV2.extract_feat (single_stage_fsd_v2.py:159-271), executed from /root/reference
through oracle/ref_loader (unmodified Python; natives as described there).  Produces tests/golden/fsd_chain.npz and
fsdv2_chain.npz (tests/golden/make_golden.py) and pins oracle/fsd_cpu.py (tests/test_fsd_chain.py)."""
import types
from functools import partial

import torch
import torch.nn as nn

from . import build_ref, ref_loader

_FSD = 'mmdet3d/models/detectors/single_stage_fsd.py'
_FSD2 = 'mmdet3d/models/detectors/single_stage_fsd_v2.py'


def _multi_apply(func, *args, **kwargs):
    """mmdet.core.multi_apply (mmdet 2.14, third party, absent here): map a function over argument lists, transpose"""
    results = map(partial(func, **kwargs) if kwargs else func, *args)
    return tuple(map(list, zip(*results)))


def reference_ops():
    R = ref_loader.load_reference_spconv()
    ref = R.base
    from scipy.sparse.csgraph import connected_components
    glb = {'scatter_v2': ref.sst_ops.scatter_v2, 'connected_components': connected_components, 'multi_apply': _multi_apply}
    for fn in ('filter_almost_empty', 'find_connected_componets', 'find_connected_componets_single_batch',
               'modify_cluster_by_class'):
        glb[fn] = ref_loader.load_reference_function(_FSD, fn, glb)
    cluster_cls = ref_loader.load_reference_class(_FSD, 'ClusterAssigner', glb)
    voxel_ext = build_ref.load()

    def voxelize(points_list, voxel_size, point_cloud_range):
        coors = []
        for b, p in enumerate(points_list):
            c = torch.zeros((p.size(0), 3), dtype=torch.int32)
            voxel_ext.dynamic_voxelize(p.contiguous(), c, list(voxel_size), list(point_cloud_range), 3)
            coors.append(torch.nn.functional.pad(c, (1, 0), value=b))
        return torch.cat(points_list), torch.cat(coors).long()

    glb2 = {'scatter_v2': ref.sst_ops.scatter_v2}
    extract = ref_loader.load_reference_method(_FSD2, 'SingleStageFSDV2', 'extract_feat', glb2)
    vox_with_batch = ref_loader.load_reference_method(_FSD2, 'SingleStageFSDV2', 'voxelize_with_batch_idx', glb2)
    clip = ref_loader.load_reference_method(_FSD2, 'SingleStageFSDV2', 'clip_points', glb2)
    fusion = ref_loader.load_reference_method(_FSD2, 'SingleStageFSDV2', 'multiscale_fusion', glb2)
    coors_proj = ref_loader.load_reference_method(_FSD2, 'SingleStageFSDV2', 'ms_coors_proj', glb2)
    recover = ref_loader.load_reference_method(_FSD2, 'SingleStageFSDV2', 'recover_point_features', glb2)

    class VirtualVoxelExtractor(nn.Module):
        """the attributes SingleStageFSDV2.__init__ sets up (single_stage_fsd_v2.py:60-105) around its own extract_feat"""

        def __init__(self, backbone, voxel_encoder, virtual_point_projector, train_cfg=None, test_cfg=None,
                     multiscale_cfg=None, bbox_head=None, as_rpn=None):
            super().__init__()
            vpp = virtual_point_projector
            self.baseline_mode, self.train_cfg, self.print_info = False, {}, {}
            self.as_rpn = bool((bbox_head or {}).get('as_rpn', False)) if as_rpn is None else bool(as_rpn)
            if self.as_rpn:                                                  # single_stage_fsd_v2.py:92-93
                self.recover_proj = ref.sst_ops.build_mlp(vpp['recover_in_channels'], vpp['recover_hidden_dims'], vpp['norm_cfg'])
            self.multiscale_cfg = multiscale_cfg
            if multiscale_cfg is not None:                                   # :99-105
                self.ms_projectors = nn.ModuleList([ref.sst_ops.build_mlp(p[0], p[1:], multiscale_cfg['norm_cfg'])
                                                    for p in multiscale_cfg['projector_hiddens']])
            self.zero_virtual_feature, self.only_virtual = vpp.get('zero_virtual_feature', False), vpp.get('only_virtual', False)
            self.virtual_voxel_size, self.point_cloud_range = voxel_encoder['voxel_size'], voxel_encoder['point_cloud_range']
            self.virtual_proj = ref.sst_ops.build_mlp(vpp['in_channels'], vpp['hidden_dims'], vpp['norm_cfg'])
            self.ori_proj = ref.sst_ops.build_mlp(vpp['ori_in_channels'], vpp['ori_hidden_dims'], vpp['norm_cfg'])
            ve = dict(voxel_encoder)
            ve.pop('type')
            self.voxel_encoder = ref.voxel_encoder.DynamicScatterVFE(**ve)
            bb = dict(backbone)
            bb.pop('type')
            self.backbone = R.sparse_unet.VirtualVoxelMixer(**bb)
            self.voxelize_with_batch_idx = types.MethodType(vox_with_batch, self)
            self.clip_points = types.MethodType(clip, self)
            self.multiscale_fusion = types.MethodType(fusion, self)
            self.ms_coors_proj = types.MethodType(coors_proj, self)
            self.recover_point_features = types.MethodType(recover, self)

        def forward(self, sampled_dict, origin_dict, gt_bboxes_3d=None, multiscale_features=None):
            return extract(self, sampled_dict, origin_dict, gt_bboxes_3d, multiscale_features)

    return types.SimpleNamespace(
        name='reference', voxelize=voxelize, scatter_v2=ref.sst_ops.scatter_v2,
        DynamicScatterVFE=ref.voxel_encoder.DynamicScatterVFE,
        PseudoMiddleEncoderForSpconvFSD=ref.input_layer_v2.PseudoMiddleEncoderForSpconvFSD,
        SimpleSparseUNet=R.sparse_unet.SimpleSparseUNet, ClusterAssigner=cluster_cls, SIR=ref.sir.SIR,
        VirtualVoxelExtractor=VirtualVoxelExtractor, DynamicPointROIExtractor=None)
