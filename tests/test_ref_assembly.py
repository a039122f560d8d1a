"""CPU, build container only (skipped where /root/reference is absent): the CPU port of the reference data flow -
oracle/cpu_pipeline.CpuSSTBackbone, which is bench.py's `cpu_baseline` and the checker behind its `parity` block - against
THE REFERENCE ITSELF wired as an assembly, on the same clouds with the same weights:

    voxel_layer.dynamic_voxelize (the reference's C++, compiled into oracle/_ref)      ops/voxel/src/voxelization_cpu.cpp
    -> DynamicVFE            (the reference's Python, unmodified, under oracle/ref_loader's stubs)   voxel_encoder.py:92-298
    -> SSTInputLayerV2                                                                  sst_input_layer_v2.py:40-318
    -> SSTv2 (BasicShiftBlockV2 x blocks, nn.MultiheadAttention on padded windows)      sst_v2.py:16-154

i.e. DynamicVoxelNet.extract_feat (detectors/dynamic_voxelnet.py:38-47).  The pieces of the port are pinned one by one in
tests/test_oracle.py; this pins the port as a whole ("parity is against the builder's port").
Not reference code: DynamicScatter (GPU-only in the reference, voxelization.h:106) is the oracle's restatement on both
sides, TorchEx's in-window rank is the stable rank on both sides (see oracle/ref_loader.py)."""
import types

import numpy as np
import pytest
import torch

from oracle import build_ref, ref_loader

pytestmark = pytest.mark.skipif(not ref_loader.available(), reason='needs the reference tree (build container)')

VOXEL_SIZE = (0.32, 0.32, 6)
PC_RANGE = [-74.88, -74.88, -2, 74.88, 74.88, 4]
DROP_TRAIN = {0: {'max_tokens': 30, 'drop_range': (0, 30)}, 1: {'max_tokens': 60, 'drop_range': (30, 60)},
              2: {'max_tokens': 100, 'drop_range': (60, 100000)}}
DROP_TEST = {0: {'max_tokens': 30, 'drop_range': (0, 30)}, 1: {'max_tokens': 60, 'drop_range': (30, 60)},
             2: {'max_tokens': 100, 'drop_range': (60, 100)}, 3: {'max_tokens': 144, 'drop_range': (100, 100000)}}


def uniform_cloud(n, seed):
    """SURVEY.md §8(d) U-cloud (= bench.make_cloud)"""
    g = torch.Generator().manual_seed(seed)
    return torch.rand(n, 3, generator=g) * torch.tensor([149.76, 149.76, 6.0]) + torch.tensor([-74.88, -74.88, -2.0])


def crowded_cloud(n, seed, centre):
    """points concentrated on a 20 m x 20 m patch: windows beyond 30 / 60 / 100 tokens, voxels dropped in training"""
    g = torch.Generator().manual_seed(seed)
    return torch.rand(n, 3, generator=g) * torch.tensor([20.0, 20.0, 6.0]) + torch.tensor([centre[0], centre[1], -2.0])


class ReferenceAssembly(torch.nn.Module):
    """DynamicVoxelNet.extract_feat of the reference without the detector shell"""

    def __init__(self, num_blocks, seed=0):
        super().__init__()
        ref = ref_loader.load_reference()
        self.voxel_ext = build_ref.load()
        torch.manual_seed(seed)
        self.voxel_encoder = ref.voxel_encoder.DynamicVFE(
            in_channels=3, feat_channels=[64, 128], with_distance=False, voxel_size=VOXEL_SIZE, with_cluster_center=True,
            with_voxel_center=True, point_cloud_range=PC_RANGE, norm_cfg=dict(type='naiveSyncBN1d', eps=1e-3, momentum=0.01))
        self.middle_encoder = ref.input_layer_v2.SSTInputLayerV2(
            drop_info=(DROP_TRAIN, DROP_TEST), window_shape=(12, 12, 1), sparse_shape=(468, 468, 1), shuffle_voxels=False,
            debug=False, mute=True)
        self.backbone = ref.sst_v2.SSTv2(
            d_model=[128] * num_blocks, nhead=[8] * num_blocks, num_blocks=num_blocks, dim_feedforward=[256] * num_blocks,
            output_shape=[468, 468], num_attached_conv=0, to_bev=False, debug=False)
        g = torch.Generator().manual_seed(seed + 1)
        with torch.no_grad():   # biases start at zero: randomise them so that a dropped bias would show
            for p in self.backbone.parameters():
                if p.dim() == 1:
                    p.add_(torch.randn(p.shape, generator=g) * 0.1)

    def voxelize(self, points_list):
        """DynamicVoxelNet.voxelize, dynamic_voxelnet.py:49-71"""
        coors = []
        for b, p in enumerate(points_list):
            c = torch.zeros((p.size(0), 3), dtype=torch.int32)
            self.voxel_ext.dynamic_voxelize(p.contiguous(), c, list(VOXEL_SIZE), PC_RANGE, 3)
            coors.append(torch.nn.functional.pad(c, (1, 0), value=b))
        return torch.cat(points_list), torch.cat(coors)

    def forward(self, points_list):
        points, coors = self.voxelize(points_list)
        voxel_feats, voxel_coors = self.voxel_encoder(points, coors)
        info = self.middle_encoder(voxel_feats, voxel_coors, len(points_list))
        out = self.backbone(info)[0]
        return out['voxel_feats'], out['voxel_coors']


def build_pair(num_blocks, train=True):
    from oracle.cpu_pipeline import CpuSSTBackbone, load_pipeline_weights
    ref = ReferenceAssembly(num_blocks).train(train)
    port = CpuSSTBackbone(VOXEL_SIZE, PC_RANGE, DROP_TRAIN if train else DROP_TEST, num_blocks=num_blocks).train(train)
    load_pipeline_weights(port, types.SimpleNamespace(voxel_encoder=ref.voxel_encoder, backbone=ref.backbone))
    return ref, port


@pytest.mark.parametrize('case', ['config0_20k', 'two_frames_crowded'])
def test_cpu_port_equals_reference_assembly_forward(case):
    if case == 'config0_20k':        # BASELINE.json configs[0]: 20 000 points, one SRA block
        clouds, blocks = [uniform_cloud(20000, 7)], 1
    else:                            # two frames, crowded: every drop level, voxels dropped (training-mode drop)
        clouds, blocks = [torch.cat([crowded_cloud(9000, 1, (-10.0, -10.0)), uniform_cloud(3000, 2)]),
                          torch.cat([crowded_cloud(6000, 3, (30.0, 5.0)), uniform_cloud(2000, 4)])], 2
    ref, port = build_pair(blocks)
    with torch.no_grad():
        want, want_coors = ref(clouds)
        got = port(clouds)
    assert np.array_equal(port.last_voxel_coors.numpy().astype(np.int64), want_coors.numpy().astype(np.int64)), \
        'kept voxels (and their row order) differ'
    if case != 'config0_20k':
        assert port.last_voxel_coors.size(0) < port.last_all_voxel_coors.size(0), 'the case must drop voxels'
    err = float((got - want).abs().max())
    assert err <= 1e-5, err


def test_cpu_port_equals_reference_assembly_gradients():
    """forward + backward through two blocks: parameter gradients of the first VFE layer, the first and the last encoder
    layer"""
    clouds = [torch.cat([crowded_cloud(5000, 11, (0.0, 0.0)), uniform_cloud(2500, 12)])]
    ref, port = build_pair(2)
    g = torch.Generator().manual_seed(5)
    out_r, _ = ref(clouds)
    up = torch.randn(out_r.shape, generator=g)
    (out_r * up).sum().backward()
    out_p = port(clouds)
    (out_p * up).sum().backward()
    pairs = [(ref.voxel_encoder.vfe_layers[0].linear.weight, port.vfe.linears[0].weight),
             (ref.voxel_encoder.vfe_layers[1].norm.weight, port.vfe.norms[1].weight),
             (ref.backbone.block_list[0].encoder_list[0].win_attn.self_attn.in_proj_weight,
              port.layers[0].self_attn.in_proj_weight),
             (ref.backbone.block_list[1].encoder_list[1].linear2.weight, port.layers[3].linear2.weight),
             (ref.backbone.block_list[1].encoder_list[1].norm2.bias, port.layers[3].norm2.bias)]
    for a, b in pairs:
        scale = max(1.0, float(a.grad.abs().max()))
        assert float((a.grad - b.grad).abs().max()) <= 1e-4 * scale


def test_cpu_port_equals_reference_assembly_eval_mode():
    """inference drop levels (a fourth level of 144 tokens, nothing dropped) and running batch-norm statistics"""
    clouds = [torch.cat([crowded_cloud(7000, 21, (12.0, -40.0)), uniform_cloud(3000, 22)])]
    ref, port = build_pair(1, train=False)
    with torch.no_grad():
        want, want_coors = ref(clouds)
        got = port(clouds)
    assert np.array_equal(port.last_voxel_coors.numpy().astype(np.int64), want_coors.numpy().astype(np.int64))
    assert float((got - want).abs().max()) <= 1e-5
