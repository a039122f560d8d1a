"""GPU: the FSD / FSDv2 hot-path pipelines of bench_workloads.py (BASELINE.json configs[3], [4]) run forward + backward on a
small cloud - an integration test of the modules wired the way the reference's detectors wire them (voxelize ->
DynamicScatterVFE -> SimpleSparseUNet -> clustering / virtual voxels -> SIR / VirtualVoxelMixer)."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


@pytest.mark.parametrize('name,points', [('fsd', 40000), ('fsdv2', 40000)])
def test_workload_pipeline_forward_backward(name, points):
    import bench_workloads as W
    torch.manual_seed(0)
    model = W.WORKLOADS[name]['cls']().to(DEV).train()
    clouds = [model.make_cloud(points, 3 + i, DEV) for i in range(2)]          # two frames: the batched paths
    loss, stats = model(clouds)
    assert torch.isfinite(loss)
    assert stats['points'] == 2 * points and stats['voxels'] > 0 and stats['fg_points'] > 0
    loss.backward()
    grads = [p.grad for p in model.parameters() if p.requires_grad and p.grad is not None]
    assert len(grads) > 20 and all(torch.isfinite(g).all() for g in grads)
    conv, seg = W._conv_roofline(model, clouds)                                 # the instrumented pass of bench.py
    assert conv is not None and conv['algorithmic_flops'] > 0 and seg is not None and seg['achieved'] > 0
