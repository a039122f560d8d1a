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


@pytest.mark.parametrize('name,points', [('fsd', 40000), ('fsdv2', 40000)])
def test_index_phase_built_ahead_gives_the_same_step(name, points):
    """prepare_index_phase (voxelisation, voxel grouping, every rulebook of the segmentor U-Net built without features, for a
    batch that is not being computed yet) + forward(prepared=...) against the plain forward: same loss, same gradients, bit for
    bit - and the prepared forward builds no rulebook of the segmentor itself."""
    import bench_workloads as W
    from sst_amd import spconv
    torch.manual_seed(0)
    model = W.WORKLOADS[name]['cls']().to(DEV).train()
    clouds = [model.make_cloud(points, 3 + i, DEV) for i in range(2)]

    def run(prepared):
        for p in model.parameters():
            p.grad = None
        loss, _ = model(clouds, prepared=prepared)
        loss.backward()
        return loss.detach().clone(), [p.grad.clone() for p in model.parameters() if p.grad is not None]
    loss_a, grads_a = run(None)
    prepared = model.prepare(clouds)
    assert prepared is not None and len(prepared['indice_dict']) >= model.seg_backbone.stage_num      # one submanifold + one strided rulebook per level
    built = []
    orig = spconv.get_indice_pairs
    spconv.get_indice_pairs = lambda *a, **k: built.append(a[2]) or orig(*a, **k)       # a[2]: the spatial shape of the call
    try:
        loss_b, grads_b = run(prepared)
    finally:
        spconv.get_indice_pairs = orig
    seg_shape = list(model.seg_backbone.sparse_shape)
    assert all(list(sh) != seg_shape for sh in built), 'the segmentor built a rulebook although they were handed over'
    assert torch.equal(loss_a, loss_b) and len(grads_a) == len(grads_b)
    for a, b in zip(grads_a, grads_b):
        assert torch.equal(a, b)
