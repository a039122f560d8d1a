"""GPU: the encoder layer as ONE library call per direction (csrc/layer_exec.hip, sst_encoder_layer_{fwd,bwd}_f32x6) against the
same launch sequence issued from Python (sst_amd/sst_basic_block.py FusedEncoderLayerFn): same kernels in the same order, so the
outputs and every gradient must agree BIT FOR BIT - through the whole pipeline, with and without gradients, GELU and ReLU."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _run(model, frames, exec_on, grad=True):
    from sst_amd import sst_basic_block as B
    B._LAYER_EXEC = 1 if exec_on else 0
    try:
        torch.manual_seed(11)                    # voxel shuffle / drop
        for p in model.parameters():
            p.grad = None
        if not grad:
            with torch.no_grad():
                return model(frames).clone(), None
        out = model(frames)
        gen = torch.Generator(device=DEV).manual_seed(3)
        out.backward(torch.randn(out.shape, device=DEV, generator=gen))
        return out.detach().clone(), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    finally:
        B._LAYER_EXEC = 1


@pytest.mark.parametrize('n_points,blocks', [(20000, 2), (116000, 1), (9000, 1)])
def test_layer_executor_equals_the_python_sequence(n_points, blocks):
    import bench
    from sst_amd import sst_basic_block as B
    torch.manual_seed(0)
    model = bench.Pipeline(blocks).to(DEV).train()
    model.backbone.set_precision('f32x6')
    frames = [bench.make_cloud(n_points, 5, DEV)]
    calls = []
    orig = B._layer_exec_fwd

    def counted(*a, **k):
        calls.append(1)
        return orig(*a, **k)
    B._layer_exec_fwd = counted
    try:
        out_e, grads_e = _run(model, frames, True)
    finally:
        B._layer_exec_fwd = orig
    assert len(calls) == 2 * blocks, 'the executor did not take the layers'
    out_p, grads_p = _run(model, frames, False)
    assert torch.equal(out_e, out_p)
    assert grads_e.keys() == grads_p.keys() and len(grads_e) > 10
    for n in grads_e:
        assert torch.equal(grads_e[n], grads_p[n]), n
    out_ne, _ = _run(model, frames, True, grad=False)
    out_np, _ = _run(model, frames, False, grad=False)
    assert torch.equal(out_ne, out_np) and torch.equal(out_ne, out_e)


def test_layer_executor_with_relu_and_two_frames():
    import bench
    torch.manual_seed(1)
    model = bench.Pipeline(2).to(DEV).train()
    model.backbone.set_precision('f32x6')
    for blk in model.backbone.block_list:
        for enc in blk.encoder_list:
            enc.act_name = 'relu'
            enc.activation = torch.nn.functional.relu
    frames = [bench.make_cloud(30000, 5, DEV), bench.make_lidar_cloud(6, DEV, beams=16, azimuth_steps=600)]
    out_e, grads_e = _run(model, frames, True)
    out_p, grads_p = _run(model, frames, False)
    assert torch.equal(out_e, out_p)
    for n in grads_e:
        assert torch.equal(grads_e[n], grads_p[n]), n


def test_other_modes_keep_the_python_sequence():
    """fp32 matrix pipe ('f32') and the three-product split: the executor is for the exact-split mode only"""
    import bench
    from sst_amd import sst_basic_block as B
    torch.manual_seed(2)
    model = bench.Pipeline(1).to(DEV).train()
    frames = [bench.make_cloud(20000, 5, DEV)]
    calls = []
    orig = B._layer_exec_fwd
    B._layer_exec_fwd = lambda *a, **k: calls.append(1) or orig(*a, **k)
    try:
        for mode in ('fp32', 'f32x3'):
            model.backbone.set_precision(mode)
            out = model(frames)
            out.sum().backward()
    finally:
        B._layer_exec_fwd = orig
        model.backbone.set_precision('fp32')
    assert not calls
