"""GPU: `fp16 = dict(loss_scale=32.0)` of the shipped config (configs/sst_refactor/sst_waymoD5_1x_3class_8heads_v2.py:82) is
HONOURED, not refused.  The reference's tools/train.py wraps the model with mmcv's wrap_fp16_model - `model.half()` plus
`fp16_enabled = True` on every module that has the attribute (sst_basic_block_v2.py:102-104: the encoder layers' auto_fp16) - and
installs Fp16OptimizerHook (fp32 master weights, loss scale).  Here: the detector built from the untouched config, those two
steps applied the way mmcv applies them, forward + backward with the loss scale:
  * every parameter is still fp32 (they are the master weights: sst_amd._lib.Fp32Master), nothing raises;
  * the encoder stack ran in its reduced-precision (bf16) mode and the features leave as float16, as the reference's half layers
    hand them on: bit-equal to the same model with `backbone.set_precision('bf16')` and no wrapping;
  * the gradients are those of the bf16 mode times the loss scale (finite, fp32)."""
import ast
import contextlib
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _detector(voxel_feats_only=True):
    import bench
    cfg = ast.literal_eval(open(os.path.join(ROOT, 'tests', 'golden', 'configs',
                                             'sst_waymoD5_1x_3class_8heads_v2.model.py')).read())
    torch.manual_seed(0)
    det = bench.Pipeline(model_cfg=cfg, voxel_feats_only=voxel_feats_only).to(DEV).train()
    det.middle_encoder.shuffle_voxels = False
    return det


def _wrap_fp16_model(model):
    """mmcv.runner.fp16_utils.wrap_fp16_model (mmcv 1.3.9): half the model, keep normalisation layers in fp32, set fp16_enabled"""
    model.half()
    for m in model.modules():
        if isinstance(m, (torch.nn.modules.batchnorm._BatchNorm, torch.nn.GroupNorm)):
            m.float()
        if hasattr(m, 'fp16_enabled'):
            m.fp16_enabled = True
    return model


def _step(det, frames, scale):
    for p in det.parameters():
        p.grad = None
    with contextlib.redirect_stdout(sys.stderr):
        out = det(frames)
    gen = torch.Generator(device=DEV).manual_seed(4)
    g = torch.randn(out.shape, device=DEV, generator=gen)
    (out.float() * g).sum().mul(scale).backward()
    return out.detach(), {n: p.grad.detach().clone() for n, p in det.named_parameters() if p.grad is not None}


def test_wrapped_model_runs_the_bf16_mode_with_fp32_master_weights():
    import bench
    frames = [bench.make_cloud(30000, 3, DEV)]
    wrapped = _wrap_fp16_model(_detector())
    assert all(p.dtype == torch.float32 for p in wrapped.parameters()), 'the parameters are the fp32 master weights'
    assert all(b.dtype != torch.float16 for b in wrapped.buffers())
    assert wrapped.backbone.half_requested and wrapped.voxel_encoder.half_requested
    out_w, grads_w = _step(wrapped, frames, 32.0)
    assert out_w.dtype == torch.float16 and torch.isfinite(out_w.float()).all()
    assert wrapped.backbone.precision == 'fp32', 'the mode of the module itself is untouched (the switch is per call)'

    plain = _detector()
    plain.backbone.set_precision('bf16')
    out_p, grads_p = _step(plain, frames, 1.0)
    assert out_p.dtype == torch.float32
    assert torch.equal(out_w.float(), out_p.half().float()), 'fp16 mode == bf16 mode of the stack, features rounded to half'
    assert grads_w.keys() == grads_p.keys() and len(grads_w) > 20
    for n in grads_w:
        assert grads_w[n].dtype == torch.float32 and torch.isfinite(grads_w[n]).all(), n
        sc = max(1e-6, float(grads_p[n].abs().max()))
        # same kernels, the upstream gradient rounded to half once more (x 32: exact in binary): a few 1e-3 of the scale
        assert float((grads_w[n] / 32.0 - grads_p[n]).abs().max()) <= 2e-2 * sc, n


def test_half_inputs_and_the_full_backbone():
    """half point features are cast (force_fp32: voxel_encoder.py:229, dynamic_voxelnet.py:50); with the BEV canvas and the
    attached convolutions the output is float16 too (the neck behind is a half module)"""
    import bench
    det = _wrap_fp16_model(_detector(voxel_feats_only=False))
    frames = [bench.make_cloud(20000, 5, DEV)]
    with contextlib.redirect_stdout(sys.stderr):
        out = det(frames)
    out = out[0] if isinstance(out, (list, tuple)) else out
    assert out.dtype == torch.float16 and out.dim() == 4 and torch.isfinite(out.float()).all()
    out.float().sum().backward()
    assert all(p.grad is None or torch.isfinite(p.grad).all() for p in det.parameters())
    # a conversion that is not to float16 still applies
    det.double()
    assert all(p.dtype == torch.float64 for p in det.parameters())
