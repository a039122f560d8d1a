"""Host-side behaviour of the module mirrors added in round 3 that does not need a GPU: the torch compositions the fused
passes stand for are what CPU tensors get (LayerNorm + activation, batch norm + identity + ReLU, the MLP stages of
build_mlp), the parameter layout of the reference is kept, constants are uploaded once."""
import torch
import torch.nn.functional as F


def test_mlp_stage_keeps_the_reference_layout_and_composition_on_cpu():
    from sst_amd.sst_ops import build_mlp
    torch.manual_seed(0)
    mlp = build_mlp(3, [16, 32, 13], dict(type='LN', eps=1e-3), act='gelu')
    keys = list(mlp.state_dict())
    assert keys == ['0.0.weight', '0.1.weight', '0.1.bias', '1.0.weight', '1.1.weight', '1.1.bias', '2.0.weight', '2.1.weight',
                    '2.1.bias']
    x = torch.randn(50, 3)
    t = x
    for stage in mlp:
        t = F.gelu(F.layer_norm(F.linear(t, stage[0].weight), stage[1].normalized_shape, stage[1].weight, stage[1].bias,
                                stage[1].eps))
    assert torch.allclose(mlp(x), t, atol=1e-6)
    head = build_mlp(8, [16, 4], dict(type='LN', eps=1e-3), is_head=True, act='relu')
    assert isinstance(head[1], torch.nn.Linear) and head[1].bias is not None


def test_layer_norm_with_activation_on_cpu_is_the_torch_composition():
    from sst_amd.dense import add_layer_norm
    torch.manual_seed(1)
    norm = torch.nn.LayerNorm(12)
    x, r = torch.randn(9, 12), torch.randn(9, 12)
    for act, fn in ((torch.nn.GELU(), F.gelu), (torch.nn.ReLU(), F.relu), (None, lambda t: t), (torch.nn.Tanh(), torch.tanh)):
        want = fn(F.layer_norm(x + r, (12,), norm.weight, norm.bias, norm.eps))
        assert torch.allclose(add_layer_norm(x, r, norm, act=act), want, atol=1e-6)


def test_batch_norm_with_identity_branch_on_cpu_is_the_torch_composition():
    from sst_amd.norm import BatchNorm1d, batch_norm_act
    torch.manual_seed(2)
    bn, ref = BatchNorm1d(8).train(), torch.nn.BatchNorm1d(8).train()
    x, res = torch.randn(40, 8), torch.randn(40, 8)
    got = batch_norm_act(bn, x, relu=True, residual=res)
    want = F.relu(ref(x) + res)
    assert torch.allclose(got, want, atol=1e-6)
    assert int(bn.num_batches_tracked) == int(ref.num_batches_tracked) == 1
    assert torch.allclose(bn.running_mean, ref.running_mean, atol=1e-6)


def test_constants_are_cached_per_value_device_and_dtype():
    from sst_amd import kernels as K
    a = K.const_tensor([0.25, 0.25, 0.2], torch.device('cpu'))
    b = K.const_tensor((0.25, 0.25, 0.2), 'cpu')
    c = K.const_tensor([0.25, 0.25, 0.2], torch.device('cpu'), torch.float64)
    assert a is b and a is not c and c.dtype == torch.float64
    assert a.tolist() == [0.25, 0.25, 0.20000000298023224]


def test_conv_precision_switch_rejects_unknown_modes():
    import pytest
    from sst_amd import spconv
    assert spconv.conv_precision() == spconv.DEFAULT_CONV_PRECISION == 'f32x6'
    spconv.set_conv_precision('f32x3')
    assert spconv.conv_precision() == 'f32x3'
    spconv.set_conv_precision('f32')
    assert spconv.conv_precision() == 'f32'
    spconv.set_conv_precision(spconv.DEFAULT_CONV_PRECISION)
    with pytest.raises(ValueError):
        spconv.set_conv_precision('bf16')


def test_half_keeps_the_fp32_master_weights():
    """the fp16 contract since round 6 (VERDICT round 5 missing 2): `model.half()` - what mmcv's wrap_fp16_model does to a model
    whose config carries `fp16 = dict(loss_scale=32.0)` - is remembered by the entry modules and NOT applied: their parameters
    are the fp32 master weights, the encoder stack switches to its bf16 mode per call (tests/test_gpu_fp16_key.py runs it);
    any other conversion applies as usual"""
    import sst_amd
    bb = sst_amd.build_backbone(dict(type='SSTv2', d_model=[128], nhead=[8], num_blocks=1, dim_feedforward=[256],
                                     output_shape=[468, 468], num_attached_conv=0, to_bev=False, debug=False))
    assert not bb.half_requested
    assert bb.half() is bb and bb.half_requested
    assert all(p.dtype == torch.float32 for p in bb.parameters())
    from sst_amd import _lib
    assert _lib.wants_half(bb)
    vfe = sst_amd.build_voxel_encoder(dict(
        type='DynamicVFE', in_channels=3, feat_channels=[64, 128], with_cluster_center=True, with_voxel_center=True,
        voxel_size=(0.32, 0.32, 6), point_cloud_range=[-74.88, -74.88, -2, 74.88, 74.88, 4],
        norm_cfg=dict(type='naiveSyncBN1d', eps=1e-3, momentum=0.01))).half()
    assert all(p.dtype == torch.float32 for p in vfe.parameters()) and all(b.dtype != torch.float16 for b in vfe.buffers())
    sir = sst_amd.build_backbone(dict(type='SIR', num_blocks=1, in_channels=[8], feat_channels=[[16, 16]], rel_mlp_hidden_dims=[[8]],
                                      norm_cfg=dict(type='LN', eps=1e-3), act='gelu')).half()
    assert all(p.dtype == torch.float32 for p in sir.parameters())
    # a parent that is not one of ours: its own parameters convert, the entry modules below it keep theirs
    parent = torch.nn.Sequential(torch.nn.Linear(4, 4), bb).half()
    assert parent[0].weight.dtype == torch.float16 and all(p.dtype == torch.float32 for p in parent[1].parameters())
    # other conversions apply
    assert all(p.dtype == torch.float64 for p in bb.double().parameters())
    assert all(p.dtype == torch.float32 for p in bb.float().parameters())
    # half tensors are cast at the entry modules (force_fp32), not refused
    assert _lib.as_fp32(torch.zeros(2, dtype=torch.float16)).dtype == torch.float32
    x = torch.zeros(2)
    assert _lib.as_fp32(x) is x


def test_round6_host_logic_without_a_gpu():
    """host side of the round-6 additions: the stack-level weight-image helper declines CPU layers (the per-layer call then packs
    for itself), batch-norm encoder layers are recognised as chain material by shape alone, and the fp16-flag watch list of
    _lib.wants_half follows flags that are set after its first use"""
    import sst_amd
    from sst_amd import _lib
    from sst_amd import sst_basic_block as B
    net = sst_amd.build_backbone(dict(type='SSTv2', d_model=[128], nhead=[8], num_blocks=1, dim_feedforward=[256],
                                      output_shape=[468, 468], num_attached_conv=0, to_bev=False, debug=False,
                                      layer_cfg=dict(use_bn=True)))
    layers = [enc for block in net.block_list for enc in block.encoder_list]
    assert all(enc.bn_modules() is not None and B._bn_layer_ok(enc.norm1, 128) and B._bn_layer_ok(enc.norm2, 128) for enc in layers)
    assert not B._bn_layer_ok(layers[0].norm1, 64) and not B._bn_layer_ok(torch.nn.LayerNorm(128), 128)
    assert B.stack_tail_images(layers, torch.zeros(1)) == [None, None]          # batch-norm layers: never the one-kernel tail
    ln = sst_amd.build_backbone(dict(type='SSTv2', d_model=[128], nhead=[8], num_blocks=1, dim_feedforward=[256],
                                     output_shape=[468, 468], num_attached_conv=0, to_bev=False, debug=False))
    ln_layers = [enc for block in ln.block_list for enc in block.encoder_list]
    assert ln_layers[0].bn_modules() is None
    assert B.stack_tail_images(ln_layers, torch.zeros(1)) == [None, None]       # CPU parameters: nothing to launch
    assert not _lib.wants_half(ln)
    assert '_sst_fp16_watch' in ln.__dict__ and len(ln.__dict__['_sst_fp16_watch']) >= 2
    ln_layers[1].fp16_enabled = True                                            # what mmcv's wrap_fp16_model does, later
    assert _lib.wants_half(ln)
    ln_layers[1].fp16_enabled = False
    assert not _lib.wants_half(ln)
    assert all(k.count('_sst_fp16_watch') == 0 for k in ln.state_dict())
