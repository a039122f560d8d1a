"""CPU: host-side logic of sst_amd.spconv / sparse_unet that needs no GPU -- output-shape formulas against the oracle,
module construction through the registries, parameter layouts, SparseSequential plumbing, loud failure on CPU
tensors (there is no CPU path)."""
import numpy as np
import pytest
import torch

import sst_amd
from sst_amd import spconv


@pytest.mark.parametrize('shape,ks,st,pd,dl', [([41, 1504, 1504], [3, 3, 3], [2, 2, 2], [1, 1, 1], [1, 1, 1]),
                                               ([32, 640, 640], [3, 3, 3], [2, 2, 2], [0, 1, 1], [1, 1, 1]),
                                               ([9, 33, 31], [3, 1, 3], [2, 1, 2], [0, 0, 1], [1, 1, 1]),
                                               ([10, 30, 30], [3, 3, 3], [1, 1, 1], [2, 2, 2], [2, 2, 2])])
def test_output_shape_formulas_match_oracle(shape, ks, st, pd, dl):
    from oracle import spconv_oracle as O
    assert spconv.get_conv_output_size(shape, ks, st, pd, dl) == O.conv_output_size(shape, ks, st, pd, dl)
    assert spconv.get_deconv_output_size(shape, ks, st, pd, dl, [0, 0, 0]) == \
        O.deconv_output_size(shape, ks, st, pd, dl, [0, 0, 0])


def test_layers_build_through_the_registries_with_reference_parameter_layout():
    conv = sst_amd.build_conv_layer(dict(type='SubMConv3d', indice_key='subm1'), 16, 32, 3, padding=1, bias=False)
    assert isinstance(conv, spconv.SubMConv3d) and conv.weight.shape == (3, 3, 3, 16, 32) and conv.bias is None
    assert conv.subm and conv.indice_key == 'subm1' and conv.padding == [1, 1, 1]
    down = sst_amd.build_conv_layer(dict(type='SparseConv3d', indice_key='spconv2'), 16, 32, 3, stride=2, padding=(0, 1, 1))
    assert down.stride == [2, 2, 2] and down.padding == [0, 1, 1] and down.bias.shape == (32,)
    inv = sst_amd.build_conv_layer(dict(type='SparseInverseConv3d', indice_key='spconv2'), 32, 16, 3, bias=False)
    assert inv.inverse and not inv.subm and inv.weight.shape == (3, 3, 3, 32, 16)
    one = spconv.SubMConv3d(4, 8, 1)
    assert one.conv1x1
    with pytest.raises(AssertionError):
        spconv.SparseConv3d(4, 8, 3, stride=2, dilation=2)     # "don't support this." (conv.py:82-83)
    block = sst_amd.make_sparse_convmodule(16, 32, 3, 'k', stride=2, padding=1, conv_type='SparseConv3d',
                                           norm_cfg=dict(type='BN1d', eps=1e-3, momentum=0.01))
    assert [type(m).__name__ for m in block] == ['SparseConv3d', 'BatchNorm1d', 'ReLU']
    pre = sst_amd.make_sparse_convmodule(16, 32, 3, 'k', conv_type='SubMConv3d', order=('conv', ),
                                         norm_cfg=dict(type='BN1d'))
    assert len(pre) == 1


def test_sparse_tensor_and_sequential_plumbing_on_cpu():
    ind = torch.tensor([[0, 1, 2, 3], [1, 0, 0, 1]], dtype=torch.int64)
    t = spconv.SparseConvTensor(torch.tensor([[1.0, -2.0], [3.0, 4.0]]), ind, [2, 3, 4], 2)
    assert t.indices.dtype == torch.int32 and t.spatial_size == 24 and abs(t.sparity - 2 / 48) < 1e-12
    d = t.dense()
    assert d.shape == (2, 2, 2, 3, 4) and d[0, :, 1, 2, 3].tolist() == [1.0, -2.0] and float(d.abs().sum()) == 10.0
    t2 = t.replace_feature(t.features * 2)
    assert t2.indice_dict is t.indice_dict and torch.equal(t2.indices, t.indices) and t2.features[1, 1] == 8.0
    seq = spconv.SparseSequential(torch.nn.BatchNorm1d(2), torch.nn.ReLU(), torch.nn.Linear(2, 3))
    out = seq(t)                                   # BatchNorm1d + ReLU take the fused route only on CUDA tensors
    assert out.features.shape == (2, 3) and len(seq) == 3 and isinstance(seq[-1], torch.nn.Linear)
    empty = spconv.SparseConvTensor(torch.zeros(0, 2), torch.zeros(0, 4, dtype=torch.int32), [2, 3, 4], 1)
    assert seq(empty).features.shape == (0, 2)     # modules are skipped on an empty tensor (modules.py:128-131)


def test_spconv_ops_fail_loudly_on_cpu_tensors():
    ind = torch.tensor([[0, 1, 2, 3]], dtype=torch.int32)
    with pytest.raises(RuntimeError):
        spconv.get_indice_pairs(ind, 1, [2, 3, 4], 3, subm=True)
    conv = spconv.SubMConv3d(2, 2, 3, padding=1)
    with pytest.raises(RuntimeError):
        conv(spconv.SparseConvTensor(torch.zeros(1, 2), ind, [2, 3, 4], 1))
    with pytest.raises(RuntimeError):
        sst_amd.dynamic_point_pool(torch.zeros(1, 7), torch.zeros(4, 3), [0, 0, 0], 8, 16)
    with pytest.raises(NotImplementedError):
        spconv.get_indice_pairs(torch.zeros(1, 3, dtype=torch.int32), 1, [3, 4], 3)   # 2-D indices


def test_fsd_config_backbone_parameter_count():
    net = sst_amd.BACKBONES.build(dict(
        type='SimpleSparseUNet', in_channels=64, sparse_shape=[32, 640, 640], order=('conv', 'norm', 'act'),
        norm_cfg=dict(type='naiveSyncBN1d', eps=1e-3, momentum=0.01), base_channels=64, output_channels=128,
        encoder_channels=((64, ), (64, 64, 64), (64, 64, 64), (128, 128, 128), (256, 256, 256)),
        encoder_paddings=((1, ), (1, 1, 1), (1, 1, 1), ((0, 1, 1), 1, 1), (1, 1, 1)),
        decoder_channels=((256, 256, 128), (128, 128, 64), (64, 64, 64), (64, 64, 64), (64, 64, 64)),
        decoder_paddings=((1, 1), (1, 0), (1, 0), (0, 0), (0, 1))))
    assert net.conv_out is None and net.stage_num == 5
    n_conv = sum(1 for m in net.modules() if isinstance(m, spconv.SparseConvolution))
    assert n_conv == 1 + 13 + 5 * 2 + 5 + 5      # stem, encoder, lateral blocks (2 convs each), merge, upsample
    keys = [k for k, m in net.named_modules() if isinstance(m, spconv.SparseConvolution)]
    assert {m.indice_key for m in net.modules() if isinstance(m, spconv.SparseConvolution)} == \
        {'subm1', 'subm2', 'subm3', 'subm4', 'subm5', 'spconv2', 'spconv3', 'spconv4', 'spconv5'}
    assert 'upsample_layer5.0' in keys and isinstance(net.upsample_layer5[0], spconv.SparseInverseConv3d)
    assert sum(p.numel() for p in net.parameters()) == 18034048
