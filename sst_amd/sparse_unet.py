"""Sparse-convolution blocks and the U-Net backbone of FSD's segmentor on top of sst_amd.spconv.

Mirrors mmdet3d/ops/sparse_block.py (``replace_feature`` :13-18, ``SparseBasicBlock`` :83-141 -- the reference derives
it from mmdet 2.14's ``resnet.BasicBlock``, whose constructor is restated here --, ``make_sparse_convmodule`` :218-289)
and mmdet3d/models/middle_encoders/sparse_unet.py (``SparseUNet`` :16-321, ``SimpleSparseUNet`` :324-414): same
constructor arguments, sub-module names (hence ``state_dict`` keys: ``conv_input.0.weight``,
``encoder_layers.encoder_layer2.0.0.weight``, ``lateral_layer3.bn1.weight``, ...), forward signature and returned dict.
Only 3-D (``ndim=3``) is built.
"""
import torch
from torch import nn

from .norm import build_conv_layer, build_norm_layer
from .registry import BACKBONES, MIDDLE_ENCODERS
from .spconv import SparseConvTensor, SparseModule, SparseSequential


def replace_feature(out, new_features):
    if 'replace_feature' in out.__dir__():
        return out.replace_feature(new_features)
    out.features = new_features
    return out


def _activation(act_type):
    act_type = act_type.lower()
    if act_type == 'relu':
        return nn.ReLU(inplace=True)
    if act_type == 'gelu':
        return nn.GELU()
    if act_type == 'silu':
        return nn.SiLU(inplace=True)
    raise NotImplementedError


class SparseBasicBlock(SparseModule):
    """two 3x3x3 submanifold convolutions with a residual connection (sparse_block.py:83-141); sub-modules conv1,
    bn1, conv2, bn2, relu, downsample as in mmdet's BasicBlock."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, conv_cfg=None, norm_cfg=None, act_type='relu'):
        super().__init__()
        self.norm1_name, norm1 = build_norm_layer(norm_cfg, planes, postfix=1)
        self.norm2_name, norm2 = build_norm_layer(norm_cfg, planes, postfix=2)
        self.conv1 = build_conv_layer(conv_cfg, inplanes, planes, 3, stride=stride, padding=1, dilation=1, bias=False)
        self.add_module(self.norm1_name, norm1)
        self.conv2 = build_conv_layer(conv_cfg, planes, planes, 3, padding=1, bias=False)
        self.add_module(self.norm2_name, norm2)
        self.relu = _activation(act_type)
        self.downsample = downsample
        self.stride = stride

    @property
    def norm1(self):
        return getattr(self, self.norm1_name)

    @property
    def norm2(self):
        return getattr(self, self.norm2_name)

    def forward(self, x):
        identity = x.features
        assert x.features.dim() == 2, f'x.features.dim()={x.features.dim()}'
        out = self.conv1(x)
        out = replace_feature(out, self.norm1(out.features))
        out = replace_feature(out, self.relu(out.features))
        out = self.conv2(out)
        out = replace_feature(out, self.norm2(out.features))
        if self.downsample is not None:
            identity = self.downsample(x)
        out = replace_feature(out, out.features + identity)
        out = replace_feature(out, self.relu(out.features))
        return out


def make_sparse_convmodule(in_channels, out_channels, kernel_size, indice_key, stride=1, padding=0,
                           conv_type='SubMConv3d', act_type='relu', norm_cfg=None, order=('conv', 'norm', 'act')):
    """sparse_block.py:218-289 -> SparseSequential(conv [, norm] [, act]) in the given order"""
    assert isinstance(order, tuple) and len(order) <= 3
    assert set(order) | {'conv', 'norm', 'act'} == {'conv', 'norm', 'act'}
    conv_cfg = dict(type=conv_type, indice_key=indice_key)
    layers = []
    for layer in order:
        if layer == 'conv':
            if conv_type not in ('SparseInverseConv4d', 'SparseInverseConv3d', 'SparseInverseConv2d',
                                 'SparseInverseConv1d'):
                layers.append(build_conv_layer(conv_cfg, in_channels, out_channels, kernel_size, stride=stride,
                                               padding=padding, bias=False))
            else:
                layers.append(build_conv_layer(conv_cfg, in_channels, out_channels, kernel_size, bias=False))
        elif layer == 'norm':
            layers.append(build_norm_layer(norm_cfg, out_channels)[1])
        elif layer == 'act':
            layers.append(_activation(act_type))
    return SparseSequential(*layers)


@MIDDLE_ENCODERS.register_module()
class SparseUNet(nn.Module):
    """encoder / decoder construction and the decoder step shared by the U-Nets (sparse_unet.py:16-321)."""

    def __init__(self, in_channels, sparse_shape, order=('conv', 'norm', 'act'),
                 norm_cfg=dict(type='BN1d', eps=1e-3, momentum=0.01), base_channels=16, output_channels=128,
                 encoder_channels=((16, ), (32, 32, 32), (64, 64, 64), (64, 64, 64)),
                 encoder_paddings=((1, ), (1, 1, 1), (1, 1, 1), ((0, 1, 1), 1, 1)),
                 decoder_channels=((64, 64, 64), (64, 64, 32), (32, 32, 16), (16, 16, 16)),
                 decoder_paddings=((1, 0), (1, 0), (0, 0), (0, 1)), ndim=3, act_type='relu', init_cfg=None):
        super().__init__()
        if ndim != 3:
            raise NotImplementedError('sst_amd.SparseUNet: ndim=3 only')
        self.init_cfg = init_cfg
        self.sparse_shape = sparse_shape
        self.in_channels = in_channels
        self.order = order
        self.base_channels = base_channels
        self.output_channels = output_channels
        self.encoder_channels = encoder_channels
        self.encoder_paddings = encoder_paddings
        self.decoder_channels = decoder_channels
        self.decoder_paddings = decoder_paddings
        self.stage_num = len(self.encoder_channels)
        self.ndim = ndim
        self.is_3d = ndim == 3
        self.fp16_enabled = False
        self.act_type = act_type
        assert isinstance(order, tuple) and len(order) == 3
        assert set(order) == {'conv', 'norm', 'act'}
        if self.order[0] != 'conv':  # pre activate
            self.conv_input = make_sparse_convmodule(in_channels, self.base_channels, 3, norm_cfg=norm_cfg, padding=1,
                                                     indice_key='subm1', conv_type=f'SubMConv{self.ndim}d',
                                                     order=('conv', ), act_type=act_type)
        else:
            self.conv_input = make_sparse_convmodule(in_channels, self.base_channels, 3, norm_cfg=norm_cfg, padding=1,
                                                     indice_key='subm1', conv_type=f'SubMConv{self.ndim}d',
                                                     act_type=act_type)
        encoder_out_channels = self.make_encoder_layers(make_sparse_convmodule, norm_cfg, self.base_channels)
        self.make_decoder_layers(make_sparse_convmodule, norm_cfg, encoder_out_channels)
        self.conv_out = make_sparse_convmodule(encoder_out_channels, self.output_channels, kernel_size=(3, 1, 1),
                                               stride=(2, 1, 1), norm_cfg=norm_cfg, padding=0,
                                               indice_key='spconv_down2', conv_type=f'SparseConv{self.ndim}d',
                                               act_type=act_type)

    def forward(self, voxel_features, coors, batch_size):
        """sparse_unet.py:114-159 -> dict(spatial_features [N, C*D, H, W], seg_features)"""
        assert self.is_3d, 'This forward function only supports 3D spconv'
        coors = coors.int()
        x = self.conv_input(SparseConvTensor(voxel_features, coors, self.sparse_shape, batch_size))
        encode_features = []
        for encoder_layer in self.encoder_layers:
            x = encoder_layer(x)
            encode_features.append(x)
        out = self.conv_out(encode_features[-1])
        spatial_features = out.dense()
        N, C, D, H, W = spatial_features.shape
        spatial_features = spatial_features.view(N, C * D, H, W)
        decode_features = []
        x = encode_features[-1]
        for i in range(self.stage_num, 0, -1):
            x = self.decoder_layer_forward(encode_features[i - 1], x, getattr(self, f'lateral_layer{i}'),
                                           getattr(self, f'merge_layer{i}'), getattr(self, f'upsample_layer{i}'))
            decode_features.append(x)
        return dict(spatial_features=spatial_features, seg_features=decode_features[-1].features)

    def decoder_layer_forward(self, x_lateral, x_bottom, lateral_layer, merge_layer, upsample_layer):
        """lateral block, concatenate with the feature from below, merge, channel-reduced residual, upsample"""
        x = lateral_layer(x_lateral)
        x = x.replace_feature(torch.cat((x_bottom.features, x.features), dim=1))
        x_merge = merge_layer(x)
        x = self.reduce_channel(x, x_merge.features.shape[1])
        x = x.replace_feature(x_merge.features + x.features)
        x = upsample_layer(x)
        return x

    @staticmethod
    def reduce_channel(x, out_channels):
        features = x.features
        n, in_channels = features.shape
        assert (in_channels % out_channels == 0) and (in_channels >= out_channels)
        return x.replace_feature(features.view(n, out_channels, -1).sum(dim=2))

    def make_encoder_layers(self, make_block, norm_cfg, in_channels):
        self.encoder_layers = SparseSequential()
        for i, blocks in enumerate(self.encoder_channels):
            blocks_list = []
            for j, out_channels in enumerate(tuple(blocks)):
                padding = tuple(self.encoder_paddings[i])[j]
                if i != 0 and j == 0:   # each stage but the first starts with a stride-2 convolution
                    blocks_list.append(make_block(in_channels, out_channels, 3, norm_cfg=norm_cfg, stride=2,
                                                  padding=padding, indice_key=f'spconv{i + 1}',
                                                  conv_type=f'SparseConv{self.ndim}d', act_type=self.act_type))
                else:
                    blocks_list.append(make_block(in_channels, out_channels, 3, norm_cfg=norm_cfg, padding=padding,
                                                  indice_key=f'subm{i + 1}', conv_type=f'SubMConv{self.ndim}d',
                                                  act_type=self.act_type))
                in_channels = out_channels
            self.encoder_layers.add_module(f'encoder_layer{i + 1}', SparseSequential(*blocks_list))
        return out_channels

    def make_decoder_layers(self, make_block, norm_cfg, in_channels):
        block_num = len(self.decoder_channels)
        for i, block_channels in enumerate(self.decoder_channels):
            paddings = self.decoder_paddings[i]
            setattr(self, f'lateral_layer{block_num - i}',
                    SparseBasicBlock(in_channels, block_channels[0],
                                     conv_cfg=dict(type=f'SubMConv{self.ndim}d', indice_key=f'subm{block_num - i}'),
                                     norm_cfg=norm_cfg, act_type=self.act_type))
            setattr(self, f'merge_layer{block_num - i}',
                    make_block(in_channels * 2, block_channels[1], 3, norm_cfg=norm_cfg, padding=paddings[0],
                               indice_key=f'subm{block_num - i}', conv_type=f'SubMConv{self.ndim}d',
                               act_type=self.act_type))
            if block_num - i != 1:
                setattr(self, f'upsample_layer{block_num - i}',
                        make_block(in_channels, block_channels[2], 3, norm_cfg=norm_cfg,
                                   indice_key=f'spconv{block_num - i}', conv_type=f'SparseInverseConv{self.ndim}d',
                                   act_type=self.act_type))
            else:
                setattr(self, f'upsample_layer{block_num - i}',
                        make_block(in_channels, block_channels[2], 3, norm_cfg=norm_cfg, padding=paddings[1],
                                   indice_key='subm1', conv_type=f'SubMConv{self.ndim}d', act_type=self.act_type))
            in_channels = block_channels[2]


@BACKBONES.register_module()
class SimpleSparseUNet(SparseUNet):
    """the U-Net without the dense output branch: FSD's segmentor backbone (sparse_unet.py:324-414)."""

    def __init__(self, in_channels, sparse_shape, order=('conv', 'norm', 'act'),
                 norm_cfg=dict(type='BN1d', eps=1e-3, momentum=0.01), base_channels=16, output_channels=128, ndim=3,
                 encoder_channels=((16, ), (32, 32, 32), (64, 64, 64), (64, 64, 64)),
                 encoder_paddings=((1, ), (1, 1, 1), (1, 1, 1), ((0, 1, 1), 1, 1)),
                 decoder_channels=((64, 64, 64), (64, 64, 32), (32, 32, 16), (16, 16, 16)),
                 decoder_paddings=((1, 0), (1, 0), (0, 0), (0, 1)), keep_coors_dims=None, act_type='relu',
                 return_multiscale_features=False, init_cfg=None):
        super().__init__(in_channels=in_channels, sparse_shape=sparse_shape, order=order, norm_cfg=norm_cfg,
                         base_channels=base_channels, output_channels=output_channels,
                         encoder_channels=encoder_channels, encoder_paddings=encoder_paddings,
                         decoder_channels=decoder_channels, decoder_paddings=decoder_paddings, ndim=ndim,
                         act_type=act_type, init_cfg=init_cfg)
        self.conv_out = None  # override
        self.ndim = ndim
        self.keep_coors_dims = keep_coors_dims
        self.return_multiscale_features = return_multiscale_features

    def forward(self, voxel_info):
        """voxel_info: dict(voxel_feats [N, C], voxel_coors [N, 4] (b, z, y, x)) -> [dict(voxel_feats, voxel_coors,
        sparse_shape, batch_size, decoder_features)] (a list, like SSTv2)."""
        coors = voxel_info['voxel_coors']
        if self.keep_coors_dims is not None:
            coors = coors[:, self.keep_coors_dims]
        voxel_features = voxel_info['voxel_feats']
        coors = coors.int()
        batch_size = coors[:, 0].max().item() + 1
        x = self.conv_input(SparseConvTensor(voxel_features, coors, self.sparse_shape, batch_size))
        encode_features = []
        decode_features = []
        for encoder_layer in self.encoder_layers:
            x = encoder_layer(x)
            encode_features.append(x)
        x = encode_features[-1]
        for i in range(self.stage_num, 0, -1):
            x = self.decoder_layer_forward(encode_features[i - 1], x, getattr(self, f'lateral_layer{i}'),
                                           getattr(self, f'merge_layer{i}'), getattr(self, f'upsample_layer{i}'))
            if self.return_multiscale_features:
                decode_features.append(x)
        ret = {'voxel_feats': x.features, 'voxel_coors': x.indices, 'sparse_shape': x.spatial_shape,
               'batch_size': x.batch_size, 'decoder_features': decode_features}
        return [ret, ]


@BACKBONES.register_module()
class VirtualVoxelMixer(SparseUNet):
    """FSDv2's backbone over real + virtual voxels (sparse_unet.py:417-504): the U-Net followed by a submanifold
    output convolution; forward(voxel_features, coors, batch_size) -> (features, indices, spatial_shape)."""

    def __init__(self, in_channels, sparse_shape, order=('conv', 'norm', 'act'),
                 norm_cfg=dict(type='BN1d', eps=1e-3, momentum=0.01), base_channels=16, output_channels=128, ndim=3,
                 encoder_channels=((16, ), (32, 32, 32), (64, 64, 64), (64, 64, 64)),
                 encoder_paddings=((1, ), (1, 1, 1), (1, 1, 1), ((0, 1, 1), 1, 1)),
                 decoder_channels=((64, 64, 64), (64, 64, 32), (32, 32, 16), (16, 16, 16)),
                 decoder_paddings=((1, 0), (1, 0), (0, 0), (0, 1)), keep_coors_dims=None, act_type='relu',
                 init_cfg=None):
        super().__init__(in_channels=in_channels, sparse_shape=sparse_shape, order=order, norm_cfg=norm_cfg,
                         base_channels=base_channels, output_channels=output_channels,
                         encoder_channels=encoder_channels, encoder_paddings=encoder_paddings,
                         decoder_channels=decoder_channels, decoder_paddings=decoder_paddings, ndim=ndim,
                         act_type=act_type, init_cfg=init_cfg)
        self.ndim = ndim
        self.keep_coors_dims = keep_coors_dims
        self.conv_out = make_sparse_convmodule(decoder_channels[-1][-1], self.output_channels, kernel_size=3, stride=1,
                                               norm_cfg=norm_cfg, padding=0, indice_key='out_conv',
                                               conv_type=f'SubMConv{self.ndim}d', act_type=act_type)

    def forward(self, voxel_features, coors, batch_size):
        if self.keep_coors_dims is not None:
            coors = coors[:, self.keep_coors_dims]
        coors = coors.int()
        x = self.conv_input(SparseConvTensor(voxel_features, coors, self.sparse_shape, batch_size))
        encode_features = []
        for encoder_layer in self.encoder_layers:
            x = encoder_layer(x)
            encode_features.append(x)
        x = encode_features[-1]
        for i in range(self.stage_num, 0, -1):
            x = self.decoder_layer_forward(encode_features[i - 1], x, getattr(self, f'lateral_layer{i}'),
                                           getattr(self, f'merge_layer{i}'), getattr(self, f'upsample_layer{i}'))
        x = self.conv_out(x)
        return x.features, x.indices, x.spatial_shape
