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

from .norm import batch_norm_act, build_conv_layer, build_norm_layer
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
        # BatchNorm1d + ReLU, and BatchNorm1d + identity + ReLU, are one pass each (csrc/bn.hip) when the block is built
        # from them - the reference's order of operations otherwise
        fused = (type(self.relu) is nn.ReLU and isinstance(self.norm1, nn.BatchNorm1d)
                 and isinstance(self.norm2, nn.BatchNorm1d) and identity.is_cuda)
        out = self.conv1(x)
        if fused:
            out = replace_feature(out, batch_norm_act(self.norm1, out.features, relu=True))
        else:
            out = replace_feature(out, self.norm1(out.features))
            out = replace_feature(out, self.relu(out.features))
        out = self.conv2(out)
        if self.downsample is not None:
            identity = self.downsample(x)
        if fused:
            return replace_feature(out, batch_norm_act(self.norm2, out.features, relu=True, residual=identity))
        out = replace_feature(out, self.norm2(out.features))
        out = replace_feature(out, out.features + identity)
        out = replace_feature(out, self.relu(out.features))
        return out


_INVERSE_CONVS = tuple(f'SparseInverseConv{d}d' for d in (1, 2, 3, 4))


def make_sparse_convmodule(in_channels, out_channels, kernel_size, indice_key, stride=1, padding=0,
                           conv_type='SubMConv3d', act_type='relu', norm_cfg=None, order=('conv', 'norm', 'act')):
    """SparseSequential of convolution / norm / activation in ``order`` (sparse_block.py:218-289); an inverse
    convolution takes its geometry from the rulebook of its ``indice_key`` instead of stride / padding."""
    assert isinstance(order, tuple) and 0 < len(order) <= 3 and set(order) <= {'conv', 'norm', 'act'}
    geometry = {} if conv_type in _INVERSE_CONVS else dict(stride=stride, padding=padding)
    make = {'conv': lambda: build_conv_layer(dict(type=conv_type, indice_key=indice_key), in_channels, out_channels,
                                             kernel_size, bias=False, **geometry),
            'norm': lambda: build_norm_layer(norm_cfg, out_channels)[1],
            'act': lambda: _activation(act_type)}
    return SparseSequential(*[make[kind]() for kind in order])


class _UNetStages(nn.Module):
    """What SparseUNet, SimpleSparseUNet and VirtualVoxelMixer share (sparse_unet.py:16-321): the input convolution, an
    encoder of submanifold stages (every stage after the first opened by a stride-2 convolution), and a decoder that
    walks back up - lateral residual block, concatenation with the coarser feature, merge convolution, channel-folded
    residual, inverse convolution onto the finer level's voxels.  Sub-module names are the reference's (they are the
    ``state_dict`` keys): conv_input, encoder_layers.encoder_layer{i}, lateral_layer{i}, merge_layer{i},
    upsample_layer{i}."""

    def _build_stages(self, in_channels, sparse_shape, order, norm_cfg, base_channels, output_channels, encoder_channels,
                      encoder_paddings, decoder_channels, decoder_paddings, ndim, act_type, init_cfg):
        if ndim != 3:
            raise NotImplementedError('sst_amd sparse U-Nets: ndim=3 only')
        assert isinstance(order, tuple) and set(order) == {'conv', 'norm', 'act'} and len(order) == 3
        self.init_cfg, self.sparse_shape, self.in_channels, self.order = init_cfg, sparse_shape, in_channels, order
        self.base_channels, self.output_channels = base_channels, output_channels
        self.encoder_channels, self.encoder_paddings = encoder_channels, encoder_paddings
        self.decoder_channels, self.decoder_paddings = decoder_channels, decoder_paddings
        self.stage_num = len(encoder_channels)
        self.ndim, self.is_3d, self.fp16_enabled, self.act_type = ndim, True, False, act_type
        subm, strided, inverse = f'SubMConv{ndim}d', f'SparseConv{ndim}d', f'SparseInverseConv{ndim}d'
        common = dict(norm_cfg=norm_cfg, act_type=act_type)
        # a pre-activation order leaves the input convolution bare
        self.conv_input = make_sparse_convmodule(in_channels, base_channels, 3, padding=1, indice_key='subm1',
                                                 conv_type=subm, order=order if order[0] == 'conv' else ('conv', ),
                                                 **common)
        # encoder
        self.encoder_layers = SparseSequential()
        width = base_channels
        for level, (channels, paddings) in enumerate(zip(encoder_channels, encoder_paddings), start=1):
            stage = []
            for j, (out, pad) in enumerate(zip(tuple(channels), tuple(paddings))):
                down = level > 1 and j == 0
                stage.append(make_sparse_convmodule(width, out, 3, padding=pad, stride=2 if down else 1,
                                                    indice_key=f'spconv{level}' if down else f'subm{level}',
                                                    conv_type=strided if down else subm, **common))
                width = out
            self.encoder_layers.add_module(f'encoder_layer{level}', SparseSequential(*stage))
        self.encoder_out_channels = width
        # decoder, coarsest level first
        for level, ((c_lat, c_merge, c_up), pads) in zip(range(len(decoder_channels), 0, -1),
                                                         zip(decoder_channels, decoder_paddings)):
            setattr(self, f'lateral_layer{level}',
                    SparseBasicBlock(width, c_lat, conv_cfg=dict(type=subm, indice_key=f'subm{level}'),
                                     norm_cfg=norm_cfg, act_type=act_type))
            setattr(self, f'merge_layer{level}',
                    make_sparse_convmodule(2 * width, c_merge, 3, padding=pads[0], indice_key=f'subm{level}',
                                           conv_type=subm, **common))
            up = (dict(indice_key=f'spconv{level}', conv_type=inverse) if level > 1
                  else dict(indice_key='subm1', conv_type=subm, padding=pads[1]))
            setattr(self, f'upsample_layer{level}', make_sparse_convmodule(width, c_up, 3, **up, **common))
            width = c_up

    def build_rulebooks(self, coors, batch_size):
        """Every rulebook of the network for the voxel set ``coors`` [N, 4] (b, z, y, x), WITHOUT features: the index half of
        encode() (the decoder re-uses the encoder's keys).  They depend on the coordinates alone, so a training loop can
        build them for the NEXT batch while the current one is still being differentiated - each rulebook costs a size
        read-back, and inside the forward pass every one of them waits for all the kernels queued before it.  -> the
        ``indice_dict`` to hand to encode() / forward()."""
        from .spconv import SparseConvolution
        x = SparseConvTensor(None, coors.int(), self.sparse_shape, batch_size)
        for m in list(self.conv_input.modules()) + list(self.encoder_layers.modules()):
            if isinstance(m, SparseConvolution):
                x = m.dry(x)
        return x.indice_dict

    def encode(self, voxel_features, coors, batch_size, indice_dict=None):
        """-> the feature of every encoder level, finest first; indice_dict: rulebooks built ahead (build_rulebooks) for
        exactly these coordinates"""
        first = SparseConvTensor(voxel_features, coors.int(), self.sparse_shape, batch_size)
        if indice_dict is not None:
            first.indice_dict = indice_dict
        x = self.conv_input(first)
        levels = []
        for stage in self.encoder_layers:
            x = stage(x)
            levels.append(x)
        return levels

    def decode(self, levels, keep_all=False):
        """walk back up from the coarsest level: -> the finest decoder feature (and every level's, if asked)"""
        x, outs = levels[-1], []
        for level in range(self.stage_num, 0, -1):
            x = self.decoder_layer_forward(levels[level - 1], x, getattr(self, f'lateral_layer{level}'),
                                           getattr(self, f'merge_layer{level}'), getattr(self, f'upsample_layer{level}'))
            if keep_all:
                outs.append(x)
        return x, outs

    def decoder_layer_forward(self, x_lateral, x_bottom, lateral_layer, merge_layer, upsample_layer):
        """one decoder step (sparse_unet.py:161-185)"""
        x = lateral_layer(x_lateral)
        x = x.replace_feature(torch.cat((x_bottom.features, x.features), dim=1))
        merged = merge_layer(x)
        x = self.reduce_channel(x, merged.features.shape[1])
        return upsample_layer(x.replace_feature(merged.features + x.features))

    @staticmethod
    def reduce_channel(x, out_channels):
        """fold the channels down by summing groups of in / out neighbours (sparse_unet.py:187-202)"""
        n, width = x.features.shape
        assert width % out_channels == 0 and width >= out_channels
        return x.replace_feature(x.features.view(n, out_channels, -1).sum(dim=2))


_UNET_DEFAULTS = dict(
    encoder_channels=((16, ), (32, 32, 32), (64, 64, 64), (64, 64, 64)),
    encoder_paddings=((1, ), (1, 1, 1), (1, 1, 1), ((0, 1, 1), 1, 1)),
    decoder_channels=((64, 64, 64), (64, 64, 32), (32, 32, 16), (16, 16, 16)),
    decoder_paddings=((1, 0), (1, 0), (0, 0), (0, 1)))


@MIDDLE_ENCODERS.register_module()
class SparseUNet(_UNetStages):
    """SECOND-style U-Net with the dense output branch (sparse_unet.py:16-159)."""

    def __init__(self, in_channels, sparse_shape, order=('conv', 'norm', 'act'),
                 norm_cfg=dict(type='BN1d', eps=1e-3, momentum=0.01), base_channels=16, output_channels=128,
                 encoder_channels=_UNET_DEFAULTS['encoder_channels'], encoder_paddings=_UNET_DEFAULTS['encoder_paddings'],
                 decoder_channels=_UNET_DEFAULTS['decoder_channels'], decoder_paddings=_UNET_DEFAULTS['decoder_paddings'],
                 ndim=3, act_type='relu', init_cfg=None):
        super().__init__()
        self._build_stages(in_channels, sparse_shape, order, norm_cfg, base_channels, output_channels, encoder_channels,
                           encoder_paddings, decoder_channels, decoder_paddings, ndim, act_type, init_cfg)
        self.conv_out = make_sparse_convmodule(self.encoder_out_channels, output_channels, kernel_size=(3, 1, 1),
                                               stride=(2, 1, 1), norm_cfg=norm_cfg, padding=0, indice_key='spconv_down2',
                                               conv_type=f'SparseConv{ndim}d', act_type=act_type)

    def forward(self, voxel_features, coors, batch_size):
        """-> dict(spatial_features [N, C*D, H, W] from the coarsest level, seg_features of the finest decoder level)"""
        levels = self.encode(voxel_features, coors, batch_size)
        dense = self.conv_out(levels[-1]).dense()
        n, c, d, h, w = dense.shape
        finest, _ = self.decode(levels)
        return dict(spatial_features=dense.view(n, c * d, h, w), seg_features=finest.features)


@BACKBONES.register_module()
class SimpleSparseUNet(_UNetStages):
    """the U-Net without the dense output branch: FSD's segmentor backbone (sparse_unet.py:324-414)."""

    def __init__(self, in_channels, sparse_shape, order=('conv', 'norm', 'act'),
                 norm_cfg=dict(type='BN1d', eps=1e-3, momentum=0.01), base_channels=16, output_channels=128, ndim=3,
                 encoder_channels=_UNET_DEFAULTS['encoder_channels'], encoder_paddings=_UNET_DEFAULTS['encoder_paddings'],
                 decoder_channels=_UNET_DEFAULTS['decoder_channels'], decoder_paddings=_UNET_DEFAULTS['decoder_paddings'],
                 keep_coors_dims=None, act_type='relu', return_multiscale_features=False, init_cfg=None):
        super().__init__()
        self._build_stages(in_channels, sparse_shape, order, norm_cfg, base_channels, output_channels, encoder_channels,
                           encoder_paddings, decoder_channels, decoder_paddings, ndim, act_type, init_cfg)
        self.conv_out = None
        self.keep_coors_dims = keep_coors_dims
        self.return_multiscale_features = return_multiscale_features

    def forward(self, voxel_info):
        """voxel_info: dict(voxel_feats [N, C], voxel_coors [N, 4] (b, z, y, x)[, batch_size]) -> [dict(voxel_feats,
        voxel_coors, sparse_shape, batch_size, decoder_features)] (a list, like SSTv2)."""
        coors = voxel_info['voxel_coors']
        if self.keep_coors_dims is not None:
            coors = coors[:, self.keep_coors_dims]
        batch_size = voxel_info.get('batch_size')
        if batch_size is None:   # the reference reads it off the coordinates (one read-back, sparse_unet.py:382)
            batch_size = int(coors[:, 0].max().item()) + 1
        finest, every = self.decode(self.encode(voxel_info['voxel_feats'], coors, batch_size, voxel_info.get('indice_dict')),
                                    keep_all=self.return_multiscale_features)
        return [{'voxel_feats': finest.features, 'voxel_coors': finest.indices, 'sparse_shape': finest.spatial_shape,
                 'batch_size': finest.batch_size, 'decoder_features': every}]


@BACKBONES.register_module()
class VirtualVoxelMixer(_UNetStages):
    """FSDv2's backbone over real + virtual voxels (sparse_unet.py:417-504): the U-Net followed by a submanifold
    output convolution; forward(voxel_features, coors, batch_size) -> (features, indices, spatial_shape)."""

    def __init__(self, in_channels, sparse_shape, order=('conv', 'norm', 'act'),
                 norm_cfg=dict(type='BN1d', eps=1e-3, momentum=0.01), base_channels=16, output_channels=128, ndim=3,
                 encoder_channels=_UNET_DEFAULTS['encoder_channels'], encoder_paddings=_UNET_DEFAULTS['encoder_paddings'],
                 decoder_channels=_UNET_DEFAULTS['decoder_channels'], decoder_paddings=_UNET_DEFAULTS['decoder_paddings'],
                 keep_coors_dims=None, act_type='relu', init_cfg=None):
        super().__init__()
        self._build_stages(in_channels, sparse_shape, order, norm_cfg, base_channels, output_channels, encoder_channels,
                           encoder_paddings, decoder_channels, decoder_paddings, ndim, act_type, init_cfg)
        self.keep_coors_dims = keep_coors_dims
        self.conv_out = make_sparse_convmodule(decoder_channels[-1][-1], output_channels, kernel_size=3, stride=1,
                                               norm_cfg=norm_cfg, padding=0, indice_key='out_conv',
                                               conv_type=f'SubMConv{ndim}d', act_type=act_type)

    def forward(self, voxel_features, coors, batch_size):
        if self.keep_coors_dims is not None:
            coors = coors[:, self.keep_coors_dims]
        finest, _ = self.decode(self.encode(voxel_features, coors, batch_size))
        out = self.conv_out(finest)
        return out.features, out.indices, out.spatial_shape
