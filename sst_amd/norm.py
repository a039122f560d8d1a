"""Normalisation layers of the path + the mmcv-style ``build_norm_layer`` the reference modules call.

Reference: mmdet3d/ops/norm.py:9-24 (AllReduce), :28-86 (NaiveSyncBatchNorm1d), :89-140 (2d).
naiveSyncBN gathers [mean || meansqr] (2C floats) across ranks in forward and all-reduces the gradient
in backward; on ROCm ``backend='nccl'`` is RCCL, so the collectives run over xGMI unchanged.  These are
the only collectives inside the forward of the hot path.
"""
import torch
from torch import distributed as dist
from torch import nn as nn
from torch.autograd.function import Function


class AllReduce(Function):

    @staticmethod
    def forward(ctx, input):
        input_list = [torch.zeros_like(input) for _ in range(dist.get_world_size())]
        # all_gather then sum, exactly as the reference (norm.py:13-18)
        dist.all_gather(input_list, input, async_op=False)
        inputs = torch.stack(input_list, dim=0)
        return torch.sum(inputs, dim=0)

    @staticmethod
    def backward(ctx, grad_output):
        grad_output = grad_output.contiguous()
        dist.all_reduce(grad_output, async_op=False)
        return grad_output


def _sync_bn_forward(self, input, reduce_dims):
    C = input.shape[1]
    mean = torch.mean(input, dim=reduce_dims)
    meansqr = torch.mean(input * input, dim=reduce_dims)
    vec = torch.cat([mean, meansqr], dim=0)
    vec = AllReduce.apply(vec) * (1.0 / dist.get_world_size())
    mean, meansqr = torch.split(vec, C)
    var = meansqr - mean * mean
    self.running_mean += self.momentum * (mean.detach() - self.running_mean)
    self.running_var += self.momentum * (var.detach() - self.running_var)
    invstd = torch.rsqrt(var + self.eps)
    scale = self.weight * invstd
    bias = self.bias - mean * scale
    return scale, bias


class NaiveSyncBatchNorm1d(nn.BatchNorm1d):
    """mmdet3d/ops/norm.py:28-86."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.fp16_enabled = False

    def forward(self, input):
        input = input.float()
        if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1 or not self.training:
            return super().forward(input)
        assert input.shape[0] > 0, 'SyncBN does not support empty inputs'
        dim_2 = input.dim() == 2
        if dim_2:
            input = input.unsqueeze(2)
        scale, bias = _sync_bn_forward(self, input, [0, 2])
        input = input * scale.reshape(1, -1, 1) + bias.reshape(1, -1, 1)
        if dim_2:
            input = input.squeeze(2)
        return input


class NaiveSyncBatchNorm2d(nn.BatchNorm2d):
    """mmdet3d/ops/norm.py:89-140."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.fp16_enabled = False

    def forward(self, input):
        input = input.float()
        if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1 or not self.training:
            return super().forward(input)
        assert input.shape[0] > 0, 'SyncBN does not support empty inputs'
        scale, bias = _sync_bn_forward(self, input, [0, 2, 3])
        return input * scale.reshape(1, -1, 1, 1) + bias.reshape(1, -1, 1, 1)


NORM_LAYERS = {
    'BN': nn.BatchNorm2d,
    'BN1d': nn.BatchNorm1d,
    'BN2d': nn.BatchNorm2d,
    'BN3d': nn.BatchNorm3d,
    'SyncBN': nn.SyncBatchNorm,
    'LN': nn.LayerNorm,
    'GN': nn.GroupNorm,
    'naiveSyncBN1d': NaiveSyncBatchNorm1d,
    'naiveSyncBN2d': NaiveSyncBatchNorm2d,
}


def build_norm_layer(cfg, num_features, postfix=''):
    """mmcv.cnn.build_norm_layer: returns (name, layer).  cfg: dict(type=..., eps=..., momentum=..., requires_grad=...)."""
    if not isinstance(cfg, dict) or 'type' not in cfg:
        raise TypeError('cfg must be a dict containing the key "type"')
    cfg_ = dict(cfg)
    layer_type = cfg_.pop('type')
    if layer_type not in NORM_LAYERS:
        raise KeyError(f'Unrecognized norm type {layer_type}')
    norm_cls = NORM_LAYERS[layer_type]
    abbr = {'LN': 'ln', 'GN': 'gn'}.get(layer_type, 'bn')
    name = abbr + str(postfix)
    requires_grad = cfg_.pop('requires_grad', True)
    cfg_.setdefault('eps', 1e-5)
    if layer_type == 'GN':
        layer = norm_cls(num_channels=num_features, **cfg_)
    else:
        layer = norm_cls(num_features, **cfg_)
    for p in layer.parameters():
        p.requires_grad = requires_grad
    return name, layer


def build_conv_layer(cfg, *args, **kwargs):
    """mmcv.cnn.build_conv_layer for the dense attached convs of SSTv2 (sst_v2.py:86-92)."""
    cfg_ = dict(type='Conv2d') if cfg is None else dict(cfg)
    layer_type = cfg_.pop('type')
    convs = {'Conv1d': nn.Conv1d, 'Conv2d': nn.Conv2d, 'Conv3d': nn.Conv3d, 'Conv': nn.Conv2d}
    if layer_type not in convs:
        raise KeyError(f'Unrecognized conv type {layer_type}')
    return convs[layer_type](*args, **kwargs, **cfg_)
