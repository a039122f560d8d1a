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


def _dist_world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size()
    return 1


class _BatchNormActFn(Function):
    """y = act(x * scale + shift (+ res)) through csrc/bn.hip; ``prep`` [4, C] = mean, invstd, scale, shift.
    ``batch_stats``: mean / invstd were computed from this batch (training), so the gradient flows through them:
    dx = scale * (g - (G1 + xhat * G2) / count), G1 = sum g, G2 = sum g * xhat taken over every rank when ``sync``
    (naiveSyncBN averages the per-rank means, ops/norm.py:53-58, hence count = world_size * N_local; its AllReduce
    backward sums the statistic gradients, :20-24).  ``res``: the identity branch of a residual block, added before
    the activation in the same pass; its gradient (dy masked by the activation) leaves the backward apply kernel."""

    @staticmethod
    def forward(ctx, x, weight, bias, prep, act, batch_stats, count, sync, res=None):
        from . import _lib
        n, c = x.shape
        y = torch.empty((n, c), dtype=torch.float32, device=x.device)
        rc = _lib.load().sst_bn_act_res_fwd_f32(_lib.ptr(x), n, c, x.stride(0),
                                                _lib.ptr(res) if res is not None else None,
                                                res.stride(0) if res is not None else 0, _lib.ptr(prep[2]),
                                                _lib.ptr(prep[3]), int(act), _lib.ptr(y), y.stride(0), _lib.stream_ptr())
        _lib.check(rc, 'sst_bn_act_res_fwd_f32')
        ctx.has_res = res is not None
        if ctx.has_res:
            ctx.save_for_backward(x, prep, res)
        else:
            ctx.save_for_backward(x, prep)
        ctx.cfg = (int(act), bool(batch_stats), float(count), bool(sync), weight is not None, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import _lib
        if ctx.has_res:
            x, prep, res = ctx.saved_tensors
        else:
            (x, prep), res = ctx.saved_tensors, None
        act, batch_stats, count, sync, has_w, has_b = ctx.cfg
        n, c = x.shape
        dy = dy.contiguous()
        lib = _lib.load()
        res_p, ldr = (_lib.ptr(res), res.stride(0)) if res is not None else (None, 0)
        sums = torch.empty((2, c), dtype=torch.float32, device=x.device)
        ws = _lib.workspace(lib.sst_bn_workspace_bytes(n, c), x.device)
        rc = lib.sst_bn_act_res_bwd_reduce_f32(_lib.ptr(dy), _lib.ptr(x), res_p, n, c, dy.stride(0), x.stride(0), ldr,
                                               _lib.ptr(prep[0]), _lib.ptr(prep[1]), _lib.ptr(prep[2]),
                                               _lib.ptr(prep[3]), act, _lib.ptr(sums[0]), _lib.ptr(sums[1]),
                                               _lib.ptr(ws), _lib.stream_ptr())
        _lib.check(rc, 'sst_bn_act_res_bwd_reduce_f32')
        total = sums
        if sync and batch_stats:
            total = sums.clone()  # the parameter gradients stay the local sums (DDP averages them afterwards)
            dist.all_reduce(total, async_op=False)
        dx = dres = None
        want_dres = res is not None and ctx.needs_input_grad[8]
        if ctx.needs_input_grad[0] or want_dres:
            dx = torch.empty((n, c), dtype=torch.float32, device=x.device)
            if want_dres:
                dres = torch.empty((n, c), dtype=torch.float32, device=x.device)
            rc = lib.sst_bn_act_res_bwd_apply_f32(_lib.ptr(dy), _lib.ptr(x), res_p, n, c, dy.stride(0), x.stride(0),
                                                  ldr, _lib.ptr(prep[0]), _lib.ptr(prep[1]), _lib.ptr(prep[2]),
                                                  _lib.ptr(prep[3]), _lib.ptr(total[0]), _lib.ptr(total[1]),
                                                  (1.0 / count) if batch_stats else 0.0, act,
                                                  _lib.ptr(dres) if dres is not None else None,
                                                  dres.stride(0) if dres is not None else 0, _lib.ptr(dx),
                                                  dx.stride(0), _lib.stream_ptr())
            _lib.check(rc, 'sst_bn_act_res_bwd_apply_f32')
            if not ctx.needs_input_grad[0]:
                dx = None
        return (dx, (sums[1] if has_w else None), (sums[0] if has_b else None), None, None, None, None, None, dres)


def _bn_kernel_ok(bn, x):
    return (isinstance(bn, nn.BatchNorm1d) and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32
            and x.size(0) > 0 and x.size(1) % 4 == 0 and x.size(1) <= 1024 and x.stride(1) == 1
            and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0)


def bn_prepare(bn, x, partials=None):
    """The statistics half of ``batch_norm_act``: -> (prep [4, C] = mean, invstd, scale, shift; batch_stats; count; sync) with
    the module's bookkeeping (running statistics, num_batches_tracked, naiveSyncBN's cross-rank average).
    ``partials``: a workspace that already holds the block partials of the column moments of ``x`` (written by the kernel that
    produced x: sst_vfe_linear_moments_f32) - x is then not read again for them."""
    from . import _lib
    n, c = x.shape
    world = _dist_world()
    sync = isinstance(bn, NaiveSyncBatchNorm1d) and world > 1 and bn.training
    batch_stats = bn.training or (bn.running_mean is None and bn.running_var is None)
    count = 1.0
    lib = _lib.load()
    prep = torch.empty((4, c), dtype=torch.float32, device=x.device)  # mean, invstd, scale, shift
    if batch_stats and not sync:
        # one call: moments, invstd / scale / shift and nn.BatchNorm1d's running-statistics bookkeeping
        count = float(n)
        factor = 0.0
        update = bn.training and bn.track_running_stats and bn.running_mean is not None
        tracked = None
        if update:
            factor = 0.0 if bn.momentum is None else bn.momentum
            if bn.num_batches_tracked is not None:
                if bn.momentum is None:   # cumulative average: the factor needs the counter on the host
                    with torch.no_grad():
                        bn.num_batches_tracked.add_(1)
                    factor = 1.0 / float(bn.num_batches_tracked)
                elif bn.num_batches_tracked.is_cuda and bn.num_batches_tracked.dtype == torch.int64:
                    tracked = bn.num_batches_tracked      # incremented by the prepare kernel: no launch of its own
                else:
                    with torch.no_grad():
                        bn.num_batches_tracked.add_(1)
        w_p = _lib.ptr(bn.weight.detach()) if bn.weight is not None else None
        b_p = _lib.ptr(bn.bias.detach()) if bn.bias is not None else None
        rm_p, rv_p = (_lib.ptr(bn.running_mean), _lib.ptr(bn.running_var)) if update else (None, None)
        if partials is not None:
            rc = lib.sst_bn_prepare_from_partials_f32(n, c, w_p, b_p, float(bn.eps), rm_p, rv_p, float(factor),
                                                      _lib.ptr(tracked) if tracked is not None else None, _lib.ptr(prep),
                                                      _lib.ptr(partials), _lib.stream_ptr())
            _lib.check(rc, 'sst_bn_prepare_from_partials_f32')
        else:
            xd = x.detach()
            ws = _lib.workspace(lib.sst_bn_workspace_bytes(n, c), x.device)
            rc = lib.sst_bn_prepare_tracked_f32(_lib.ptr(xd), n, c, xd.stride(0), w_p, b_p, float(bn.eps), rm_p, rv_p,
                                                float(factor), _lib.ptr(tracked) if tracked is not None else None,
                                                _lib.ptr(prep), _lib.ptr(ws), _lib.stream_ptr())
            _lib.check(rc, 'sst_bn_prepare_tracked_f32')
    else:
        if batch_stats:  # naiveSyncBN across ranks
            stats = torch.empty((2, c), dtype=torch.float32, device=x.device)
            if partials is not None:
                rc = lib.sst_bn_stats_from_partials_f32(n, c, _lib.ptr(stats[0]), _lib.ptr(stats[1]), _lib.ptr(partials),
                                                        _lib.stream_ptr())
                _lib.check(rc, 'sst_bn_stats_from_partials_f32')
            else:
                xd = x.detach()
                ws = _lib.workspace(lib.sst_bn_workspace_bytes(n, c), x.device)
                rc = lib.sst_bn_stats_f32(_lib.ptr(xd), n, c, xd.stride(0), _lib.ptr(stats[0]), _lib.ptr(stats[1]),
                                          _lib.ptr(ws), _lib.stream_ptr())
                _lib.check(rc, 'sst_bn_stats_f32')
            mean, var = stats[0], stats[1]
            vec = torch.cat([mean, var + mean * mean], dim=0)  # [mean || meansqr], ops/norm.py:53
            dist.all_reduce(vec, async_op=False)
            vec = vec * (1.0 / world)
            mean, meansqr = vec[:c], vec[c:]
            var = (meansqr - mean * mean).clamp_(min=0)
            with torch.no_grad():
                bn.running_mean += bn.momentum * (mean - bn.running_mean)
                bn.running_var += bn.momentum * (var - bn.running_var)
            count = float(world * n)
        else:
            mean, var = bn.running_mean, bn.running_var
        with torch.no_grad():
            invstd = torch.rsqrt(var + bn.eps)
            scale = invstd if bn.weight is None else bn.weight * invstd
            shift = -mean * scale if bn.bias is None else bn.bias - mean * scale
            prep[0], prep[1], prep[2], prep[3] = mean, invstd, scale, shift
    return prep, batch_stats, count, sync


def batch_norm_act(bn, x, relu=False, residual=None):
    """``relu(bn(x))`` / ``bn(x)`` / ``relu(bn(x) + residual)`` for a BatchNorm1d-family module on [N, C] point features.

    CUDA float32 inputs go through the fused kernels (csrc/bn.hip) with the module's exact bookkeeping:
    nn.BatchNorm1d (torch/nn/modules/batchnorm.py: biased batch variance for the output, unbiased for
    running_var, num_batches_tracked) or, for NaiveSyncBatchNorm1d in distributed training, the reference's
    averaging of per-rank [mean || meansqr] and its ``running += momentum * (stat - running)`` update
    (mmdet3d/ops/norm.py:50-66).  Anything else (CPU tensors, 3-D inputs) takes the module's own forward."""
    if residual is not None and not (residual.shape == x.shape and residual.dtype == torch.float32 and residual.is_cuda
                                     and residual.stride(1) == 1 and residual.stride(0) % 4 == 0
                                     and residual.data_ptr() % 16 == 0):
        y = batch_norm_act(bn, x, relu=False) + residual
        return torch.relu(y) if relu else y
    if not _bn_kernel_ok(bn, x):
        y = bn(x)
        if residual is not None:
            y = y + residual
        return torch.relu(y) if relu else y
    prep, batch_stats, count, sync = bn_prepare(bn, x)
    return _BatchNormActFn.apply(x, bn.weight, bn.bias, prep, bool(relu), batch_stats, count, sync, residual)


class BatchNorm1d(nn.BatchNorm1d):
    """nn.BatchNorm1d (same parameters / buffers / state_dict) whose [N, C] CUDA forward and backward run through
    csrc/bn.hip; what ``build_norm_layer(dict(type='BN1d'))`` returns."""

    def forward(self, input):
        if _bn_kernel_ok(self, input):
            return batch_norm_act(self, input, relu=False)
        return super().forward(input)


class LayerNorm(nn.LayerNorm):
    """nn.LayerNorm (same parameters / state_dict) whose [N, C] CUDA float32 forward / backward run through the row
    kernels of csrc/dense.hip; what ``build_norm_layer(dict(type='LN'))`` returns (FSD's SIR / rel_mlp layers)."""

    def forward(self, input):
        if input.dim() == 2 and input.is_cuda and input.dtype == torch.float32:
            from .dense import add_layer_norm
            return add_layer_norm(input, None, self)  # falls back to torch for shapes the kernel is not built for
        return super().forward(input)


class NaiveSyncBatchNorm1d(nn.BatchNorm1d):
    """mmdet3d/ops/norm.py:28-86."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.fp16_enabled = False

    def forward(self, input):
        input = input.float()
        if _bn_kernel_ok(self, input):
            return batch_norm_act(self, input, relu=False)
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
    'BN1d': BatchNorm1d,
    'BN2d': nn.BatchNorm2d,
    'BN3d': nn.BatchNorm3d,
    'SyncBN': nn.SyncBatchNorm,
    'LN': LayerNorm,
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
        from .spconv import CONV_LAYERS  # sparse layers register there as they do in mmcv.cnn.CONV_LAYERS
        sparse = CONV_LAYERS.get(layer_type)
        if sparse is not None:
            return sparse(*args, **kwargs, **cfg_)
        raise KeyError(f'Unrecognized conv type {layer_type}')
    return convs[layer_type](*args, **kwargs, **cfg_)
