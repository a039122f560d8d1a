"""DynamicVFE's two-layer stack as ONE autograd node over fused passes.

Reference data flow (mmdet3d/models/voxel_encoders/voxel_encoder.py:258-296, utils.py:107-144), per layer:
``Linear(bias=False) -> BN1d -> ReLU -> DynamicScatter(max) -> cat([point feats, pooled[voxel of the point]])``.  Module by
module that is, per step, three library GEMMs, two [N, 128] concatenations / activations that exist only to be read once, a
dense [N, 128] matrix of the pooling's gradient that is 99 % zeros, and ~25 autograd nodes.  Here:

  layer 0   y0 = x W0^T (K = 10: VALU kernel, W0^T in LDS) with the batch-norm moment partials from the same pass
            pf0 = relu(bn0(y0));  pooled0, arg0 = segmented max of pf0 over the voxels
  layer 1   SPLIT WEIGHT: cat([pf0, pooled0[v]]) W1^T = pf0 W1[:, :64]^T + (pooled0 W1[:, 64:]^T)[v]: the pooled half is
            multiplied once per voxel, gathered in the epilogue of the point half's product; no concatenated matrix
            out, arg1 = segmented max of relu(bn1(y1)), the norm + activation applied WHILE pooling: the activated [N, 128]
            matrix is never written (only the voxel features are wanted: voxel_encoder.py:296)
  backward  the gradient of a pooling is routed to the recorded arg-max rows inside the batch-norm backward passes
            (sst_bn_act_pool_bwd_*): no dense scatter; d(pooled0) from d(t) = segmented SUM of dy1 over the voxels
            (the plan's deterministic hand-back gradient) times W1[:, 64:]: [M, 128] x [128, 64] instead of [N, 128] x [128, 64]

All products on the bf16 matrix pipe from the exact three-way split (csrc/dense_f32x6.hip; same results as fp32 products to
fp32 rounding) or, for K = 10, in plain fp32 FMAs.  naiveSyncBN's cross-rank averaging sits between the halves of the
statistics (norm.bn_prepare), as in batch_norm_act.
"""
import torch
from torch import nn
from torch.autograd.function import Function

from . import _lib
from . import kernels as K
from .dense import weight_bias_grad
from .norm import bn_prepare

EPI_BIAS = 0


def _x6_linear(x, w_ptr, ldw, trans_w, m, k, n, name):
    y = torch.empty((m, n), dtype=torch.float32, device=x.device)
    rc = _lib.load().sst_tall_linear_epi_f32x6(_lib.ptr(x), x.stride(0), w_ptr, ldw, int(trans_w), None, m, k, n, EPI_BIAS,
                                               None, None, 0, _lib.ptr(y), n, _lib.stream_ptr())
    _lib.check(rc, name)
    return y


def _float_ptr(t, offset_elems=0):
    import ctypes
    return ctypes.c_void_p(t.data_ptr() + 4 * offset_elems)


def fused_vfe2_ok(encoder, x, plan):
    """two layers Linear(k <= 16 -> 64) / Linear(128 -> 128 or 64), BatchNorm1d-family norms with affine parameters, ReLU, max
    pooling, fp32 CUDA input without a gradient, a plan that offers the raw pooling and the hand-back gradient"""
    layers = getattr(encoder, 'vfe_layers', None)
    if layers is None or len(layers) != 2 or encoder.mode != 'max' or encoder.return_point_feats:
        return False
    l0, l1 = layers
    for l in (l0, l1):
        # DynamicVFELayer is Linear -> norm -> ReLU by construction; the V2 layer names its activation and may drop out
        relu = isinstance(getattr(l, 'act', None), nn.ReLU) if hasattr(l, 'act') else True
        if not (isinstance(l.norm, nn.BatchNorm1d) and l.norm.weight is not None and l.norm.bias is not None
                and relu and getattr(l, 'dropout', None) is None and l.linear.bias is None):
            return False
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.size(0) > 0 and not x.requires_grad
            and x.size(1) == l0.linear.in_features and x.size(1) <= 16 and l0.linear.out_features == 64
            and l1.linear.in_features == 128 and l1.linear.out_features in (64, 128)
            and l0.linear.weight.is_contiguous() and l1.linear.weight.is_contiguous()
            and hasattr(plan, 'raw_max') and hasattr(plan, 'group_sum')
            and getattr(plan, 'num_voxels', 1) != 0)     # no kept voxel at all: the hand-back has no row 0 to read (layer-wise path)


class FusedVFE2(Function):

    @staticmethod
    def forward(ctx, x, w0, g0, b0, w1, g1, b1, plan, bn0, bn1):
        lib = _lib.load()
        x = x.contiguous()
        n, k = x.shape
        c0, c1 = w0.size(0), w1.size(0)
        dev = x.device
        st = _lib.stream_ptr
        # layer 0: product + moment partials, statistics, norm + activation, pooling
        y0 = torch.empty((n, c0), dtype=torch.float32, device=dev)
        ws = _lib.workspace(lib.sst_bn_workspace_bytes(n, c0), dev)
        rc = lib.sst_vfe_linear_moments_f32(_lib.ptr(x), x.stride(0), n, k, _lib.ptr(w0), w0.stride(0), c0, _lib.ptr(y0), c0,
                                            _lib.ptr(ws), st())
        _lib.check(rc, 'sst_vfe_linear_moments_f32')
        prep0, bs0, cnt0, sync0 = bn_prepare(bn0, y0, partials=ws)
        pf0 = torch.empty((n, c0), dtype=torch.float32, device=dev)
        rc = lib.sst_bn_act_fwd_f32(_lib.ptr(y0), n, c0, c0, _lib.ptr(prep0[2]), _lib.ptr(prep0[3]), 1, _lib.ptr(pf0), c0, st())
        _lib.check(rc, 'sst_bn_act_fwd_f32')
        pooled0, arg0 = plan.raw_max(pf0)
        rows = pooled0.size(0)
        # layer 1 in its split-weight form
        t = _x6_linear(pooled0, _float_ptr(w1, c0), w1.stride(0), 0, rows, c0, c1, 'sst_tall_linear_epi_f32x6 (pooled half)')
        y1 = torch.empty((n, c1), dtype=torch.float32, device=dev)
        index = plan.coors_map
        rc = lib.sst_tall_linear_add_rows_f32x6(_lib.ptr(pf0), c0, _lib.ptr(w1), w1.stride(0), n, c0, c1, _lib.ptr(t), c1,
                                                _lib.ptr(index), _lib.ptr(y1), c1, st())
        _lib.check(rc, 'sst_tall_linear_add_rows_f32x6')
        prep1, bs1, cnt1, sync1 = bn_prepare(bn1, y1)
        out, arg1 = plan.raw_max(y1, scale_shift=(prep1[2], prep1[3]))
        ctx.save_for_backward(x, y0, pf0, pooled0, arg0, y1, arg1, prep0, prep1, w0, w1, index)
        ctx.plan = plan
        ctx.cfg = ((bs0, cnt0, sync0), (bs1, cnt1, sync1))
        return out

    @staticmethod
    def backward(ctx, dout):
        from torch import distributed as dist
        x, y0, pf0, pooled0, arg0, y1, arg1, prep0, prep1, w0, w1, index = ctx.saved_tensors
        plan = ctx.plan
        (bs0, cnt0, sync0), (bs1, cnt1, sync1) = ctx.cfg
        lib = _lib.load()
        st = _lib.stream_ptr
        n, k = x.shape
        c0, c1 = w0.size(0), w1.size(0)
        dev = x.device
        dout = dout.contiguous()

        def bn_pool_bwd(dy, xin, c, prep, arg, dpool, batch_stats, count, sync):
            sums = torch.empty((2, c), dtype=torch.float32, device=dev)
            ws = _lib.workspace(lib.sst_bn_workspace_bytes(n, c), dev)
            rc = lib.sst_bn_act_pool_bwd_reduce_f32(_lib.ptr(dy), _lib.ptr(xin), n, c, c, c, _lib.ptr(prep[0]), _lib.ptr(prep[1]),
                                                    _lib.ptr(prep[2]), _lib.ptr(prep[3]), 1, _lib.ptr(index), _lib.ptr(arg),
                                                    _lib.ptr(dpool), dpool.stride(0), _lib.ptr(sums[0]), _lib.ptr(sums[1]),
                                                    _lib.ptr(ws), st())
            _lib.check(rc, 'sst_bn_act_pool_bwd_reduce_f32')
            total = sums
            if sync and batch_stats:
                total = sums.clone()   # the parameter gradients stay the local sums (DDP averages them afterwards)
                dist.all_reduce(total, async_op=False)
            dx = torch.empty((n, c), dtype=torch.float32, device=dev)
            rc = lib.sst_bn_act_pool_bwd_apply_f32(_lib.ptr(dy), _lib.ptr(xin), n, c, c, c, _lib.ptr(prep[0]), _lib.ptr(prep[1]),
                                                   _lib.ptr(prep[2]), _lib.ptr(prep[3]), _lib.ptr(total[0]), _lib.ptr(total[1]),
                                                   (1.0 / count) if batch_stats else 0.0, 1, _lib.ptr(index), _lib.ptr(arg),
                                                   _lib.ptr(dpool), dpool.stride(0), _lib.ptr(dx), c, st())
            _lib.check(rc, 'sst_bn_act_pool_bwd_apply_f32')
            return dx, sums

        # layer 1: pooling + ReLU + batch norm backward from the voxel gradient alone
        dy1, sums1 = bn_pool_bwd(None, y1, c1, prep1, arg1, dout, bs1, cnt1, sync1)
        dt = plan.group_sum(dy1)                                   # d(pooled0 W1b^T): segmented sum over the voxels (+ row-0 quirk)
        m = getattr(plan, 'num_voxels', None)
        if m is None:
            m = int(plan.d_counts[0].item())
        dw1a, _ = weight_bias_grad(dy1, pf0, False)                # [c1, c0]
        dw1b, _ = weight_bias_grad(dt[:m], pooled0[:m], False)     # [c1, c0]
        dw1 = torch.cat([dw1a, dw1b], dim=1)
        dpf0 = _x6_linear(dy1, _lib.ptr(w1), w1.stride(0), 1, n, c1, c0, 'sst_tall_linear_epi_f32x6 (d point half)')
        dpooled0 = _x6_linear(dt, _float_ptr(w1, c0), w1.stride(0), 1, dt.size(0), c1, c0,
                              'sst_tall_linear_epi_f32x6 (d pooled half)')
        # layer 0: dense part + the pooling's share
        dy0, sums0 = bn_pool_bwd(dpf0, y0, c0, prep0, arg0, dpooled0, bs0, cnt0, sync0)
        dw0, _ = weight_bias_grad(dy0, x, False)
        return None, dw0, sums0[1], sums0[0], dw1, sums1[1], sums1[0], None, None, None


def fused_vfe2(encoder, x, plan):
    l0, l1 = encoder.vfe_layers
    return FusedVFE2.apply(x, l0.linear.weight, l0.norm.weight, l0.norm.bias, l1.linear.weight, l1.norm.weight, l1.norm.bias,
                           plan, l0.norm, l1.norm)


class UniquePlanAdapter(object):
    """what FusedVFE2 needs of a grouping, on a plain sorted-unique (kernels.UniquePlan): the grouping of scatter_v2 /
    DynamicScatterVFE (ops/sst/sst_ops.py:151-182) - every group is kept, no point is discarded"""

    def __init__(self, plan):
        self.plan = plan
        self.coors_map = plan.inverse
        self.num_voxels = plan.m

    def raw_max(self, feats, scale_shift=None):
        p = self.plan
        return K._segment_reduce_fwd(feats, p.perm, p.offsets, p.m, K.REDUCE['max'], True, None, None, p, scale_shift)

    def group_sum(self, part):
        return K.segment_reduce(part, self.plan, 'sum')
