"""Drop-in replacements for the NATIVE modules the reference's Python imports - integration path B of INTEGRATION.md:
every reference .py file stays as it is, only the compiled extensions are swapped.

    import sst_amd.native_shims as shims, sys
    sys.modules['mmdet3d.ops.voxel.voxel_layer'] = shims.voxel_layer            # voxelization.cpp:6-11 (pybind module)
    sys.modules['ingroup_indices'] = shims.ingroup_indices                      # TorchEx, call site sst_ops.py:249-257
    sys.modules['dynamic_point_pool_ext'] = shims.dynamic_point_pool_ext        # TorchEx, dynamic_point_pool_op.py:36, 86-88
    sys.modules['mmdet3d.ops.spconv.sparse_conv_ext'] = shims.sparse_conv_ext   # ops/spconv/ops.py:93-183
    sys.modules['torch_scatter'] = shims.torch_scatter                          # sst_ops.py:172-177

Each namespace has exactly the functions, argument orders and in / out conventions of the module it stands for (outputs
the reference pre-allocates are written in place).  Everything runs on the GPU through the C ABI of libsst_amd.so
(`sst_amd/_lib.py`); there is no CPU path, as in the reference (`voxelization.h:106`).  Executed by
tests/test_gpu_native_shims.py.
"""
import types

import torch

from . import _lib
from . import kernels as K
from . import spconv as _spconv
from . import voxel as _voxel
from .dynamic_point_pool import _pool
from .sst_ops import plan_of_inverse


# ------------------------------------------------------------------------------------------------------------------
# mmdet3d/ops/voxel/voxel_layer  (src/voxelization.h:71-128)
# ------------------------------------------------------------------------------------------------------------------
def _dynamic_point_to_voxel_backward(grad_feats, grad_reduced_feats, feats, reduced_feats, coors_idx, reduce_count,
                                     reduce_type):
    """scatter_points_cuda.cu:236-303: writes grad_feats in place.  coors_idx = the forward's coors_map (-1: the point's
    voxel was discarded), reduce_count = points per kept voxel.  MAX: the gradient of a (voxel, channel) goes to the
    point with the smallest index whose feature equals the maximum (max_reduce_traceback_scatter_idx_kernel)."""
    if reduce_type not in ('max', 'sum', 'mean'):
        raise RuntimeError('do not support reduce type ' + str(reduce_type))
    _lib.require_cuda(grad_feats, grad_reduced_feats, feats, reduced_feats, coors_idx)
    n, c = feats.shape
    grad_feats.zero_()
    if n == 0 or reduced_feats.size(0) == 0:
        return
    valid = coors_idx >= 0
    row = coors_idx.clamp(min=0).long()
    g = grad_reduced_feats[row]
    if reduce_type == 'mean':
        g = g / reduce_count[row].to(g.dtype).unsqueeze(1)
    if reduce_type in ('sum', 'mean'):
        grad_feats.copy_(torch.where(valid.unsqueeze(1), g, torch.zeros_like(g)))
        return
    ar = torch.arange(n, device=feats.device).unsqueeze(1).expand(n, c)
    hit = (feats == reduced_feats[row]) & valid.unsqueeze(1)
    cand = torch.where(hit, ar, torch.full_like(ar, n))
    first = torch.full((reduced_feats.size(0), c), n, dtype=cand.dtype, device=feats.device)
    first.scatter_reduce_(0, row.unsqueeze(1).expand(n, c), cand, 'amin', include_self=True)
    take = (first[row] == ar) & valid.unsqueeze(1)
    grad_feats.copy_(torch.where(take, g, torch.zeros_like(g)))


voxel_layer = types.SimpleNamespace(
    dynamic_voxelize=_voxel.dynamic_voxelize,                               # (points, coors, voxel_size, coors_range, NDim=3)
    hard_voxelize=_voxel.hard_voxelize,
    dynamic_point_to_voxel_forward=_voxel.dynamic_point_to_voxel_forward,   # -> [reduced, out_coors, coors_map, count]
    dynamic_point_to_voxel_backward=_dynamic_point_to_voxel_backward)


# ------------------------------------------------------------------------------------------------------------------
# TorchEx ingroup_indices
# ------------------------------------------------------------------------------------------------------------------
def _ingroup_forward(group_inds, out_inds):
    """out_inds (int64, pre-filled with -1 by the caller) <- rank of every element inside its group"""
    out_inds.copy_(K.ingroup_rank(group_inds.contiguous()))


ingroup_indices = types.SimpleNamespace(forward=_ingroup_forward)


# ------------------------------------------------------------------------------------------------------------------
# TorchEx dynamic_point_pool_ext
# ------------------------------------------------------------------------------------------------------------------
def _dpp_forward(rois, pts, extra_wlh, max_inbox_point, out_pts_idx, out_roi_idx, out_pts_feats, rois_batch=None,
                 pts_batch=None):
    """outputs are the caller's pre-filled tensors (-1, -1, zeros): the pairs found are written as a prefix, the rest
    stays untouched - the reference then masks out_pts_idx >= 0 (dynamic_point_pool_op.py:38-40)"""
    a, b, f, n = _pool(rois, rois_batch, pts, pts_batch, extra_wlh, max_inbox_point, out_pts_idx.numel())
    out_pts_idx.copy_(a)
    out_roi_idx.copy_(b)
    out_pts_feats.copy_(f)
    return n


def _dpp_mixed(rois, rois_batch, pts, pts_batch, extra_wlh, max_inbox_point, out_pts_idx, out_roi_idx, out_pts_feats):
    return _dpp_forward(rois, pts, extra_wlh, max_inbox_point, out_pts_idx, out_roi_idx, out_pts_feats, rois_batch,
                        pts_batch)


dynamic_point_pool_ext = types.SimpleNamespace(forward=_dpp_forward, dynamic_point_pool_mixed_gpu=_dpp_mixed)


# ------------------------------------------------------------------------------------------------------------------
# mmdet3d/ops/spconv/sparse_conv_ext  (3-D, int32 indices, fp32; ops/spconv/ops.py:93-183)
# ------------------------------------------------------------------------------------------------------------------
def _get_indice_pairs_3d(indices, batch_size, out_shape, spatial_shape, ksize, stride, padding, dilation, out_padding,
                         subm, transpose):
    return list(_spconv.get_indice_pairs(indices, batch_size, spatial_shape, ksize, stride, padding, dilation,
                                         out_padding, bool(subm), bool(transpose)))


def _indice_conv_fp32(features, filters, indice_pairs, indice_pair_num, num_activate_out, inverse, subm):
    return _spconv.indice_conv(features, filters, indice_pairs, indice_pair_num, num_activate_out, bool(inverse),
                               bool(subm))


def _indice_conv_backward_fp32(features, filters, out_bp, indice_pairs, indice_pair_num, inverse, subm):
    return list(_spconv.indice_conv_backward(features, filters, out_bp, indice_pairs, indice_pair_num, bool(inverse),
                                             bool(subm)))


sparse_conv_ext = types.SimpleNamespace(
    get_indice_pairs_3d=_get_indice_pairs_3d, indice_conv_fp32=_indice_conv_fp32,
    indice_conv_backward_fp32=_indice_conv_backward_fp32, indice_maxpool_fp32=_spconv.indice_maxpool,
    indice_maxpool_backward_fp32=_spconv.indice_maxpool_backward)


# ------------------------------------------------------------------------------------------------------------------
# torch_scatter: the two functions sst_ops.py:172-177 calls (dim = 0, index = the inverse map of a unique)
# ------------------------------------------------------------------------------------------------------------------
def _groups_of(index, src):
    if index.dim() != 1 or index.numel() != src.size(0):
        raise RuntimeError('torch_scatter shim: a 1-D index over dim 0 is expected (the reference passes unq_inv)')
    m = int(index.max().item()) + 1 if index.numel() else 0
    return plan_of_inverse(index, m), m


def _scatter_max(src, index, dim=0):
    """-> (out [G, C], argmax [G, C] int64): row index of the first point attaining the maximum"""
    assert dim == 0
    flat = src.reshape(src.size(0), -1).contiguous().float()
    plan, m = _groups_of(index, flat)
    out, arg = K.segment_argmax(flat, plan)
    return out.reshape((m,) + src.shape[1:]).to(src.dtype), arg.long().reshape((m,) + src.shape[1:])


def _scatter(src, index, dim=0, reduce='sum'):
    assert dim == 0
    if reduce not in ('sum', 'add', 'mean'):
        raise NotImplementedError(reduce)
    flat = src.reshape(src.size(0), -1).contiguous().float()
    plan, m = _groups_of(index, flat)
    out = K.segment_reduce(flat, plan, 'mean' if reduce == 'mean' else 'sum')
    return out.reshape((m,) + src.shape[1:]).to(src.dtype)


torch_scatter = types.SimpleNamespace(scatter_max=_scatter_max, scatter=_scatter)
