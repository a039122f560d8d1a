"""Host-side mirror of mmdet3d/ops/sst/sst_ops.py (same function names, argument meaning and return
types), with the index arithmetic done by the HIP kernels of libsst_amd.so.

Reference lines: get_flat2win_inds :26-64, flat2window :67-104, window2flat :106-132, *_v2 :134-149,
scatter_v2 :151-182, get_inner_win_inds (TorchEx ingroup_indices) :244-264, get_window_coors :266-314,
make_continuous_inds :316-331, build_mlp :334-361, get_activation(_layer) :363-391.

These functions keep the reference's tensor-dict API (per-drop-level dictionaries, int64 indices) so
code written against mmdet3d.ops keeps working.  The SST backbone of this package does NOT go through
them: it consumes the window CSR produced by kernels.region_batching directly (no padded [W,T,C]
tensors).
"""
import traceback

import torch
import torch.nn as nn

from . import kernels as K
from .norm import build_norm_layer


@torch.no_grad()
def get_window_coors(coors, sparse_shape, window_shape, do_shift):
    """coors [M,4] (b,z,y,x) long -> (batch_win_inds [M] long, coors_in_win [M,3] long (z,y,x))."""
    if len(window_shape) == 2:
        win_shape_x, win_shape_y = window_shape
        win_shape_z = sparse_shape[-1]
    else:
        win_shape_x, win_shape_y, win_shape_z = window_shape
    sparse_shape_x, sparse_shape_y, sparse_shape_z = sparse_shape
    assert sparse_shape_z < sparse_shape_x, 'Usually holds... in case of wrong order'
    win0, ciw0, win1, ciw1 = K.window_coors(coors.contiguous(), [sparse_shape_x, sparse_shape_y, sparse_shape_z],
                                            [win_shape_x, win_shape_y, win_shape_z])
    if do_shift:
        return win1.long(), ciw1.long()
    return win0.long(), ciw0.long()


def get_inner_win_inds(win_inds):
    """IngroupIndicesFunction.apply: rank of every element inside its group (stable order)."""
    with torch.no_grad():
        return K.ingroup_rank(win_inds.contiguous())


@torch.no_grad()
def make_continuous_inds(inds):
    """Compress ids to 0..K-1 preserving order (sst_ops.py:316-331)."""
    if inds.numel() == 0:
        return inds.clone()
    plan = K.unique_rows(inds.reshape(-1, 1).contiguous())
    return plan.inverse.to(inds.dtype)


def _levels_present(voxel_drop_lvl, drop_info):
    """[(level key, int64 positions of its voxels)] for the drop levels that occur, in the order of ``drop_info``"""
    out = []
    for key in drop_info:
        where = torch.nonzero(voxel_drop_lvl == key).squeeze(1)
        if where.numel() > 0:
            out.append((key, where))
    return out


@torch.no_grad()
def get_flat2win_inds(batch_win_inds, voxel_drop_lvl, drop_info, debug=True):
    """{level: (slot of every voxel of the level in its padded [W * T] window tensor, (positions of those voxels,))}
    (sst_ops.py:26-64): windows of a level numbered 0..W-1 in ascending window-id order, slot = window * T + rank of
    the voxel inside its window (ascending voxel index)."""
    table = {}
    for key, where in _levels_present(voxel_drop_lvl, drop_info):
        cap = drop_info[key]['max_tokens']
        win = make_continuous_inds(batch_win_inds[where])
        rank = get_inner_win_inds(win)
        slots = win * cap + rank
        table[key] = (slots, (where,))
        if debug:
            n_win = int(win.max().item()) + 1
            assert int(rank.max().item()) < cap, f'Max inner inds({int(rank.max())}) larger(equal) than {cap}'
            top = int(slots.max().item())
            assert int(slots.min().item()) >= 0 and (n_win - 1) * cap <= top < n_win * cap, \
                f'slot range of level {key} inconsistent: max {top}, {n_win} windows of {cap} tokens'
    return table


class _ScatterRowsFn(torch.autograd.Function):
    """out[slots[i]] = rows[i] on a [n_slots, C] tensor filled with ``padding`` (slots unique): sst_scatter_rows_f32;
    backward = the row gather of the upstream gradient (sst_gather_rows_f32).  `feat_3d[this_inds] = feat` of
    ops/sst/sst_ops.py:98 with its autograd."""

    @staticmethod
    def forward(ctx, rows, slots, n_slots, padding):
        out = torch.full((n_slots, rows.shape[-1]), padding, dtype=rows.dtype, device=rows.device)
        K.scatter_rows(rows.contiguous(), slots, out)
        ctx.save_for_backward(slots)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        slots, = ctx.saved_tensors
        return K.gather_rows(grad_out.contiguous(), slots), None, None, None


class _GatherRowsFn(torch.autograd.Function):
    """rows2d[slots] (slots unique): sst_gather_rows_f32; backward = row scatter of the upstream gradient into zeros.
    `feat[inds]` of ops/sst/sst_ops.py:124-125 with its autograd."""

    @staticmethod
    def forward(ctx, rows2d, slots):
        ctx.save_for_backward(slots)
        ctx.n_src = rows2d.size(0)
        return K.gather_rows(rows2d.contiguous(), slots)

    @staticmethod
    def backward(ctx, grad_out):
        slots, = ctx.saved_tensors
        grad_src = torch.zeros((ctx.n_src, grad_out.size(1)), dtype=grad_out.dtype, device=grad_out.device)
        K.scatter_rows(grad_out.contiguous(), slots, grad_src)
        return grad_src, None


def _kernel_rows(t):
    return t.is_cuda and t.dtype == torch.float32 and t.dim() == 2


def _rows_to_slots(rows, slots, n_slots, padding):
    """out[slots[i]] = rows[i] on a [n_slots, C] tensor filled with ``padding``: the row-scatter kernel for fp32 CUDA
    features - with autograd, so training keeps the kernel -, index assignment otherwise (integer / bool payloads of the
    reference-format dictionaries)"""
    if _kernel_rows(rows):
        return _ScatterRowsFn.apply(rows, slots.to(torch.int32).contiguous(), int(n_slots), float(padding))
    out = torch.full((n_slots, rows.shape[-1]), padding, dtype=rows.dtype, device=rows.device)
    out[slots] = rows
    return out


def flat2window(feat, voxel_drop_lvl, flat2win_inds_dict, drop_info, padding=0):
    """[N, C] -> {level: [W, T, C]} padded window tensors (sst_ops.py:67-104)."""
    windows = {}
    for key, where in _levels_present(voxel_drop_lvl, drop_info):
        slots = flat2win_inds_dict[key][0]
        cap = drop_info[key]['max_tokens']
        n_win = int(torch.div(slots, cap, rounding_mode='floor').max().item()) + 1
        windows[key] = _rows_to_slots(feat[where], slots, n_win * cap, padding).reshape(n_win, cap, feat.shape[-1])
    return windows


def window2flat(feat_3d_dict, inds_dict):
    """{level: [W, T, C]} -> [N, C]: every voxel reads its slot back (sst_ops.py:106-132)."""
    total = sum(inds_dict[key][0].shape[0] for key in inds_dict)
    any_level = next(iter(feat_3d_dict.values()))
    flat = torch.zeros((total, any_level.shape[-1]), device=any_level.device, dtype=any_level.dtype)
    for key, padded in feat_3d_dict.items():
        slots, where = inds_dict[key]
        rows2d = padded.reshape(-1, padded.shape[-1])
        if _kernel_rows(rows2d):
            flat[where] = _GatherRowsFn.apply(rows2d, slots.to(torch.int32).contiguous())
        else:
            flat[where] = rows2d[slots]
    return flat


def _numeric_levels(inds_dict):
    return {key: val for key, val in inds_dict.items() if not isinstance(key, str)}


def get_flat2win_inds_v2(batch_win_inds, voxel_drop_lvl, drop_info, debug=True):
    """v2 dictionaries additionally carry the level of every voxel and the batching table (sst_ops.py:134-139)."""
    table = get_flat2win_inds(batch_win_inds, voxel_drop_lvl, drop_info, debug)
    table.update(voxel_drop_level=voxel_drop_lvl, batching_info=drop_info)
    return table


def window2flat_v2(feat_3d_dict, inds_dict):
    return window2flat(feat_3d_dict, _numeric_levels(inds_dict))


def flat2window_v2(feat, inds_dict, padding=0):
    assert 'voxel_drop_level' in inds_dict, 'voxel_drop_level should be in inds_dict in v2 function'
    return flat2window(feat, inds_dict['voxel_drop_level'], _numeric_levels(inds_dict), inds_dict['batching_info'],
                       padding=padding)


# --------------------------------------------------------------------------------------------
# scatter_v2
# --------------------------------------------------------------------------------------------
def _attach_plan(unq_inv, plan):
    unq_inv._sst_plan = plan  # lets later scatter_v2(..., unq_inv=...) calls reuse the CSR (unique_once)
    return unq_inv


def unique_with_plan(coors, return_counts=False):
    """torch.unique(coors, return_inverse=True, dim=0) computed by the radix-sort unique kernel.
    Returns (new_coors, unq_inv[, counts]); unq_inv carries the CSR plan for scatter_v2 reuse."""
    coors = coors.contiguous()
    plan = K.unique_rows(coors)
    new_coors = K.unpack_unique_rows(plan, coors.dtype)
    unq_inv = _attach_plan(plan.inverse.long(), plan)
    if return_counts:
        return new_coors, unq_inv, plan.counts().long()
    return new_coors, unq_inv


def plan_of_inverse(unq_inv, num_groups):
    """the CSR (kernels.UniquePlan) behind an inverse map: the one it was produced with, or rebuilt from the ids"""
    plan = getattr(unq_inv, '_sst_plan', None)
    if plan is not None and plan.m == num_groups:
        return plan
    # an inverse produced elsewhere (torch.unique, a slice / clone of ours): regroup by the ids themselves.  Like
    # torch_scatter (output sized by the number of groups, absent ids keep an empty row) every id 0..num_groups-1
    # gets a group: one sentinel row per id is appended, so the CSR has exactly num_groups groups in id order and
    # the sentinels (positions >= N) are dropped from the permutation again.
    n = unq_inv.numel()
    ids = torch.cat([unq_inv.reshape(-1), torch.arange(num_groups, device=unq_inv.device, dtype=unq_inv.dtype)])
    full = K.unique_rows(ids.reshape(-1, 1).contiguous(), [0], [max(int(num_groups), 1)])
    assert full.m == num_groups, 'unq_inv holds an id outside [0, len(new_coors))'
    keep = full.perm < n                                   # stable sort: a group's sentinel is its last member
    plan = K.UniquePlan()
    plan.n, plan.m, plan.ncols = n, num_groups, 1
    plan.mins, plan.extents = full.mins, full.extents
    plan.perm = full.perm[keep].contiguous()
    plan.inverse = full.inverse[:n].contiguous()
    plan.offsets = (full.offsets[:num_groups + 1]
                    - torch.arange(num_groups + 1, device=ids.device, dtype=torch.int32)).contiguous()
    plan.ukeys = full.ukeys
    _attach_plan(unq_inv, plan)
    return plan


def scatter_v2(feat, coors, mode, return_inv=True, min_points=0, unq_inv=None, new_coors=None):
    """sst_ops.py:151-182: unique (unless unq_inv is given) + segmented max / mean / sum."""
    assert feat.size(0) == coors.size(0)
    if mode == 'avg':
        mode = 'mean'
    plan = None
    if unq_inv is None:
        coors_c = coors.contiguous()
        plan = K.unique_rows(coors_c)
        new_coors = K.unpack_unique_rows(plan, coors.dtype)
        unq_inv = _attach_plan(plan.inverse.long(), plan)
        unq_cnt = plan.counts().long() if min_points > 0 else None
    else:
        assert new_coors is not None, \
            'please pass new_coors for interface consistency, caller: {}'.format(traceback.extract_stack()[-2][2])
        plan = plan_of_inverse(unq_inv, new_coors.size(0))
        unq_cnt = plan.counts().long() if min_points > 0 else None

    if min_points > 0:
        cnt_per_point = unq_cnt[unq_inv]
        valid_mask = cnt_per_point >= min_points
        feat = feat[valid_mask]
        coors = coors[valid_mask].contiguous()
        plan = K.unique_rows(coors)
        new_coors = K.unpack_unique_rows(plan, coors.dtype)
        unq_inv = _attach_plan(plan.inverse.long(), plan)

    if mode not in ('max', 'mean', 'sum'):
        raise NotImplementedError
    new_feat = K.segment_reduce(feat.contiguous(), plan, mode)

    if not return_inv:
        return new_feat, new_coors
    return new_feat, new_coors, unq_inv


# --------------------------------------------------------------------------------------------
# MLP helpers (sst_ops.py:334-391): pure nn glue; the Sequential layout fixes the state_dict keys (``{i}.0.weight`` ...)
# --------------------------------------------------------------------------------------------
_ACTIVATION_LAYERS = {
    'relu': lambda dim: nn.ReLU(inplace=True),
    'gelu': lambda dim: nn.GELU(),
    'leakyrelu': lambda dim: nn.LeakyReLU(inplace=True),
    'prelu': lambda dim: nn.PReLU(num_parameters=dim),
    'swish': lambda dim: nn.SiLU(inplace=True),
    'silu': lambda dim: nn.SiLU(inplace=True),
    'glu': lambda dim: nn.GLU(),
    'elu': lambda dim: nn.ELU(inplace=True),
}


def get_activation_layer(act, dim=None):
    try:
        return _ACTIVATION_LAYERS[act.lower()](dim)
    except KeyError:
        raise NotImplementedError(act)


def get_activation(activation):
    fns = {'relu': torch.nn.functional.relu, 'gelu': torch.nn.functional.gelu, 'glu': torch.nn.functional.glu}
    if activation not in fns:
        raise RuntimeError(F"activation should be relu/gelu, not {activation}.")
    return fns[activation]


class _MLPStage(nn.Sequential):
    """[Linear, norm, act (, dropout)] with the children and parameter names of the nn.Sequential the reference builds;
    LayerNorm + activation run as one pass (csrc/dense.hip) on CUDA float32 features."""

    def forward(self, x):
        norm = self[1]
        if isinstance(norm, nn.LayerNorm) and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32:
            from .dense import add_layer_norm
            y = add_layer_norm(self[0](x), None, norm, act=self[2])
            for extra in list(self)[3:]:
                y = extra(y)
            return y
        return super().forward(x)


def build_mlp(in_channel, hidden_dims, norm_cfg, is_head=False, act='relu', bias=False, dropout=0):
    """Sequential of [Linear -> norm -> act (-> dropout)] stages; with ``is_head`` the last stage is a bare Linear
    with bias (sst_ops.py:334-361)."""
    dims = [hidden_dims] if isinstance(hidden_dims, int) else list(hidden_dims)
    stages, width = [], in_channel
    for i, out in enumerate(dims):
        if is_head and i == len(dims) - 1:
            stages.append(nn.Linear(width, out, bias=True))
        else:
            # activation, then norm: the construction order of the reference, which decides the RNG draws of the init
            act_layer = get_activation_layer(act, out)
            norm_layer = build_norm_layer(norm_cfg, out)[1]
            parts = [nn.Linear(width, out, bias=bias), norm_layer, act_layer]
            if dropout > 0:
                parts.append(nn.Dropout(dropout))
            stages.append(_MLPStage(*parts))
        width = out
    return nn.Sequential(*stages)
