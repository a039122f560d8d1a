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


@torch.no_grad()
def get_flat2win_inds(batch_win_inds, voxel_drop_lvl, drop_info, debug=True):
    flat2window_inds_dict = {}
    for dl in drop_info:
        dl_mask = voxel_drop_lvl == dl
        if not dl_mask.any():
            continue
        conti_win_inds = make_continuous_inds(batch_win_inds[dl_mask])
        max_tokens = drop_info[dl]['max_tokens']
        inner_win_inds = get_inner_win_inds(conti_win_inds)
        flat2window_inds = conti_win_inds * max_tokens + inner_win_inds
        flat2window_inds_dict[dl] = (flat2window_inds, torch.where(dl_mask))
        if debug:
            num_windows = len(torch.unique(conti_win_inds))
            assert inner_win_inds.max() < max_tokens, \
                f'Max inner inds({inner_win_inds.max()}) larger(equal) than {max_tokens}'
            assert (flat2window_inds >= 0).all()
            max_ind = flat2window_inds.max().item()
            assert max_ind < num_windows * max_tokens, \
                f'max_ind({max_ind}) larger than upper bound({num_windows * max_tokens})'
            assert max_ind >= (num_windows - 1) * max_tokens, \
                f'max_ind({max_ind}) less than lower bound({(num_windows - 1) * max_tokens})'
    return flat2window_inds_dict


def flat2window(feat, voxel_drop_lvl, flat2win_inds_dict, drop_info, padding=0):
    """[N,C] -> {level: [num_windows, max_tokens, C]} (sst_ops.py:67-104)."""
    dtype = feat.dtype
    device = feat.device
    feat_dim = feat.shape[-1]
    feat_3d_dict = {}
    for dl in drop_info:
        dl_mask = voxel_drop_lvl == dl
        if not dl_mask.any():
            continue
        feat_this_dl = feat[dl_mask]
        this_inds = flat2win_inds_dict[dl][0]
        max_tokens = drop_info[dl]['max_tokens']
        num_windows = (this_inds // max_tokens).max().item() + 1
        feat_3d = torch.full((num_windows * max_tokens, feat_dim), padding, dtype=dtype, device=device)
        feat_3d[this_inds] = feat_this_dl
        feat_3d_dict[dl] = feat_3d.reshape((num_windows, max_tokens, feat_dim))
    return feat_3d_dict


def window2flat(feat_3d_dict, inds_dict):
    num_all_voxel = 0
    for dl in inds_dict:
        num_all_voxel += inds_dict[dl][0].shape[0]
    first = feat_3d_dict[list(feat_3d_dict.keys())[0]]
    all_flat_feat = torch.zeros((num_all_voxel, first.shape[-1]), device=first.device, dtype=first.dtype)
    for dl in feat_3d_dict:
        feat = feat_3d_dict[dl]
        feat_dim = feat.shape[-1]
        inds, flat_pos = inds_dict[dl]
        feat = feat.reshape(-1, feat_dim)
        all_flat_feat[flat_pos] = feat[inds]
    return all_flat_feat


def get_flat2win_inds_v2(batch_win_inds, voxel_drop_lvl, drop_info, debug=True):
    transform_dict = get_flat2win_inds(batch_win_inds, voxel_drop_lvl, drop_info, debug)
    transform_dict['voxel_drop_level'] = voxel_drop_lvl
    transform_dict['batching_info'] = drop_info
    return transform_dict


def window2flat_v2(feat_3d_dict, inds_dict):
    inds_v1 = {k: inds_dict[k] for k in inds_dict if not isinstance(k, str)}
    return window2flat(feat_3d_dict, inds_v1)


def flat2window_v2(feat, inds_dict, padding=0):
    assert 'voxel_drop_level' in inds_dict, 'voxel_drop_level should be in inds_dict in v2 function'
    inds_v1 = {k: inds_dict[k] for k in inds_dict if not isinstance(k, str)}
    batching_info = inds_dict['batching_info']
    return flat2window(feat, inds_dict['voxel_drop_level'], inds_v1, batching_info, padding=padding)


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


def _plan_from_inverse(unq_inv, num_groups):
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
        plan = _plan_from_inverse(unq_inv, new_coors.size(0))
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
# MLP helpers (pure nn glue, sst_ops.py:334-391)
# --------------------------------------------------------------------------------------------
def build_mlp(in_channel, hidden_dims, norm_cfg, is_head=False, act='relu', bias=False, dropout=0):
    layer_list = []
    last_channel = in_channel
    if isinstance(hidden_dims, int):
        hidden_dims = [hidden_dims, ]
    for i, c in enumerate(hidden_dims):
        act_layer = get_activation_layer(act, c)
        norm_layer = build_norm_layer(norm_cfg, c)[1]
        if i == len(hidden_dims) - 1 and is_head:
            layer_list.append(nn.Linear(last_channel, c, bias=True), )
        else:
            sq = [nn.Linear(last_channel, c, bias=bias), norm_layer, act_layer]
            if dropout > 0:
                sq.append(nn.Dropout(dropout))
            layer_list.append(nn.Sequential(*sq))
        last_channel = c
    return nn.Sequential(*layer_list)


def get_activation(activation):
    if activation == "relu":
        return torch.nn.functional.relu
    if activation == "gelu":
        return torch.nn.functional.gelu
    if activation == "glu":
        return torch.nn.functional.glu
    raise RuntimeError(F"activation should be relu/gelu, not {activation}.")


def get_activation_layer(act, dim=None):
    act = act.lower()
    if act == 'relu':
        return nn.ReLU(inplace=True)
    if act == 'gelu':
        return nn.GELU()
    if act == 'leakyrelu':
        return nn.LeakyReLU(inplace=True)
    if act == 'prelu':
        return nn.PReLU(num_parameters=dim)
    if act in ('swish', 'silu'):
        return nn.SiLU(inplace=True)
    if act == 'glu':
        return nn.GLU()
    if act == 'elu':
        return nn.ELU(inplace=True)
    raise NotImplementedError
