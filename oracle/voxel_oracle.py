"""TEST INFRASTRUCTURE ONLY — CPU restatement of dynamic voxelization and DynamicScatter.

dynamic_voxelize   follows mmdet3d/ops/voxel/src/voxelization_cpu.cpp:7-41,144-169 (== the CUDA kernel
                   voxelization_cuda.cu:24-65): fp32 (p - min) / v, floor, clamp to [0, grid-1], (z,y,x).
dynamic scatter    follows mmdet3d/ops/voxel/src/scatter_points_cuda.cu:183-234 (forward) and :236-303
                   (backward); python wrapper mmdet3d/ops/voxel/scatter_points.py:9-107.
"""
import numpy as np
import torch
from torch import nn
from torch.autograd import Function


def dynamic_voxelize_grid(voxel_size, coors_range):
    v = np.asarray(voxel_size, dtype=np.float32)
    r = np.asarray(coors_range, dtype=np.float32)
    return np.ceil((r[3:] - r[:3]) / v).astype(np.int64)  # (gx, gy, gz), voxelization_cpu.cpp:155-158


def dynamic_voxelize(points, voxel_size, coors_range):
    """points [N, >=3] float32 (numpy or torch CPU) -> int32 [N,3] (z,y,x)."""
    p = np.asarray(points, dtype=np.float32)[:, :3]
    v = np.asarray(voxel_size, dtype=np.float32)
    lo = np.asarray(coors_range[:3], dtype=np.float32)
    grid = dynamic_voxelize_grid(voxel_size, coors_range)
    q = np.floor((p - lo[None, :]) / v[None, :])           # fp32 subtract, fp32 divide, floor
    q = np.clip(q, -1.0, grid[None, :].astype(np.float32))  # keep the int cast in range, then the real clamp
    c = q.astype(np.int64)
    c = np.clip(c, 0, grid[None, :] - 1)                    # this fork: clamp, never -1 (cpp:23-31)
    return np.ascontiguousarray(c[:, ::-1]).astype(np.int32)


def dynamic_point_to_voxel_forward(feats, coors, reduce_type, reference_compat=True):
    """torch CPU tensors.  Returns [reduced_feats, out_coors, coors_map(int32), reduce_count(int32)]."""
    assert reduce_type in ('max', 'sum', 'mean')
    n, c = feats.shape
    if n == 0:  # scatter_points_cuda.cu:192-196
        return [feats.clone().detach(), coors.clone().detach(), coors.new_empty((0,), dtype=torch.int32),
                coors.new_empty((0,), dtype=torch.int32)]
    clean = coors.masked_fill(coors.lt(0).any(-1, True), -1)                       # :200
    out_coors, coors_map, cnt = torch.unique(clean, dim=0, sorted=True, return_inverse=True,
                                             return_counts=True)                  # :202-205
    if reference_compat:
        drop = 1                                                                   # :207-210 unconditional
    else:
        drop = 1 if bool((out_coors[0] < 0).any()) else 0
    out_coors = out_coors[drop:]
    cnt = cnt[drop:].to(torch.int32)
    coors_map = coors_map.to(torch.int32) - drop
    m = out_coors.size(0)
    valid = coors_map >= 0
    idx = coors_map[valid].long().view(-1, 1).expand(-1, c)
    if reduce_type == 'max':
        red = torch.full((m, c), float('-inf'), dtype=feats.dtype)
        red = red.scatter_reduce(0, idx, feats[valid], reduce='amax', include_self=True)
    else:
        red = torch.zeros((m, c), dtype=feats.dtype)
        red = red.scatter_reduce(0, idx, feats[valid], reduce='sum', include_self=True)
        if reduce_type == 'mean':
            red = red / cnt.unsqueeze(-1).to(red.dtype)                            # :228-229
    return [red, out_coors, coors_map, cnt]


def dynamic_point_to_voxel_backward(grad_reduced, feats, reduced, coors_map, reduce_count, reduce_type):
    """Returns grad_feats (scatter_points_cuda.cu:236-303)."""
    n, c = feats.shape
    m = reduced.size(0)
    grad_feats = torch.zeros_like(feats)
    if n == 0 or m == 0:
        return grad_feats
    valid = coors_map >= 0
    vmap = coors_map.long().clamp(min=0)
    if reduce_type in ('sum', 'mean'):
        g = grad_reduced[vmap]
        if reduce_type == 'mean':
            g = g / reduce_count[vmap].unsqueeze(-1).to(g.dtype)
        grad_feats[valid] = g[valid]
        return grad_feats
    # max: gradient goes to the SMALLEST point index whose feature equals the max (atomicMin, :154-157)
    pt = torch.arange(n).view(-1, 1).expand(-1, c)
    is_max = (feats == reduced[vmap]) & valid.view(-1, 1)
    cand = torch.where(is_max, pt, torch.full_like(pt, n))
    reduce_from = torch.full((m, c), n, dtype=torch.long)
    reduce_from = reduce_from.scatter_reduce(0, vmap.view(-1, 1).expand(-1, c)[valid], cand[valid], reduce='amin',
                                             include_self=True)
    ch = torch.arange(c).view(1, -1).expand(m, -1)
    ok = reduce_from < n
    grad_feats[reduce_from[ok], ch[ok]] = grad_reduced[ok]
    return grad_feats


class _DynamicScatterFn(Function):

    @staticmethod
    def forward(ctx, feats, coors, reduce_type, reference_compat):
        red, out_coors, cmap, cnt = dynamic_point_to_voxel_forward(feats, coors, reduce_type, reference_compat)
        ctx.reduce_type = reduce_type
        ctx.save_for_backward(feats, red, cmap, cnt)
        ctx.mark_non_differentiable(out_coors)
        return red, out_coors

    @staticmethod
    def backward(ctx, g, _gc=None):
        feats, red, cmap, cnt = ctx.saved_tensors
        return dynamic_point_to_voxel_backward(g.contiguous(), feats, red, cmap, cnt, ctx.reduce_type), None, None, None


def dynamic_scatter(feats, coors, reduce_type='max', reference_compat=True):
    return _DynamicScatterFn.apply(feats, coors, reduce_type, reference_compat)


class DynamicScatterOracle(nn.Module):
    """CPU stand-in for mmdet3d.ops.DynamicScatter (scatter_points.py:53-107), including the per-sample loop."""

    def __init__(self, voxel_size, point_cloud_range, average_points, reference_compat=True):
        super().__init__()
        self.voxel_size = voxel_size
        self.point_cloud_range = point_cloud_range
        self.average_points = average_points
        self.reference_compat = reference_compat

    def forward_single(self, points, coors):
        reduce = 'mean' if self.average_points else 'max'
        return dynamic_scatter(points.contiguous(), coors.contiguous(), reduce, self.reference_compat)

    def forward(self, points, coors):
        if coors.size(-1) == 3:
            return self.forward_single(points, coors)
        batch_size = int(coors[-1, 0]) + 1
        voxels, voxel_coors = [], []
        for i in range(batch_size):
            inds = torch.where(coors[:, 0] == i)
            voxel, voxel_coor = self.forward_single(points[inds], coors[inds][:, 1:])
            voxel_coors.append(nn.functional.pad(voxel_coor, (1, 0), mode='constant', value=i))
            voxels.append(voxel)
        return torch.cat(voxels, dim=0), torch.cat(voxel_coors, dim=0)


def hard_voxelize(points, voxel_size, coors_range, max_points, max_voxels):
    """voxelization_cpu.cpp:43-100 + 102-142 without the sequential loop (the formulation the GPU path uses): dynamic
    voxelization clamped to the round() grid of :127-130, voxels numbered by the first appearance of one of their
    points and cut at max_voxels, the first max_points points of a voxel kept in input order.
    -> (voxels [V, max_points, C] zero padded, coors [V, 3] int32 (z, y, x), num_points [V] int32)."""
    p = np.asarray(points, dtype=np.float32)
    c = dynamic_voxelize(p, voxel_size, coors_range).astype(np.int64)
    v = np.asarray(voxel_size, dtype=np.float32)
    r = np.asarray(coors_range, dtype=np.float32)
    # C round(): halves away from zero (numpy / Python round to even), on the float32 quotient
    grid = np.floor(((r[3:] - r[:3]) / v).astype(np.float32) + np.float32(0.5)).astype(np.int64)   # (gx, gy, gz)
    c = np.minimum(c, grid[::-1][None, :] - 1)
    uniq, first, inv, counts = np.unique(c, axis=0, return_index=True, return_inverse=True, return_counts=True)
    inv = inv.reshape(-1)
    order = np.argsort(first, kind='stable')
    vrank = np.empty(len(uniq), dtype=np.int64)
    vrank[order] = np.arange(len(uniq))
    num = len(uniq) if max_voxels == -1 else min(len(uniq), max_voxels)
    # rank of every point inside its voxel, in input order
    by_voxel = np.argsort(inv, kind='stable')
    start = np.concatenate([[0], np.cumsum(counts)])[:-1]
    rank = np.empty(len(p), dtype=np.int64)
    rank[by_voxel] = np.arange(len(p)) - start[inv[by_voxel]]
    keep = (vrank[inv] < num) & ((rank < max_points) if max_points != -1 else True)
    width = max_points if max_points != -1 else int(counts.max())
    voxels = np.zeros((num, width, p.shape[1]), dtype=np.float32)
    voxels[vrank[inv][keep], rank[keep]] = p[keep]
    npts = counts if max_points == -1 else np.minimum(counts, max_points)
    return voxels, uniq[order[:num]].astype(np.int32), npts[order[:num]].astype(np.int32)
