"""Host-side mirror of mmdet3d/ops/voxel (Voxelization, DynamicScatter) backed by libsst_amd.so.

Reference interfaces mirrored here:
  Voxelization / voxelization      mmdet3d/ops/voxel/voxelize.py:10-122
  DynamicScatter / dynamic_scatter mmdet3d/ops/voxel/scatter_points.py:9-107
  voxel_layer.{dynamic_voxelize, dynamic_point_to_voxel_forward, dynamic_point_to_voxel_backward}
                                   mmdet3d/ops/voxel/src/voxelization.h:71-128 (pybind, voxelization.cpp:6-11)

Reference quirk reproduced by default (``reference_compat=True``): dynamic_point_to_voxel_forward always
discards row 0 of the sorted-unique voxel list (scatter_points_cuda.cu:207-210, "the first element of
out_coors is always (-1,-1,-1)").  With this fork's clamped voxelization no coordinate is negative, so
that row is a real voxel whose points get coors_map = -1.  ``reference_compat=False`` drops row 0 only
when it really is the invalid (-1,-1,-1) group.
"""
import torch
from torch import nn
from torch.nn.modules.utils import _pair

from . import kernels as K


# ------------------------------------------------------------------------------------------------
# voxel_layer-level functions (same names / argument meaning as the reference pybind module)
# ------------------------------------------------------------------------------------------------
def dynamic_voxelize(points, coors, voxel_size, coors_range, NDim=3):
    """voxel_layer.dynamic_voxelize: fills the caller-allocated int32 ``coors`` [N,3] with (z,y,x)."""
    if NDim != 3:
        raise RuntimeError('sst_amd.dynamic_voxelize supports NDim == 3 only')
    K.dynamic_voxelize(points.contiguous(), list(voxel_size), list(coors_range), coors=coors)


def hard_voxelize(points, voxels, coors, num_points_per_voxel, voxel_size, coors_range, max_points, max_voxels,
                  NDim=3):
    """voxel_layer.hard_voxelize (voxelization.h:51-69; algorithm voxelization_cpu.cpp:43-100): fills the
    caller-allocated ``voxels`` [max_voxels, max_points, C], ``coors`` [max_voxels, 3] int32 (z, y, x) and
    ``num_points_per_voxel`` [max_voxels] int32, returns the number of voxels.

    Same result as the reference's sequential loop: voxels numbered by the first appearance of one of their points,
    at most ``max_voxels`` of them (later voxels are dropped), inside a voxel the first ``max_points`` points in
    input order.  Here without a sequential pass: dynamic voxelization (this fork clamps, grid = round(), :127-130),
    sorted-unique of the coordinates, voxels ordered by their smallest point index, in-voxel rank from the CSR
    position.  No SST / FSD config uses it (they all set max_num_points = -1); composed of the existing kernels."""
    if NDim != 3:
        raise RuntimeError('sst_amd.hard_voxelize supports NDim == 3 only')
    K._lib.require_cuda(points)
    n = points.size(0)
    if n == 0:
        return 0
    with torch.no_grad():
        c = K.dynamic_voxelize(points.contiguous(), list(voxel_size), list(coors_range))           # [N, 3] (z, y, x)
        # C round() on the float32 quotient (halves away from zero; Python's round() goes to even)
        f32 = torch.tensor(list(coors_range) + list(voxel_size), dtype=torch.float32)
        grid = [int(g) for g in torch.floor((f32[3:6] - f32[0:3]) / f32[6:9] + 0.5).tolist()]         # x, y, z
        c = torch.minimum(c, torch.tensor([grid[2] - 1, grid[1] - 1, grid[0] - 1], dtype=torch.int32, device=c.device))
        plan = K.unique_rows(c.contiguous(), [0, 0, 0], [grid[2], grid[1], grid[0]])
        m = plan.m
        perm, offsets, inv = plan.perm.long(), plan.offsets[:m + 1].long(), plan.inverse.long()
        first_point = perm[offsets[:m]]                       # smallest point index of every voxel (perm is ascending in a group)
        order = torch.argsort(first_point)                    # voxels in order of first appearance
        vrank = torch.empty(m, dtype=torch.long, device=c.device)
        vrank[order] = torch.arange(m, device=c.device)
        pos = torch.empty(n, dtype=torch.long, device=c.device)
        pos[perm] = torch.arange(n, device=c.device)
        rank = pos - offsets[inv]                             # position of the point inside its voxel, input order
        num = min(m, int(max_voxels)) if max_voxels != -1 else m
        v_of_point = vrank[inv]
        keep = v_of_point < num
        if max_points != -1:
            keep &= rank < max_points
        voxels[v_of_point[keep], rank[keep]] = points[keep].to(voxels.dtype)
        counts = plan.counts().long()
        if max_points != -1:
            counts = counts.clamp(max=max_points)
        sel = order[:num]
        num_points_per_voxel[:num] = counts[sel].to(num_points_per_voxel.dtype)
        coors[:num] = K.unpack_unique_rows(plan, torch.int32)[sel].to(coors.dtype)
    return num


class ScatterPlan(object):
    """Grouping of N points into voxels, computed ONCE per frame batch and shared by every scatter call
    on the same coordinates (the reference re-runs at::unique_dim in each of DynamicVFE's 3 scatters)."""

    def __init__(self, plan, first, keep_idx, voxel_coors, coors_map, reduce_count, dropped_idx=None):
        self.plan = plan                # kernels.UniquePlan over ALL sorted-unique rows
        self.first = first              # groups [first, plan.m) are kept when keep_idx is None
        self.keep_idx = keep_idx        # int64 indices of kept groups (batched case) or None
        self.voxel_coors = voxel_coors  # [M, 3 or 4] int32
        self.coors_map = coors_map      # [N] int32, point -> kept voxel (-1: discarded)
        self.reduce_count = reduce_count  # [M] int32
        self.num_voxels = voxel_coors.size(0)
        self._keep_i32 = keep_idx.to(torch.int32) if keep_idx is not None else None
        self._dropped_i32 = dropped_idx   # int32 indices of the discarded groups (their points read voxel row 0), or None

    def raw_max(self, feats, scale_shift=None):
        """(max, arg-max rows) over the kept voxels without an autograd node (vfe_fused.FusedVFE2)"""
        if self.keep_idx is None:
            return K._segment_reduce_fwd(feats, self.plan.perm, self.plan.offsets[self.first:], self.plan.m - self.first,
                                         K.REDUCE['max'], True, None, None, self.plan, scale_shift)
        return K._segment_reduce_fwd(feats, self.plan.perm, self.plan.offsets, self._keep_i32.numel(), K.REDUCE['max'], True,
                                     self._keep_i32, None, self.plan, scale_shift)

    def group_sum(self, part):
        """gradient of ``pooled[coors_map]`` without atomics: CSR sum over the kept voxels + the discarded groups' rows on
        row 0 (frame_plan.FramePlan.group_sum)"""
        return K.add_group_rows_to_row0(self.reduce(part, 'sum'), part, self.plan, self._dropped_i32)

    def reduce(self, feats, mode):
        if self.keep_idx is None:
            return K.segment_reduce(feats, self.plan, mode, first=self.first)
        # kept groups addressed through an index inside the kernel: no gather of the result (and no index_add in
        # its backward)
        return K.segment_reduce(feats, self.plan, mode, group_index=self._keep_i32, inverse=self.coors_map)


def build_scatter_plan(coors, grid_zyx=None, reference_compat=True):
    """coors: [N,3] (z,y,x) or [N,4] (b,z,y,x) int32/int64 CUDA tensor.

    grid_zyx (optional): extents of the voxel grid; saves the min/max pass + host sync of the generic path.
    """
    if coors.dim() != 2 or coors.size(1) not in (3, 4):
        raise RuntimeError('coors must be [N,3] or [N,4]')
    coors = coors.contiguous()
    n, k = coors.shape
    batched = k == 4
    bmax_known = None
    if batched:
        if grid_zyx is not None:
            bmax = int(coors[-1, 0].item()) if n > 0 else 0  # the reference reads coors[-1,0] too
            bmax_known = bmax
            mins = [0, -1, -1, -1]
            extents = [bmax + 1] + [int(g) + 1 for g in grid_zyx]
        elif n > 0:
            hi = coors.amax(0).tolist()
            mins = [0, -1, -1, -1]
            extents = [max(int(hi[0]), 0) + 1] + [max(int(h), 0) + 2 for h in hi[1:]]
        else:
            mins, extents = [0, -1, -1, -1], [1, 1, 1, 1]
        plan = K.unique_rows(coors, mins, extents, invalid_if_negative=2)
    else:
        if grid_zyx is not None:
            mins, extents = [0, 0, 0], [int(g) for g in grid_zyx]
            plan = K.unique_rows(coors, mins, extents, invalid_if_negative=1)
        else:
            plan = K.unique_rows(coors, invalid_if_negative=1)
    m_all = plan.m
    dev = coors.device
    if n == 0 or m_all == 0:
        empty_c = torch.empty((0, k), dtype=coors.dtype, device=dev)
        e32 = torch.empty(0, dtype=torch.int32, device=dev)
        return ScatterPlan(plan, 0, None, empty_c, e32, e32)

    all_coors = K.unpack_unique_rows(plan, coors.dtype)  # [m_all, k]; invalid groups decode to -1 columns
    counts = plan.counts()
    if not batched:
        if reference_compat:
            first = 1
        else:
            first = 1 if bool((plan.ukeys[0] == 0).item()) else 0
        return ScatterPlan(plan, first, None, all_coors[first:], plan.inverse - first, counts[first:].contiguous(),
                           torch.arange(first, dtype=torch.int32, device=dev))
    if reference_compat and bmax_known == 0:
        # a single sample: "drop the first row of every sample" is the contiguous first = 1 case (no keep index,
        # no compaction, no extra readback)
        return ScatterPlan(plan, 1, None, all_coors[1:], plan.inverse - 1, counts[1:].contiguous(),
                           torch.zeros(1, dtype=torch.int32, device=dev))
    # batched: the reference loops over samples, so the "first row" is dropped once PER SAMPLE
    b = all_coors[:, 0]
    is_first = torch.ones_like(b, dtype=torch.bool)
    is_first[1:] = b[1:] != b[:-1]
    if reference_compat:
        drop = is_first
    else:
        drop = is_first & (all_coors[:, 1] < 0)
    keep = ~drop
    newid = torch.cumsum(keep.to(torch.int32), 0, dtype=torch.int32) - 1
    newid = torch.where(keep, newid, torch.full_like(newid, -1))
    keep_idx = torch.nonzero(keep).squeeze(1)
    coors_map = newid[plan.inverse.long()]
    return ScatterPlan(plan, 0, keep_idx, all_coors[keep_idx], coors_map, counts[keep_idx].contiguous(),
                       torch.nonzero(drop).squeeze(1).to(torch.int32))


def dynamic_point_to_voxel_forward(feats, coors, reduce_type, reference_compat=True):
    """voxel_layer.dynamic_point_to_voxel_forward -> [reduced_feats, out_coors, coors_map, reduce_count]
    (scatter_points_cuda.cu:183-234).  coors: [N,3] int."""
    if reduce_type not in ('max', 'sum', 'mean'):
        raise RuntimeError('do not support reduce type ' + str(reduce_type))
    if not feats.is_cuda or not coors.is_cuda:
        raise RuntimeError('feats and coors must be CUDA tensors (the reference has no CPU path either: '
                           'voxelization.h:106 "do not support cpu yet")')
    if not feats.is_contiguous() or not coors.is_contiguous():
        raise RuntimeError('feats / coors must be contiguous')
    if feats.size(0) == 0:  # scatter_points_cuda.cu:192-196
        e = torch.empty(0, dtype=torch.int32, device=coors.device)
        return [feats.clone().detach(), coors.clone().detach(), e, e.clone()]
    sp = build_scatter_plan(coors, reference_compat=reference_compat)
    reduced = sp.reduce(feats, reduce_type)
    return [reduced, sp.voxel_coors, sp.coors_map, sp.reduce_count]


def dynamic_scatter(feats, coors, reduce_type='max', plan=None, reference_compat=True):
    """_dynamic_scatter.apply (scatter_points.py:9-50): returns (voxel_feats, voxel_coors);
    differentiable w.r.t. feats, voxel_coors is an index tensor."""
    if feats.size(0) == 0:
        return feats.clone(), coors.clone()
    if plan is None:
        plan = build_scatter_plan(coors, reference_compat=reference_compat)
    return plan.reduce(feats.contiguous(), reduce_type), plan.voxel_coors


class _VoxelizationFn(object):
    """voxelization(points, voxel_size, coors_range, max_points, max_voxels) (voxelize.py:10-61)."""

    @staticmethod
    def apply(points, voxel_size, coors_range, max_points=35, max_voxels=20000):
        if max_points == -1 or max_voxels == -1:
            with torch.no_grad():
                return K.dynamic_voxelize(points.contiguous(), list(voxel_size), list(coors_range))
        # voxelize.py:47-58
        voxels = points.new_zeros(size=(max_voxels, max_points, points.size(1)))
        coors = points.new_zeros(size=(max_voxels, 3), dtype=torch.int)
        num_points_per_voxel = points.new_zeros(size=(max_voxels, ), dtype=torch.int)
        voxel_num = hard_voxelize(points, voxels, coors, num_points_per_voxel, voxel_size, coors_range, max_points,
                                  max_voxels, 3)
        return voxels[:voxel_num], coors[:voxel_num], num_points_per_voxel[:voxel_num]


voxelization = _VoxelizationFn.apply


class Voxelization(nn.Module):
    """Same constructor / forward as mmdet3d.ops.Voxelization (voxelize.py:64-122)."""

    def __init__(self, voxel_size, point_cloud_range, max_num_points, max_voxels=20000):
        super(Voxelization, self).__init__()
        self.voxel_size = voxel_size
        self.point_cloud_range = point_cloud_range
        self.max_num_points = max_num_points
        if isinstance(max_voxels, tuple):
            self.max_voxels = max_voxels
        else:
            self.max_voxels = _pair(max_voxels)
        point_cloud_range = torch.tensor(point_cloud_range, dtype=torch.float32)
        voxel_size = torch.tensor(voxel_size, dtype=torch.float32)
        grid_size = (point_cloud_range[3:] - point_cloud_range[:3]) / voxel_size
        grid_size = torch.round(grid_size).long()
        input_feat_shape = grid_size[:2]
        self.grid_size = grid_size
        self.pcd_shape = [*input_feat_shape, 1][::-1]

    def forward(self, input):
        max_voxels = self.max_voxels[0] if self.training else self.max_voxels[1]
        return voxelization(input, self.voxel_size, self.point_cloud_range, self.max_num_points, max_voxels)

    def voxelize_batch(self, points_list):
        """DynamicVoxelNet.voxelize (detectors/dynamic_voxelnet.py:49-71) without the pad/cat round trips:
        every sample writes its (b,z,y,x) rows straight into one [sum N, 4] tensor."""
        sizes = [int(p.size(0)) for p in points_list]
        total = sum(sizes)
        dev = points_list[0].device
        coors = torch.empty((total, 4), dtype=torch.int32, device=dev)
        off = 0
        with torch.no_grad():
            for b, p in enumerate(points_list):
                if sizes[b] > 0:
                    K.dynamic_voxelize(p.contiguous(), list(self.voxel_size), list(self.point_cloud_range),
                                       coors=coors[off:off + sizes[b]], batch_idx=b)
                off += sizes[b]
        points = torch.cat(points_list, dim=0) if len(points_list) > 1 else points_list[0]
        return points, coors

    def __repr__(self):
        tmpstr = self.__class__.__name__ + '('
        tmpstr += 'voxel_size=' + str(self.voxel_size)
        tmpstr += ', point_cloud_range=' + str(self.point_cloud_range)
        tmpstr += ', max_num_points=' + str(self.max_num_points)
        tmpstr += ', max_voxels=' + str(self.max_voxels)
        tmpstr += ')'
        return tmpstr


class DynamicScatter(nn.Module):
    """Same constructor / forward as mmdet3d.ops.DynamicScatter (scatter_points.py:53-107)."""

    def __init__(self, voxel_size, point_cloud_range, average_points: bool, reference_compat=True):
        super(DynamicScatter, self).__init__()
        self.voxel_size = voxel_size
        self.point_cloud_range = point_cloud_range
        self.average_points = average_points
        self.reference_compat = reference_compat

    def forward_single(self, points, coors, plan=None):
        reduce = 'mean' if self.average_points else 'max'
        return dynamic_scatter(points.contiguous(), coors.contiguous(), reduce, plan=plan,
                               reference_compat=self.reference_compat)

    def forward(self, points, coors, plan=None):
        """points [N,C], coors [N,3] or [N,4] -> (voxel_feats [M,C], voxel_coors [M,3/4]).
        The batched form is done in one pass (the reference loops over samples with torch.where)."""
        return self.forward_single(points, coors, plan=plan)

    def __repr__(self):
        tmpstr = self.__class__.__name__ + '('
        tmpstr += 'voxel_size=' + str(self.voxel_size)
        tmpstr += ', point_cloud_range=' + str(self.point_cloud_range)
        tmpstr += ', average_points=' + str(self.average_points)
        tmpstr += ')'
        return tmpstr
