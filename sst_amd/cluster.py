"""FSD's cluster assignment on the GPU.

Mirrors the module-level helpers and ``ClusterAssigner`` of mmdet3d/models/detectors/single_stage_fsd.py
(:30-68 filter_almost_empty / find_connected_componets*, :144-151 modify_cluster_by_class, :922-999
ClusterAssigner): same names, arguments and return values, so ``single_stage_fsd.py`` can import them from here.

Reference cost being removed: per class and per sample a dense N x N distance matrix, a device->host copy, scipy's
connected_components on the CPU and a host->device copy (the documented source of FSD's training-time instability,
docs/overall_instructions.md:51).  Here one lock-free union-find launch over all samples (csrc/cluster.hip); labels
are bit-identical (components numbered by their smallest member, scipy's order of first appearance).
"""
import torch

from . import _lib
from . import kernels as K
from .sst_ops import scatter_v2


def connected_components_xy(points, batch_idx, dist, return_count=False):
    """labels [N] int32 of the components of {same sample, xy distance < dist}; samples must be stored one after
    the other (ascending ``batch_idx``) for the numbering to equal the reference's running-base numbering."""
    if not points.is_cuda:
        raise RuntimeError('sst_amd.connected_components_xy: CUDA tensors required (no CPU fallback)')
    n = points.size(0)
    pts = points.float()
    if pts.stride(1) != 1:
        pts = pts.contiguous()
    b = batch_idx.to(torch.int32).contiguous()
    labels = torch.empty(n, dtype=torch.int32, device=points.device)
    count = torch.zeros(1, dtype=torch.int32, device=points.device)
    lib = _lib.load()
    ws = _lib.workspace(lib.sst_connected_components_workspace_bytes(n), points.device)
    rc = lib.sst_connected_components_xy_f32(_lib.ptr(pts), pts.stride(0) if n > 0 else 2, _lib.ptr(b), n, float(dist),
                                             _lib.ptr(labels), _lib.ptr(count), _lib.ptr(ws), _lib.stream_ptr())
    _lib.check(rc, 'sst_connected_components_xy_f32')
    return (labels, count) if return_count else labels


def filter_almost_empty(coors, min_points):
    """mask of the rows whose coordinate occurs at least ``min_points`` times (single_stage_fsd.py:30-34)."""
    if coors.size(0) == 0:
        return torch.zeros(0, dtype=torch.bool, device=coors.device)
    plan = K.unique_rows(coors.contiguous())
    cnt_per_point = plan.counts()[plan.inverse.long()]
    return cnt_per_point >= min_points


def find_connected_componets(points, batch_idx, dist):
    """single_stage_fsd.py:45-68 (training path): int labels, numbered per sample with a running base."""
    assert len(points) > 0
    b = batch_idx.int()
    sorted_already = True
    if b.numel() > 1 and points.is_cuda:
        # the centres come out of a sorted-unique, i.e. sample after sample; anything else is brought into that
        # order first (stable), labelled, and put back
        sorted_already = bool((b[1:] >= b[:-1]).all().item()) if b.numel() < (1 << 16) else None
        if sorted_already is None:
            sorted_already = bool((b[1:] >= b[:-1]).all().item())
    if sorted_already:
        labels = connected_components_xy(points, b, dist)
    else:
        order = torch.sort(b, stable=True)[1]
        lab_sorted = connected_components_xy(points[order], b[order], dist)
        labels = torch.empty_like(lab_sorted)
        labels[order] = lab_sorted
    return labels.to(batch_idx.dtype) if batch_idx.dtype != torch.int32 else labels


def find_connected_componets_single_batch(points, batch_idx, dist):
    """single_stage_fsd.py:70-84 (test path): ONE graph over all points, the sample index is ignored."""
    zeros = torch.zeros(points.size(0), dtype=torch.int32, device=points.device)
    return connected_components_xy(points, zeros, dist)


def modify_cluster_by_class(cluster_inds_list):
    """prepend the class index as column 0 (single_stage_fsd.py:144-151)."""
    new_list = []
    for i, inds in enumerate(cluster_inds_list):
        cls_pad = inds.new_ones((len(inds),)) * i
        new_list.append(torch.cat([cls_pad[:, None], inds], 1))
    return new_list


class ClusterAssigner(torch.nn.Module):
    """Cluster centres per class and the assignment of every foreground point to one of them
    (single_stage_fsd.py:922-999; same constructor and forward)."""

    def __init__(self, cluster_voxel_size, min_points, point_cloud_range, connected_dist,
                 class_names=['Car', 'Cyclist', 'Pedestrian'], gpu_clustering=(False, False)):
        super().__init__()
        self.cluster_voxel_size = cluster_voxel_size
        self.min_points = min_points
        self.connected_dist = connected_dist
        self.point_cloud_range = point_cloud_range
        self.class_names = class_names
        self.gpu_clustering = gpu_clustering
        self.num_classes = len(class_names)

    def _per_class(self, table, class_name):
        if isinstance(table, dict):
            return table[class_name]
        if isinstance(table, list):
            return table[self.class_names.index(class_name)]
        return table

    @torch.no_grad()
    def forward(self, points_list, batch_idx_list, gt_bboxes_3d=None, gt_labels_3d=None, origin_points=None):
        assert self.num_classes == len(self.class_names)
        outs = [self.forward_single_class(p, b, c, origin_points)
                for p, b, c in zip(points_list, batch_idx_list, self.class_names)]
        cluster_inds_list = modify_cluster_by_class([o[0] for o in outs])
        return cluster_inds_list, [o[1] for o in outs]

    def forward_single_class(self, points, batch_idx, class_name, origin_points):
        """One class: cluster-voxel grouping of the voted centres -> centroids of the voxels with at least ``min_points``
        votes -> connected components of the centroids -> every surviving point inherits its voxel's component
        (single_stage_fsd.py:953-999).  ONE sorted-unique grouping serves the occupancy filter, the centroid reduction and
        the point -> centroid map (the reference groups the same keys three times: filter_almost_empty, scatter_v2 and its
        inverse), and the two data-dependent lengths (surviving points, surviving voxels) are read back together."""
        dev = points.device
        batch_idx = batch_idx.int()
        cell = K.const_tensor(self._per_class(self.cluster_voxel_size, class_name), dev, points.dtype)
        origin = K.const_tensor(self.point_cloud_range[:3], dev, points.dtype)
        cells = torch.cat([batch_idx[:, None], torch.div(points - origin, cell, rounding_mode='floor').int()], dim=1)
        n = points.size(0)
        if n == 0:
            empty = torch.zeros((0, 2), dtype=torch.int32, device=dev)
            return empty, torch.zeros(0, dtype=torch.bool, device=dev)
        groups = K.unique_rows(cells.contiguous())
        occupancy = groups.counts()                                         # votes per cluster voxel, sorted-voxel order
        crowded = occupancy >= self.min_points
        point_group = groups.inverse.long()
        valid_mask = crowded[point_group]
        sizes = torch.stack([valid_mask.sum(), crowded.sum()]).tolist()     # the one read-back of this class
        if sizes[0] == 0:
            # nothing survives the filter: the reference then keeps EVERY point (valid_mask = ~valid_mask, :968-970)
            valid_mask = torch.ones_like(valid_mask)
            crowded = torch.ones_like(crowded)
            sizes = [n, groups.m]
        keep_points = _nonzero_known(valid_mask, sizes[0])
        keep_groups = _nonzero_known(crowded, sizes[1])
        # centroids of the surviving voxels: the mean over ALL votes of a voxel (a voxel survives or falls as a whole)
        centroids = K.segment_reduce(points.float().contiguous(), groups, 'mean').index_select(0, keep_groups)
        centroid_sample = K.unpack_unique_rows(groups, torch.int32).index_select(0, keep_groups)[:, 0]
        dist = self._per_class(self.connected_dist, class_name)
        if self.training:
            component = connected_components_xy(centroids, centroid_sample, dist)   # sorted-unique order = sample after sample
        else:
            # the reference's test path clusters all samples as one graph (both of its variants, :36-43 / :70-84)
            component = find_connected_componets_single_batch(centroids, centroid_sample, dist)
        assert component.numel() == sizes[1]
        rank_of_group = torch.cumsum(crowded.int(), 0) - 1                   # surviving voxel -> row of `centroids`
        point_component = component[rank_of_group[point_group[keep_points]].long()]
        valid_mask._sst_keep = keep_points     # the indices of the set entries ride along: callers need not search again
        return torch.stack([batch_idx[keep_points], point_component.int()], 1), valid_mask


def _nonzero_known(mask, count):
    """indices of the set entries of a 1-D mask whose number is already on the host: no further read-back"""
    if hasattr(torch, 'nonzero_static'):
        return torch.nonzero_static(mask, size=int(count)).squeeze(1)
    return torch.nonzero(mask).squeeze(1)
