"""TEST INFRASTRUCTURE ONLY — CPU restatement of FSD's connected-components clustering.

Follows mmdet3d/models/detectors/single_stage_fsd.py:45-68 (find_connected_componets) and :70-84
(find_connected_componets_single_batch): dense xy distance matrix in float32, ``< dist``, then
scipy.sparse.csgraph.connected_components — the SAME third-party routine the reference calls (scipy is present in
this image), so the component numbering needs no restating.  Pinned against the reference function itself, executed
from its own source text (oracle/ref_loader.load_reference_function), in tests/test_oracle.py when /root/reference
is present, and through tests/golden/cluster.npz on the GPU box."""
import numpy as np
from scipy.sparse.csgraph import connected_components


def _components(xy, dist):
    xy = np.asarray(xy, dtype=np.float32)
    d = xy[:, None, :] - xy[None, :, :]                      # float32
    dist_mat = np.sqrt((d * d).sum(2, dtype=np.float32))     # (dx^2 + dy^2) ** 0.5 in float32
    adj = dist_mat < np.float32(dist)
    return connected_components(adj, directed=False)[1].astype(np.int32)


def find_connected_components(points, batch_idx, dist):
    """single_stage_fsd.py:45-68: per sample, labels shifted by a running base; -1 never remains."""
    points = np.asarray(points, dtype=np.float32)
    batch_idx = np.asarray(batch_idx)
    out = np.zeros(len(points), dtype=np.int32) - 1
    base = 0
    for i in range(int(batch_idx.max()) + 1):
        mask = batch_idx == i
        if mask.any():
            c = _components(points[mask, :2], dist) + base
            base = int(c.max()) + 1
            out[mask] = c
    return out


def find_connected_components_single_batch(points, dist):
    """single_stage_fsd.py:70-84: one graph, the sample index is ignored."""
    return _components(np.asarray(points, dtype=np.float32)[:, :2], dist)
