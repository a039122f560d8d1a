"""TEST INFRASTRUCTURE ONLY — CPU restatement of the dynamic point pool.

PARITY UNPINNED for the 13 per-pair features: the routine the reference calls (dynamic_point_pool_ext, TorchEx; call
sites mmdet3d/ops/dynamic_point_pool_op.py:36 and :86-88) is not in the reference tree and the reference holds no test
or fixture for it.  What IS pinned:
  * the box convention and the membership test — restated from the reference's own points-in-boxes routine
    (mmdet3d/ops/roiaware_pool3d/src/points_in_boxes_cpu.cpp:16-43: z is the bottom centre, extents (w, l, h), local
    frame rotated by rz + pi/2, local x against l and local y against w, |z| inclusive, x / y strict) and checked
    against that routine compiled from its source (oracle/build_ref.build_points_in_boxes; tests/test_oracle.py)
    and through tests/golden/point_pool.npz;
  * the invariants the reference's extractor asserts on the features
    (mmdet3d/models/roi_heads/roi_extractors/dynamic_point_roi_extractor.py:96-105), see ``check_invariants``.
The feature layout [xyz | local xyz | 6 face distances | is_in_margin] follows the slicing in
dynamic_point_roi_extractor.py:86-89.  Order and caps: deterministic (RoI, then point index; first
``max_inbox_point`` per RoI, first ``max_all_pts`` overall) — one of the outcomes the reference's atomics allow."""
import numpy as np

F32 = np.float32


def box_frame(rois):
    """per RoI: centre (z lifted by h/2), cos / sin of rz + pi/2 in float32 (points_in_boxes_cpu.cpp:19-21, 33)"""
    r = np.asarray(rois, dtype=F32)
    cz = (r[:, 2] + r[:, 5] * F32(0.5)).astype(F32)
    rot = (r[:, 6].astype(np.float64) + np.pi / 2).astype(F32)
    return r[:, 0], r[:, 1], cz, np.cos(rot).astype(F32), np.sin(rot).astype(F32)


def local_coords(rois, pts):
    """[R, P] local x, y, z (points_in_boxes_cpu.cpp:22-23: x*cos + y*(-sin), x*sin + y*cos), float32 step by step"""
    cx, cy, cz, cosa, sina = box_frame(rois)
    p = np.asarray(pts, dtype=F32)
    sx = (p[None, :, 0] - cx[:, None]).astype(F32)
    sy = (p[None, :, 1] - cy[:, None]).astype(F32)
    lx = ((sx * cosa[:, None]).astype(F32) + (sy * (-sina)[:, None]).astype(F32)).astype(F32)
    ly = ((sx * sina[:, None]).astype(F32) + (sy * cosa[:, None]).astype(F32)).astype(F32)
    lz = (p[None, :, 2] - cz[:, None]).astype(F32)
    return lx, ly, lz


def inside(lx, ly, lz, w, l, h):
    """points_in_boxes_cpu.cpp:36-41 on local coordinates; w, l, h are [R] float32"""
    hw, hl, hh = (w * F32(0.5))[:, None], (l * F32(0.5))[:, None], (h * F32(0.5))[:, None]
    return ~(np.abs(lz) > hh) & (lx > -hl) & (lx < hl) & (ly > -hw) & (ly < hw)


def face_clearance(rois, pts, extra_wlh):
    """[R, P] smallest distance of a point to any face of the box or of the enlarged box: pairs with a tiny
    clearance may legitimately flip between implementations (1-ulp differences in cos / sin)."""
    r = np.asarray(rois, dtype=F32)
    lx, ly, lz = local_coords(r, pts)
    out = np.full(lx.shape, np.inf)
    for e in (np.zeros(3, dtype=F32), np.asarray(extra_wlh, dtype=F32)):
        for loc, ext in ((lx, r[:, 4] + e[1]), (ly, r[:, 3] + e[0]), (lz, r[:, 5] + e[2])):
            out = np.minimum(out, np.abs(np.abs(loc.astype(np.float64)) - ext[:, None].astype(np.float64) * 0.5))
    return out


def dynamic_point_pool(rois, pts, extra_wlh, max_inbox_point, max_all_pts, rois_batch=None, pts_batch=None):
    r = np.asarray(rois, dtype=F32)
    p = np.asarray(pts, dtype=F32)[:, :3]
    e = np.asarray(extra_wlh, dtype=F32)
    lx, ly, lz = local_coords(r, p)
    w, l, h = r[:, 3], r[:, 4], r[:, 5]
    large = inside(lx, ly, lz, (w + e[0]).astype(F32), (l + e[1]).astype(F32), (h + e[2]).astype(F32))
    small = inside(lx, ly, lz, w, l, h)
    if rois_batch is not None:
        large &= np.asarray(rois_batch)[:, None] == np.asarray(pts_batch)[None, :]
    rank = np.cumsum(large, axis=1) - 1
    keep = large & (rank < max_inbox_point)
    roi_idx, pts_idx = np.nonzero(keep)  # row-major: sorted by (RoI, point)
    roi_idx, pts_idx = roi_idx[:max_all_pts], pts_idx[:max_all_pts]
    a, b, c = lx[roi_idx, pts_idx], ly[roi_idx, pts_idx], lz[roi_idx, pts_idx]
    sl, sw, sh = l[roi_idx] * F32(0.5), w[roi_idx] * F32(0.5), h[roi_idx] * F32(0.5)
    feats = np.stack([p[pts_idx, 0], p[pts_idx, 1], p[pts_idx, 2], a, b, c, a + sl, b + sw, c + sh, sl - a, sw - b,
                      sh - c, (~small[roi_idx, pts_idx]).astype(F32)], axis=1).astype(F32)
    return pts_idx.astype(np.int64), roi_idx.astype(np.int64), feats


def check_invariants(rois, pts, extra_wlh, pts_idx, roi_idx, feats):
    """dynamic_point_roi_extractor.py:96-105, on any implementation's output"""
    r = np.asarray(rois, dtype=F32)[roi_idx]
    assert np.allclose(np.asarray(pts, dtype=F32)[pts_idx, :3], feats[:, :3])
    off = feats[:, 6:12]
    assert np.allclose(off[:, 0] + off[:, 3], r[:, 4], atol=1e-5)
    assert np.allclose(off[:, 1] + off[:, 4], r[:, 3], atol=1e-5)
    assert np.allclose(off[:, 2] + off[:, 5], r[:, 5], atol=1e-5)
    assert (np.abs(feats[:, 3]) < r[:, 4] + extra_wlh[0] + 1e-5).all()
    assert (np.abs(feats[:, 4]) < r[:, 3] + extra_wlh[1] + 1e-5).all()
    assert (np.abs(feats[:, 5]) < r[:, 5] + extra_wlh[2] + 1e-5).all()
