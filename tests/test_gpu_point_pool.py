"""GPU: dynamic point pool (csrc/point_pool.hip, sst_amd/dynamic_point_pool.py) against the oracle
(oracle/point_pool_oracle.py), the reference-derived membership fixture (tests/golden/point_pool.npz) and the
invariants the reference's extractor asserts.  Integer outputs are compared exactly away from box faces (pairs
within 1e-5 of a face may flip: device cosf / sinf differ from the host's in the last bit); features to 1e-5."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _run(rois, pts, extra, max_inbox, max_all, rb=None, pb=None):
    import sst_amd
    r, p = torch.from_numpy(rois).to(DEV), torch.from_numpy(pts).to(DEV)
    if rb is None:
        out = sst_amd.dynamic_point_pool(r, p, list(extra), max_inbox, max_all)
    else:
        out = sst_amd.dynamic_point_pool_mixed(r, torch.from_numpy(rb).to(DEV), p, torch.from_numpy(pb).to(DEV),
                                               list(extra), max_inbox, max_all)
    return [o.cpu().numpy() for o in out]


def _compare(rois, pts, extra, got, want):
    from oracle import point_pool_oracle as O
    (gp, gr, gf), (wp, wr, wf) = got, want
    gs = {(r, p): i for i, (r, p) in enumerate(zip(gr.tolist(), gp.tolist()))}
    ws = {(r, p): i for i, (r, p) in enumerate(zip(wr.tolist(), wp.tolist()))}
    diff = set(gs) ^ set(ws)
    if diff:
        clear = O.face_clearance(rois, pts, extra)
        assert all(clear[r, p] < 1e-5 for r, p in diff), 'membership differs away from any face'
    common = sorted(set(gs) & set(ws))
    gi, wi = np.array([gs[k] for k in common]), np.array([ws[k] for k in common])
    assert np.abs(gf[gi, :12] - wf[wi, :12]).max() < 1e-5
    flips = gf[gi, 12] != wf[wi, 12]
    if flips.any():
        clear = O.face_clearance(rois, pts, extra)
        assert all(clear[r, p] < 1e-5 for (r, p), f in zip(common, flips) if f)
    return len(common), len(diff)


@pytest.mark.parametrize('tag', ['veh', 'ped'])
def test_point_pool_matches_oracle_and_reference_membership(tag):
    from oracle import point_pool_oracle as O
    g = load_golden('point_pool.npz')
    rois, pts, extra = g[f'in::{tag}::rois'], g[f'in::{tag}::pts'], g[f'in::{tag}::extra_wlh']
    got = _run(rois, pts, extra, 4096, 100000)
    want = O.dynamic_point_pool(rois, pts, extra, 4096, 100000)
    n_common, n_diff = _compare(rois, pts, extra, got, want)
    assert n_common > 500 and n_diff <= 2
    gp, gr, gf = got
    assert gp.dtype == np.int64 and gr.dtype == np.int64 and gf.dtype == np.float32 and gf.shape[1] == 13
    assert (np.diff(gr * (len(pts) + 1) + gp) > 0).all()  # sorted by (roi, point): deterministic order
    O.check_invariants(rois, pts, extra, gp, gr, gf)
    # membership against the pairs the reference's own points_in_boxes_cpu produced
    clear = O.face_clearance(rois, pts, extra)
    mine = set(zip(gr.tolist(), gp.tolist()))
    ref_large = set(map(tuple, g[f'out::{tag}::pairs_in_enlarged_box'].tolist()))
    ref_small = set(map(tuple, g[f'out::{tag}::pairs_in_box'].tolist()))
    assert all(clear[r, p] < 1e-5 for r, p in mine ^ ref_large)
    inner = {(r, p) for (r, p), m in zip(zip(gr.tolist(), gp.tolist()), gf[:, 12].tolist()) if m == 0.0}
    assert all(clear[r, p] < 1e-5 for r, p in inner ^ ref_small)


def test_point_pool_caps_keep_the_first_points():
    from oracle import point_pool_oracle as O
    g = load_golden('point_pool.npz')
    rois, pts, extra = g['in::veh::rois'], g['in::veh::pts'], g['in::veh::extra_wlh']
    for max_inbox, max_all in ((8, 100000), (8, 100), (1, 50000), (4096, 37)):
        got = _run(rois, pts, extra, max_inbox, max_all)
        want = O.dynamic_point_pool(rois, pts, extra, max_inbox, max_all)
        assert len(got[0]) == len(want[0]) <= max_all
        assert np.bincount(got[1]).max() <= max_inbox
        np.testing.assert_array_equal(got[0], want[0])
        np.testing.assert_array_equal(got[1], want[1])
        assert np.abs(got[2] - want[2]).max() < 1e-5
    again = _run(rois, pts, extra, 8, 100)
    for a, b in zip(again, _run(rois, pts, extra, 8, 100)):
        np.testing.assert_array_equal(a, b)


def test_point_pool_empty_result_and_argument_errors():
    import sst_amd
    rois = np.array([[0, 0, 0, 1, 1, 1, 0.3]], dtype=np.float32)
    pts = np.array([[10, 10, 10], [20, 20, 20]], dtype=np.float32)
    p, r, f = _run(rois, pts, (0.5, 0.5, 0.5), 16, 100)
    assert p.tolist() == [-1] and r.tolist() == [-1] and f.shape == (1, 13) and not f.any()  # the "fake" row
    with pytest.raises(RuntimeError):
        sst_amd.dynamic_point_pool(torch.from_numpy(rois), torch.from_numpy(pts), [0, 0, 0], 16, 100)  # CPU tensors
    with pytest.raises(AssertionError):
        sst_amd.dynamic_point_pool(torch.zeros(0, 7, device=DEV), torch.from_numpy(pts).to(DEV), [0, 0, 0], 16, 100)
    out = sst_amd.dynamic_point_pool(torch.from_numpy(rois).to(DEV).requires_grad_(True), torch.from_numpy(pts).to(DEV),
                                     [0, 0, 0], 16, 100)
    assert not any(o.requires_grad for o in out)


def test_point_pool_mixed_batches_and_strided_points():
    from oracle import point_pool_oracle as O
    import sst_amd
    g = load_golden('point_pool.npz')
    rois, pts, extra = g['in::ped::rois'], g['in::ped::pts'], g['in::ped::extra_wlh']
    rb = np.sort(np.random.default_rng(0).integers(0, 3, len(rois))).astype(np.int32)
    pb = np.sort(np.random.default_rng(1).integers(0, 3, len(pts))).astype(np.int32)
    got = _run(rois, pts, extra, 64, 200000, rb, pb)
    want = O.dynamic_point_pool(rois, pts, extra, 64, 200000, rb, pb)
    n_common, n_diff = _compare(rois, pts, extra, got, want)
    assert n_common > 200 and n_diff <= 2
    # points as a strided view of a wider tensor ([P, 5], xyz first)
    wide = torch.cat([torch.from_numpy(pts), torch.rand(len(pts), 2)], 1).to(DEV)
    out = sst_amd.dynamic_point_pool(torch.from_numpy(rois).to(DEV), wide[:, :3], list(extra), 64, 200000)
    base = _run(rois, pts, extra, 64, 200000)
    for a, b in zip(out, base):
        np.testing.assert_array_equal(a.cpu().numpy(), b)


def test_roi_extractor_mirrors_reference_flow():
    """DynamicPointROIExtractor: per-sample calls with index bases (dynamic_point_roi_extractor.py:51-80), the
    single-sample fast path, the debug invariants, registry construction."""
    from oracle import point_pool_oracle as O
    import sst_amd
    g = load_golden('point_pool.npz')
    rois, pts, extra = g['in::veh::rois'], g['in::veh::pts'], g['in::veh::extra_wlh']
    rb = np.sort(np.random.default_rng(2).integers(0, 2, len(rois))).astype(np.int32)
    pb = np.sort(np.random.default_rng(3).integers(0, 2, len(pts))).astype(np.int32)
    ext = sst_amd.ROI_EXTRACTORS.build(dict(type='DynamicPointROIExtractor', extra_wlh=list(map(float, extra)),
                                            max_inbox_point=256, max_all_pts=50000))
    rois8 = torch.cat([torch.from_numpy(rb).float()[:, None], torch.from_numpy(rois)], 1).to(DEV)
    inds, roi_inds, info = ext(torch.from_numpy(pts).to(DEV), torch.from_numpy(pb).to(DEV), rois8)
    want = O.dynamic_point_pool(rois, pts, extra, 256, 1 << 30, rb, pb)
    feats = torch.cat([torch.from_numpy(pts).to(DEV)[inds], info['local_xyz'], info['boundary_offset'],
                       info['is_in_margin'][:, None]], 1)
    n_common, n_diff = _compare(rois, pts, extra, (inds.cpu().numpy(), roi_inds.cpu().numpy(), feats.cpu().numpy()),
                                want)
    assert n_common > 500 and n_diff <= 2
    assert info['local_xyz'].shape[1] == 3 and info['boundary_offset'].shape[1] == 6
    one = ext(torch.from_numpy(pts).to(DEV), torch.zeros(len(pts), device=DEV), rois8, batch_size=1)
    ref1 = _run(rois, pts, extra, 256, 50000)
    np.testing.assert_array_equal(one[0].cpu().numpy(), ref1[0])
    np.testing.assert_array_equal(one[1].cpu().numpy(), ref1[1])


def test_roi_extractor_caps_every_sample_on_its_own():
    """sample 0 far above max_all_pts must not take pairs away from sample 1 (the reference caps each
    sample in its own call, dynamic_point_roi_extractor.py:51-80); a sample without pairs gets the (-1, -1, zeros) row;
    unsorted batch indices are refused whatever `debug` says (the reference asserts them unconditionally, :44-46)."""
    import sst_amd
    rng = np.random.default_rng(5)
    rois = np.array([[0, 0, 0, 0, 4, 4, 2, 0.3], [0, 10, 0, 0, 4, 4, 2, 0.0], [1, 0, 0, 0, 4, 4, 2, -0.2],
                     [2, 50, 50, 0, 1, 1, 1, 0.0]], np.float32)
    p0 = np.concatenate([rng.uniform(-1.5, 1.5, (3000, 3)), rng.uniform(-1.5, 1.5, (3000, 3)) + [10, 0, 0]]) * [1, 1, 0.5]
    p1 = rng.uniform(-1.5, 1.5, (300, 3)) * [1, 1, 0.5]
    p2 = rng.uniform(-1.5, 1.5, (50, 3))                      # nowhere near the RoI of sample 2
    pts = np.concatenate([p0, p1, p2]).astype(np.float32)
    pb = np.concatenate([np.zeros(len(p0)), np.ones(len(p1)), np.full(len(p2), 2)]).astype(np.int64)
    ext = sst_amd.DynamicPointROIExtractor(extra_wlh=[0.2, 0.2, 0.2], max_inbox_point=4096, max_all_pts=500, debug=False)
    inds, roi_inds, info = ext(torch.from_numpy(pts).to(DEV), torch.from_numpy(pb).to(DEV), torch.from_numpy(rois).to(DEV))
    inds, roi_inds = inds.cpu().numpy(), roi_inds.cpu().numpy()
    s0 = np.isin(roi_inds, [0, 1])
    assert s0.sum() == 500                                     # sample 0: thousands of candidates, capped at 500
    s1 = roi_inds == 2
    single = sst_amd.dynamic_point_pool(torch.from_numpy(rois[2:3, 1:]).to(DEV), torch.from_numpy(p1.astype(np.float32)).to(DEV),
                                        [0.2, 0.2, 0.2], 4096, 500)
    assert s1.sum() == len(single[0]) > 100                    # sample 1: every one of its own pairs, none lost
    np.testing.assert_array_equal(inds[s1] - len(p0), single[0].cpu().numpy())
    assert (pb[inds[s0]] == 0).all() and (pb[inds[s1]] == 1).all()
    fake = ~(s0 | s1)
    assert fake.sum() == 1 and inds[fake][0] == -1 and roi_inds[fake][0] == -1     # sample 2: the fake row
    assert float(info['local_xyz'][torch.from_numpy(fake).to(DEV)].abs().max()) == 0.0
    with pytest.raises(AssertionError):
        ext(torch.from_numpy(pts).to(DEV), torch.from_numpy(pb[::-1].copy()).to(DEV), torch.from_numpy(rois).to(DEV))


def test_point_pool_full_size_properties():
    """FSD second-stage scale (2e5 points, 2000 RoIs): invariants, caps, sortedness, determinism, and the pair count
    against a float64 membership count away from faces."""
    from oracle import point_pool_oracle as O
    import sst_amd
    rng = np.random.default_rng(9)
    n_rois, n_pts = 2000, 200000
    rois = np.concatenate([rng.uniform(-70, 70, (n_rois, 2)), rng.uniform(-2, 1, (n_rois, 1)),
                           rng.uniform(0.5, 5, (n_rois, 3)), rng.uniform(-4, 4, (n_rois, 1))], 1).astype(np.float32)
    k = rng.integers(0, n_rois, n_pts)
    pts = (rois[k, :3] + rng.normal(0, 1.0, (n_pts, 3)) + np.array([0, 0, 1.0])).astype(np.float32)
    extra = (0.5, 0.5, 0.5)
    p, r, f = _run(rois, pts, extra, 256, 100000)
    assert 0 < len(p) <= 100000 and np.bincount(r).max() <= 256
    assert (np.diff(r * (n_pts + 1) + p) > 0).all()
    O.check_invariants(rois, pts, extra, p, r, f)
    p2, r2, f2 = _run(rois, pts, extra, 256, 100000)
    np.testing.assert_array_equal(p, p2)
    np.testing.assert_array_equal(f, f2)
    # uncapped pair count of the first 64 RoIs against the oracle
    sub = rois[:64]
    ps, rs, fs = _run(sub, pts, extra, 1 << 20, 1 << 22)
    wp, wr, wf = O.dynamic_point_pool(sub, pts, extra, 1 << 20, 1 << 22)
    assert abs(len(ps) - len(wp)) <= 2
