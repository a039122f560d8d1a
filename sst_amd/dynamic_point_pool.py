"""Dynamic point pool and its RoI extractor on the GPU.

Mirrors mmdet3d/ops/dynamic_point_pool_op.py (``dynamic_point_pool`` :9-58, ``dynamic_point_pool_mixed`` :61-113)
and mmdet3d/models/roi_heads/roi_extractors/dynamic_point_roi_extractor.py:9-136 (``DynamicPointROIExtractor``):
same names, arguments, return values, "fake non-empty" result and non-differentiable outputs.

The CUDA extension those wrappers call belongs to TorchEx and is not in the reference tree, so the per-pair
arithmetic is a restatement (see csrc/point_pool.hip): PARITY UNPINNED beyond the reference's box convention and the
invariants its extractor asserts in debug mode, which ``debug=True`` checks here too.  Differences by design: the
output is deterministic (sorted by RoI, then point index; above the caps the lowest point indices survive) where the
reference's atomics leave order and survivors to the race; the valid rows are sliced off with one read-back of
their count instead of a boolean mask over ``max_all_pts`` rows.
"""
import torch
from torch import nn

from . import _lib
from .registry import ROI_EXTRACTORS


def _pool(rois, rois_batch, pts, pts_batch, extra_wlh, max_inbox_point, max_all_pts):
    if not (rois.is_cuda and pts.is_cuda):
        raise RuntimeError('sst_amd.dynamic_point_pool: CUDA tensors required (no CPU fallback)')
    assert len(rois) > 0  # dynamic_point_pool_op.py:35
    assert rois.size(1) == 7 and pts.size(1) >= 3 and len(extra_wlh) == 3
    dev = pts.device
    rois_f = rois.float().contiguous()
    pts_f = pts.float()
    if pts_f.dim() != 2 or pts_f.stride(1) != 1:
        pts_f = pts_f.contiguous()
    n_rois, n_pts = rois_f.size(0), pts_f.size(0)
    rb = pb = None
    if rois_batch is not None:
        rb = rois_batch.to(torch.int32).contiguous()
        pb = pts_batch.to(torch.int32).contiguous()
        assert rb.numel() == n_rois and pb.numel() == n_pts
    out_pts_idx = torch.full((max_all_pts,), -1, dtype=torch.long, device=dev)
    out_roi_idx = torch.full((max_all_pts,), -1, dtype=torch.long, device=dev)
    out_pts_feats = torch.zeros((max_all_pts, 13), dtype=torch.float, device=dev)
    num_out = torch.zeros(1, dtype=torch.long, device=dev)
    lib = _lib.load()
    ws = _lib.workspace(lib.sst_dynamic_point_pool_workspace_bytes(n_rois, n_pts), dev)
    extra = _lib.farray(extra_wlh)  # host array, read during the call
    rc = lib.sst_dynamic_point_pool_f32(_lib.ptr(rois_f), _lib.ptr(rb), n_rois, _lib.ptr(pts_f),
                                        pts_f.stride(0) if n_pts > 0 else 3, _lib.ptr(pb), n_pts,
                                        extra, int(max_inbox_point),
                                        int(max_all_pts), _lib.ptr(out_pts_idx), _lib.ptr(out_roi_idx),
                                        _lib.ptr(out_pts_feats), _lib.ptr(num_out), _lib.ptr(ws), _lib.stream_ptr())
    _lib.check(rc, 'sst_dynamic_point_pool_f32')
    return out_pts_idx, out_roi_idx, out_pts_feats, num_out


def _valid_rows(out_pts_idx, out_roi_idx, out_pts_feats, num_out):
    """slice the pairs off the max_all_pts-row buffers (one read-back of their count)"""
    n = int(num_out.item())
    if n == 0:
        # "fake a non-empty input" (dynamic_point_pool_op.py:41-45): one row of (-1, -1, zeros)
        return out_pts_idx[0:1], out_roi_idx[0:1], out_pts_feats[0:1, :]
    return out_pts_idx[:n], out_roi_idx[:n], out_pts_feats[:n]


class DynamicPointPoolFunction(torch.autograd.Function):
    """dynamic_point_pool_op.py:9-56."""

    @staticmethod
    def forward(ctx, rois, pts, extra_wlh, max_inbox_point, max_all_pts=50000):
        out = _valid_rows(*_pool(rois, None, pts, None, extra_wlh, max_inbox_point, max_all_pts))
        ctx.mark_non_differentiable(*out)
        return out

    @staticmethod
    def backward(ctx, g1, g2, g3):
        return None, None, None, None, None


dynamic_point_pool = DynamicPointPoolFunction.apply


class DynamicPointPoolMixedFunction(torch.autograd.Function):
    """dynamic_point_pool_op.py:61-111: all samples in one call, a pair needs equal sample indices."""

    @staticmethod
    def forward(ctx, rois, rois_batch, pts, pts_batch, extra_wlh, max_inbox_point, max_all_pts=200000):
        out = _valid_rows(*_pool(rois, rois_batch, pts, pts_batch, extra_wlh, max_inbox_point, max_all_pts))
        ctx.mark_non_differentiable(*out)
        return out

    @staticmethod
    def backward(ctx, g1, g2, g3):
        return None, None, None, None, None, None, None


dynamic_point_pool_mixed = DynamicPointPoolMixedFunction.apply


@ROI_EXTRACTORS.register_module()
class DynamicPointROIExtractor(nn.Module):
    """Point-wise RoI extractor (dynamic_point_roi_extractor.py:9-136).

    forward(pts_xyz [P,3], batch_inds [P] (sorted), rois [R,8] = (sample, x, y, z, w, l, h, rz)) ->
    (point indices, RoI indices, dict(local_xyz, boundary_offset, is_in_margin))."""

    def __init__(self, init_cfg=None, debug=True, extra_wlh=[0, 0, 0], max_inbox_point=512, max_all_pts=50000):
        super().__init__()
        self.init_cfg = init_cfg
        self.debug = debug
        self.extra_wlh = extra_wlh
        self.max_inbox_point = max_inbox_point
        self.max_all_pts = max_all_pts

    def forward(self, pts_xyz, batch_inds, rois, max_inbox_point=None, batch_size=None):
        """All samples in ONE pool launch (the batched entry point pairs a point only with RoIs of its own sample) and
        one read-back of the per-sample pair counts; the reference runs one launch and two read-backs per sample
        (:51-80).  What its loop defines is kept: RoI / point indices are global, the pairs of a sample are capped at
        ``max_all_pts``, a sample without any pair contributes the fake (-1, -1, zeros) row."""
        if batch_size == 1:
            return self.fast_single_sample_forward(pts_xyz, rois, max_inbox_point)
        assert len(pts_xyz) > 0 and len(batch_inds) > 0 and len(rois) > 0
        cap_in_box = self.max_inbox_point if max_inbox_point is None else max_inbox_point
        roi_sample = rois[:, 0].to(torch.int32)
        pts_sample = batch_inds.to(torch.int32)
        # the reference asserts sorted batch indices unconditionally (:44-46): the two flags ride on the one read-back
        order_ok = torch.stack([(pts_sample[1:] >= pts_sample[:-1]).all(), (roi_sample[1:] >= roi_sample[:-1]).all()])
        n_samples = int(batch_size) if batch_size is not None else int(batch_inds.max().item()) + 1
        rows = n_samples * self.max_all_pts
        pts_idx, roi_idx, feats, num_out = _pool(rois[:, 1:], roi_sample, pts_xyz, pts_sample, self.extra_wlh,
                                                 cap_in_box, rows)
        # pairs come out sorted by (RoI, point), RoIs are sorted by sample: the pairs of a sample are one run
        row_sample = roi_sample.long()[roi_idx.clamp(min=0)]
        row_sample = torch.where(torch.arange(rows, device=row_sample.device) < num_out, row_sample,
                                 torch.full_like(row_sample, n_samples))
        host = torch.cat([torch.bincount(row_sample, minlength=n_samples + 1)[:n_samples], order_ok.long(),
                          num_out.reshape(1).long()]).tolist()                                  # the one read-back
        per_sample, (pts_sorted, rois_sorted), total = host[:n_samples], host[n_samples:n_samples + 2], host[-1]
        assert pts_sorted, 'points must be sorted by sample'
        assert rois_sorted, 'RoIs must be sorted by sample'
        if total >= rows and n_samples > 1:      # one sample: the shared cap IS its own cap
            # the shared buffer filled up: a sample above its own cap may have pushed later samples' pairs out.  The
            # reference caps every sample on its own (:51-80) - do what it does, sample by sample
            return self._per_sample_forward(pts_xyz, pts_sample, rois, roi_sample, n_samples, cap_in_box)
        pieces, start = [], 0
        for count in per_sample:
            if count == 0:
                pieces.append(self._fake_row(pts_idx.device))
            else:
                keep = slice(start, start + min(count, self.max_all_pts))
                pieces.append((pts_idx[keep], roi_idx[keep], feats[keep]))
            start += count
        if len(pieces) == 1:
            all_inds, all_roi_inds, info = pieces[0]
        else:
            all_inds, all_roi_inds, info = (torch.cat(col, dim=0) for col in zip(*pieces))
        if self.debug:
            self.check_invariants(pts_xyz, rois[..., 1:], all_inds, all_roi_inds, info[:, :3], info[:, 3:6], info[:, 6:-1])
        return all_inds, all_roi_inds, dict(local_xyz=info[:, 3:6], boundary_offset=info[:, 6:-1],
                                            is_in_margin=info[:, -1])

    @staticmethod
    def _fake_row(dev):
        """what the reference emits for a sample without any pair (:70-75): indices -1, zero features"""
        return (torch.full((1,), -1, dtype=torch.long, device=dev), torch.full((1,), -1, dtype=torch.long, device=dev),
                torch.zeros((1, 13), dtype=torch.float, device=dev))

    def _per_sample_forward(self, pts_xyz, pts_sample, rois, roi_sample, n_samples, cap_in_box):
        """one pool call per sample, each capped at max_all_pts on its own (the reference's loop, :51-80)"""
        pieces = []
        for b in range(n_samples):
            p_sel = torch.nonzero(pts_sample == b).squeeze(1)
            r_sel = torch.nonzero(roi_sample == b).squeeze(1)
            if p_sel.numel() == 0 or r_sel.numel() == 0:
                pieces.append(self._fake_row(pts_xyz.device))
                continue
            pi, ri, ft = _valid_rows(*_pool(rois[r_sel, 1:].contiguous(), None, pts_xyz[p_sel].contiguous(), None,
                                            self.extra_wlh, cap_in_box, self.max_all_pts))
            if int(pi[0]) < 0:   # the op's own fake row: no pair in this sample
                pieces.append(self._fake_row(pts_xyz.device))
            else:
                pieces.append((p_sel[pi], r_sel[ri], ft))
        all_inds, all_roi_inds, info = (torch.cat(col, dim=0) for col in zip(*pieces))
        return all_inds, all_roi_inds, dict(local_xyz=info[:, 3:6], boundary_offset=info[:, 6:-1],
                                            is_in_margin=info[:, -1])

    def check_invariants(self, pts_xyz, rois7, inds, roi_inds, out_xyz, local_xyz, offset):
        """the reference's debug block (:96-105); pairs of the fake row (-1) are skipped"""
        ok = inds >= 0
        if not bool(ok.any()):
            return
        roi_per_pts = rois7[roi_inds[ok]]
        out_xyz, local_xyz, offset = out_xyz[ok], local_xyz[ok], offset[ok]
        assert torch.isclose(pts_xyz[inds[ok]], out_xyz).all()
        assert torch.isclose(offset[:, 0] + offset[:, 3], roi_per_pts[:, 4]).all()
        assert torch.isclose(offset[:, 1] + offset[:, 4], roi_per_pts[:, 3]).all()
        assert torch.isclose(offset[:, 2] + offset[:, 5], roi_per_pts[:, 5]).all()
        assert (local_xyz[:, 0].abs() < roi_per_pts[:, 4] + self.extra_wlh[0] + 1e-5).all()
        assert (local_xyz[:, 1].abs() < roi_per_pts[:, 3] + self.extra_wlh[1] + 1e-5).all()
        assert (local_xyz[:, 2].abs() < roi_per_pts[:, 5] + self.extra_wlh[2] + 1e-5).all()

    def fast_single_sample_forward(self, pts_xyz, rois, max_inbox_point=None):
        """one sample (:107-133; the reference passes no max_all_pts here: the op's default of 50000 applies)"""
        cap_in_box = self.max_inbox_point if max_inbox_point is None else max_inbox_point
        pts_idx, roi_idx, info = dynamic_point_pool(rois[..., 1:].contiguous(), pts_xyz.contiguous(), self.extra_wlh,
                                                    cap_in_box)
        return pts_idx, roi_idx, dict(local_xyz=info[:, 3:6], boundary_offset=info[:, 6:-1], is_in_margin=info[:, -1])
