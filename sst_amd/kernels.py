"""Tensor-level wrappers around the C ABI (include/sst_amd.h) + the autograd Functions of the path.

Every function here launches HIP kernels through libsst_amd.so on the current torch stream.
Nothing in this module computes on the CPU or through ATen eager kernels, except for allocating
output tensors and (where the reference API itself needs a host-side size) one ``.item()`` readback.
"""
import math

import os

import torch
from torch.autograd import Function

from . import _lib

REDUCE = {'sum': 0, 'mean': 1, 'avg': 1, 'max': 2}


# ----------------------------------------------------------------------------------------------
# (a1) dynamic voxelization
# ----------------------------------------------------------------------------------------------
_CONSTANTS = {}


def const_tensor(values, device, dtype=torch.float32):
    """a small constant (voxel size, range, normaliser ...) as a device tensor, uploaded ONCE per (values, device, dtype):
    ``torch.tensor(list, device=cuda)`` is a pageable host-to-device copy, i.e. a host synchronisation per call.
    The tensor is shared: never write to it in place."""
    key = (tuple(float(v) for v in values), str(device), dtype)
    t = _CONSTANTS.get(key)
    if t is None:
        t = torch.tensor([float(v) for v in values], dtype=dtype).to(device)
        _CONSTANTS[key] = t
    return t


def voxel_grid(voxel_size, coors_range):
    """grid (x, y, z) the kernel clamps to: fp32 ceil((max-min)/v), voxelization_cuda.cu:355-357."""
    import ctypes
    g = (ctypes.c_int32 * 3)()
    _lib.load().sst_dynamic_voxelize_grid(_lib.farray(voxel_size), _lib.farray(coors_range), g)
    return [int(g[0]), int(g[1]), int(g[2])]


def dynamic_voxelize(points, voxel_size, coors_range, coors=None, batch_idx=-1):
    """points [N, C>=3] fp32 (cuda) -> coors [N,3] int32 (z,y,x), or [N,4] (b,z,y,x) if batch_idx >= 0."""
    if points.dtype != torch.float32:
        raise RuntimeError('sst_amd.dynamic_voxelize: points must be float32')
    _lib.require_cuda(points)
    n = points.size(0)
    ncol = 4 if batch_idx >= 0 else 3
    if coors is None:
        coors = torch.empty((n, ncol), dtype=torch.int32, device=points.device)
    else:
        _lib.require_cuda(coors)
        if coors.dtype != torch.int32 or coors.size(0) != n or coors.size(1) != ncol:
            raise RuntimeError('sst_amd.dynamic_voxelize: coors must be int32 [N,%d]' % ncol)
    rc = _lib.load().sst_dynamic_voxelize_f32(
        _lib.ptr(points), n, points.stride(0), _lib.farray(voxel_size), _lib.farray(coors_range),
        _lib.ptr(coors), coors.stride(0) if n > 0 else ncol, ncol - 3, batch_idx, _lib.stream_ptr())
    _lib.check(rc, 'sst_dynamic_voxelize_f32')
    return coors


# ----------------------------------------------------------------------------------------------
# device primitives (exposed for tests)
# ----------------------------------------------------------------------------------------------
def exclusive_scan_i32(x):
    _lib.require_cuda(x)
    assert x.dtype == torch.int32 and x.dim() == 1
    n = x.numel()
    lib = _lib.load()
    out = torch.empty_like(x)
    total = torch.zeros(1, dtype=torch.int32, device=x.device)
    ws = _lib.workspace(lib.sst_scan_workspace_bytes(n), x.device)
    rc = lib.sst_exclusive_scan_i32(_lib.ptr(x), _lib.ptr(out), n, _lib.ptr(total), _lib.ptr(ws), _lib.stream_ptr())
    _lib.check(rc, 'sst_exclusive_scan_i32')
    return out, total


def sort_pairs_u64(keys, key_bits):
    """keys: int64 tensor holding non-negative values < 2**key_bits. Returns (sorted_keys, perm int32)."""
    _lib.require_cuda(keys)
    assert keys.dtype == torch.int64 and keys.dim() == 1
    n = keys.numel()
    lib = _lib.load()
    kin = keys.clone()
    kout = torch.empty_like(keys)
    perm = torch.empty(n, dtype=torch.int32, device=keys.device)
    ws = _lib.workspace(lib.sst_sort_workspace_bytes(n), keys.device)
    rc = lib.sst_sort_pairs_u64(_lib.ptr(kin), _lib.ptr(kout), _lib.ptr(perm), n, int(key_bits), _lib.ptr(ws),
                                _lib.stream_ptr())
    _lib.check(rc, 'sst_sort_pairs_u64')
    return kout, perm


class UniquePlan(object):
    """Result of unique_rows: the grouping of N rows into M sorted-unique rows (all device int32)."""
    __slots__ = ('n', 'm', 'num', 'perm', 'inverse', 'offsets', 'ukeys', 'mins', 'extents', 'ncols', 'has_invalid_group',
                 'scratch')

    def counts(self):
        return self.offsets[1:self.m + 1] - self.offsets[:self.m]


def unique_rows(coors, mins=None, extents=None, invalid_if_negative=False, defer_count=False):
    """Sorted-unique of integer rows (torch.unique(dim=0, return_inverse) / at::unique_dim semantics).

    coors: [N, k] int32 or int64 (cuda, contiguous).  mins/extents: per-column lower bound and extent;
    computed from the data (one host sync, like torch.unique itself) when not given.
    invalid_if_negative: 0 = plain lexicographic unique; 1 = rows containing a negative entry form ONE group
    that sorts first; 2 = batched form (column 0 = batch index, see include/sst_amd.h).
    """
    _lib.require_cuda(coors)
    if coors.dtype not in (torch.int32, torch.int64) or coors.dim() != 2:
        raise RuntimeError('sst_amd.unique_rows: coors must be a 2-D int32/int64 tensor')
    n, k = coors.shape
    dev = coors.device
    lib = _lib.load()
    if n > 0 and (mins is None or extents is None):
        lo = coors.amin(0)
        hi = coors.amax(0)
        lohi = torch.stack([lo, hi]).tolist()  # host sync
        mins = [int(v) for v in lohi[0]]
        if int(invalid_if_negative) == 1:
            mins = [max(v, 0) for v in mins]
        extents = [max(int(h) - int(l) + 1, 1) for l, h in zip(mins, lohi[1])]
    elif n == 0:
        mins = [0] * k
        extents = [1] * k
    plan = UniquePlan()
    plan.n, plan.ncols = n, k
    plan.mins, plan.extents = list(mins), list(extents)
    plan.perm = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    plan.inverse = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    plan.offsets = torch.empty(n + 1, dtype=torch.int32, device=dev)
    plan.ukeys = torch.empty(max(n, 1), dtype=torch.int64, device=dev)
    num = torch.zeros(1, dtype=torch.int32, device=dev)
    ws = _lib.workspace(lib.sst_unique_workspace_bytes(n), dev)
    rc = lib.sst_unique_rows(_lib.ptr(coors), 1 if coors.dtype == torch.int64 else 0, n, k,
                             coors.stride(0) if n > 0 else k, _lib.i64array(mins), _lib.i64array(extents),
                             int(invalid_if_negative), _lib.ptr(plan.perm), _lib.ptr(plan.inverse),
                             _lib.ptr(plan.offsets), _lib.ptr(plan.ukeys), _lib.ptr(num), _lib.ptr(ws),
                             _lib.stream_ptr())
    _lib.check(rc, 'sst_unique_rows')
    plan.num = num
    if defer_count:
        plan.m = None             # stays on the device (plan.num); the caller works with the upper bound n
    else:
        plan.m = int(num.item())  # the one host sync the reference API forces (output shape [M, ...])
    plan.inverse = plan.inverse[:n]
    plan.perm = plan.perm[:n]
    return plan


def unpack_unique_rows(plan, dtype, first=0, count=None, out_cols=None, col0=0, out=None):
    """Decode the packed keys of groups [first, first+count) back into coordinate rows."""
    lib = _lib.load()
    m = plan.m - first if count is None else count
    k = plan.ncols
    dev = plan.ukeys.device
    width = k if out_cols is None else out_cols
    if out is None:
        out = torch.empty((m, width), dtype=dtype, device=dev)
    if m > 0:
        keys = plan.ukeys[first:first + m]
        rc = lib.sst_unpack_keys(_lib.ptr(keys), m, k, _lib.i64array(plan.mins), _lib.i64array(plan.extents),
                                 _lib.ptr(out), 1 if dtype == torch.int64 else 0, out.stride(0), col0,
                                 _lib.stream_ptr())
        _lib.check(rc, 'sst_unpack_keys')
    return out


# ----------------------------------------------------------------------------------------------
# (a3/a5/a15) segmented reduce with autograd
# ----------------------------------------------------------------------------------------------
def _long_work_list(plan, n, m, c, dev):
    """work list of the long groups of a grouping (csrc/scatter.hip seg_reduce_fwd_work_k / seg_reduce_merge_k): counters zeroed
    once per plan, the kernels leave them zeroed; one list per stream (launches on different streams may overlap) and width"""
    cache = getattr(plan, 'scratch', None)
    if cache is None:
        cache = plan.scratch = {}
    key = ('work', _lib.stream_ptr().value, (c + 3) // 4)
    buf = cache.get(key)
    need = int(_lib.load().sst_segment_reduce_work_words(n, max(int(m), int(getattr(plan, 'm', 0) or 0)), c))
    if buf is None or buf.numel() < need:
        buf = cache[key] = torch.zeros(need, dtype=torch.int32, device=dev)
    return buf


def _segment_reduce_fwd(feats, perm, offsets, m, mode, want_argmax, group_index=None, m_limit=None, plan=None,
                        scale_shift=None):
    """scale_shift = (scale [c], shift [c]): values are read as relu(x * scale + shift) (needs a plan and c % 4 == 0)"""
    n, c = feats.shape
    out = torch.empty((m, c), dtype=torch.float32, device=feats.device)
    argmax = torch.empty((m, c), dtype=torch.int32, device=feats.device) if want_argmax else None
    # with a plan to keep the work list on: long groups (a voxel next to the sensor holds thousands of points of a real sweep)
    # are reduced by a workgroup each instead of by one thread per channel vector
    work = _long_work_list(plan, n, m, c, feats.device) if (plan is not None and n > 0
                                                      and os.environ.get('SST_SEG_WORK', '1') != '0') else None
    rc = _lib.load().sst_segment_reduce_fwd_work_f32(_lib.ptr(feats), n, c, _lib.ptr(perm), _lib.ptr(offsets),
                                                     _lib.ptr(group_index), m, mode, _lib.ptr(out), _lib.ptr(argmax),
                                                     _lib.ptr(m_limit), _lib.ptr(work), 0 if work is None else work.numel(),
                                                     _lib.ptr(scale_shift[0]) if scale_shift is not None else None,
                                                     _lib.ptr(scale_shift[1]) if scale_shift is not None else None,
                                                     _lib.stream_ptr())
    _lib.check(rc, 'sst_segment_reduce_fwd_work_f32')
    return out, argmax


def _long_group_scratch(plan, n, m, c, dev):
    """counters zeroed once per (grouping, width); seg_tiles_k leaves them zeroed (csrc/scatter.hip)"""
    cache = getattr(plan, 'scratch', None)
    if cache is None:
        cache = plan.scratch = {}
    buf = cache.get((n, m, c))
    if buf is None:
        buf = cache[(n, m, c)] = torch.zeros(int(_lib.load().sst_segment_long_scratch_bytes(n, m, c)), dtype=torch.uint8,
                                             device=dev)
    return buf


def _long_group_applies(n, m, c, inverse, group_index, m_limit):
    """few, long, uneven groups (FSD's clusters, RoI point sets): n >= 8 m"""
    cv = c // 4 if c % 4 == 0 else c
    return (group_index is None and m_limit is None and inverse is not None and m > 0 and n >= 8 * m and c <= 256
            and (256 % cv == 0 or cv * min(32, 256 // cv) <= 256) and os.environ.get('SST_SEG_LONG', '1') != '0')


class SegmentReduce(Function):
    """out[g] = reduce(feats[perm[offsets[g]:offsets[g+1]]]) for groups first..first+m-1 of a plan.

    ``inverse_shift`` is added to plan.inverse to obtain the output row of a point (used by the
    DynamicScatter "first row" quirk, where group 0 is discarded: shift = -1).
    """

    @staticmethod
    def forward(ctx, feats, perm, offsets, inverse, m, mode, inverse_shift, group_index=None, m_limit=None, plan=None):
        if feats.dtype != torch.float32:
            raise RuntimeError('sst_amd: features must be float32')
        feats = feats.contiguous()
        _lib.require_cuda(feats, perm, offsets)
        n, c = feats.shape
        if plan is not None and _long_group_applies(n, m, c, inverse, group_index, m_limit) \
                and (c % 4 != 0 or feats.data_ptr() % 16 == 0):
            out = torch.empty((m, c), dtype=torch.float32, device=feats.device)
            argmax = torch.empty((m, c), dtype=torch.int32, device=feats.device) if mode == 2 else None
            scratch = _long_group_scratch(plan, n, m, c, feats.device)
            rc = _lib.load().sst_segment_reduce_long_f32(_lib.ptr(feats), n, c, _lib.ptr(perm), _lib.ptr(inverse),
                                                         int(inverse_shift), _lib.ptr(offsets), m, mode, _lib.ptr(scratch),
                                                         _lib.ptr(out), _lib.ptr(argmax), _lib.stream_ptr())
            _lib.check(rc, 'sst_segment_reduce_long_f32')
        else:
            out, argmax = _segment_reduce_fwd(feats, perm, offsets, m, mode, mode == 2, group_index, m_limit, plan)
        ctx.mode, ctx.m, ctx.shift = mode, m, inverse_shift
        ctx.shape = feats.shape
        ctx.has_gidx = group_index is not None
        ctx.m_limit = m_limit   # device int32 [1]: rows of the output that exist (m is an upper bound), or None
        ctx.save_for_backward(offsets, inverse, argmax if argmax is not None else offsets,
                              group_index if group_index is not None else offsets)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        offsets, inverse, argmax, gidx = ctx.saved_tensors
        n, c = ctx.shape
        grad_out = grad_out.contiguous()
        grad_feats = torch.empty((n, c), dtype=torch.float32, device=grad_out.device)
        rc = _lib.load().sst_segment_reduce_bwd_f32(
            _lib.ptr(grad_out), ctx.m, c, _lib.ptr(inverse), ctx.shift, _lib.ptr(offsets),
            _lib.ptr(gidx) if ctx.has_gidx else None, _lib.ptr(argmax) if ctx.mode == 2 else None, n, ctx.mode,
            _lib.ptr(grad_feats), _lib.ptr(ctx.m_limit), _lib.stream_ptr())
        _lib.check(rc, 'sst_segment_reduce_bwd_f32')
        return grad_feats, None, None, None, None, None, None, None, None, None


def segment_reduce(feats, plan, mode, first=0, group_index=None, inverse=None, m_limit=None):
    """Reduce rows of feats over groups [first, plan.m) of a UniquePlan; mode in 'sum'|'mean'|'max'.
    group_index ([m'] int32, with ``inverse`` = point -> output row map): output row g reduces group
    group_index[g] (subset / re-ordering of the groups without gathering the result).
    m_limit (device int32 [1], with group_index): the number of valid rows of group_index when only the device
    knows it; the output then has group_index.numel() rows of which the first *m_limit are written."""
    if group_index is not None:
        return SegmentReduce.apply(feats, plan.perm, plan.offsets, inverse, group_index.numel(), REDUCE[mode], 0,
                                   group_index, m_limit, plan)
    m = plan.m - first
    offsets = plan.offsets[first:]
    return SegmentReduce.apply(feats, plan.perm, offsets, plan.inverse, m, REDUCE[mode], -first, None, None, plan)


def add_group_rows_to_row0(dg, feats, plan, groups_i32):
    """dg[0] += sum of the rows of ``feats`` that belong to the CSR groups ``groups_i32`` (negative entries: none), in a fixed
    order: the share of the points of discarded voxels in the gradient of a gather that sent them to row 0."""
    if groups_i32 is None or groups_i32.numel() == 0 or dg.size(0) == 0:
        return dg
    extra, _ = _segment_reduce_fwd(feats.contiguous(), plan.perm, plan.offsets, groups_i32.numel(), REDUCE['sum'], False,
                                   groups_i32, None, plan)
    dg[0] += extra[0] if extra.size(0) == 1 else extra.sum(0)
    return dg


def segment_argmax(feats, plan, first=0):
    """(max, argmax row index) per group — torch_scatter.scatter_max's second output."""
    feats = feats.contiguous()
    return _segment_reduce_fwd(feats, plan.perm, plan.offsets[first:], plan.m - first, 2, True, plan=plan)


# ----------------------------------------------------------------------------------------------
# (a7) in-group rank
# ----------------------------------------------------------------------------------------------
def ingroup_rank(group_inds, key_bits=None):
    _lib.require_cuda(group_inds)
    if group_inds.dtype != torch.int64 or group_inds.dim() != 1:
        raise RuntimeError('sst_amd.ingroup_rank: group_inds must be a 1-D int64 tensor')
    n = group_inds.numel()
    out = torch.empty_like(group_inds)
    if n == 0:
        return out
    if key_bits is None:
        mx = int(group_inds.max().item())
        if int(group_inds.min().item()) < 0:
            raise RuntimeError('sst_amd.ingroup_rank: negative group index')
        key_bits = max(1, int(mx).bit_length())
    lib = _lib.load()
    ws = _lib.workspace(lib.sst_ingroup_rank_workspace_bytes(n), group_inds.device)
    rc = lib.sst_ingroup_rank_i64(_lib.ptr(group_inds), n, int(key_bits), _lib.ptr(out), _lib.ptr(ws),
                                  _lib.stream_ptr())
    _lib.check(rc, 'sst_ingroup_rank_i64')
    return out


# ----------------------------------------------------------------------------------------------
# (a6, a7-a9) window coordinates and region batching
# ----------------------------------------------------------------------------------------------
def window_coors(coors, sparse_shape, window_shape):
    """coors [M,4] (b,z,y,x) int32/int64 -> (win0, ciw0, win1, ciw1), all int32."""
    _lib.require_cuda(coors)
    if coors.dtype not in (torch.int32, torch.int64) or coors.dim() != 2 or coors.size(1) != 4:
        raise RuntimeError('sst_amd.window_coors: coors must be [M,4] int32/int64')
    m = coors.size(0)
    dev = coors.device
    win0 = torch.empty(m, dtype=torch.int32, device=dev)
    win1 = torch.empty(m, dtype=torch.int32, device=dev)
    ciw0 = torch.empty((m, 3), dtype=torch.int32, device=dev)
    ciw1 = torch.empty((m, 3), dtype=torch.int32, device=dev)
    rc = _lib.load().sst_window_coors(_lib.ptr(coors), 1 if coors.dtype == torch.int64 else 0, m,
                                      _lib.i32array(sparse_shape), _lib.i32array(window_shape), _lib.ptr(win0),
                                      _lib.ptr(ciw0), _lib.ptr(win1), _lib.ptr(ciw1), _lib.stream_ptr())
    _lib.check(rc, 'sst_window_coors')
    return win0, ciw0, win1, ciw1


def region_batching(win0, win1, win_bits, levels):
    """levels: list of (max_tokens, lower, upper).  Returns a dict of device int32 tensors + counts."""
    _lib.require_cuda(win0, win1)
    m = win0.numel()
    dev = win0.device
    lib = _lib.load()

    def e(n):
        return torch.empty(n, dtype=torch.int32, device=dev)

    r = dict(keep=e(m), newidx=e(m), level0=e(m), level1=e(m), inner0=e(m), inner1=e(m), flat2win0=e(m),
             flat2win1=e(m), tok0=e(m), tok1=e(m), winoff0=e(m + 1), winoff1=e(m + 1), winlevel0=e(max(m, 1)),
             winlevel1=e(max(m, 1)))
    counts = torch.zeros(8, dtype=torch.int32, device=dev)
    flat = []
    for (cap, lo, hi) in levels:
        flat += [int(cap), int(lo), int(min(hi, 2 ** 31 - 1))]
    ws = _lib.workspace(lib.sst_region_batching_workspace_bytes(m), dev)
    rc = lib.sst_region_batching(
        _lib.ptr(win0), _lib.ptr(win1), m, int(win_bits), _lib.i32array(flat), len(levels), _lib.ptr(r['keep']),
        _lib.ptr(r['newidx']), _lib.ptr(r['level0']), _lib.ptr(r['level1']), _lib.ptr(r['inner0']),
        _lib.ptr(r['inner1']), _lib.ptr(r['flat2win0']), _lib.ptr(r['flat2win1']), _lib.ptr(r['tok0']),
        _lib.ptr(r['tok1']), _lib.ptr(r['winoff0']), _lib.ptr(r['winoff1']), _lib.ptr(r['winlevel0']),
        _lib.ptr(r['winlevel1']), _lib.ptr(counts), _lib.ptr(ws), _lib.stream_ptr())
    _lib.check(rc, 'sst_region_batching')
    r['counts'] = counts
    return r


# ----------------------------------------------------------------------------------------------
# (a12) SRA attention core with autograd
# ----------------------------------------------------------------------------------------------
class WindowPlan(object):
    """Window CSR consumed by the SRA kernels: tokens of window w are tok[winoff[w]:winoff[w+1]]."""
    __slots__ = ('tok', 'winoff', 'n_windows', 'n_tokens', 'max_tokens', '_order', 'rows_in_window_order')

    def __init__(self, tok, winoff, n_windows, n_tokens, max_tokens, rows_in_window_order=False):
        self.tok, self.winoff = tok, winoff
        self.n_windows, self.n_tokens, self.max_tokens = int(n_windows), int(n_tokens), int(max_tokens)
        self._order = None
        # tok == arange(n_tokens): the feature rows themselves are in window order (the frame plan numbers the kept voxels by
        # their place in the unshifted partition), so the fp32 register-resident kernels skip the token list - one dependent
        # load less at the head of every wave
        self.rows_in_window_order = bool(rows_in_window_order)

    def tok_ptr(self, impl):
        if self.rows_in_window_order and impl in (0, 3) and 0 < self.max_tokens <= 144:
            return None
        return _lib.ptr(self.tok)

    @property
    def order(self):
        """launch order of the windows for the register-resident kernels: ascending token count (they dispatch from the
        end: largest windows first).  Two small launches, once per plan - a plan serves every encoder layer of its shift,
        forward and backward.  None for small plans, where every workgroup is resident at once anyway."""
        if self._order is None and self.n_windows >= WINDOW_ORDER_MIN:
            order = torch.empty(self.n_windows, dtype=torch.int32, device=self.winoff.device)
            rc = _lib.load().sst_window_order_i32(_lib.ptr(self.winoff), self.n_windows, self.max_tokens, _lib.ptr(order),
                                                  _lib.stream_ptr()) if (self.winoff.is_cuda and self.winoff.dtype == torch.int32
                                                                         and not os.environ.get('SST_AMD_WINDOW_ORDER_TORCH')) \
                else _lib.SST_ERR_UNSUPPORTED
            if rc == _lib.SST_ERR_UNSUPPORTED:       # windows of 512 tokens and more: the library sort
                off = self.winoff[:self.n_windows + 1]
                order = torch.sort(off[1:] - off[:-1], stable=True)[1].to(torch.int32)
            else:
                _lib.check(rc, 'sst_window_order_i32')
            self._order = order
        return self._order


WINDOW_ORDER_MIN = 512    # windows below which the launch order is left alone (0 switches the ordering off: set to 1 << 30)


def _row_stride(t):
    if t.dim() != 2 or t.stride(1) != 1:
        raise RuntimeError('sst_amd.sra_attention: q/k/v must be 2-D with unit inner stride')
    return t.stride(0)


# bench.py hooks: when a list is installed here every SRA launch group is bracketed by a pair of HIP events
# recorded on the launch stream (torch's current stream) and (kind, start, end, n_tokens) is appended.
EVENT_SINK = None


class _KernelEvents(object):
    """A pair of HIP events bound to one kernel launch (sst_sra_attn_profile_next_fwd): elapsed_time = the kernel's
    own begin -> end on its launch stream."""

    def __init__(self, lib):
        self.lib = lib
        self.start, self.stop = lib.sst_event_create(), lib.sst_event_create()

    def elapsed_time(self, _other=None):
        return float(self.lib.sst_event_elapsed_ms(self.start, self.stop))

    def __del__(self):
        try:
            self.lib.sst_event_destroy(self.start)
            self.lib.sst_event_destroy(self.stop)
        except Exception:
            pass


EVENT_STRIDE = 1        # bench.py: time every EVENT_STRIDE-th launch of a kind (the marks cost a few us of queue drain)
EVENT_KINDS = ('sra_fwd', 'sra_bwd')
_event_counter = {'sra_fwd': 0, 'sra_bwd': 0}


def _bracket(kind, n_tokens, fn):
    """Run ``fn`` (one C-ABI call that ends in one launch of the register-resident SRA kernel of ``kind``) with HIP
    events attached to that launch itself (hipExtLaunchKernelGGL start / stop, armed through the library's one-shot
    hooks): the interval is the kernel's own begin -> end on its stream."""
    if EVENT_SINK is None or kind not in EVENT_KINDS:
        return fn()
    _event_counter[kind] += 1
    if (_event_counter[kind] - 1) % EVENT_STRIDE != 0:
        return fn()
    lib = _lib.load()
    arm = lib.sst_sra_attn_profile_next_fwd if kind == 'sra_fwd' else lib.sst_sra_attn_profile_next_bwd
    ke = _KernelEvents(lib)
    arm(ke.start, ke.stop)
    r = fn()
    arm(None, None)  # disarm if the call took another kernel path
    EVENT_SINK.append((kind, ke, ke, n_tokens))
    return r


def _sra_fwd(q, k, v, plan, n_heads, scale, impl):
    for t in (q, k, v):
        if t.dtype != torch.float32 or not t.is_cuda:
            raise RuntimeError('sst_amd.sra_attention: q, k, v must be float32 CUDA tensors')
    m, c = q.shape
    if c != n_heads * 16:
        raise RuntimeError('sst_amd.sra_attention: head_dim must be 16')
    if plan.n_tokens < m:
        o = torch.zeros((m, c), dtype=torch.float32, device=q.device)
    else:
        o = torch.empty((m, c), dtype=torch.float32, device=q.device)
    lse = torch.empty((m, n_heads), dtype=torch.float32, device=q.device)
    order = plan.order

    def call(tok_p):
        return _bracket('sra_fwd', plan.n_tokens, lambda: _lib.load().sst_sra_attn_fwd_ord_f32(
            _lib.ptr(q), _lib.ptr(k), _lib.ptr(v), _row_stride(q), _row_stride(k), _row_stride(v),
            tok_p, _lib.ptr(plan.winoff), _lib.ptr(order) if order is not None else None, plan.n_windows,
            n_heads, float(scale), plan.max_tokens, impl, _lib.ptr(o), o.stride(0), _lib.ptr(lse), _lib.stream_ptr()))
    tok_p = plan.tok_ptr(impl)
    rc = call(tok_p)
    if rc == _lib.SST_ERR_UNSUPPORTED and tok_p is None:   # layouts only the generic kernels take: they want the list
        rc = call(_lib.ptr(plan.tok))
    _lib.check(rc, 'sst_sra_attn_fwd_ord_f32')
    return o, lse


def _sra_bwd(q, k, v, o, lse, grad_o, plan, n_heads, scale, impl, dq, dk, dv):
    m = q.size(0)
    lib = _lib.load()
    ws = _lib.workspace(lib.sst_sra_attn_bwd_workspace_bytes(m, n_heads), q.device)
    order = plan.order

    def call(tok_p):
        return _bracket('sra_bwd', plan.n_tokens, lambda: lib.sst_sra_attn_bwd_ord_f32(
            _lib.ptr(q), _lib.ptr(k), _lib.ptr(v), _lib.ptr(o), _lib.ptr(grad_o), _lib.ptr(lse), _row_stride(q),
            _row_stride(k), _row_stride(v), o.stride(0), grad_o.stride(0), tok_p,
            _lib.ptr(plan.winoff), _lib.ptr(order) if order is not None else None, plan.n_windows, m, n_heads, scale,
            plan.max_tokens, impl, _lib.ptr(dq), _lib.ptr(dk), _lib.ptr(dv), _row_stride(dq), _row_stride(dk),
            _row_stride(dv), _lib.ptr(ws), _lib.stream_ptr()))
    tok_p = plan.tok_ptr(impl)
    rc = call(tok_p)
    if rc == _lib.SST_ERR_UNSUPPORTED and tok_p is None:
        rc = call(_lib.ptr(plan.tok))
    _lib.check(rc, 'sst_sra_attn_bwd_ord_f32')


def cosine_kernels_ok(plan, n_heads, impl=0):
    """scaled cosine attention inside the register-resident kernels (sst_sra_attn_cos_{fwd,bwd}_f32): windows of <= 144 tokens,
    heads in groups of four, the default kernel choice; anything else normalises q, k outside the kernel"""
    return impl == 0 and n_heads % 4 == 0 and 0 < plan.max_tokens <= 144


def _sra_cos_fwd(q, k, v, plan, n_heads, head_scale):
    """softmax(normalize(q) normalize(k)^T * head_scale[h]) v per window; head_scale: [n_heads] fp32 on the device.
    -> (o, lse), or None when the library does not take the layout (the caller normalises outside)"""
    m, c = q.shape
    o = (torch.zeros if plan.n_tokens < m else torch.empty)((m, c), dtype=torch.float32, device=q.device)
    lse = torch.empty((m, n_heads), dtype=torch.float32, device=q.device)
    order = plan.order
    rc = _bracket('sra_fwd', plan.n_tokens, lambda: _lib.load().sst_sra_attn_cos_fwd_f32(
        _lib.ptr(q), _lib.ptr(k), _lib.ptr(v), _row_stride(q), _row_stride(k), _row_stride(v), plan.tok_ptr(0),
        _lib.ptr(plan.winoff), _lib.ptr(order) if order is not None else None, plan.n_windows, n_heads, _lib.ptr(head_scale),
        plan.max_tokens, _lib.ptr(o), o.stride(0), _lib.ptr(lse), _lib.stream_ptr()))
    if rc == _lib.SST_ERR_UNSUPPORTED:
        return None
    _lib.check(rc, 'sst_sra_attn_cos_fwd_f32')
    return o, lse


def _sra_cos_bwd(q, k, v, o, lse, grad_o, plan, n_heads, head_scale, dq, dk, dv):
    """-> r [M, n_heads] = normalize(q) . d normalize(q): d head_scale = r.sum(0) / head_scale"""
    m = q.size(0)
    r = (torch.zeros if plan.n_tokens < m else torch.empty)((m, n_heads), dtype=torch.float32, device=q.device)
    order = plan.order
    rc = _bracket('sra_bwd', plan.n_tokens, lambda: _lib.load().sst_sra_attn_cos_bwd_f32(
        _lib.ptr(q), _lib.ptr(k), _lib.ptr(v), _lib.ptr(o), _lib.ptr(grad_o), _lib.ptr(lse), _row_stride(q), _row_stride(k),
        _row_stride(v), o.stride(0), grad_o.stride(0), plan.tok_ptr(0), _lib.ptr(plan.winoff),
        _lib.ptr(order) if order is not None else None, plan.n_windows, m, n_heads, _lib.ptr(head_scale), plan.max_tokens,
        _lib.ptr(dq), _lib.ptr(dk), _lib.ptr(dv), _row_stride(dq), _row_stride(dk), _row_stride(dv), _lib.ptr(r),
        _lib.stream_ptr()))
    _lib.check(rc, 'sst_sra_attn_cos_bwd_f32')
    return r


def head_scale_grad(r, head_scale):
    """gradient of the per-head score scale from the backward kernel's r [M, H] (column sums / scale)"""
    from .dense import colsum
    return colsum(r) / head_scale


class SRACosineAttentionQKV(Function):
    """scaled cosine attention on q|k packed as [M, 2C] and v [M, C] (cosine_msa.py:123-170): normalisation and the per-head
    scale 1 / clamp(tau) inside the kernels; differentiable in qk, v and head_scale ([H], device)."""

    @staticmethod
    def forward(ctx, qk, v, head_scale, plan, n_heads):
        c = v.size(1)
        head_scale = head_scale.contiguous()
        res = _sra_cos_fwd(qk[:, :c], qk[:, c:], v, plan, n_heads, head_scale)
        if res is None:
            raise RuntimeError('sst_amd: cosine attention kernels do not take this layout (check cosine_kernels_ok first)')
        o, lse = res
        ctx.plan, ctx.n_heads = plan, n_heads
        ctx.save_for_backward(qk, v, o, lse, head_scale)
        return o

    @staticmethod
    def backward(ctx, grad_o):
        qk, v, o, lse, head_scale = ctx.saved_tensors
        c = v.size(1)
        grad_o = grad_o.contiguous()
        full = ctx.plan.n_tokens == v.size(0)
        dqk = _grad_buf(qk.shape, qk.device, full)
        dv = _grad_buf(v.shape, v.device, full)
        r = _sra_cos_bwd(qk[:, :c], qk[:, c:], v, o, lse, grad_o, ctx.plan, ctx.n_heads, head_scale, dqk[:, :c], dqk[:, c:], dv)
        return dqk, dv, head_scale_grad(r, head_scale), None, None


def sra_cosine_attention_qk_v(qk, v, head_scale, plan, n_heads):
    return SRACosineAttentionQKV.apply(qk, v, head_scale, plan, n_heads)


def _grad_buf(shape, device, full):
    return (torch.empty if full else torch.zeros)(shape, dtype=torch.float32, device=device)


class SRAAttention(Function):
    """q, k, v given as three tensors (each [M, C], row-strided views allowed)."""

    @staticmethod
    def forward(ctx, q, k, v, plan, n_heads, scale, impl):
        o, lse = _sra_fwd(q, k, v, plan, n_heads, scale, impl)
        ctx.plan, ctx.n_heads, ctx.scale, ctx.impl = plan, n_heads, float(scale), impl
        ctx.save_for_backward(q, k, v, o, lse)
        return o

    @staticmethod
    def backward(ctx, grad_o):
        q, k, v, o, lse = ctx.saved_tensors
        grad_o = grad_o.contiguous()
        full = ctx.plan.n_tokens == q.size(0)
        dq = _grad_buf(q.shape, q.device, full)
        dk = _grad_buf(q.shape, q.device, full)
        dv = _grad_buf(q.shape, q.device, full)
        _sra_bwd(q, k, v, o, lse, grad_o, ctx.plan, ctx.n_heads, ctx.scale, ctx.impl, dq, dk, dv)
        return dq, dk, dv, None, None, None, None


class SRAAttentionQKV(Function):
    """q|k packed as one [M, 2C] tensor (the q = k = x + pos projection) and v [M, C]: the gradients come
    back as one contiguous [M, 2C] and one [M, C] tensor, so the projection backward needs no re-packing."""

    @staticmethod
    def forward(ctx, qk, v, plan, n_heads, scale, impl):
        c = v.size(1)
        q, k = qk[:, :c], qk[:, c:]
        o, lse = _sra_fwd(q, k, v, plan, n_heads, scale, impl)
        ctx.plan, ctx.n_heads, ctx.scale, ctx.impl = plan, n_heads, float(scale), impl
        ctx.save_for_backward(qk, v, o, lse)
        return o

    @staticmethod
    def backward(ctx, grad_o):
        qk, v, o, lse = ctx.saved_tensors
        c = v.size(1)
        grad_o = grad_o.contiguous()
        full = ctx.plan.n_tokens == v.size(0)
        dqk = _grad_buf(qk.shape, qk.device, full)
        dv = _grad_buf(v.shape, v.device, full)
        _sra_bwd(qk[:, :c], qk[:, c:], v, o, lse, grad_o, ctx.plan, ctx.n_heads, ctx.scale, ctx.impl,
                 dqk[:, :c], dqk[:, c:], dv)
        return dqk, dv, None, None, None, None


def sra_attention(q, k, v, plan, n_heads, scale=None, impl=0):
    """softmax(q k^T * scale) v inside each window of ``plan``; q,k,v: [M, n_heads*16] fp32 (row-strided ok)."""
    if scale is None:
        scale = 1.0 / math.sqrt(16.0)
    return SRAAttention.apply(q, k, v, plan, n_heads, scale, impl)


def sra_attention_qk_v(qk, v, plan, n_heads, scale=None, impl=0):
    if scale is None:
        scale = 1.0 / math.sqrt(16.0)
    return SRAAttentionQKV.apply(qk, v, plan, n_heads, scale, impl)


# ----------------------------------------------------------------------------------------------
# (a10) row gather / scatter
# ----------------------------------------------------------------------------------------------
def gather_rows(src, idx, fill=0.0):
    _lib.require_cuda(src, idx)
    assert src.dtype == torch.float32 and idx.dtype == torch.int32 and src.dim() == 2
    n, c = idx.numel(), src.size(1)
    out = torch.empty((n, c), dtype=torch.float32, device=src.device)
    rc = _lib.load().sst_gather_rows_f32(_lib.ptr(src), src.stride(0), _lib.ptr(idx), n, c, float(fill),
                                         _lib.ptr(out), c, _lib.stream_ptr())
    _lib.check(rc, 'sst_gather_rows_f32')
    return out


class _AddTableRows(torch.autograd.Function):
    """x + table[idx] with the gradient of x passed through (the table rows are constants here: positional embedding)"""

    @staticmethod
    def forward(ctx, x, table, idx):
        return _add_table_rows(x, table, idx)

    @staticmethod
    def backward(ctx, grad):
        return grad, None, None


def add_table_rows(x, table, idx):
    """x + table[idx] (x [M, C] fp32, idx int32 [M]) in one launch (index cast + index_select + add otherwise); differentiable
    with respect to x"""
    return _AddTableRows.apply(x, table, idx)


def _add_table_rows(x, table, idx):
    _lib.require_cuda(x, table, idx)
    m, c = x.shape
    if not (x.dtype == torch.float32 and table.dtype == torch.float32 and idx.dtype == torch.int32 and c % 4 == 0
            and x.stride(1) == 1 and table.stride(1) == 1 and x.stride(0) % 4 == 0 and table.stride(0) % 4 == 0
            and x.data_ptr() % 16 == 0 and table.data_ptr() % 16 == 0 and idx.numel() == m and table.size(1) == c):
        return x + table.index_select(0, idx.long())
    out = torch.empty((m, c), dtype=torch.float32, device=x.device)
    rc = _lib.load().sst_add_table_rows_f32(_lib.ptr(x), x.stride(0), _lib.ptr(table), table.stride(0), _lib.ptr(idx), m, c,
                                            _lib.ptr(out), c, _lib.stream_ptr())
    _lib.check(rc, 'sst_add_table_rows_f32')
    return out


def scatter_rows(src, idx, out):
    _lib.require_cuda(src, idx, out)
    assert src.dtype == torch.float32 and idx.dtype == torch.int32 and src.dim() == 2
    rc = _lib.load().sst_scatter_rows_f32(_lib.ptr(src), src.stride(0), _lib.ptr(idx), idx.numel(), src.size(1),
                                          _lib.ptr(out), out.stride(0), _lib.stream_ptr())
    _lib.check(rc, 'sst_scatter_rows_f32')
    return out


def vfe_decorate(features, inverse, voxel_mean, cluster_div, coors, voxel_size, offsets, with_cluster, with_center):
    """[features | xyz - voxel mean (/ cluster_div) | xyz - voxel centre] in one launch (csrc/scatter.hip,
    sst_vfe_decorate_f32); the decorate step of DynamicVFE / DynamicScatterVFE (voxel_encoder.py:252-271, 569-589)."""
    _lib.require_cuda(features)
    n, c = features.shape
    out = torch.empty((n, c + 3 * int(bool(with_cluster)) + 3 * int(bool(with_center))), dtype=torch.float32,
                      device=features.device)
    if n == 0:
        return out
    inv = mean = None
    if with_cluster:
        inv = inverse if inverse.dtype == torch.int32 else inverse.int()
        inv = inv.contiguous()
        mean = voxel_mean if voxel_mean.stride(1) == 1 else voxel_mean.contiguous()
    cc = None
    if with_center:
        cc = coors if coors.stride(1) == 1 else coors.contiguous()
        if cc.dtype not in (torch.int32, torch.int64):
            cc = cc.long()
    lib = _lib.load()
    rc = lib.sst_vfe_decorate_f32(_lib.ptr(features), features.stride(0), n, c, _lib.ptr(inv), _lib.ptr(mean),
                                  mean.stride(0) if mean is not None else 3, float(cluster_div), _lib.ptr(cc),
                                  int(cc is not None and cc.dtype == torch.int64), cc.stride(0) if cc is not None else 4,
                                  _lib.farray(voxel_size), _lib.farray(offsets), int(bool(with_cluster)),
                                  int(bool(with_center)), _lib.ptr(out), out.stride(0), _lib.stream_ptr())
    _lib.check(rc, 'sst_vfe_decorate_f32')
    return out


# ----------------------------------------------------------------------------------------------
# (a15 / f1) gather by the inverse map + concatenation between the layers of the point-group encoders
# ----------------------------------------------------------------------------------------------
class ConcatGather(Function):
    """cat([x, g[idx]], dim=1) in one pass (csrc/scatter.hip).  Backward: d(x) is the left column block of the incoming
    gradient; d(g) is a segmented sum over the members of every group, taken through ``group_sum`` (a callable
    [N, C] -> [G, C], the deterministic CSR reduce of the grouping the index came from) when the caller supplies one,
    else an index_add."""

    @staticmethod
    def forward(ctx, x, g, idx32, group_sum):
        n, c1 = x.shape
        c2 = g.size(1)
        x = x if x.stride(1) == 1 else x.contiguous()
        g = g if g.stride(1) == 1 else g.contiguous()
        out = torch.empty((n, c1 + c2), dtype=torch.float32, device=x.device)
        rc = _lib.load().sst_concat_gather_f32(_lib.ptr(x), x.stride(0), c1, _lib.ptr(g), g.stride(0), c2, _lib.ptr(idx32),
                                               n, _lib.ptr(out), _lib.stream_ptr())
        _lib.check(rc, 'sst_concat_gather_f32')
        ctx.save_for_backward(idx32)
        ctx.c1, ctx.rows, ctx.group_sum = c1, g.size(0), group_sum
        return out

    @staticmethod
    def backward(ctx, dy):
        (idx32,) = ctx.saved_tensors
        c1 = ctx.c1
        dx = dy[:, :c1] if ctx.needs_input_grad[0] else None
        dg = None
        if ctx.needs_input_grad[1]:
            part = dy[:, c1:].contiguous()
            if ctx.group_sum is not None:
                dg = ctx.group_sum(part)
            else:
                dg = torch.zeros((ctx.rows, part.size(1)), dtype=part.dtype, device=part.device)
                dg.index_add_(0, idx32.long().clamp(min=0), part)
        return dx, dg, None, None


def concat_gather(x, g, idx, group_sum=None):
    """[x | g[idx]] for [N, C1] point features, [G, C2] group features and the point -> group map ``idx`` (int32 or
    int64, negative entries read row 0).  Shapes the kernel is not built for go through torch."""
    ok = (x.is_cuda and x.dtype == torch.float32 and g.dtype == torch.float32 and x.dim() == 2 and g.dim() == 2
          and x.size(1) % 4 == 0 and g.size(1) % 4 == 0 and x.size(1) >= 4 and g.size(1) >= 4 and g.size(0) > 0
          and x.data_ptr() % 16 == 0 and g.data_ptr() % 16 == 0 and x.stride(0) % 4 == 0 and g.stride(0) % 4 == 0)
    if not ok:
        return torch.cat([x, g[idx.long().clamp(min=0)]], dim=1)
    idx32 = idx if idx.dtype == torch.int32 else idx.to(torch.int32)
    return ConcatGather.apply(x, g, idx32.contiguous(), group_sum)
