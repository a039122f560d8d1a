"""Sparse 3-D convolution on the GPU (first part): rulebook, SubM / strided / transposed / inverse
convolution with forward, data gradient and weight gradient.

Mirrors the Python surface of the reference's vendored spconv (mmdet3d/ops/spconv): ``SparseConvTensor``
(structure.py:21-73), ``SparseModule`` / ``SparseSequential`` / ``ToDense`` (modules.py:36-215), ``get_conv_output_size``
/ ``get_deconv_output_size`` / ``get_indice_pairs`` / ``indice_conv`` / ``indice_conv_backward`` (ops.py:20-160), the
autograd functions ``indice_conv`` / ``indice_inverse_conv`` / ``indice_subm_conv`` (functional.py:20-75) and
``SparseConvolution`` with its 3-D subclasses (conv.py:37-480): same names, constructor arguments, ``weight`` layout
``[kz, ky, kx, Cin, Cout]`` and ``indice_dict`` entries, so checkpoints and the reference's backbones
(middle_encoders/sparse_unet.py) see the same objects.

What differs underneath (csrc/spconv.hip): the rulebook is kept as two dense int32 maps next to the reference's pair
lists; a convolution is ONE gathered-GEMM launch (fp32 MFMA, no atomics) instead of K x (gather, mm, scatter-add);
output voxels of a strided / transposed convolution are numbered by ascending (b, z, y, x) -- the order of the
reference's GPU path (``torch::_unique`` of the linear indices, spconv_ops.h:128); pairs of an offset are ordered by
input row (the reference's CPU order; its GPU order is left to atomics).  3-D int32 indices and fp32 features only.
"""
import math
import os
import weakref

import numpy as np
import torch
from torch import nn
from torch.nn import init
from torch.nn.parameter import Parameter

from . import _lib
from . import kernels as K
from .registry import Registry

CONV_LAYERS = Registry('conv layer')  # mmcv.cnn.CONV_LAYERS


# ------------------------------------------------------------------------------------------------ structure.py
def scatter_nd(indices, updates, shape):
    """structure.py:5-18 (no repeated indices)"""
    ret = torch.zeros(*shape, dtype=updates.dtype, device=updates.device)
    ndim = indices.shape[-1]
    output_shape = list(indices.shape[:-1]) + shape[indices.shape[-1]:]
    flat = indices.view(-1, ndim)
    slices = [flat[:, i] for i in range(ndim)] + [Ellipsis]
    ret[tuple(slices)] = updates.view(*output_shape)
    return ret


class SparseConvTensor(object):
    """features [N, C] fp32, indices [N, 4] int32 (batch, z, y, x), spatial_shape (z, y, x), batch_size."""

    def __init__(self, features, indices, spatial_shape, batch_size, grid=None):
        self.features = features
        self.indices = indices if indices.dtype == torch.int32 else indices.int()
        self.spatial_shape = spatial_shape
        self.batch_size = batch_size
        self.indice_dict = {}
        self.grid = grid

    @property
    def spatial_size(self):
        return np.prod(self.spatial_shape)

    def find_indice_pair(self, key):
        if key is None:
            return None
        return self.indice_dict.get(key)

    def replace_feature(self, new_features):
        """spconv 2.x API used by the reference's blocks (ops/sparse_block.py:13-18): same voxels, new features"""
        out = SparseConvTensor(new_features, self.indices, self.spatial_shape, self.batch_size, self.grid)
        out.indice_dict = self.indice_dict
        return out

    def dense(self, channels_first=True):
        output_shape = [self.batch_size] + list(self.spatial_shape) + [self.features.shape[1]]
        res = scatter_nd(self.indices.long(), self.features, output_shape)
        if not channels_first:
            return res
        ndim = len(self.spatial_shape)
        order = list(range(0, ndim + 1))
        order.insert(1, ndim + 1)
        return res.permute(*order).contiguous()

    @property
    def sparity(self):
        return self.indices.shape[0] / np.prod(self.spatial_shape) / self.batch_size


# ------------------------------------------------------------------------------------------------ ops.py
def _triple(v, ndim=3):
    return [int(e) for e in v] if isinstance(v, (list, tuple)) else [int(v)] * ndim


def get_conv_output_size(input_size, kernel_size, stride, padding, dilation):
    """ops.py:20-30"""
    out = []
    for i in range(len(input_size)):
        size = (input_size[i] + 2 * padding[i] - dilation[i] * (kernel_size[i] - 1) - 1) // stride[i] + 1
        out.append(1 if kernel_size[i] == -1 else size)
    return out


def get_deconv_output_size(input_size, kernel_size, stride, padding, dilation, output_padding):
    """ops.py:33-43"""
    out = []
    for i in range(len(input_size)):
        if kernel_size[i] == -1:
            raise ValueError("deconv don't support kernel_size < 0")
        out.append((input_size[i] - 1) * stride[i] - 2 * padding[i] + kernel_size[i] + output_padding[i])
    return out


class Rulebook(object):
    """in2out [K, n], out2in [K, m] int32 maps (-1 = no partner) + the reference's pair lists.

    The pair-list tensor carries its Rulebook as an attribute (it rides along wherever the reference passes
    ``indice_pairs`` around); the Rulebook refers back to the tensor only weakly, so the pair lists and the two dense
    maps (~4 K n int32 each) are released by reference counting as soon as the tensor is, not by the cycle collector."""
    __slots__ = ('in2out', 'out2in', '_pairs_ref', 'num', 'n', 'm', 'kvol', '_total', '_orders')

    def tile_order(self, mapping, rows, tile_rows):
        """launch order of the row tiles of the output-stationary kernel for one of the two maps: heaviest tile first
        (csrc/spconv_os.hip; two small launches + a sort ONCE per rulebook, map and tile height - a submanifold rulebook
        serves every convolution of its level, forward and backward)"""
        if getattr(self, '_orders', None) is None:
            self._orders = {}
        key = (mapping.data_ptr(), tile_rows)
        order = self._orders.get(key)
        if order is None:
            lib = _lib.load()
            work = torch.empty(-(-rows // tile_rows), dtype=torch.int32, device=mapping.device)
            rc = lib.sst_spconv_os_tile_work_i32(_lib.ptr(mapping), rows, mapping.size(0), tile_rows, _lib.ptr(work),
                                                 _lib.stream_ptr())
            _lib.check(rc, 'sst_spconv_os_tile_work_i32')
            order = torch.sort(work, descending=True, stable=True)[1].to(torch.int32)
            self._orders[key] = order
        return order

    def known_total_pairs(self):
        """the pair count if somebody has read it back already, else -1 (callers then size by the upper bound K x n)"""
        return -1 if self._total is None else self._total

    @property
    def total_pairs(self):
        """sum of the pairs over the offsets: ONE host read-back, on first use only - the convolutions themselves never
        need it (the rulebook of a submanifold level is then built without any host synchronisation)"""
        if self._total is None:
            self._total = int(self.num.sum().item()) if self.n > 0 else 0
        return self._total

    @property
    def density(self):
        """populated share of the (offset, row) slots (picks the form of the first-generation kernels)"""
        return self.total_pairs / max(1, self.kvol * max(self.n, self.m))

    @property
    def pairs(self):
        t = self._pairs_ref()
        if t is None:
            raise RuntimeError('sst_amd.spconv: the indice_pairs tensor of this rulebook has been released')
        return t


def _i32(vals):
    return _lib.i32array([int(v) for v in vals])


def _finish_rulebook(in2out, kvol, n, m, dev, out2in=None):
    lib = _lib.load()
    rb = Rulebook()
    rb.in2out, rb.kvol, rb.n, rb.m = in2out, kvol, n, m
    if out2in is not None:
        rb.out2in = out2in
    else:
        rb.out2in = torch.empty((kvol, m), dtype=torch.int32, device=dev)
        _lib.check(lib.sst_spconv_invert_map_i32(_lib.ptr(in2out), kvol, n, m, _lib.ptr(rb.out2in), _lib.stream_ptr()),
                   'sst_spconv_invert_map_i32')
    pairs = torch.empty((kvol, 2, n), dtype=torch.int32, device=dev)
    rb.num = torch.zeros(kvol, dtype=torch.int32, device=dev)
    ws = _lib.workspace(lib.sst_spconv_pair_lists_workspace_bytes(kvol, n), dev)
    _lib.check(lib.sst_spconv_pair_lists_i32(_lib.ptr(in2out), kvol, n, _lib.ptr(pairs), _lib.ptr(rb.num),
                                             _lib.ptr(ws), _lib.stream_ptr()), 'sst_spconv_pair_lists_i32')
    rb._pairs_ref = weakref.ref(pairs)
    pairs._sst_rulebook = rb
    rb._total = None if n > 0 else 0
    return rb, pairs


def get_indice_pairs(indices, batch_size, spatial_shape, ksize=3, stride=1, padding=0, dilation=1, out_padding=0,
                     subm=False, transpose=False, grid=None):
    """ops.py:46-102 -> (outids [M, 4] int32, indice_pairs [K, 2, N] int32, indice_pair_num [K] int32)."""
    if indices.dim() != 2 or indices.shape[1] != 4:
        raise NotImplementedError('sst_amd.spconv: 3-D indices (batch, z, y, x) only')
    if not indices.is_cuda:
        raise RuntimeError('sst_amd.spconv: CUDA tensors required (no CPU fallback)')
    ksize, stride, padding = _triple(ksize), _triple(stride), _triple(padding)
    dilation, out_padding = _triple(dilation), _triple(out_padding)
    for d, s in zip(dilation, stride):
        assert any([s == 1, d == 1]), "don't support this."
    spatial_shape = [int(s) for s in spatial_shape]
    if not subm:
        if transpose:
            out_shape = get_deconv_output_size(spatial_shape, ksize, stride, padding, dilation, out_padding)
        else:
            out_shape = get_conv_output_size(spatial_shape, ksize, stride, padding, dilation)
    else:
        out_shape = spatial_shape
    indices = indices.int().contiguous()
    n, dev = indices.size(0), indices.device
    kvol = int(np.prod(ksize))
    lib = _lib.load()
    in2out = torch.empty((kvol, n), dtype=torch.int32, device=dev)
    if n == 0:
        rb, pairs = _finish_rulebook(in2out, kvol, 0, 0, dev)
        return indices, pairs, rb.num
    # dense cell grids (csrc/spconv.hip, second builder) whenever batch x shape fits 2^28 cells (1 GB of int32);
    # SST_SPCONV_RULEBOOK=sort forces the sort-based builder
    n_cells = int(batch_size) * int(np.prod(spatial_shape if subm else out_shape))
    debug = os.environ.get('SST_AMD_DEBUG', '0') == '1'     # debug: the sort-based builder, which can tell a voxel listed twice
    use_grid = (os.environ.get('SST_SPCONV_RULEBOOK', 'grid') == 'grid' and 0 < n_cells <= (1 << 28) and min(out_shape) > 0
                and not debug)
    if use_grid and subm:
        grid = torch.empty(n_cells, dtype=torch.int32, device=dev)
        out2in = torch.empty((kvol, n), dtype=torch.int32, device=dev)
        rc = lib.sst_spconv_grid_subm_i32(_lib.ptr(indices), n, int(batch_size), _i32(spatial_shape), _i32(ksize),
                                          _i32(dilation), _lib.ptr(grid), _lib.ptr(in2out), _lib.ptr(out2in),
                                          _lib.stream_ptr())
        _lib.check(rc, 'sst_spconv_grid_subm_i32')
        rb, pairs = _finish_rulebook(in2out, kvol, n, n, dev, out2in=out2in)
        return indices, pairs, rb.num
    if use_grid:
        geom = (_i32(spatial_shape), _i32(out_shape), _i32(ksize), _i32(stride), _i32(padding), _i32(dilation))
        ws = _lib.workspace(lib.sst_spconv_grid_conv_workspace_bytes(n_cells), dev)
        num_out = torch.empty(1, dtype=torch.int32, device=dev)
        rc = lib.sst_spconv_grid_conv_count_i32(_lib.ptr(indices), n, int(batch_size), *geom, int(bool(transpose)),
                                                _lib.ptr(ws), _lib.ptr(num_out), _lib.stream_ptr())
        _lib.check(rc, 'sst_spconv_grid_conv_count_i32')
        m = int(num_out.item())      # the one read-back an output tensor of data-dependent length needs
        outids = torch.empty((m, 4), dtype=torch.int32, device=dev)
        out2in = torch.empty((kvol, m), dtype=torch.int32, device=dev)
        rc = lib.sst_spconv_grid_conv_maps_i32(_lib.ptr(indices), n, int(batch_size), *geom, int(bool(transpose)),
                                               _lib.ptr(ws), m, _lib.ptr(outids), _lib.ptr(in2out), _lib.ptr(out2in),
                                               _lib.stream_ptr())
        _lib.check(rc, 'sst_spconv_grid_conv_maps_i32')
        rb, pairs = _finish_rulebook(in2out, kvol, n, m, dev, out2in=out2in)
        return outids, pairs, rb.num
    if subm:
        # no host synchronisation on this path: the group count stays on the device.  SST_AMD_DEBUG=1 reads it back and
        # checks that no voxel is listed twice (a SparseConvTensor never does; the reference does not check either)
        plan = K.unique_rows(indices, [0, 0, 0, 0], [int(batch_size)] + out_shape, invalid_if_negative=0,
                             defer_count=not debug)
        if debug and plan.m != n:
            raise RuntimeError('sst_amd.spconv: a SparseConvTensor must not contain a voxel twice')
        rc = lib.sst_spconv_subm_map_i32(_lib.ptr(indices), n, _i32(out_shape), _i32(ksize), _i32(dilation),
                                         _lib.ptr(plan.ukeys), _lib.ptr(plan.perm), _lib.ptr(in2out),
                                         _lib.stream_ptr())
        _lib.check(rc, 'sst_spconv_subm_map_i32')
        rb, pairs = _finish_rulebook(in2out, kvol, n, n, dev)
        return indices, pairs, rb.num
    rows = torch.empty((kvol * n + 1, 4), dtype=torch.int32, device=dev)
    rc = lib.sst_spconv_candidates_i32(_lib.ptr(indices), n, _i32(spatial_shape), _i32(out_shape), _i32(ksize),
                                       _i32(stride), _i32(padding), _i32(dilation), int(bool(transpose)),
                                       _lib.ptr(rows), _lib.stream_ptr())
    _lib.check(rc, 'sst_spconv_candidates_i32')
    plan = K.unique_rows(rows, [0, 0, 0, 0], [int(batch_size)] + out_shape, invalid_if_negative=1)
    m = plan.m - 1  # group 0 collects the invalid candidates (the extra last row guarantees that it exists)
    outids = K.unpack_unique_rows(plan, torch.int32, first=1, count=m)
    _lib.check(lib.sst_spconv_inverse_to_map_i32(_lib.ptr(plan.inverse), kvol * n, _lib.ptr(in2out),
                                                 _lib.stream_ptr()), 'sst_spconv_inverse_to_map_i32')
    rb, pairs = _finish_rulebook(in2out, kvol, n, m, dev)
    return outids, pairs, rb.num


def rulebook_of(indice_pairs, indice_pair_num, num_out):
    """the dense maps behind a pair-list tensor; rebuilt from the lists when they did not come from get_indice_pairs"""
    rb = getattr(indice_pairs, '_sst_rulebook', None)
    if rb is not None:
        return rb
    kvol, _, n = indice_pairs.shape
    dev = indice_pairs.device
    valid = torch.arange(n, device=dev)[None, :] < indice_pair_num[:, None].to(dev)
    koff = torch.arange(kvol, device=dev)[:, None].expand(kvol, n)[valid]
    src, dst = indice_pairs[:, 0][valid].long(), indice_pairs[:, 1][valid]
    in2out = torch.full((kvol, n), -1, dtype=torch.int32, device=dev)
    in2out[koff, src] = dst
    rb, _ = _finish_rulebook(in2out, kvol, n, int(num_out), dev)
    rb._pairs_ref = weakref.ref(indice_pairs)   # the caller's lists (same pairs per offset, possibly another order)
    indice_pairs._sst_rulebook = rb
    return rb


def _conv_kernel_choice():
    """SST_SPCONV_KERNEL=os (default: output-stationary implicit GEMM, csrc/spconv_os.hip) | legacy (the first-generation
    kernels of csrc/spconv.hip, kept for A/B measurements and for shapes the new kernel does not take)"""
    return os.environ.get('SST_SPCONV_KERNEL', 'os')


_OS_ORDER_MIN_ROWS = 16384   # below: fewer tiles than workgroup slots on the chip, the order cannot matter


def _os_tile_order_enabled():
    return os.environ.get('SST_SPCONV_OS_ORDER', '1') != '0'


# the exact split is the default since round 5 (what a model built from a shipped config runs without any call; the FSD / FSDv2
# bench lines have carried it since round 4); set_conv_precision('f32') is the opt-out onto the fp32 matrix pipe
DEFAULT_CONV_PRECISION = 'f32x6'
_CONV_PRECISION = DEFAULT_CONV_PRECISION


def set_conv_precision(mode):
    """How the forward contraction and the data gradient of every sparse convolution multiply (filter gradients stay on the
    fp32 matrix pipe: they are gather-bound):
      'f32'    fp32 matrix pipe (csrc/spconv_os.hip): the opt-out;
      'f32x6'  (default) exact three-way bf16 split of both operands, six products, fp32 accumulation (csrc/spconv_os_x6.hip): the same
               arithmetic class (error vs float64 <= 2 x the fp32 kernel's, tests/test_gpu_spconv.py), 2.7 x less pipe time;
      'f32x3'  two-way split, three products (csrc/spconv_os_x3.hip; ~1e-5 of the output scale per layer): a leg only.
    PROCESS-GLOBAL, like sst_amd.dense.set_matmul_mode."""
    global _CONV_PRECISION
    if mode not in ('f32', 'f32x3', 'f32x6'):
        raise ValueError(mode)
    _CONV_PRECISION = mode


def conv_precision():
    return _CONV_PRECISION


def _gather_gemm(x, mapping, rows, weight3, trans_w, cout, density=None, tile_cfg=0):
    # density: a number, None, or the Rulebook (its `density` costs a host read-back: only the legacy path asks for it)
    """Y[r] = sum_k X[mapping[k][r]] W[k]; weight3 is [K, cin, cout], or [K, cout, cin] with trans_w."""
    lib = _lib.load()
    x = x if x.stride(1) == 1 else x.contiguous()
    if x.size(0) == 0:   # nothing to gather from (e.g. the data gradient of a layer whose output side is empty)
        return torch.zeros((rows, cout), dtype=torch.float32, device=x.device)
    y = torch.empty((rows, cout), dtype=torch.float32, device=x.device)
    if rows == 0:
        return y
    kvol, cin = weight3.size(0), x.size(1)
    weight3 = weight3 if weight3.is_contiguous() else weight3.contiguous()
    if (_conv_kernel_choice() == 'os' and kvol <= 32 and cin % 4 == 0 and x.stride(0) % 4 == 0
            and x.data_ptr() % 16 == 0):
        split = _CONV_PRECISION if (_CONV_PRECISION != 'f32' and int(tile_cfg) == 0) else None
        if split == 'f32x6':     # room for the partial tiles of a thin level too (csrc/spconv_os_x6.hip: offsets dealt out)
            ws = _lib.workspace(lib.sst_spconv_conv_os_f32x6_workspace_bytes_rows(kvol, cin, cout, rows), x.device)
        else:
            ws = _lib.workspace(lib.sst_spconv_conv_os_workspace_bytes(kvol, cin, cout), x.device)
        order = None
        x3 = split is not None
        if isinstance(density, Rulebook) and rows >= _OS_ORDER_MIN_ROWS and _os_tile_order_enabled():
            # the split-precision kernel always works on 64-row tiles (it ignores the SST_SPCONV_OS_TILE override the exact
            # kernel's os_pick honours): its launch order must be computed for ITS tile height
            tile_rows = 64 if x3 else lib.sst_spconv_conv_os_tile_rows(rows, cout, int(tile_cfg))
            order = density.tile_order(mapping, rows, tile_rows)
        if split == 'f32x6':
            rc = lib.sst_spconv_conv_os_rows_f32x6(_lib.ptr(x), x.stride(0), _lib.ptr(mapping), rows, kvol, _lib.ptr(weight3), cin,
                                                   cout, int(trans_w), None, _lib.ptr(y), y.stride(0), 0,
                                                   _lib.ptr(order) if order is not None else None, _lib.ptr(ws), ws.numel(),
                                                   _lib.stream_ptr())
            _lib.check(rc, 'sst_spconv_conv_os_rows_f32x6')
            return y
        entry = getattr(lib, 'sst_spconv_conv_os_' + (split or 'f32'))
        rc = entry(_lib.ptr(x), x.stride(0), _lib.ptr(mapping), rows, kvol, _lib.ptr(weight3), cin, cout, int(trans_w), None,
                   _lib.ptr(y), y.stride(0), int(tile_cfg), _lib.ptr(order) if order is not None else None, _lib.ptr(ws),
                   _lib.stream_ptr())
        _lib.check(rc, 'sst_spconv_conv_os_f32')
        return y
    # first-generation kernels: compacted rows pay off when few offsets are populated or one 64-column group covers the
    # layer; they read W as [cin, cout] only
    if trans_w:
        weight3 = weight3.transpose(1, 2).contiguous()
    if isinstance(density, Rulebook):
        density = density.density
    form = 2 if (cout <= 64 or (density is not None and density < 0.2)) else 1
    rc = lib.sst_spconv_gather_gemm_f32(_lib.ptr(x), x.stride(0), _lib.ptr(mapping), rows, kvol, _lib.ptr(weight3), cin,
                                        cout, 0, None, _lib.ptr(y), y.stride(0), form, _lib.stream_ptr())
    _lib.check(rc, 'sst_spconv_gather_gemm_f32')
    return y


def _wgrad(x, dy, rb, pairs, x_side, shape):
    lib = _lib.load()
    kvol, cin, cout = rb.kvol, x.size(1), dy.size(1)
    dw = torch.empty((kvol, cin, cout), dtype=torch.float32, device=x.device)
    x = x if x.stride(1) == 1 else x.contiguous()
    dy = dy if dy.stride(1) == 1 else dy.contiguous()
    ldx, lddy = (x.stride(0) if x.size(0) else cin), (dy.stride(0) if dy.size(0) else cout)
    total = rb.known_total_pairs()     # -1 while nobody has read the count back: launch and workspace use the upper bound
    if (_conv_kernel_choice() == 'os' and cin % 4 == 0 and cout % 4 == 0 and ldx % 4 == 0 and lddy % 4 == 0
            and x.data_ptr() % 16 == 0 and dy.data_ptr() % 16 == 0):
        ws = _lib.workspace(lib.sst_spconv_wgrad_os_workspace_bytes(kvol, rb.n, total, cin, cout), x.device)
        # the filter gradient multiplies the way the convolution does: exact three-way bf16 split ('f32x6', the default) or the
        # fp32 matrix pipe ('f32'; 'f32x3' has no filter-gradient kernel of its own and takes the fp32 one)
        name = ('sst_spconv_wgrad_os_f32x6' if (_CONV_PRECISION == 'f32x6' and os.environ.get('SST_SPCONV_WGRAD_X6', '1') != '0')
                else 'sst_spconv_wgrad_os_f32')     # SST_SPCONV_WGRAD_X6=0: A/B against the fp32-pipe kernel
        rc = getattr(lib, name)(_lib.ptr(x), ldx, _lib.ptr(dy), lddy, _lib.ptr(pairs), rb.n, total, x_side,
                                _lib.ptr(rb.num), kvol, cin, cout, _lib.ptr(dw), _lib.ptr(ws), _lib.stream_ptr())
        _lib.check(rc, name)
        return dw.view(shape)
    ws = _lib.workspace(lib.sst_spconv_wgrad_workspace_bytes(kvol, rb.n, total, cin, cout), x.device)
    rc = lib.sst_spconv_wgrad_f32(_lib.ptr(x), ldx, _lib.ptr(dy), lddy, _lib.ptr(pairs), rb.n, total, x_side,
                                  _lib.ptr(rb.num), kvol, cin, cout, _lib.ptr(dw), _lib.ptr(ws), _lib.stream_ptr())
    _lib.check(rc, 'sst_spconv_wgrad_f32')
    return dw.view(shape)


def _check(features, filters):
    if not features.is_cuda:
        raise RuntimeError('sst_amd.spconv: CUDA tensors required (no CPU fallback)')
    if features.dtype != torch.float32 or filters.dtype != torch.float32:
        raise NotImplementedError('sst_amd.spconv: fp32 only')


def indice_conv(features, filters, indice_pairs, indice_pair_num, num_activate_out, inverse=False, subm=False):
    """ops.py:105-124: out[pairs[k][1 - inverse]] += features[pairs[k][inverse]] @ filters[k]"""
    _check(features, filters)
    rb = rulebook_of(indice_pairs, indice_pair_num, features.size(0) if inverse else num_activate_out)
    w3 = filters.reshape(-1, filters.shape[-2], filters.shape[-1])
    if inverse:
        return _gather_gemm(features, rb.in2out, rb.n, w3, False, w3.size(2), rb)
    return _gather_gemm(features, rb.out2in, rb.m, w3, False, w3.size(2), rb)


def indice_conv_backward(features, filters, out_bp, indice_pairs, indice_pair_num, inverse=False, subm=False):
    """ops.py:139-156 -> (input gradient, filter gradient)"""
    _check(features, filters)
    rb = rulebook_of(indice_pairs, indice_pair_num, features.size(0) if inverse else out_bp.size(0))
    w3 = filters.reshape(-1, filters.shape[-2], filters.shape[-1])
    out_bp = out_bp.contiguous()
    # the data gradient is the same contraction with W[k]^T: the forward weights are read with the roles swapped
    if inverse:
        input_bp = _gather_gemm(out_bp, rb.out2in, rb.m, w3, True, w3.size(1), rb)
        filters_bp = _wgrad(features, out_bp, rb, indice_pairs, 1, filters.shape)
    else:
        input_bp = _gather_gemm(out_bp, rb.in2out, rb.n, w3, True, w3.size(1), rb)
        filters_bp = _wgrad(features, out_bp, rb, indice_pairs, 0, filters.shape)
    return input_bp, filters_bp


def indice_maxpool(features, indice_pairs, indice_pair_num, num_activate_out):
    """ops.py:162-171 / pool_ops.h:24-58: out = max(0, max over the paired inputs)"""
    if not features.is_cuda:
        raise RuntimeError('sst_amd.spconv: CUDA tensors required (no CPU fallback)')
    if features.dtype != torch.float32:
        raise NotImplementedError('sst_amd.spconv: fp32 only')
    rb = rulebook_of(indice_pairs, indice_pair_num, num_activate_out)
    x = features if features.stride(1) == 1 else features.contiguous()
    y = torch.empty((rb.m, x.size(1)), dtype=torch.float32, device=x.device)
    rc = _lib.load().sst_spconv_maxpool_fwd_f32(_lib.ptr(x), x.stride(0) if x.size(0) else x.size(1), _lib.ptr(rb.out2in),
                                                rb.m, rb.kvol, x.size(1), _lib.ptr(y), y.stride(0), _lib.stream_ptr())
    _lib.check(rc, 'sst_spconv_maxpool_fwd_f32')
    return y


def indice_maxpool_backward(features, out_features, out_bp, indice_pairs, indice_pair_num):
    """ops.py:174-183 / pool_ops.h:60-96"""
    rb = rulebook_of(indice_pairs, indice_pair_num, out_features.size(0))
    x = features if features.stride(1) == 1 else features.contiguous()
    y = out_features if out_features.stride(1) == 1 else out_features.contiguous()
    dy = out_bp if out_bp.stride(1) == 1 else out_bp.contiguous()
    if y.size(0) == 0 or x.size(0) == 0:   # empty output side: no input voxel has a partner
        return torch.zeros_like(x)
    dx = torch.empty_like(x)
    c = x.size(1)
    rc = _lib.load().sst_spconv_maxpool_bwd_f32(_lib.ptr(x), x.stride(0) if x.size(0) else c, _lib.ptr(y),
                                                y.stride(0) if y.size(0) else c, _lib.ptr(dy),
                                                dy.stride(0) if dy.size(0) else c, _lib.ptr(rb.in2out), rb.n, rb.kvol, c,
                                                _lib.ptr(dx), dx.stride(0) if dx.size(0) else c, _lib.stream_ptr())
    _lib.check(rc, 'sst_spconv_maxpool_bwd_f32')
    return dx


# ------------------------------------------------------------------------------------------------ functional.py
class SparseMaxPoolFunction(torch.autograd.Function):
    """functional.py:78-96"""

    @staticmethod
    def forward(ctx, features, indice_pairs, indice_pair_num, num_activate_out):
        out = indice_maxpool(features, indice_pairs, indice_pair_num, num_activate_out)
        ctx.save_for_backward(indice_pairs, indice_pair_num, features, out)
        ctx.rulebook = getattr(indice_pairs, '_sst_rulebook', None)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        indice_pairs, indice_pair_num, features, out = ctx.saved_tensors
        if ctx.rulebook is not None:
            indice_pairs._sst_rulebook = ctx.rulebook
        return indice_maxpool_backward(features, out, grad_output, indice_pairs, indice_pair_num), None, None, None


indice_maxpool_fn = SparseMaxPoolFunction.apply


class SparseConvFunction(torch.autograd.Function):

    @staticmethod
    def forward(ctx, features, filters, indice_pairs, indice_pair_num, num_activate_out):
        ctx.save_for_backward(indice_pairs, indice_pair_num, features, filters)
        ctx.rulebook = getattr(indice_pairs, '_sst_rulebook', None)
        return indice_conv(features, filters, indice_pairs, indice_pair_num, num_activate_out, False)

    @staticmethod
    def backward(ctx, grad_output):
        indice_pairs, indice_pair_num, features, filters = ctx.saved_tensors
        if ctx.rulebook is not None:
            indice_pairs._sst_rulebook = ctx.rulebook
        input_bp, filters_bp = indice_conv_backward(features, filters, grad_output, indice_pairs, indice_pair_num, False)
        return input_bp, filters_bp, None, None, None


class SparseInverseConvFunction(torch.autograd.Function):

    @staticmethod
    def forward(ctx, features, filters, indice_pairs, indice_pair_num, num_activate_out):
        ctx.save_for_backward(indice_pairs, indice_pair_num, features, filters)
        ctx.rulebook = getattr(indice_pairs, '_sst_rulebook', None)
        return indice_conv(features, filters, indice_pairs, indice_pair_num, num_activate_out, True, False)

    @staticmethod
    def backward(ctx, grad_output):
        indice_pairs, indice_pair_num, features, filters = ctx.saved_tensors
        if ctx.rulebook is not None:
            indice_pairs._sst_rulebook = ctx.rulebook
        input_bp, filters_bp = indice_conv_backward(features, filters, grad_output, indice_pairs, indice_pair_num, True,
                                                    False)
        return input_bp, filters_bp, None, None, None


class SubMConvFunction(torch.autograd.Function):

    @staticmethod
    def forward(ctx, features, filters, indice_pairs, indice_pair_num, num_activate_out):
        ctx.save_for_backward(indice_pairs, indice_pair_num, features, filters)
        ctx.rulebook = getattr(indice_pairs, '_sst_rulebook', None)
        return indice_conv(features, filters, indice_pairs, indice_pair_num, num_activate_out, False, True)

    @staticmethod
    def backward(ctx, grad_output):
        indice_pairs, indice_pair_num, features, filters = ctx.saved_tensors
        if ctx.rulebook is not None:
            indice_pairs._sst_rulebook = ctx.rulebook
        input_bp, filters_bp = indice_conv_backward(features, filters, grad_output, indice_pairs, indice_pair_num, False,
                                                    True)
        return input_bp, filters_bp, None, None, None


indice_conv_fn = SparseConvFunction.apply
indice_inverse_conv = SparseInverseConvFunction.apply
indice_subm_conv = SubMConvFunction.apply


# ------------------------------------------------------------------------------------------------ modules.py
class SparseModule(nn.Module):
    """marker: modules deriving from it receive the SparseConvTensor itself inside a SparseSequential"""
    pass


def is_spconv_module(module):
    return isinstance(module, SparseModule)


def is_sparse_conv(module):
    return isinstance(module, SparseConvolution)


class SparseSequential(SparseModule):
    """Ordered container (modules.py:42-140): sparse modules receive the SparseConvTensor, everything else its features.
    Accepts positional modules, one dict of modules, and keyword modules; children are named '0', '1', ... or by their keys
    (these names are the ``state_dict`` prefixes of the reference's checkpoints)."""

    def __init__(self, *modules, **named):
        super().__init__()
        if len(modules) == 1 and isinstance(modules[0], dict):
            entries = list(modules[0].items())
        else:
            entries = [(str(position), module) for position, module in enumerate(modules)]
        for key, module in entries + list(named.items()):
            if key in self._modules:
                raise ValueError('name exists.')
            self.add_module(key, module)
        self._sparity_dict = {}

    def __len__(self):
        return len(self._modules)

    def __getitem__(self, idx):
        children = list(self._modules.values())
        if not -len(children) <= idx < len(children):
            raise IndexError('index {} is out of range'.format(idx))
        return children[idx]

    @property
    def sparity_dict(self):
        return self._sparity_dict

    def add(self, module, name=None):
        key = str(len(self._modules)) if name is None else name
        if name is None and key in self._modules:
            raise KeyError('name exists')
        self.add_module(key, module)

    def forward(self, input):
        from .norm import batch_norm_act
        mods = list(self._modules.items())
        i = 0
        while i < len(mods):
            k, module = mods[i]
            i += 1
            if is_spconv_module(module):
                assert isinstance(input, SparseConvTensor)
                self._sparity_dict[k] = input.sparity
                input = module(input)
            else:
                # BatchNorm1d directly followed by ReLU on the features: one fused kernel pair (csrc/bn.hip) instead
                # of norm + a separate ReLU pass -- the `conv -> norm -> act` order of make_sparse_convmodule
                fuse = (isinstance(module, nn.BatchNorm1d) and i < len(mods) and type(mods[i][1]) is nn.ReLU)
                if isinstance(input, SparseConvTensor):
                    if input.indices.shape[0] != 0:
                        input.features = batch_norm_act(module, input.features, relu=True) if fuse \
                            else module(input.features)
                        if fuse:
                            i += 1
                else:
                    input = module(input)
        return input


class ToDense(SparseModule):

    def forward(self, x):
        return x.dense()


class RemoveGrid(SparseModule):

    def forward(self, x):
        x.grid = None
        return x


# ------------------------------------------------------------------------------------------------ conv.py
def _calculate_fan_in_and_fan_out_hwio(tensor):
    """conv.py:28-46: the weight is [*kernel, Cin, Cout]"""
    dimensions = tensor.ndimension()
    if dimensions < 2:
        raise ValueError('fan in and fan out can not be computed for tensor with fewer than 2 dimensions')
    if dimensions == 2:
        return tensor.size(-2), tensor.size(-1)
    receptive = 1
    if tensor.dim() > 2:
        receptive = tensor[..., 0, 0].numel()
    return tensor.size(-2) * receptive, tensor.size(-1) * receptive


class SparseConvolution(SparseModule):
    """conv.py:49-230.  fused_bn is accepted and ignored (the reference marks it "don't use")."""

    def __init__(self, ndim, in_channels, out_channels, kernel_size=3, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, subm=False, output_padding=0, transposed=False, inverse=False, indice_key=None,
                 fused_bn=False):
        super().__init__()
        assert groups == 1
        if ndim != 3:
            raise NotImplementedError('sst_amd.spconv: 3-D convolutions only')
        kernel_size, stride, padding = _triple(kernel_size), _triple(stride), _triple(padding)
        dilation, output_padding = _triple(dilation), _triple(output_padding)
        for d, s in zip(dilation, stride):
            assert any([s == 1, d == 1]), "don't support this."
        self.ndim = ndim
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel_size = kernel_size
        self.conv1x1 = np.prod(kernel_size) == 1
        self.stride = stride
        self.padding = padding
        self.dilation = dilation
        self.transposed = transposed
        self.inverse = inverse
        self.output_padding = output_padding
        self.groups = groups
        self.subm = subm
        self.indice_key = indice_key
        self.fused_bn = fused_bn
        self.weight = Parameter(torch.Tensor(*kernel_size, in_channels, out_channels))
        if bias:
            self.bias = Parameter(torch.Tensor(out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in, _ = _calculate_fan_in_and_fan_out_hwio(self.weight)
            bound = 1 / math.sqrt(fan_in)
            init.uniform_(self.bias, -bound, bound)

    def _geometry(self, input):
        """-> (outids, indice_pairs, indice_pair_num, out_spatial_shape): the index half of forward() - output voxels and the
        rulebook, found under ``indice_key`` in the tensor's ``indice_dict`` or built (and filed there) - features untouched"""
        indices = input.indices
        spatial_shape = input.spatial_shape
        batch_size = input.batch_size
        if not self.subm:
            if self.transposed:
                out_spatial_shape = get_deconv_output_size(spatial_shape, self.kernel_size, self.stride, self.padding,
                                                           self.dilation, self.output_padding)
            else:
                out_spatial_shape = get_conv_output_size(spatial_shape, self.kernel_size, self.stride, self.padding,
                                                         self.dilation)
        else:
            out_spatial_shape = spatial_shape
        datas = input.find_indice_pair(self.indice_key)
        if self.inverse:
            assert datas is not None and self.indice_key is not None
            _, outids, indice_pairs, indice_pair_num, out_spatial_shape = datas
            assert indice_pairs.shape[0] == np.prod(self.kernel_size), \
                'inverse conv must have same kernel size as its couple conv'
        else:
            if self.indice_key is not None and datas is not None:
                outids, _, indice_pairs, indice_pair_num, _ = datas
            else:
                outids, indice_pairs, indice_pair_num = get_indice_pairs(
                    indices, batch_size, spatial_shape, self.kernel_size, self.stride, self.padding, self.dilation,
                    self.output_padding, self.subm, self.transposed, grid=input.grid)
                input.indice_dict[self.indice_key] = (outids, indices, indice_pairs, indice_pair_num, spatial_shape)
        return outids, indice_pairs, indice_pair_num, out_spatial_shape

    def dry(self, input):
        """The layer on the voxel SET of ``input`` only: its output voxels, with the rulebook filed under ``indice_key`` - for
        building the rulebooks of a network ahead of its forward pass (they depend on the coordinates alone: sparse_unet.
        _UNetStages.build_rulebooks); features are neither read nor produced."""
        if self.conv1x1:
            return input
        outids, _, _, out_spatial_shape = self._geometry(input)
        out = SparseConvTensor(None, outids, out_spatial_shape, input.batch_size)
        out.indice_dict = input.indice_dict
        out.grid = input.grid
        return out

    def forward(self, input):
        assert isinstance(input, SparseConvTensor)
        features = input.features
        batch_size = input.batch_size
        if self.conv1x1:
            features = torch.mm(input.features, self.weight.view(self.in_channels, self.out_channels))
            if self.bias is not None:
                features += self.bias
            out_tensor = SparseConvTensor(features, input.indices, input.spatial_shape, input.batch_size)
            out_tensor.indice_dict = input.indice_dict
            out_tensor.grid = input.grid
            return out_tensor
        outids, indice_pairs, indice_pair_num, out_spatial_shape = self._geometry(input)
        if self.subm:
            out_features = indice_subm_conv(features, self.weight, indice_pairs, indice_pair_num, outids.shape[0])
        elif self.inverse:
            out_features = indice_inverse_conv(features, self.weight, indice_pairs, indice_pair_num, outids.shape[0])
        else:
            out_features = indice_conv_fn(features, self.weight, indice_pairs, indice_pair_num, outids.shape[0])
        if self.bias is not None:
            out_features += self.bias
        out_tensor = SparseConvTensor(out_features, outids, out_spatial_shape, batch_size)
        out_tensor.indice_dict = input.indice_dict
        out_tensor.grid = input.grid
        return out_tensor

@CONV_LAYERS.register_module()
class SparseConv3d(SparseConvolution):

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias,
                         indice_key=indice_key)


@CONV_LAYERS.register_module()
class SparseConvTranspose3d(SparseConvolution):

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias,
                         transposed=True, indice_key=indice_key)


@CONV_LAYERS.register_module()
class SparseInverseConv3d(SparseConvolution):

    def __init__(self, in_channels, out_channels, kernel_size, indice_key, bias=True):
        super().__init__(3, in_channels, out_channels, kernel_size, bias=bias, inverse=True, indice_key=indice_key)


@CONV_LAYERS.register_module()
class SubMConv3d(SparseConvolution):

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias, True,
                         indice_key=indice_key)


# ------------------------------------------------------------------------------------------------ pool.py
class SparseMaxPool(SparseModule):
    """pool.py:20-72"""

    def __init__(self, ndim, kernel_size, stride=1, padding=0, dilation=1, subm=False):
        super().__init__()
        if ndim != 3:
            raise NotImplementedError('sst_amd.spconv: 3-D pooling only')
        self.ndim = ndim
        self.kernel_size = _triple(kernel_size)
        self.stride = _triple(stride)
        self.padding = _triple(padding)
        self.subm = subm
        self.dilation = _triple(dilation)

    def forward(self, input):
        assert isinstance(input, SparseConvTensor)
        if not self.subm:
            out_spatial_shape = get_conv_output_size(input.spatial_shape, self.kernel_size, self.stride, self.padding,
                                                     self.dilation)
        else:
            out_spatial_shape = input.spatial_shape
        outids, indice_pairs, indice_pairs_num = get_indice_pairs(input.indices, input.batch_size, input.spatial_shape,
                                                                  self.kernel_size, self.stride, self.padding,
                                                                  self.dilation, 0, self.subm)
        out_features = indice_maxpool_fn(input.features, indice_pairs, indice_pairs_num, outids.shape[0])
        out_tensor = SparseConvTensor(out_features, outids, out_spatial_shape, input.batch_size)
        out_tensor.indice_dict = input.indice_dict
        out_tensor.grid = input.grid
        return out_tensor


class SparseMaxPool3d(SparseMaxPool):

    def __init__(self, kernel_size, stride=1, padding=0, dilation=1):
        super().__init__(3, kernel_size, stride, padding, dilation)
