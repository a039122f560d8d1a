"""Dense / row-wise pieces of an encoder layer arranged for MI355X.

* ``tall_linear``: y = x W^T + b for a tall x [M, in] (M ~ 1e5 tokens).  Forward and the data gradient are plain
  library GEMMs (hipBLASLt).  The weight gradient dW = dY^T X has a tiny output (<= 384 x 256) and a reduction
  of length M: the library's single-pass kernel uses ~48 workgroups of the 256 CUs (273 us per call, 16.7 ms
  per training step in profiles/r01/a_first_path_kernel_stats.csv).  Here it is a split-K batched GEMM
  (S chunks -> [S, out, in] partials -> sum) that fills the chip; the bias gradient is a column-sum kernel.
* ``add_layer_norm``: fused residual-add + LayerNorm (HIP kernels in csrc/dense.hip) replacing
  ``norm(src + src2)`` (mmdet3d/models/sst/sst_basic_block_v2.py:113-118).
"""
import torch
from torch.autograd import Function

from . import _lib


def colsum(x):
    """sum over rows of a 2-D fp32 tensor (row-strided ok)."""
    m, c = x.shape
    if x.dtype != torch.float32 or (c % 4) != 0 or c > 1024 or x.stride(1) != 1 or not x.is_cuda:
        return x.sum(0)
    out = torch.empty(c, dtype=torch.float32, device=x.device)
    lib = _lib.load()
    ws = _lib.workspace(lib.sst_colsum_workspace_bytes(m, c), x.device)
    rc = lib.sst_colsum_f32(_lib.ptr(x), m, c, x.stride(0), _lib.ptr(out), _lib.ptr(ws), _lib.stream_ptr())
    _lib.check(rc, 'sst_colsum_f32')
    return out


def weight_grad_splitk(dy, x, chunk=1024):
    """dW[out, in] = dy[M, out]^T x[M, in] as a batched split-K GEMM."""
    m = dy.size(0)
    s = m // chunk
    if s < 4:
        return dy.t() @ x
    body = s * chunk
    dw = torch.bmm(dy[:body].view(s, chunk, -1).transpose(1, 2), x[:body].view(s, chunk, -1)).sum(0)
    if body < m:
        dw = dw + dy[body:].t() @ x[body:]
    return dw


class TallLinear(Function):

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        if bias is not None:
            return torch.addmm(bias, x, weight.t())
        return x @ weight.t()

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = dy @ weight
        if ctx.needs_input_grad[1]:
            dw = weight_grad_splitk(dy, x.contiguous())
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = colsum(dy)
        return dx, dw, db


def tall_linear(x, weight, bias=None):
    if x.dim() != 2 or x.dtype != torch.float32 or not x.is_cuda:
        return torch.nn.functional.linear(x, weight, bias)
    return TallLinear.apply(x, weight, bias)


class AddLayerNorm(Function):
    """y = LayerNorm(x + res); the gradient w.r.t. x and res is the same tensor."""

    @staticmethod
    def forward(ctx, x, res, weight, bias, eps):
        x = x.contiguous()
        m, c = x.shape
        if res is not None:
            res = res.contiguous()
        y = torch.empty_like(x)
        need_sum = res is not None
        s = torch.empty_like(x) if need_sum else x
        stats = torch.empty((m, 2), dtype=torch.float32, device=x.device)
        rc = _lib.load().sst_add_layernorm_fwd_f32(_lib.ptr(x), _lib.ptr(res), _lib.ptr(weight), _lib.ptr(bias), m, c,
                                                   float(eps), _lib.ptr(y), _lib.ptr(s) if need_sum else None,
                                                   _lib.ptr(stats), _lib.stream_ptr())
        _lib.check(rc, 'sst_add_layernorm_fwd_f32')
        ctx.save_for_backward(s, stats, weight)
        ctx.has_res = res is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        s, stats, weight = ctx.saved_tensors
        dy = dy.contiguous()
        m, c = s.shape
        dx = torch.empty_like(s)
        dw = torch.empty(c, dtype=torch.float32, device=s.device)
        db = torch.empty(c, dtype=torch.float32, device=s.device)
        lib = _lib.load()
        ws = _lib.workspace(lib.sst_add_layernorm_bwd_workspace_bytes(m, c), s.device)
        rc = lib.sst_add_layernorm_bwd_f32(_lib.ptr(dy), _lib.ptr(s), _lib.ptr(stats), _lib.ptr(weight), m, c,
                                           _lib.ptr(dx), _lib.ptr(dw), _lib.ptr(db), _lib.ptr(ws), _lib.stream_ptr())
        _lib.check(rc, 'sst_add_layernorm_bwd_f32')
        return dx, (dx if ctx.has_res else None), dw, db, None


def add_layer_norm(x, res, norm):
    """norm(x + res) for an nn.LayerNorm ``norm`` (res may be None); falls back to torch for shapes the kernel
    is not built for (never for the SST configs: C = 128 / 192)."""
    c = x.size(-1)
    ok = (isinstance(norm, torch.nn.LayerNorm) and x.dim() == 2 and x.dtype == torch.float32 and x.is_cuda
          and norm.elementwise_affine and norm.bias is not None and c % 4 == 0 and c <= 512)
    if not ok:
        return norm(x + res if res is not None else x)
    return AddLayerNorm.apply(x, res, norm.weight, norm.bias, norm.eps)
