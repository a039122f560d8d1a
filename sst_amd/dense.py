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


def weight_bias_grad(dy, x, want_bias, out_w=None, out_b=None):
    """(dW, db) of y = x W^T + b for tall dy [M, out], x [M, in] through the split-K MFMA kernel
    (csrc/wgrad.hip); shapes it is not built for go through the batched library GEMM.
    out_w / out_b: optional contiguous destinations (e.g. row slices of a packed in_proj gradient)."""
    m, out = dy.shape
    inn = x.size(1)
    ok = (dy.stride(1) == 1 and x.stride(1) == 1 and m >= 4096 and out <= 4096 and inn <= 4096
          and dy.dtype == torch.float32 and x.dtype == torch.float32 and dy.is_cuda)
    if not ok:
        dw = weight_grad_splitk(dy, x.contiguous())
        db = colsum(dy) if want_bias else None
        if out_w is not None:
            out_w.copy_(dw)
            dw = out_w
        if out_b is not None and db is not None:
            out_b.copy_(db)
            db = out_b
        return dw, db
    lib = _lib.load()
    dw = out_w if out_w is not None else torch.empty((out, inn), dtype=torch.float32, device=dy.device)
    db = None
    if want_bias:
        db = out_b if out_b is not None else torch.empty(out, dtype=torch.float32, device=dy.device)
    assert dw.is_contiguous() and (db is None or db.is_contiguous())
    ws = _lib.workspace(lib.sst_weight_grad_workspace_bytes(m, out, inn), dy.device)
    rc = lib.sst_weight_grad_f32(_lib.ptr(dy), _lib.ptr(x), m, out, inn, dy.stride(0), x.stride(0), _lib.ptr(dw),
                                 _lib.ptr(db), _lib.ptr(ws), _lib.stream_ptr())
    _lib.check(rc, 'sst_weight_grad_f32')
    return dw, db


def weight_bias_grad_group(problems):
    """Weight / bias gradients of several linears at once: problems = [(dy, x, out_w, out_b | None[, (rows, index)]), ...] with
    contiguous fp32 destinations; the split-K kernels of csrc/wgrad.hip one after the other and ONE reduction launch for all of
    them (sst_weight_grad_group_f32).  The optional fifth element: the X operand is x + rows[index] (the positional rows of an
    encoder layer), added on load - exact-split group only; elsewhere the sum is formed first.  Problems the kernel is not built
    for go through weight_bias_grad one by one."""
    problems = [tuple(p) + (None,) * (5 - len(p)) for p in problems]
    ok = all(dy.stride(1) == 1 and x.stride(1) == 1 and dy.size(0) >= 4096 and dy.size(1) <= 4096 and x.size(1) <= 4096
             and dy.dtype == torch.float32 and x.dtype == torch.float32 and dy.is_cuda and ow.is_contiguous()
             and (ob is None or ob.is_contiguous()) for dy, x, ow, ob, _ in problems)
    lib = _lib.load()
    if ok and len(problems) <= 8 and _MATMUL_MODE == 'f32x6':
        arr = (_lib.WgradProblemF32 * len(problems))()
        for q, (dy, x, ow, ob, xadd) in zip(arr, problems):
            q.dy, q.x, q.m, q.ld_dy, q.ld_x = dy.data_ptr(), x.data_ptr(), dy.size(0), dy.stride(0), x.stride(0)
            q.dw, q.db, q.out, q.inn = ow.data_ptr(), (ob.data_ptr() if ob is not None else None), dy.size(1), x.size(1)
            if xadd is not None:
                q.x_add_rows, q.x_add_index = xadd[0].data_ptr(), xadd[1].data_ptr()
        # exact three-way bf16 split on the bf16 matrix pipe (csrc/wgrad_x6.hip), one launch for the group
        need = lib.sst_weight_grad_group_f32x6_workspace_bytes(arr, len(problems))
        if need >= 0:
            ws = _lib.workspace(need, problems[0][0].device)
            _lib.check(lib.sst_weight_grad_group_f32x6(arr, len(problems), _lib.ptr(ws), _lib.stream_ptr()),
                       'sst_weight_grad_group_f32x6')
            return
    # every other path takes X as a tensor: form x + rows[index] where it was asked for
    problems = [(dy, x if xadd is None else x + xadd[0].index_select(0, xadd[1].long()), ow, ob) for dy, x, ow, ob, xadd in problems]
    if not ok or len(problems) > 8:
        for dy, x, ow, ob in problems:
            weight_bias_grad(dy, x, ob is not None, out_w=ow, out_b=ob)
        return
    arr = (_lib.WgradProblemF32 * len(problems))()
    for q, (dy, x, ow, ob) in zip(arr, problems):
        q.dy, q.x, q.m, q.ld_dy, q.ld_x = dy.data_ptr(), x.data_ptr(), dy.size(0), dy.stride(0), x.stride(0)
        q.dw, q.db, q.out, q.inn = ow.data_ptr(), (ob.data_ptr() if ob is not None else None), dy.size(1), x.size(1)
    if _MATMUL_MODE == 'f32x6':
        need = lib.sst_weight_grad_group_f32x6_workspace_bytes(arr, len(problems))
        if need >= 0:
            ws = _lib.workspace(need, problems[0][0].device)
            _lib.check(lib.sst_weight_grad_group_f32x6(arr, len(problems), _lib.ptr(ws), _lib.stream_ptr()),
                       'sst_weight_grad_group_f32x6')
            return
    ws = _lib.workspace(lib.sst_weight_grad_group_workspace_bytes(arr, len(problems)), problems[0][0].device)
    _lib.check(lib.sst_weight_grad_group_f32(arr, len(problems), _lib.ptr(ws), _lib.stream_ptr()), 'sst_weight_grad_group_f32')


def tall_gemm(x, w, bias=None, trans_w=False, out=None, accumulate=False):
    """out (+)= x @ (w if trans_w else w.t()) + bias through csrc/tall_gemm.hip (W resident in LDS, fp32 MFMA);
    returns None when the shape is not one the kernel is built for (the caller then uses the library GEMM)."""
    m, k = x.shape
    n = w.size(1) if trans_w else w.size(0)
    kw = w.size(0) if trans_w else w.size(1)
    if (n != 128 or k not in (128, 256) or kw != k or x.dtype != torch.float32 or not x.is_cuda
            or x.stride(1) != 1 or w.stride(1) != 1 or x.stride(0) % 4 or w.stride(0) % 4
            or x.data_ptr() % 16 or w.data_ptr() % 16):
        return None
    if out is None:
        assert not accumulate
        out = torch.empty((m, n), dtype=torch.float32, device=x.device)
    assert out.stride(1) == 1 and out.size(0) == m and out.size(1) == n
    rc = _lib.load().sst_tall_linear_f32(_lib.ptr(x), x.stride(0), _lib.ptr(w), w.stride(0), _lib.ptr(bias), m, n, k,
                                         int(trans_w), int(accumulate), _lib.ptr(out), out.stride(0), _lib.stream_ptr())
    _lib.check(rc, 'sst_tall_linear_f32')
    return out


EPI_BIAS, EPI_GELU, EPI_RELU, EPI_MUL_GELU_GRAD, EPI_MUL_RELU_GRAD, EPI_ADD = range(6)
_LDS_LINEAR_SHAPES = ((128, 128), (128, 256), (256, 128))
# How the LDS-resident linears multiply:
#   'f32'   exact fp32 on the fp32 matrix pipe (v_mfma_f32_16x16x4_f32, csrc/dense_f32.hip);
#   'f32x6' exact three-way bf16 split of both operands, the six products x_i w_j with i + j <= 2 on the bf16 matrix pipe with
#           fp32 accumulation (csrc/dense_f32x6.hip): what is dropped is below the rounding of an fp32 FMA chain - same
#           arithmetic class, 2.7 x less matrix-pipe time; admissible as the headline only while tests/test_gpu_dense_f32x6.py
#           holds (error against float64 <= 2 x the fp32 kernel's, every shape and the 12-layer stack);
#   'f32x3' two-way split, three products (csrc/dense_f32x3.hip; ~1e-5 relative, tighter than the TF32 products of the
#           reference's own torch 1.8 on Ampere): a leg beside the headline, never the headline.
# The DEFAULT is 'f32x6' (round 5): it is admissible as exact fp32 (tests/test_gpu_dense_f32x6.py) and it is what a model
# built from a shipped config gets without any call; 'f32' (SSTv2.set_precision('fp32') / set_matmul_mode('f32')) is the opt-out.
# SSTv2.set_precision switches the mode of THAT module only: it keeps its own mode and runs its stack inside
# matmul_mode_scope(); the global below is the default for code outside any scope (set_matmul_mode changes it).
DEFAULT_MATMUL_MODE = 'f32x6'
_MATMUL_MODE = DEFAULT_MATMUL_MODE
_MATMUL_MODES = ('f32', 'f32x3', 'f32x6')


def set_matmul_mode(mode):
    global _MATMUL_MODE
    if mode not in _MATMUL_MODES:
        raise ValueError(mode)
    _MATMUL_MODE = mode


def matmul_mode():
    return _MATMUL_MODE


import contextlib  # noqa: E402


@contextlib.contextmanager
def matmul_mode_scope(mode):
    """The mode for the duration of a block (None: leave it alone).  A backbone runs its encoder stack inside the scope of ITS
    OWN mode and every fused layer node re-enters the mode of its forward pass for its backward pass, so two models of one
    process (an EMA copy, a two-stage detector, a test beside a benchmark) no longer follow each other's set_precision()
    (ADVICE round 3); set_matmul_mode() stays the process-wide default for code outside any scope."""
    global _MATMUL_MODE
    if mode is None:
        yield
        return
    if mode not in _MATMUL_MODES:
        raise ValueError(mode)
    prev = _MATMUL_MODE
    _MATMUL_MODE = mode
    try:
        yield
    finally:
        _MATMUL_MODE = prev


def lds_linear_ok(x, w, trans_w=False):
    """shapes / layouts csrc/dense_f32.hip is built for"""
    k, n = (w.size(0), w.size(1)) if trans_w else (w.size(1), w.size(0))
    return (x.is_cuda and x.dtype == torch.float32 and w.dtype == torch.float32 and x.dim() == 2 and x.size(1) == k
            and (k, n) in _LDS_LINEAR_SHAPES and x.stride(1) == 1 and w.stride(1) == 1 and x.stride(0) % 4 == 0
            and w.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0 and w.data_ptr() % 16 == 0)


def lds_linear(x, w, bias=None, epilogue=EPI_BIAS, trans_w=False, aux_in=None, want_pre=False, out=None):
    """y = epilogue(x @ (w if trans_w else w.t()) + bias) in exact fp32 with the weight matrix resident in LDS
    (csrc/dense_f32.hip).  GELU / RELU with want_pre -> (y, pre-activation); MUL_*_GRAD / ADD read ``aux_in`` ([M, N]);
    ``out``: written in place (with EPI_ADD and aux_in = out: out += product)."""
    m = x.size(0)
    k, n = (w.size(0), w.size(1)) if trans_w else (w.size(1), w.size(0))
    y = out if out is not None else torch.empty((m, n), dtype=torch.float32, device=x.device)
    pre = torch.empty((m, n), dtype=torch.float32, device=x.device) if want_pre else None
    aux = aux_in if aux_in is not None else pre
    if aux is not None and (aux.stride(1) != 1 or aux.data_ptr() % 16 or aux.stride(0) % 4):
        raise RuntimeError('sst_amd.dense.lds_linear: aux tensor must be row-major and 16-byte aligned')
    lib = _lib.load()
    entry = getattr(lib, 'sst_tall_linear_epi_' + _MATMUL_MODE)
    rc = entry(_lib.ptr(x), x.stride(0), _lib.ptr(w), w.stride(0), int(trans_w), _lib.ptr(bias), m, k, n, int(epilogue),
               _lib.ptr(aux_in), _lib.ptr(pre), aux.stride(0) if aux is not None else 0, _lib.ptr(y), y.stride(0),
               _lib.stream_ptr())
    _lib.check(rc, 'sst_tall_linear_epi_' + _MATMUL_MODE)
    return (y, pre) if want_pre else y


def lds_linear_dqkv_ok(dqkv, w_in):
    """d(x) of the whole in-projection as one product over K = 384 (f32x6 mode only: csrc/dense_f32x6.hip)"""
    return (_MATMUL_MODE == 'f32x6' and w_in.shape == (384, 128) and w_in.is_contiguous() and dqkv.is_cuda
            and dqkv.dtype == torch.float32 and dqkv.dim() == 2 and dqkv.size(1) == 384 and dqkv.is_contiguous()
            and dqkv.data_ptr() % 16 == 0 and w_in.data_ptr() % 16 == 0)


def lds_linear_qkv_ok(xp, x, w_in):
    """q | k | v as ONE launch: the f32x6 kernel's column groups may read different inputs (csrc/dense_f32x6.hip)"""
    return (_MATMUL_MODE == 'f32x6' and w_in.shape == (384, 128) and lds_linear_ok(xp, w_in[:256]) and lds_linear_ok(x, w_in[256:])
            and xp.shape == x.shape and xp.stride(0) == x.stride(0) and w_in.is_contiguous())


def lds_linear_qkv(xp, x, w_in, b_in):
    """[M, 384] = [(xp W_q^T | xp W_k^T) | x W_v^T] + b: q = k = feat + pos, v = feat (sst_basic_block_v2.py:56-62)"""
    m = x.size(0)
    y = torch.empty((m, 384), dtype=torch.float32, device=x.device)
    rc = _lib.load().sst_tall_linear_epi2_f32x6(_lib.ptr(xp), _lib.ptr(x), 256, x.stride(0), _lib.ptr(w_in), w_in.stride(0), 0,
                                               _lib.ptr(b_in), m, 128, 384, EPI_BIAS, None, None, 0, _lib.ptr(y), 384,
                                               _lib.stream_ptr())
    _lib.check(rc, 'sst_tall_linear_epi2_f32x6')
    return y


def inproj_pos_ok(x, pos_spec, w_in):
    """the in-projection from x + positional rows formed on load (sst_inproj_pos_f32x6): exact-split mode, d_model 128"""
    if _MATMUL_MODE != 'f32x6' or pos_spec is None:
        return False
    table, idx = pos_spec
    return (x.dim() == 2 and x.size(1) == 128 and w_in.shape == (384, 128) and w_in.is_contiguous() and x.is_contiguous()
            and x.is_cuda and x.dtype == torch.float32 and x.data_ptr() % 16 == 0 and w_in.data_ptr() % 16 == 0
            and table.dtype == torch.float32 and table.dim() == 2 and table.size(1) == 128 and table.is_contiguous()
            and table.data_ptr() % 16 == 0 and idx.dtype == torch.int32 and idx.is_contiguous() and idx.numel() == x.size(0))


def inproj_pos(x, pos_spec, w_in, b_in):
    """[M, 384] = [((x + table[idx]) W_q^T | (x + table[idx]) W_k^T) | x W_v^T] + b (sst_basic_block_v2.py:56-62) - "x + pos" is
    never a tensor"""
    m = x.size(0)
    y = torch.empty((m, 384), dtype=torch.float32, device=x.device)
    rc = _lib.load().sst_inproj_pos_f32x6(_lib.ptr(x), 128, _lib.ptr(pos_spec[0]), _lib.ptr(pos_spec[1]), _lib.ptr(w_in), 128,
                                          _lib.ptr(b_in), m, _lib.ptr(y), 384, _lib.stream_ptr())
    _lib.check(rc, 'sst_inproj_pos_f32x6')
    return y


def _gelu_gemm_ok(x, w, n, k):
    return (x.is_cuda and x.dtype == torch.float32 and w.dtype == torch.float32 and k == 128 and n % 128 == 0
            and x.dim() == 2 and x.stride(1) == 1 and w.stride(1) == 1 and x.stride(0) % 4 == 0 and w.stride(0) % 4 == 0
            and x.data_ptr() % 16 == 0 and w.data_ptr() % 16 == 0 and (x.size(0) + 32) * n < (1 << 30))


def linear_gelu(x, w1, b1):
    """(pre, h) = (x w1^T + b1, gelu(pre)) for the first FFN layer: bias and activation ride on the GEMM's epilogue
    (csrc/tall_gemm.hip, one launch per 128 output columns); None when the shape is not one the kernel is built for."""
    m, k = x.shape
    n = w1.size(0)
    if w1.size(1) != k or not _gelu_gemm_ok(x, w1, n, k) or b1 is None:
        return None
    pre = torch.empty((m, n), dtype=torch.float32, device=x.device)
    h = torch.empty((m, n), dtype=torch.float32, device=x.device)
    lib = _lib.load()
    for j in range(0, n, 128):
        rc = lib.sst_tall_linear_gelu_f32(_lib.ptr(x), x.stride(0), _lib.ptr(w1[j:j + 128]), w1.stride(0),
                                          _lib.ptr(b1[j:j + 128]), m, 128, k, 0, 0, _lib.ptr(h[:, j:]),
                                          _lib.ptr(pre[:, j:]), n, _lib.stream_ptr())
        _lib.check(rc, 'sst_tall_linear_gelu_f32')
    return pre, h


def dgrad_gelu(dy, w2, pre):
    """d(pre) = (dy w2) * gelu'(pre) for the second FFN layer (w2: [out = k, in = n]): the activation's derivative is
    applied in the epilogue of the data-gradient GEMM; None when the shape is not one the kernel is built for."""
    m, k = dy.shape
    n = w2.size(1)
    if w2.size(0) != k or not _gelu_gemm_ok(dy, w2, n, k) or pre.shape != (m, n) or not pre.is_contiguous():
        return None
    out = torch.empty((m, n), dtype=torch.float32, device=dy.device)
    lib = _lib.load()
    for j in range(0, n, 128):
        rc = lib.sst_tall_linear_gelu_f32(_lib.ptr(dy), dy.stride(0), _lib.ptr(w2[:, j:]), w2.stride(0), None, m, 128,
                                          k, 1, 1, _lib.ptr(pre[:, j:]), _lib.ptr(out[:, j:]), n, _lib.stream_ptr())
        _lib.check(rc, 'sst_tall_linear_gelu_f32')
    return out


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
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            dw, db = weight_bias_grad(dy, x, want_b)
        elif want_b:
            db = colsum(dy)
        return dx, dw, db


def tall_linear(x, weight, bias=None):
    if x.dim() != 2 or x.dtype != torch.float32 or not x.is_cuda:
        return torch.nn.functional.linear(x, weight, bias)
    return TallLinear.apply(x, weight, bias)


_LN_ACTS = {None: 0, 'gelu': 1, 'relu': 2}


def add_ln_fwd(x, res, weight, bias, eps, save_sum=True, act=None, pos=None):
    """-> (y, s, stats) with s = x + res (== x when res is None), stats [M,2] = (mean, rstd).
    save_sum=False (inference): the sum is not written (s is None), a quarter of the kernel's traffic.
    act: None | 'gelu' | 'relu' applied to the norm's output in the same pass.
    pos = (table [P, C], row index int32 [M]): -> (y, s, stats, y + table[index]) from the same pass (act None, C % 4 == 0)."""
    x = x.contiguous()
    m, c = x.shape
    if res is not None:
        res = res.contiguous()
    y = torch.empty_like(x)
    s = (torch.empty_like(x) if save_sum else None) if res is not None else x
    stats = torch.empty((m, 2), dtype=torch.float32, device=x.device)
    if pos is not None:
        assert act is None
        yp = torch.empty_like(x)
        rc = _lib.load().sst_add_layernorm_pos_fwd_f32(_lib.ptr(x), _lib.ptr(res), _lib.ptr(weight), _lib.ptr(bias), m, c, float(eps),
                                                       _lib.ptr(y), _lib.ptr(s) if (res is not None and save_sum) else None,
                                                       _lib.ptr(stats), _lib.ptr(pos[0]), _lib.ptr(pos[1]), _lib.ptr(yp),
                                                       _lib.stream_ptr())
        _lib.check(rc, 'sst_add_layernorm_pos_fwd_f32')
        return y, s, stats, yp
    rc = _lib.load().sst_add_layernorm_act_fwd_f32(_lib.ptr(x), _lib.ptr(res), _lib.ptr(weight), _lib.ptr(bias), m, c,
                                                   float(eps), _LN_ACTS[act], _lib.ptr(y),
                                                   _lib.ptr(s) if (res is not None and save_sum) else None,
                                                   _lib.ptr(stats), _lib.stream_ptr())
    _lib.check(rc, 'sst_add_layernorm_act_fwd_f32')
    return y, s, stats


def add_ln_act_bwd(dy, s, stats, weight, bias, act):
    """-> (d(x + res), dweight, dbias) of y = act(LayerNorm(x + res)); dy arrives at the activation's output"""
    dy = dy.contiguous()
    m, c = s.shape
    dx = torch.empty_like(s)
    dw = torch.empty(c, dtype=torch.float32, device=s.device)
    db = torch.empty(c, dtype=torch.float32, device=s.device)
    lib = _lib.load()
    ws = _lib.workspace(lib.sst_add_layernorm_bwd_workspace_bytes(m, c), s.device)
    rc = lib.sst_add_layernorm_act_bwd_f32(_lib.ptr(dy), _lib.ptr(s), _lib.ptr(stats), _lib.ptr(weight), _lib.ptr(bias),
                                           _LN_ACTS[act], m, c, _lib.ptr(dx), _lib.ptr(dw), _lib.ptr(db), _lib.ptr(ws),
                                           _lib.stream_ptr())
    _lib.check(rc, 'sst_add_layernorm_act_bwd_f32')
    return dx, dw, db


def add_ln_bwd(dy, s, stats, weight, dy2=None):
    """-> (d(x + res), dweight, dbias); dy2: optional second gradient arriving at the LayerNorm output (c = 128)."""
    dy = dy.contiguous()
    m, c = s.shape
    dx = torch.empty_like(s)
    dw = torch.empty(c, dtype=torch.float32, device=s.device)
    db = torch.empty(c, dtype=torch.float32, device=s.device)
    lib = _lib.load()
    ws = _lib.workspace(lib.sst_add_layernorm_bwd_workspace_bytes(m, c), s.device)
    if dy2 is not None and c != 128:
        dy, dy2 = dy + dy2, None
    rc = lib.sst_add_layernorm_bwd2_f32(_lib.ptr(dy), _lib.ptr(dy2.contiguous() if dy2 is not None else None), _lib.ptr(s),
                                        _lib.ptr(stats), _lib.ptr(weight), m, c, _lib.ptr(dx), _lib.ptr(dw), _lib.ptr(db),
                                        _lib.ptr(ws), _lib.stream_ptr())
    _lib.check(rc, 'sst_add_layernorm_bwd2_f32')
    return dx, dw, db


def lds_linear_add_ln_ok(x, w, res, c):
    return (c == 128 and lds_linear_ok(x, w) and w.size(0) == 128 and res.is_cuda and res.dtype == torch.float32
            and res.shape == (x.size(0), 128) and res.is_contiguous() and res.data_ptr() % 16 == 0)


def lds_linear_add_ln(x, w, bias, res, ln_weight, ln_bias, eps, save_sum=True, pos=None):
    """(y, s, stats, y_plus_pos) with y = LayerNorm(x @ w^T + bias + res) in exact fp32: the projection and
    `norm(src + src2)` in one kernel (csrc/dense_f32.hip, kEpiAddLN); w [128, K], K = 128 | 256; pos = (table fp32 [P, 128],
    row index int32 [M]) adds the second output y + table[index]."""
    m, k = x.shape
    y = torch.empty((m, 128), dtype=torch.float32, device=x.device)
    s = torch.empty((m, 128), dtype=torch.float32, device=x.device) if save_sum else None
    stats = torch.empty((m, 2), dtype=torch.float32, device=x.device)
    yp = torch.empty((m, 128), dtype=torch.float32, device=x.device) if pos is not None else None
    lib = _lib.load()
    if _MATMUL_MODE == 'f32x6' and k != 128:
        # three bf16 images of a 128-column group at K = 256 (203 KB) exceed the LDS and a LayerNorm epilogue needs the whole row
        # in one workgroup: the product runs over 64-column groups with the residual in its epilogue (53 us in the step), the
        # LayerNorm (+ the next layer's positional input) as one pass over the sum (~20 us) - against 95 us for the fused
        # kernel on the fp32 matrix pipe
        s = lds_linear(x, w, bias, EPI_ADD, aux_in=res)
        out = add_ln_fwd(s, None, ln_weight, ln_bias, eps, pos=pos)
        return out[0], s, out[2], (out[3] if pos is not None else None)
    entry = getattr(lib, 'sst_tall_linear_ln_' + _MATMUL_MODE)
    rc = entry(
        _lib.ptr(x), x.stride(0), _lib.ptr(w), w.stride(0), _lib.ptr(bias), m, k, _lib.ptr(res), 128, _lib.ptr(ln_weight),
        _lib.ptr(ln_bias), float(eps), _lib.ptr(y), _lib.ptr(s), _lib.ptr(stats),
        _lib.ptr(pos[0]) if pos is not None else None, _lib.ptr(pos[1]) if pos is not None else None, _lib.ptr(yp),
        _lib.stream_ptr())
    _lib.check(rc, 'sst_tall_linear_ln_' + _MATMUL_MODE)
    return y, s, stats, yp


def encoder_tail_ok(o, x, w_out, w1, w2):
    """the one-kernel tail of an encoder layer (csrc/layer_tail_x6.hip): exact-split mode, d_model 128, feed-forward 256"""
    if _MATMUL_MODE != 'f32x6':
        return False
    if not (o.shape == x.shape and x.dim() == 2 and x.size(1) == 128 and w_out.shape == (128, 128) and w1.shape == (256, 128)
            and w2.shape == (128, 256)):
        return False
    return all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.data_ptr() % 16 == 0
               for t in (o, x, w_out, w1, w2))


def encoder_tail_pack(w_out, w1, w2, out=None):
    """the chunk images (three bf16 parts, LDS layout) of a layer's out-projection / linear1 / linear2 weights that
    encoder_tail_fwd / _bwd fetch with LDS-DMA: one small launch per layer call (uint8 tensor, ~1.1 MB)"""
    lib = _lib.load()
    if out is None:
        out = torch.empty(int(lib.sst_encoder_tail_pack_bytes()), dtype=torch.uint8, device=w_out.device)
    _lib.check(lib.sst_encoder_tail_pack_f32x6(_lib.ptr(w_out), _lib.ptr(w1), _lib.ptr(w2), _lib.ptr(out), _lib.stream_ptr()),
               'sst_encoder_tail_pack_f32x6')
    return out


def encoder_tail_fwd(o, x, packed, b_out, b1, b2, n1w, n1b, n2w, n2b, eps, act, save=True, pos=None, out=None):
    """y1 = LN1(x + o W_o^T + b_o); pre = y1 W_1^T + b_1; h = act(pre); s2 = y1 + h W_2^T + b_2; y2 = LN2(s2) as ONE kernel
    (sst_basic_block_v2.py:113-118); packed = encoder_tail_pack(W_o, W_1, W_2).  -> dict(s1 (None unless save), st1, y1, pre,
    h, s2, st2, y2, y2p (pos = (table, index))); ``out``: optional dict of preallocated tensors by the same names."""
    m = x.size(0)
    dev = x.device
    out = dict(out) if out else {}

    def e(name, cols):
        if name not in out or out[name] is None:
            out[name] = torch.empty((m, cols), dtype=torch.float32, device=dev)
        return out[name]
    s1 = e('s1', 128) if save else None
    out['s1'] = s1
    st1, y1, pre, h, s2, st2, y2 = e('st1', 2), e('y1', 128), e('pre', 256), e('h', 256), e('s2', 128), e('st2', 2), e('y2', 128)
    y2p = e('y2p', 128) if pos is not None else None
    out['y2p'] = y2p
    P = lambda t: None if t is None else t.data_ptr()   # noqa: E731
    args = _lib.EncoderTailFwdArgs(
        m, 1 if act == 'gelu' else 2, 0, float(eps), 0.0, P(o), P(x), P(packed), P(b_out), P(b1), P(b2), P(n1w), P(n1b),
        P(n2w), P(n2b), P(pos[0]) if pos is not None else None, P(pos[1]) if pos is not None else None,
        P(s1), P(st1), P(y1), P(pre), P(h), P(s2), P(st2), P(y2), P(y2p))
    import ctypes
    _lib.check(_lib.load().sst_encoder_tail_fwd_f32x6(ctypes.byref(args), _lib.stream_ptr()), 'sst_encoder_tail_fwd_f32x6')
    return out


def encoder_tail_bwd(dy2, dy2p, s2, st2, pre, s1, st1, packed, n1w, n2w, act):
    """the backward of encoder_tail_fwd up to the attention output: -> (ds2, dpre, ds1, d_o, dn [4, 128] = dn2w | dn2b | dn1w | dn1b);
    the weight gradients of the three linears are left to the caller (ds2 with h, dpre with y1, ds1 with o)."""
    m = dy2.size(0)
    dev = dy2.device

    def e(cols):
        return torch.empty((m, cols), dtype=torch.float32, device=dev)
    ds2, dpre, ds1, d_o = e(128), e(256), e(128), e(128)
    dn = torch.empty((4, 128), dtype=torch.float32, device=dev)
    lib = _lib.load()
    ws = _lib.workspace(lib.sst_encoder_tail_bwd_workspace_bytes(m), dev)
    P = lambda t: None if t is None else t.data_ptr()   # noqa: E731
    d0 = dn.data_ptr()
    args = _lib.EncoderTailBwdArgs(m, 1 if act == 'gelu' else 2, 0, P(dy2), P(dy2p), P(s2), P(st2), P(pre), P(s1), P(st1), P(packed),
                                   P(n1w), P(n2w), P(ds2), P(dpre), P(ds1), P(d_o), d0, d0 + 512, d0 + 1024, d0 + 1536, P(ws))
    import ctypes
    _lib.check(lib.sst_encoder_tail_bwd_f32x6(ctypes.byref(args), _lib.stream_ptr()), 'sst_encoder_tail_bwd_f32x6')
    return ds2, dpre, ds1, d_o, dn


class AddLayerNorm(Function):
    """y = act(LayerNorm(x + res)) (act None | 'gelu' | 'relu'); the gradient w.r.t. x and res is the same tensor."""

    @staticmethod
    def forward(ctx, x, res, weight, bias, eps, act=None):
        y, s, stats = add_ln_fwd(x, res, weight, bias, eps, act=act)
        ctx.save_for_backward(s, stats, weight, bias)
        ctx.has_res, ctx.act = res is not None, act
        return y

    @staticmethod
    def backward(ctx, dy):
        s, stats, weight, bias = ctx.saved_tensors
        if ctx.act is None:
            dx, dw, db = add_ln_bwd(dy, s, stats, weight)
        else:
            dx, dw, db = add_ln_act_bwd(dy, s, stats, weight, bias, ctx.act)
        return dx, (dx if ctx.has_res else None), dw, db, None, None


def _act_name(act):
    """None / 'gelu' / 'relu' for the activations the LayerNorm kernels fold in, False for anything else"""
    if act is None or isinstance(act, torch.nn.Identity):
        return None
    if isinstance(act, torch.nn.GELU) and getattr(act, 'approximate', 'none') == 'none':
        return 'gelu'
    if type(act) is torch.nn.ReLU:
        return 'relu'
    return False


def add_layer_norm(x, res, norm, act=None):
    """act(norm(x + res)) (res may be None; act: an nn.GELU / nn.ReLU module or None, folded into the same pass when the
    kernel runs): the fused add + LayerNorm kernel for an nn.LayerNorm, the module's own forward
    for any other norm layer; torch for LayerNorm shapes the kernel is not built for (C > 512, norms over more than the
    last dimension)."""
    c = x.size(-1)
    post = (lambda t: t) if act is None else act
    if not isinstance(norm, torch.nn.LayerNorm):
        # layer_cfg use_bn=True (sst_basic_block_v2.py:92-99, configs/fsd/fsd_waymoD1_1x_sst_encoder.py): norm1 / norm2
        # are naiveSyncBN1d modules, whose own forward runs the batch-norm kernels of csrc/bn.hip
        return post(norm(x + res if res is not None else x))
    ok = (x.dim() == 2 and x.dtype == torch.float32 and x.is_cuda
          and norm.elementwise_affine and norm.bias is not None and 1 <= c <= 512 and len(norm.normalized_shape) == 1)
    if not ok:
        y = x + res if res is not None else x
        return post(torch.nn.functional.layer_norm(y, norm.normalized_shape, norm.weight, norm.bias, norm.eps))
    name = _act_name(act)
    if name is False:
        return act(AddLayerNorm.apply(x, res, norm.weight, norm.bias, norm.eps))
    return AddLayerNorm.apply(x, res, norm.weight, norm.bias, norm.eps, name)
