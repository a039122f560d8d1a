"""Reduced-precision (bf16) mode of the SRA encoder stack.

The reference trains SST with ``fp16 = dict(loss_scale=32.0)`` (configs/sst_refactor/sst_waymoD5_1x_3class_8heads_v2.py:82;
mmcv's Fp16OptimizerHook: half-precision activations and weights inside the wrapped modules, fp32 master weights in the
optimizer, normalisation layers computing in fp32; the VFE is pinned to fp32 by ``@force_fp32``,
voxel_encoder.py:229).  On MI355X the counterpart is bf16 - the fp32 exponent range, so no loss scaling - and it is
where the matrix cores are: exact fp32 MFMA runs at 1/16 of the bf16 rate.

What this mode does to one encoder layer (sst_basic_block_v2.py:104-119):
  * activations between kernels are bf16 in HBM (x, q|k, v, o, the FFN intermediate, both LayerNorm outputs);
  * every product accumulates in fp32 (bf16 MFMA), the softmax / log-sum-exp of the attention core and the LayerNorm
    statistics are fp32 (csrc/sra_attn_bf16.hip, csrc/dense.hip);
  * parameters stay fp32 (the module's own tensors are the master weights); the GEMMs read bf16 shadows that are
    refreshed when a parameter's version counter moves; parameter gradients are produced in fp32;
  * "x + positional embedding" for the next layer's q / k is a second output of the LayerNorm kernel (no add pass, no
    [M, C] positional tensor).
The dense products are library GEMMs on bf16 tensors (hipBLASLt through torch), the weight gradients a batched
split-K product with an fp32 reduction; the sparse / row-wise kernels are this package's.  The fp32 path and its
1e-3 parity tests are untouched; parity of this mode is stated at bf16 resolution (tests/test_gpu_bf16.py).
"""
import math

import torch
import torch.nn.functional as F
from torch.autograd import Function

from . import _lib
from . import kernels as K

BF16 = torch.bfloat16
# residual + LayerNorm in the epilogue of the projection in front of it (SST_AMD_BF16_FUSED_LN=0: separate LayerNorm kernel)
_FUSED_LN = int(__import__('os').environ.get('SST_AMD_BF16_FUSED_LN', '1'))


# ------------------------------------------------------------------------------------------------------------------
# kernel wrappers
# ------------------------------------------------------------------------------------------------------------------
def _ld(t):
    if t.dim() != 2 or t.stride(1) != 1:
        raise RuntimeError('sst_amd.bf16: 2-D tensors with unit inner stride expected')
    return t.stride(0)


def sra_fwd(q, k, v, plan, n_heads, scale):
    m, c = q.shape
    for t in (q, k, v):
        if t.dtype != BF16 or not t.is_cuda:
            raise RuntimeError('sst_amd.bf16.sra_fwd: bf16 CUDA tensors expected')
    if c != n_heads * 16:
        raise RuntimeError('sst_amd.bf16.sra_fwd: head_dim must be 16')
    alloc = torch.empty if plan.n_tokens == m else torch.zeros
    o = alloc((m, c), dtype=BF16, device=q.device)
    lse = torch.empty((m, n_heads), dtype=torch.float32, device=q.device)
    lib = _lib.load()
    order = plan.order
    rc = _timed('sra_fwd_bf16', plan.n_tokens, 0, lambda: lib.sst_sra_attn_fwd_ord_bf16(
        _lib.ptr(q), _lib.ptr(k), _lib.ptr(v), _ld(q), _ld(k), _ld(v), plan.tok_ptr(0), _lib.ptr(plan.winoff),
        _lib.ptr(order) if order is not None else None, plan.n_windows, n_heads, float(scale), plan.max_tokens,
        _lib.ptr(o), _ld(o), _lib.ptr(lse), _lib.stream_ptr()))
    _lib.check(rc, 'sst_sra_attn_fwd_ord_bf16')
    return o, lse


def sra_bwd(q, k, v, o, lse, do, plan, n_heads, scale, dq, dk, dv):
    lib = _lib.load()
    order = plan.order
    rc = _timed('sra_bwd_bf16', plan.n_tokens, 1, lambda: lib.sst_sra_attn_bwd_ord_bf16(
        _lib.ptr(q), _lib.ptr(k), _lib.ptr(v), _lib.ptr(o), _lib.ptr(do), _lib.ptr(lse), _ld(q), _ld(k), _ld(v), _ld(o),
        _ld(do), plan.tok_ptr(0), _lib.ptr(plan.winoff), _lib.ptr(order) if order is not None else None,
        plan.n_windows, n_heads, float(scale), plan.max_tokens, _lib.ptr(dq), _lib.ptr(dk), _lib.ptr(dv), _ld(dq), _ld(dk),
        _ld(dv), _lib.stream_ptr()))
    _lib.check(rc, 'sst_sra_attn_bwd_ord_bf16')


def sra_cos_fwd(q, k, v, plan, n_heads, head_scale):
    """scaled cosine attention on bf16 tensors (sst_sra_attn_cos_fwd_bf16): head_scale [n_heads] fp32 on the device"""
    m, c = q.shape
    alloc = torch.empty if plan.n_tokens == m else torch.zeros
    o = alloc((m, c), dtype=BF16, device=q.device)
    lse = torch.empty((m, n_heads), dtype=torch.float32, device=q.device)
    lib = _lib.load()
    order = plan.order
    rc = _timed('sra_fwd_bf16', plan.n_tokens, 0, lambda: lib.sst_sra_attn_cos_fwd_bf16(
        _lib.ptr(q), _lib.ptr(k), _lib.ptr(v), _ld(q), _ld(k), _ld(v), plan.tok_ptr(0), _lib.ptr(plan.winoff),
        _lib.ptr(order) if order is not None else None, plan.n_windows, n_heads, _lib.ptr(head_scale), plan.max_tokens,
        _lib.ptr(o), _ld(o), _lib.ptr(lse), _lib.stream_ptr()))
    _lib.check(rc, 'sst_sra_attn_cos_fwd_bf16')
    return o, lse


def sra_cos_bwd(q, k, v, o, lse, do, plan, n_heads, head_scale, dq, dk, dv):
    """-> r [M, n_heads] fp32 (normalize(q) . d normalize(q)): d head_scale = colsum(r) / head_scale"""
    m = q.size(0)
    r = (torch.empty if plan.n_tokens == m else torch.zeros)((m, n_heads), dtype=torch.float32, device=q.device)
    lib = _lib.load()
    order = plan.order
    rc = _timed('sra_bwd_bf16', plan.n_tokens, 1, lambda: lib.sst_sra_attn_cos_bwd_bf16(
        _lib.ptr(q), _lib.ptr(k), _lib.ptr(v), _lib.ptr(o), _lib.ptr(do), _lib.ptr(lse), _ld(q), _ld(k), _ld(v), _ld(o),
        _ld(do), plan.tok_ptr(0), _lib.ptr(plan.winoff), _lib.ptr(order) if order is not None else None,
        plan.n_windows, n_heads, _lib.ptr(head_scale), plan.max_tokens, _lib.ptr(dq), _lib.ptr(dk), _lib.ptr(dv), _ld(dq),
        _ld(dk), _ld(dv), _lib.ptr(r), _lib.stream_ptr()))
    _lib.check(rc, 'sst_sra_attn_cos_bwd_bf16')
    return r


def _timed(kind, n_tokens, backward, fn):
    """bench.py hook: kernel-bound HIP events on every kernels.EVENT_STRIDE-th launch (same scheme as the fp32 kernels)"""
    if K.EVENT_SINK is None or kind not in K.EVENT_KINDS:
        return fn()
    K._event_counter[kind] = K._event_counter.get(kind, 0) + 1
    if (K._event_counter[kind] - 1) % K.EVENT_STRIDE != 0:
        return fn()
    lib = _lib.load()
    ke = K._KernelEvents(lib)
    lib.sst_sra_attn_bf16_profile_next(backward, ke.start, ke.stop)
    r = fn()
    lib.sst_sra_attn_bf16_profile_next(backward, None, None)
    K.EVENT_SINK.append((kind, ke, ke, n_tokens))
    return r


class SRAAttentionBF16(Function):
    """softmax(q k^T * scale) v per window on bf16 tensors (fp32 softmax / accumulation); gradients in bf16."""

    @staticmethod
    def forward(ctx, q, k, v, plan, n_heads, scale):
        o, lse = sra_fwd(q, k, v, plan, n_heads, scale)
        ctx.plan, ctx.n_heads, ctx.scale = plan, n_heads, float(scale)
        ctx.save_for_backward(q, k, v, o, lse)
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse = ctx.saved_tensors
        full = ctx.plan.n_tokens == q.size(0)
        alloc = torch.empty if full else torch.zeros
        dq, dk, dv = (alloc(q.shape, dtype=BF16, device=q.device) for _ in range(3))
        sra_bwd(q, k, v, o, lse, do.contiguous(), ctx.plan, ctx.n_heads, ctx.scale, dq, dk, dv)
        return dq, dk, dv, None, None, None


def sra_attention(q, k, v, plan, n_heads, scale=None):
    return SRAAttentionBF16.apply(q, k, v, plan, n_heads, 1.0 / math.sqrt(16.0) if scale is None else scale)


def add_ln_fwd(x, res, weight, bias, eps, save_sum=True, pos=None):
    """-> (y, s, stats, y_plus_pos): bf16 y = LN(x + res), s = x + res (bf16, kept for the backward pass), fp32 stats;
    pos = (table fp32 [P, C], index int32 [M]) adds the second output y + table[index]."""
    m, c = x.shape
    y = torch.empty_like(x)
    s = torch.empty_like(x) if (save_sum and res is not None) else (x if res is None else None)
    stats = torch.empty((m, 2), dtype=torch.float32, device=x.device)
    yp = torch.empty_like(x) if pos is not None else None
    rc = _lib.load().sst_add_layernorm_fwd_bf16(
        _lib.ptr(x), _lib.ptr(res), _lib.ptr(weight), _lib.ptr(bias), m, c, float(eps), _lib.ptr(y),
        _lib.ptr(s) if (res is not None and save_sum) else None, _lib.ptr(stats),
        _lib.ptr(pos[0]) if pos is not None else None, _lib.ptr(pos[1]) if pos is not None else None, _lib.ptr(yp),
        _lib.stream_ptr())
    _lib.check(rc, 'sst_add_layernorm_fwd_bf16')
    return y, s, stats, yp


def add_ln_bwd(dy, dy2, s, stats, weight):
    """-> (d(x + res) bf16, dweight fp32, dbias fp32); dy2: optional second gradient arriving at the LayerNorm output"""
    m, c = s.shape
    dx = torch.empty_like(s)
    dw = torch.empty(c, dtype=torch.float32, device=s.device)
    db = torch.empty(c, dtype=torch.float32, device=s.device)
    lib = _lib.load()
    ws = _lib.workspace(lib.sst_add_layernorm_bwd_workspace_bytes(m, c), s.device)
    rc = lib.sst_add_layernorm_bwd_bf16(_lib.ptr(dy.contiguous()), _lib.ptr(dy2.contiguous() if dy2 is not None else None),
                                        _lib.ptr(s), _lib.ptr(stats), _lib.ptr(weight), m, c, _lib.ptr(dx), _lib.ptr(dw),
                                        _lib.ptr(db), _lib.ptr(ws), _lib.stream_ptr())
    _lib.check(rc, 'sst_add_layernorm_bwd_bf16')
    return dx, dw, db


def cast_add_pos(x, pos=None):
    """bf16(x [+ table[index]]) for an fp32 or bf16 [M, C] tensor"""
    x = x.contiguous()
    m, c = x.shape
    out = torch.empty((m, c), dtype=BF16, device=x.device)
    rc = _lib.load().sst_cast_add_pos_bf16(_lib.ptr(x), int(x.dtype == BF16), m, c,
                                           _lib.ptr(pos[0]) if pos is not None else None,
                                           _lib.ptr(pos[1]) if pos is not None else None, _lib.ptr(out), _lib.stream_ptr())
    _lib.check(rc, 'sst_cast_add_pos_bf16')
    return out


EPI_BIAS, EPI_GELU, EPI_RELU, EPI_MUL_GELU_GRAD, EPI_MUL_RELU_GRAD, EPI_ADD = range(6)
_LINEAR_SHAPES = ((128, 128), (128, 256), (256, 128))


def tall_linear(x, w, bias=None, epilogue=EPI_BIAS, aux_in=None, want_pre=False):
    """y (bf16 [M, N]) = epilogue(x (bf16 [M, K]) @ w (bf16 [N, K])^T + bias (fp32)); csrc/dense_bf16.hip.
    epilogue GELU / RELU with want_pre: -> (y, pre-activation bf16); MUL_*_GRAD / ADD read ``aux_in`` (bf16 [M, N])."""
    m, k = x.shape
    n = w.size(0)
    if (k, n) not in _LINEAR_SHAPES or w.size(1) != k or x.dtype != BF16 or w.dtype != BF16 or not w.is_contiguous():
        raise RuntimeError(f'sst_amd.bf16.tall_linear: unsupported operands {tuple(x.shape)} x {tuple(w.shape)}')
    y = torch.empty((m, n), dtype=BF16, device=x.device)
    pre = torch.empty((m, n), dtype=BF16, device=x.device) if want_pre else None
    aux = aux_in if aux_in is not None else pre
    rc = _lib.load().sst_tall_linear_bf16(_lib.ptr(x), _ld(x), _lib.ptr(w), _lib.ptr(bias), m, k, n, int(epilogue),
                                          _lib.ptr(aux_in), _lib.ptr(pre), _ld(aux) if aux is not None else 0,
                                          _lib.ptr(y), n, _lib.stream_ptr())
    _lib.check(rc, 'sst_tall_linear_bf16')
    return (y, pre) if want_pre else y


def linear_add_ln(x, w, bias, res, ln_weight, ln_bias, eps, save_sum=True, pos=None):
    """(y, s, stats, y_plus_pos) with y = LayerNorm(x @ w^T + bias + res): the projection and `norm(src + src2)` in one
    kernel (csrc/dense_bf16.hip, kEpiAddLN); w bf16 [128, K], K = 128 | 256."""
    m, k = x.shape
    if w.shape != (128, k) or k not in (128, 256) or x.dtype != BF16 or res.shape != (m, 128) or not res.is_contiguous():
        raise RuntimeError('sst_amd.bf16.linear_add_ln: unsupported operands')
    y = torch.empty((m, 128), dtype=BF16, device=x.device)
    s = torch.empty((m, 128), dtype=BF16, device=x.device) if save_sum else None
    stats = torch.empty((m, 2), dtype=torch.float32, device=x.device)
    yp = torch.empty((m, 128), dtype=BF16, device=x.device) if pos is not None else None
    rc = _lib.load().sst_tall_linear_ln_bf16(
        _lib.ptr(x), _ld(x), _lib.ptr(w), _lib.ptr(bias), m, k, _lib.ptr(res), 128, _lib.ptr(ln_weight), _lib.ptr(ln_bias),
        float(eps), _lib.ptr(y), _lib.ptr(s), _lib.ptr(stats), _lib.ptr(pos[0]) if pos is not None else None,
        _lib.ptr(pos[1]) if pos is not None else None, _lib.ptr(yp), _lib.stream_ptr())
    _lib.check(rc, 'sst_tall_linear_ln_bf16')
    return y, s, stats, yp


def wgrad_group(problems):
    """Weight / bias gradients of several tall products in ONE launch (csrc/dense_bf16.hip).
    problems: list of (a bf16 [M, P], b bf16 [M, 128], out_w fp32, out_b fp32 | None, bias_side, transpose_out);
    out_w[P][128] = a^T b (or its transpose [128][P]); out_b = column sums of a (bias_side 1) or b (bias_side 2)."""
    lib = _lib.load()
    arr = (_lib.WgradProblemBF16 * len(problems))()
    for q, (a, b, out_w, out_b, side, tr) in zip(arr, problems):
        if a.dtype != BF16 or b.dtype != BF16 or b.size(1) != 128 or a.size(1) not in (128, 256) or a.size(0) != b.size(0):
            raise RuntimeError('sst_amd.bf16.wgrad_group: operands must be bf16 [M, 128|256] and [M, 128]')
        if out_w.dtype != torch.float32 or not out_w.is_contiguous() or out_w.numel() != a.size(1) * 128:
            raise RuntimeError('sst_amd.bf16.wgrad_group: out_w must be a contiguous fp32 tensor of P x 128 values')
        q.a, q.b, q.lda, q.ldb, q.m = a.data_ptr(), b.data_ptr(), _ld(a), _ld(b), a.size(0)
        q.out_w, q.out_b = out_w.data_ptr(), (out_b.data_ptr() if out_b is not None else None)
        q.p, q.bias_side, q.transpose_out = a.size(1), int(side if out_b is not None else 0), int(tr)
    dev = problems[0][0].device
    ws = _lib.workspace(lib.sst_wgrad_group_workspace_bytes(arr, len(problems)), dev)
    _lib.check(lib.sst_wgrad_group_bf16(arr, len(problems), _lib.ptr(ws), _lib.stream_ptr()), 'sst_wgrad_group_bf16')


# ------------------------------------------------------------------------------------------------------------------
# host pieces
# ------------------------------------------------------------------------------------------------------------------
_shadows = {}   # id(parameter) -> (weak reference to it, {(rows, transposed): bf16 copy})


def _shadow_slot(p):
    import weakref
    entry = _shadows.get(id(p))
    if entry is None or entry[0]() is not p:
        pid = id(p)
        entry = _shadows[pid] = (weakref.ref(p, lambda _r, pid=pid: _shadows.pop(pid, None)), {})
    return entry[1]


def refresh_shadows(specs):
    """(Re)make the bf16 copies of fp32 parameters: specs = iterable of (parameter, rows | None, transposed).  ONE launch
    for all of them (sst_cast_group_bf16); the destination buffers are kept per parameter object and overwritten in place.
    Called by run_encoder_stack at EVERY forward: the copies follow the parameters whatever wrote them - optimizer steps,
    but also `.data` writes (mmcv's EMAHook swap, a master-to-model copy the Fp16OptimizerHook way), which do not move
    the version counter a cache could watch."""
    specs = list(specs)
    if not specs:
        return None
    arr = (_lib.CastProblemBF16 * len(specs))()
    for q, (p, rows, transposed) in zip(arr, specs):
        if p.dtype != torch.float32 or p.dim() != 2 or not p.is_cuda or p.stride(1) != 1:
            raise RuntimeError('sst_amd.bf16.refresh_shadows: fp32 CUDA matrices with unit column stride only')
        lo, hi = (0, p.size(0)) if rows is None else rows
        r, c = hi - lo, p.size(1)
        per_param = _shadow_slot(p)
        dst = per_param.get((rows, transposed))
        shape = (c, r) if transposed else (r, c)
        if dst is None or dst.shape != shape or dst.device != p.device:
            dst = per_param[(rows, transposed)] = torch.empty(shape, dtype=BF16, device=p.device)
        q.src, q.dst, q.ld_src = p.data_ptr() + 4 * lo * p.stride(0), dst.data_ptr(), p.stride(0)
        q.rows, q.cols, q.transpose = r, c, int(bool(transposed))
    _lib.check(_lib.load().sst_cast_group_bf16(arr, len(specs), _lib.stream_ptr()), 'sst_cast_group_bf16')
    return arr


def invalidate_shadows():
    """Drop every bf16 copy (they are re-made by the next forward anyway; frees their memory)."""
    _shadows.clear()


def shadow(p, rows=None, transposed=False):
    """bf16 copy of an fp32 parameter - of the row range ``rows`` = (lo, hi) of it, transposed if asked (the operand of
    a data gradient).  Inside an encoder stack the copies were refreshed by this forward's refresh_shadows call; a copy
    that does not exist yet is made on the spot.  The cache belongs to the parameter OBJECT (weak reference, dropped
    when it dies): a new parameter that reuses the id of a dead one never sees its copy."""
    hit = _shadow_slot(p).get((rows, transposed))
    if hit is None or hit.device != p.device:
        refresh_shadows([(p, rows, transposed)])
        hit = _shadow_slot(p)[(rows, transposed)]
    return hit


def _refresh_stack_shadows(layers, with_grad):
    """refresh_shadows for a whole encoder stack with the problem array kept between calls: as long as the parameters sit
    where they sat (same storage addresses), a forward costs the address check and ONE C call - the copies themselves are
    re-made every time"""
    params = [p for enc in layers for p in (enc.win_attn.self_attn.in_proj_weight, enc.win_attn.self_attn.out_proj.weight,
                                            enc.linear1.weight, enc.linear2.weight)]
    signature = (with_grad,) + tuple(p.data_ptr() for p in params)
    cached = getattr(layers[0], '_bf16_refresh', None)
    if cached is not None and cached[0] == signature and all(_shadow_slot(p) for p in params):
        _lib.check(_lib.load().sst_cast_group_bf16(cached[1], cached[2], _lib.stream_ptr()), 'sst_cast_group_bf16')
        return
    specs = [spec for enc in layers for spec in layer_shadow_specs(enc, with_grad)]
    arr = refresh_shadows(specs)
    layers[0]._bf16_refresh = (signature, arr, len(specs))


def layer_shadow_specs(enc, with_grad):
    """the copies one encoder layer's kernels read: q|k rows and v rows of in_proj, out_proj, linear1, linear2, and, for
    the backward pass, the transposes of all five"""
    attn = enc.win_attn.self_attn
    c = attn.out_proj.weight.size(0)
    mats = [(attn.in_proj_weight, (0, 2 * c)), (attn.in_proj_weight, (2 * c, 3 * c)), (attn.out_proj.weight, None),
            (enc.linear1.weight, None), (enc.linear2.weight, None)]
    specs = [(p, rows, False) for p, rows in mats]
    if with_grad:
        specs += [(p, rows, True) for p, rows in mats]
    return specs


def tail_pack(w_out, w1, w2, out=None):
    """bf16 chunk images (LDS layout) of a layer's out-projection / linear1 / linear2 weights for tail_fwd / tail_bwd, from the
    fp32 masters: one small launch per layer call (uint8 tensor, 320 KB)"""
    lib = _lib.load()
    if out is None:
        out = torch.empty(int(lib.sst_encoder_tail_pack_bf16_bytes()), dtype=torch.uint8, device=w_out.device)
    _lib.check(lib.sst_encoder_tail_pack_bf16(_lib.ptr(w_out), _lib.ptr(w1), _lib.ptr(w2), _lib.ptr(out), _lib.stream_ptr()),
               'sst_encoder_tail_pack_bf16')
    return out


def tail_weights_ok(w_out, w1, w2):
    return (w_out.shape == (128, 128) and w1.shape == (256, 128) and w2.shape == (128, 256)
            and all(w.dtype == torch.float32 and w.is_contiguous() and w.is_cuda for w in (w_out, w1, w2)))


def tail_ok(o, x, w_out, w1, w2):
    return (o.shape == x.shape and x.dim() == 2 and x.size(1) == 128 and w_out.shape == (128, 128) and w1.shape == (256, 128)
            and w2.shape == (128, 256) and o.dtype == BF16 and x.dtype == BF16 and o.is_contiguous() and x.is_contiguous()
            and o.is_cuda and o.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0
            and all(w.dtype == torch.float32 and w.is_contiguous() for w in (w_out, w1, w2)))


def tail_fwd(o, x, packed, b_out, b1, b2, n1w, n1b, n2w, n2b, eps, act, save=True, pos=None, out=None):
    """out-projection -> + x -> norm1 -> linear1 -> act -> linear2 -> + y1 -> norm2 (sst_basic_block_v2.py:113-118) as ONE kernel,
    bf16 storage (csrc/layer_tail_bf16.hip) -> dict(s1, st1, y1, pre, h, s2, st2, y2, y2p)"""
    import ctypes
    m = x.size(0)
    dev = x.device
    out = dict(out) if out else {}

    def e(name, cols, dtype=BF16):
        if name not in out or out[name] is None:
            out[name] = torch.empty((m, cols), dtype=dtype, device=dev)
        return out[name]
    s1 = e('s1', 128) if save else None
    s2 = e('s2', 128) if save else None
    out['s1'], out['s2'] = s1, s2
    st1, st2 = e('st1', 2, torch.float32), e('st2', 2, torch.float32)
    y1, pre, h, y2 = e('y1', 128), e('pre', 256), e('h', 256), e('y2', 128)
    y2p = e('y2p', 128) if pos is not None else None
    out['y2p'] = y2p
    P = lambda t: None if t is None else t.data_ptr()   # noqa: E731
    args = _lib.EncoderTailFwdBF16Args(
        m, 1 if act == 'gelu' else 2, 0, float(eps), 0.0, P(o), P(x), P(packed), P(b_out), P(b1), P(b2), P(n1w), P(n1b),
        P(n2w), P(n2b), P(pos[0]) if pos is not None else None, P(pos[1]) if pos is not None else None,
        P(s1), P(st1), P(y1), P(pre), P(h), P(s2), P(st2), P(y2), P(y2p))
    _lib.check(_lib.load().sst_encoder_tail_fwd_bf16(ctypes.byref(args), _lib.stream_ptr()), 'sst_encoder_tail_fwd_bf16')
    return out


def tail_bwd(dy2, dy2p, s2, st2, pre, s1, st1, packed, n1w, n2w, act):
    """-> (ds2, dpre, ds1, d_o (bf16), dn fp32 [4, 128] = dn2w | dn2b | dn1w | dn1b)"""
    import ctypes
    m = dy2.size(0)
    dev = dy2.device

    def e(cols):
        return torch.empty((m, cols), dtype=BF16, device=dev)
    ds2, dpre, ds1, d_o = e(128), e(256), e(128), e(128)
    dn = torch.empty((4, 128), dtype=torch.float32, device=dev)
    lib = _lib.load()
    ws = _lib.workspace(lib.sst_encoder_tail_bwd_bf16_workspace_bytes(m), dev)
    P = lambda t: None if t is None else t.data_ptr()   # noqa: E731
    d0 = dn.data_ptr()
    args = _lib.EncoderTailBwdBF16Args(m, 1 if act == 'gelu' else 2, 0, P(dy2), P(dy2p), P(s2), P(st2), P(pre), P(s1), P(st1),
                                       P(packed), P(n1w), P(n2w), P(ds2), P(dpre), P(ds1), P(d_o), d0, d0 + 512, d0 + 1024,
                                       d0 + 1536, P(ws))
    _lib.check(lib.sst_encoder_tail_bwd_bf16(ctypes.byref(args), _lib.stream_ptr()), 'sst_encoder_tail_bwd_bf16')
    return ds2, dpre, ds1, d_o, dn


def tail_images(w_out, w1, w2, refresh=False):
    """the layer's tail weight images (tail_pack), kept with the out-projection parameter like its bf16 copies; refreshed by
    run_encoder_stack at every forward (one launch for the whole stack), made on the spot when missing"""
    slot = _shadow_slot(w_out)
    t = slot.get('tail')
    if t is None or t.device != w_out.device:
        t = slot['tail'] = tail_pack(w_out, w1, w2)
    elif refresh:
        tail_pack(w_out, w1, w2, out=t)
    return t


def _refresh_stack_tail_images(layers):
    """tail images of every layer of a stack in ONE launch (sst_encoder_tail_pack_bf16_many), every forward: they follow the fp32
    masters whatever wrote them (see refresh_shadows)"""
    import ctypes
    lib = _lib.load()
    nbytes = int(lib.sst_encoder_tail_pack_bf16_bytes())
    triples = [(enc.win_attn.self_attn.out_proj.weight, enc.linear1.weight, enc.linear2.weight) for enc in layers]
    bufs = []
    for wo, w1, w2 in triples:
        slot = _shadow_slot(wo)
        t = slot.get('tail')
        if t is None or t.device != wo.device or t.numel() != nbytes:
            t = slot['tail'] = torch.empty(nbytes, dtype=torch.uint8, device=wo.device)
        bufs.append(t)
    n = len(triples)
    P = ctypes.c_void_p * n
    a = P(*[t[0].data_ptr() for t in triples]), P(*[t[1].data_ptr() for t in triples]), P(*[t[2].data_ptr() for t in triples])
    d = P(*[b.data_ptr() for b in bufs])
    _lib.check(lib.sst_encoder_tail_pack_bf16_many(a[0], a[1], a[2], d, n, _lib.stream_ptr()), 'sst_encoder_tail_pack_bf16_many')


def weight_grad(dy, x, chunk=2048):
    """dW [out, in] (fp32) = dy^T x for tall bf16 operands: batched split-K product over row chunks (each chunk a full
    MFMA-shaped GEMM for the library), partials reduced in fp32; bias gradient = fp32 column sum."""
    m = dy.size(0)
    s = m // chunk
    if s < 2:
        return (dy.t() @ x).float()
    body = s * chunk
    part = torch.bmm(dy[:body].view(s, chunk, dy.size(1)).transpose(1, 2), x[:body].view(s, chunk, x.size(1)))
    dw = part.sum(0, dtype=torch.float32)
    if body < m:
        dw += (dy[body:].t() @ x[body:]).float()
    return dw


def bias_grad(dy):
    return dy.sum(0, dtype=torch.float32)


class _CastIn(Function):
    """fp32 features -> (bf16 copy, bf16 copy + positional embedding); the two gradients are summed in fp32"""

    @staticmethod
    def forward(ctx, x, table, index):
        return cast_add_pos(x), cast_add_pos(x, (table, index))

    @staticmethod
    def backward(ctx, dx, dxp):
        return dx.float() + dxp.float(), None, None


# The whole layer as ONE library call per direction (csrc/layer_exec.hip, sst_encoder_layer_{fwd,bwd}_bf16: the launch sequence
# of EncoderLayerBF16Fn below issued from C).  The Python side of that sequence is 6 + 10 foreign calls per layer with their
# argument marshalling and ~25 tensor allocations: 4.7-5.1 ms of host time per step against 6.0 ms of kernels - on a slower or
# busier host the reduced-precision step was bound by the interpreter (driver-run 8.1 ms against 6.1 ms here).
# SST_AMD_LAYER_EXEC_BF16=0: the Python sequence (same kernels, same order, same bits: tests/test_gpu_bf16.py).
_LAYER_EXEC = int(__import__('os').environ.get('SST_AMD_LAYER_EXEC_BF16', '1'))
# what a layer keeps for its backward pass, in ONE allocation: (name, columns, bytes per element); pieces padded to 256 bytes
_SLAB = (('qk', 256, 2), ('v', 128, 2), ('o', 128, 2), ('lse', 8, 4), ('s1', 128, 2), ('st1', 2, 4), ('y1', 128, 2),
         ('pre', 256, 2), ('h', 256, 2), ('s2', 128, 2), ('st2', 2, 4))
# scratch of the backward call (bf16 columns), one allocation: ds2 | dpre | dy1 | ds1 | d_o | dqkv
_SCRATCH = (('ds2', 128), ('dpre', 256), ('dy1', 128), ('ds1', 128), ('d_o', 128), ('dqkv', 384))


def _slab_offsets(m):
    off, out = 0, {}
    for name, cols, size in _SLAB:
        out[name] = off
        off += (m * cols * size + 255) // 256 * 256
    return out, off


def _exec_ok(x, xp, plan, nhead, act, w_in, w1, w2):
    m, c = x.shape
    return (_LAYER_EXEC and _FUSED_LN and c == 128 and nhead == 8 and act in ('gelu', 'relu') and m > 0
            and w_in.shape == (384, 128) and w1.shape == (256, 128) and w2.shape == (128, 256) and plan.n_tokens == m
            and 0 < plan.max_tokens <= 144 and x.is_contiguous() and xp.is_contiguous() and x.dtype == BF16 and xp.dtype == BF16
            and x.data_ptr() % 16 == 0 and xp.data_ptr() % 16 == 0)


def _exec_fwd(x, xp, plan, nhead, act, eps, scale, pos_next, params, need_bwd, head_scale=None):
    """-> (slab, y2, y2p); head_scale ([nhead] fp32, device): scaled cosine attention"""
    import ctypes
    w_in, b_in, w_out, b_out, w1, b1, w2, b2, n1w, n1b, n2w, n2b = params
    m, c = x.shape
    dev = x.device
    offs, total = _slab_offsets(m)
    slab = torch.empty(total, dtype=torch.uint8, device=dev)
    base = slab.data_ptr()
    y2 = torch.empty((m, c), dtype=BF16, device=dev)
    y2p = torch.empty((m, c), dtype=BF16, device=dev) if pos_next is not None else None
    P = lambda t: None if t is None else t.data_ptr()   # noqa: E731
    S = lambda name: base + offs[name]                   # noqa: E731
    order = plan.order
    args = _lib.EncoderLayerFwdBF16Args(
        m, plan.n_windows, nhead, 1 if act == 'gelu' else 2, plan.max_tokens, 0, float(eps), float(scale),
        P(x), P(xp), P(shadow(w_in, (0, 2 * c))), P(shadow(w_in, (2 * c, 3 * c))), P(shadow(w_out)), P(shadow(w1)), P(shadow(w2)),
        P(b_in), P(b_out), P(b1), P(b2), P(n1w), P(n1b), P(n2w), P(n2b),
        None if plan.tok_ptr(0) is None else plan.tok.data_ptr(), P(plan.winoff), P(order),
        P(pos_next[0]) if pos_next is not None else None, P(pos_next[1]) if pos_next is not None else None,
        S('qk'), S('v'), S('o'), S('lse'), S('y1'), S('s1') if need_bwd else None, S('st1'), S('pre'), S('h'),
        S('s2') if need_bwd else None, S('st2'), P(y2), P(y2p), P(head_scale), P(tail_images(w_out, w1, w2)))
    lib = _lib.load()
    rc = _timed('sra_fwd_bf16', plan.n_tokens, 0, lambda: lib.sst_encoder_layer_fwd_bf16(ctypes.byref(args), _lib.stream_ptr()))
    _lib.check(rc, 'sst_encoder_layer_fwd_bf16')
    return slab, y2, y2p


def _exec_bwd(ctx, dy2, dy2p):
    import ctypes
    x, xp, slab, w_in, w_out, w1, w2, n1w, n2w = ctx.saved_tensors[:9]
    head_scale = ctx.saved_tensors[9] if ctx.cosine else None
    m, c = x.shape
    dev = x.device
    plan = ctx.plan
    offs, _ = _slab_offsets(m)
    cos_r = torch.empty((m, ctx.nhead), dtype=torch.float32, device=dev) if ctx.cosine else None
    base = slab.data_ptr()
    scratch = torch.empty((m, sum(cols for _, cols in _SCRATCH)), dtype=BF16, device=dev)
    sp, so = {}, scratch.data_ptr()
    for name, cols in _SCRATCH:
        sp[name] = so
        so += 2 * m * cols
    dx = torch.empty((m, c), dtype=BF16, device=dev)
    dxp = torch.empty((m, c), dtype=BF16, device=dev)
    f32 = dict(dtype=torch.float32, device=dev)
    dw_in, db_in = torch.empty((3 * c, c), **f32), torch.empty(3 * c, **f32)
    dwo, dbo = torch.empty((c, c), **f32), torch.empty(c, **f32)
    dw1, db1 = torch.empty((256, c), **f32), torch.empty(256, **f32)
    dw2, db2 = torch.empty((c, 256), **f32), torch.empty(c, **f32)
    dn = torch.empty((4, c), **f32)
    lib = _lib.load()
    ws = _lib.workspace(lib.sst_encoder_layer_bwd_bf16_workspace_bytes(m), dev)
    dy2 = dy2.contiguous()
    dy2p = dy2p.contiguous() if dy2p is not None else None
    P = lambda t: None if t is None else t.data_ptr()   # noqa: E731
    S = lambda name: base + offs[name]                   # noqa: E731
    dnp = dn.data_ptr()
    order = plan.order
    args = _lib.EncoderLayerBwdBF16Args(
        m, plan.n_windows, ctx.nhead, 1 if ctx.act == 'gelu' else 2, plan.max_tokens, 0, 0.0, float(ctx.scale),
        P(dy2), P(dy2p), P(x), P(xp), S('qk'), S('v'), S('o'), S('lse'), S('s1'), S('st1'), S('y1'), S('pre'), S('h'), S('s2'),
        S('st2'), P(shadow(w_in, (0, 2 * c), transposed=True)), P(shadow(w_in, (2 * c, 3 * c), transposed=True)),
        P(shadow(w_out, transposed=True)), P(shadow(w1, transposed=True)), P(shadow(w2, transposed=True)), P(n1w), P(n2w),
        None if plan.tok_ptr(0) is None else plan.tok.data_ptr(), P(plan.winoff), P(order),
        sp['ds2'], sp['dpre'], sp['dy1'], sp['ds1'], sp['d_o'], sp['dqkv'], P(dxp), P(dx),
        P(dw_in), P(db_in), P(dwo), P(dbo), P(dw1), P(db1), P(dw2), P(db2), dnp, dnp + 4 * c, dnp + 8 * c, dnp + 12 * c, P(ws),
        P(head_scale), P(cos_r), P(tail_images(w_out, w1, w2)))
    rc = _timed('sra_bwd_bf16', plan.n_tokens, 1, lambda: lib.sst_encoder_layer_bwd_bf16(ctypes.byref(args), _lib.stream_ptr()))
    _lib.check(rc, 'sst_encoder_layer_bwd_bf16')
    d_scale = K.head_scale_grad(cos_r, head_scale) if ctx.cosine else None
    return (dx, dxp, None, None, None, None, None, dw_in, db_in, dwo, dbo, dw1, db1, dw2, db2, dn[0], dn[1], dn[2], dn[3], d_scale)


class EncoderLayerBF16Fn(Function):
    """One post-norm SRA encoder layer (sst_basic_block_v2.py:104-119) in the reduced-precision mode, as one autograd node.
    Inputs x and xp = x + positional embedding (bf16); outputs the layer result and (when ``pos_next`` is given) the
    result + the next layer's positional embedding.  Kernel sequence, forward: 2 projections, attention core,
    out-projection+add+LayerNorm, linear1+activation, linear2+add+LayerNorm (6 launches); backward: 2 LayerNorm, 5 data
    gradients, attention core, ONE grouped weight-gradient launch + its reduction (10 launches)."""

    @staticmethod
    def forward(ctx, x, xp, plan, nhead, act, eps, pos_next, w_in, b_in, w_out, b_out, w1, b1, w2, b2, n1w, n1b, n2w, n2b,
                head_scale=None):
        """head_scale ([nhead] fp32 on the device, differentiable): scaled cosine attention, normalisation inside the kernels"""
        c = x.size(1)
        ctx.exec = False
        ctx.cosine = head_scale is not None
        if ctx.cosine:
            head_scale = head_scale.float().contiguous()
        if _exec_ok(x, xp, plan, nhead, act, w_in, w1, w2):
            need_bwd = any(ctx.needs_input_grad)
            scale = 1.0 / math.sqrt(16.0)
            slab, y2, y2p = _exec_fwd(x, xp, plan, nhead, act, eps, scale, pos_next,
                                      (w_in, b_in, w_out, b_out, w1, b1, w2, b2, n1w, n1b, n2w, n2b), need_bwd, head_scale)
            if need_bwd:
                ctx.save_for_backward(x, xp, slab, w_in, w_out, w1, w2, n1w, n2w, *((head_scale,) if ctx.cosine else ()))
                ctx.plan, ctx.nhead, ctx.act, ctx.scale, ctx.two = plan, nhead, act, scale, y2p is not None
                ctx.exec = True
            return y2 if y2p is None else (y2, y2p)
        # q | k from x + pos, v from x (sst_basic_block_v2.py:58-63); written side by side for the core
        qk = tall_linear(xp, shadow(w_in, (0, 2 * c)), b_in[:2 * c])
        v = tall_linear(x, shadow(w_in, (2 * c, 3 * c)), b_in[2 * c:])
        scale = 1.0 / math.sqrt(16.0)
        if ctx.cosine:
            o, lse = sra_cos_fwd(qk[:, :c], qk[:, c:], v, plan, nhead, head_scale)
        else:
            o, lse = sra_fwd(qk[:, :c], qk[:, c:], v, plan, nhead, scale)
        need_bwd = any(ctx.needs_input_grad)
        ctx.tail = False
        if _FUSED_LN and c == 128 and tail_ok(o, x, w_out, w1, w2) and act in ('gelu', 'relu'):
            # everything behind the attention core as ONE kernel (csrc/layer_tail_bf16.hip), as csrc/layer_exec.hip issues it
            t = tail_fwd(o, x, tail_images(w_out, w1, w2), b_out, b1, b2, n1w, n1b, n2w, n2b, eps, act, save=True, pos=pos_next)
            s1, st1, y1, pre, h, s2, st2, y2, y2p = (t[k] for k in ('s1', 'st1', 'y1', 'pre', 'h', 's2', 'st2', 'y2', 'y2p'))
            ctx.tail = True
        elif _FUSED_LN:
            # out-projection + residual + LayerNorm, linear1 + activation, linear2 + residual + LayerNorm: three launches
            y1, s1, st1, _ = linear_add_ln(o, shadow(w_out), b_out, x, n1w, n1b, eps, save_sum=need_bwd)
            h, pre = tall_linear(y1, shadow(w1), b1, EPI_GELU if act == 'gelu' else EPI_RELU, want_pre=True)
            y2, s2, st2, y2p = linear_add_ln(h, shadow(w2), b2, y1, n2w, n2b, eps, save_sum=need_bwd, pos=pos_next)
        else:
            a = tall_linear(o, shadow(w_out), b_out)
            y1, s1, st1, _ = add_ln_fwd(x, a, n1w, n1b, eps, save_sum=need_bwd)
            h, pre = tall_linear(y1, shadow(w1), b1, EPI_GELU if act == 'gelu' else EPI_RELU, want_pre=True)
            f = tall_linear(h, shadow(w2), b2)
            y2, s2, st2, y2p = add_ln_fwd(y1, f, n2w, n2b, eps, save_sum=need_bwd, pos=pos_next)
        if need_bwd:
            ctx.save_for_backward(x, xp, qk, v, o, lse, s1, st1, y1, pre, h, s2, st2, w_in, w_out, w1, w2, n1w, n2w,
                                  *((head_scale,) if ctx.cosine else ()))
            ctx.plan, ctx.nhead, ctx.act, ctx.scale, ctx.two = plan, nhead, act, scale, y2p is not None
        if y2p is None:
            return y2
        return y2, y2p

    @staticmethod
    def backward(ctx, dy2, dy2p=None):
        if ctx.exec:
            return _exec_bwd(ctx, dy2, dy2p if ctx.two else None)
        x, xp, qk, v, o, lse, s1, st1, y1, pre, h, s2, st2, w_in, w_out, w1, w2, n1w, n2w = ctx.saved_tensors[:19]
        head_scale = ctx.saved_tensors[19] if ctx.cosine else None
        c = x.size(1)
        dev = x.device
        if ctx.tail:
            ds2, dpre, ds1, do, dn = tail_bwd(dy2.contiguous(), dy2p.contiguous() if (ctx.two and dy2p is not None) else None,
                                              s2, st2, pre, s1, st1, tail_images(w_out, w1, w2), n1w, n2w, ctx.act)
            dn2w, dn2b, dn1w, dn1b = dn[0], dn[1], dn[2], dn[3]
        else:
            ds2, dn2w, dn2b = add_ln_bwd(dy2, dy2p if ctx.two else None, s2, st2, n2w)   # = d(y1 residual) = d(f)
            dpre = tall_linear(ds2, shadow(w2, transposed=True), None,
                               EPI_MUL_GELU_GRAD if ctx.act == 'gelu' else EPI_MUL_RELU_GRAD, aux_in=pre)
            dy1 = tall_linear(dpre, shadow(w1, transposed=True), None, EPI_ADD, aux_in=ds2)   # residual + FFN branch
            ds1, dn1w, dn1b = add_ln_bwd(dy1, None, s1, st1, n1w)                             # = d(x residual) = d(a)
            do = tall_linear(ds1, shadow(w_out, transposed=True))
        dqkv = torch.empty((x.size(0), 3 * c), dtype=BF16, device=dev)
        d_scale = None
        if ctx.cosine:
            r = sra_cos_bwd(qk[:, :c], qk[:, c:], v, o, lse, do, ctx.plan, ctx.nhead, head_scale, dqkv[:, :c], dqkv[:, c:2 * c],
                            dqkv[:, 2 * c:])
            d_scale = K.head_scale_grad(r, head_scale)
        else:
            sra_bwd(qk[:, :c], qk[:, c:], v, o, lse, do, ctx.plan, ctx.nhead, ctx.scale, dqkv[:, :c], dqkv[:, c:2 * c],
                    dqkv[:, 2 * c:])
        dxp = tall_linear(dqkv[:, :2 * c], shadow(w_in, (0, 2 * c), transposed=True))
        dx = tall_linear(dqkv[:, 2 * c:], shadow(w_in, (2 * c, 3 * c), transposed=True), None, EPI_ADD, aux_in=ds1)
        # every parameter gradient of the layer in one launch
        f32 = dict(dtype=torch.float32, device=dev)
        dw_in, db_in = torch.empty((3 * c, c), **f32), torch.empty(3 * c, **f32)
        dwo, dbo = torch.empty((c, c), **f32), torch.empty(c, **f32)
        dw1, db1 = torch.empty_like(w1, dtype=torch.float32), torch.empty(w1.size(0), **f32)
        dw2, db2 = torch.empty_like(w2, dtype=torch.float32), torch.empty(w2.size(0), **f32)
        wgrad_group([(dqkv[:, :2 * c], xp, dw_in[:2 * c], db_in[:2 * c], 1, 0),
                     (dqkv[:, 2 * c:], x, dw_in[2 * c:], db_in[2 * c:], 1, 0),
                     (ds1, o, dwo, dbo, 1, 0),
                     (dpre, y1, dw1, db1, 1, 0),
                     (h, ds2, dw2, db2, 2, 1)])     # dW2 [128][256] = ds2^T h: operands swapped, stored transposed
        return (dx, dxp, None, None, None, None, None, dw_in, db_in, dwo, dbo, dw1, db1, dw2, db2, dn1w, dn1b, dn2w, dn2b,
                d_scale)


def layer_supported(enc, plan, m):
    wa = enc.win_attn
    return (enc.post_norm and isinstance(enc.norm1, torch.nn.LayerNorm)
            and isinstance(enc.norm2, torch.nn.LayerNorm) and enc.act_name in ('gelu', 'relu')
            and isinstance(plan, K.WindowPlan) and plan.n_tokens == m and plan.max_tokens <= 144
            and wa.d_model == 128 and wa.head_dim == 16 and enc.linear1.out_features == 256      # the shapes csrc/dense_bf16.hip is built for
            and not (enc.training and (wa.attn_dropout > 0 or enc.dropout.p > 0)))


def run_encoder_stack(blocks, feats, plans, pos_specs):
    """The shift blocks in the reduced-precision mode: feats fp32 [M, C] -> fp32 [M, C].  plans: the two WindowPlans;
    pos_specs: per partition (positional table fp32 [P, C], row index int32 [M])."""
    layers = [enc for block in blocks for enc in block.encoder_list]
    _refresh_stack_shadows(layers, torch.is_grad_enabled())   # one launch, every forward
    if _FUSED_LN and all(tail_weights_ok(enc.win_attn.self_attn.out_proj.weight, enc.linear1.weight, enc.linear2.weight)
                         for enc in layers):
        _refresh_stack_tail_images(layers)                    # ... and one for the tail kernels' weight images
    from .sst_basic_block import stack_head_scales
    scales = stack_head_scales(layers)      # cosine layers: 1 / clamp(tau) of the whole stack in one pass (None: standard attention)
    x, xp = _CastIn.apply(feats, pos_specs[0][0], pos_specs[0][1])
    for li, enc in enumerate(layers):
        attn = enc.win_attn.self_attn
        pos_next = pos_specs[(li + 1) % 2] if li + 1 < len(layers) else None
        out = EncoderLayerBF16Fn.apply(
            x, xp, plans[li % 2], enc.win_attn.nhead, enc.act_name, enc.norm1.eps, pos_next, attn.in_proj_weight,
            attn.in_proj_bias, attn.out_proj.weight, attn.out_proj.bias, enc.linear1.weight, enc.linear1.bias,
            enc.linear2.weight, enc.linear2.bias, enc.norm1.weight, enc.norm1.bias, enc.norm2.weight, enc.norm2.bias, scales[li])
        x, xp = out if pos_next is not None else (out, None)
    return x.float()
