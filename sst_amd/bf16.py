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
    rc = _timed('sra_fwd_bf16', plan.n_tokens, 0, lambda: lib.sst_sra_attn_fwd_bf16(
        _lib.ptr(q), _lib.ptr(k), _lib.ptr(v), _ld(q), _ld(k), _ld(v), _lib.ptr(plan.tok), _lib.ptr(plan.winoff),
        plan.n_windows, n_heads, float(scale), plan.max_tokens, _lib.ptr(o), _ld(o), _lib.ptr(lse), _lib.stream_ptr()))
    _lib.check(rc, 'sst_sra_attn_fwd_bf16')
    return o, lse


def sra_bwd(q, k, v, o, lse, do, plan, n_heads, scale, dq, dk, dv):
    lib = _lib.load()
    rc = _timed('sra_bwd_bf16', plan.n_tokens, 1, lambda: lib.sst_sra_attn_bwd_bf16(
        _lib.ptr(q), _lib.ptr(k), _lib.ptr(v), _lib.ptr(o), _lib.ptr(do), _lib.ptr(lse), _ld(q), _ld(k), _ld(v), _ld(o),
        _ld(do), _lib.ptr(plan.tok), _lib.ptr(plan.winoff), plan.n_windows, n_heads, float(scale), plan.max_tokens,
        _lib.ptr(dq), _lib.ptr(dk), _lib.ptr(dv), _ld(dq), _ld(dk), _ld(dv), _lib.stream_ptr()))
    _lib.check(rc, 'sst_sra_attn_bwd_bf16')


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


# ------------------------------------------------------------------------------------------------------------------
# host pieces
# ------------------------------------------------------------------------------------------------------------------
_shadows = {}


def shadow(p):
    """bf16 copy of an fp32 parameter (or of a slice of one), re-made when the parameter changes"""
    key = (p.data_ptr(), tuple(p.shape))
    hit = _shadows.get(key)
    if hit is not None and hit[0] == p._version and hit[1].device == p.device:
        return hit[1]
    s = p.detach().to(BF16).contiguous()
    _shadows[key] = (p._version, s)
    return s


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


class EncoderLayerBF16Fn(Function):
    """One post-norm SRA encoder layer (sst_basic_block_v2.py:104-119) in the reduced-precision mode, as one autograd node.
    Inputs x and xp = x + positional embedding (bf16); outputs the layer result and (when ``pos_next`` is given) the
    result + the next layer's positional embedding."""

    @staticmethod
    def forward(ctx, x, xp, plan, nhead, act, eps, pos_next, w_in, b_in, w_out, b_out, w1, b1, w2, b2, n1w, n1b, n2w, n2b):
        c = x.size(1)
        ws_in, bs_in = shadow(w_in), shadow(b_in)
        qk = torch.addmm(bs_in[:2 * c], xp, ws_in[:2 * c].t())
        v = torch.addmm(bs_in[2 * c:], x, ws_in[2 * c:].t())
        scale = 1.0 / math.sqrt(16.0)
        o, lse = sra_fwd(qk[:, :c], qk[:, c:], v, plan, nhead, scale)
        a = torch.addmm(shadow(b_out), o, shadow(w_out).t())
        need_bwd = any(ctx.needs_input_grad)
        y1, s1, st1, _ = add_ln_fwd(x, a, n1w, n1b, eps, save_sum=need_bwd)
        pre = torch.addmm(shadow(b1), y1, shadow(w1).t())
        h = F.gelu(pre) if act == 'gelu' else F.relu(pre)
        f = torch.addmm(shadow(b2), h, shadow(w2).t())
        y2, s2, st2, y2p = add_ln_fwd(y1, f, n2w, n2b, eps, save_sum=need_bwd, pos=pos_next)
        if need_bwd:
            ctx.save_for_backward(x, xp, qk, v, o, lse, s1, st1, y1, pre, h, s2, st2, w_in, w_out, w1, w2, n1w, n2w)
            ctx.plan, ctx.nhead, ctx.act, ctx.scale, ctx.two = plan, nhead, act, scale, y2p is not None
        if y2p is None:
            return y2
        return y2, y2p

    @staticmethod
    def backward(ctx, dy2, dy2p=None):
        x, xp, qk, v, o, lse, s1, st1, y1, pre, h, s2, st2, w_in, w_out, w1, w2, n1w, n2w = ctx.saved_tensors
        c = x.size(1)
        ds2, dn2w, dn2b = add_ln_bwd(dy2, dy2p if ctx.two else None, s2, st2, n2w)   # = d(y1 residual) = d(f)
        dw2, db2 = weight_grad(ds2, h), bias_grad(ds2)
        dh = ds2 @ shadow(w2)
        dpre = torch.ops.aten.gelu_backward(dh, pre) if ctx.act == 'gelu' else dh * (pre > 0).to(dh.dtype)
        dw1, db1 = weight_grad(dpre, y1), bias_grad(dpre)
        dy1 = torch.addmm(ds2, dpre, shadow(w1))                                      # residual + FFN branch
        ds1, dn1w, dn1b = add_ln_bwd(dy1, None, s1, st1, n1w)                         # = d(x residual) = d(a)
        dwo, dbo = weight_grad(ds1, o), bias_grad(ds1)
        do = ds1 @ shadow(w_out)
        dqkv = torch.empty((x.size(0), 3 * c), dtype=BF16, device=x.device)
        sra_bwd(qk[:, :c], qk[:, c:], v, o, lse, do, ctx.plan, ctx.nhead, ctx.scale, dqkv[:, :c], dqkv[:, c:2 * c],
                dqkv[:, 2 * c:])
        dw_in = torch.cat([weight_grad(dqkv[:, :2 * c], xp), weight_grad(dqkv[:, 2 * c:], x)], dim=0)
        db_in = bias_grad(dqkv)
        ws_in = shadow(w_in)
        dxp = dqkv[:, :2 * c] @ ws_in[:2 * c]
        dx = torch.addmm(ds1, dqkv[:, 2 * c:], ws_in[2 * c:])
        return (dx, dxp, None, None, None, None, None, dw_in, db_in, dwo, dbo, dw1, db1, dw2, db2, dn1w, dn1b, dn2w, dn2b)


def layer_supported(enc, plan, m):
    wa = enc.win_attn
    return (enc.post_norm and not wa.cosine and isinstance(enc.norm1, torch.nn.LayerNorm)
            and isinstance(enc.norm2, torch.nn.LayerNorm) and enc.act_name in ('gelu', 'relu')
            and isinstance(plan, K.WindowPlan) and plan.n_tokens == m and plan.max_tokens <= 144
            and wa.d_model % 32 == 0 and not (enc.training and (wa.attn_dropout > 0 or enc.dropout.p > 0)))


def run_encoder_stack(blocks, feats, plans, pos_specs):
    """The shift blocks in the reduced-precision mode: feats fp32 [M, C] -> fp32 [M, C].  plans: the two WindowPlans;
    pos_specs: per partition (positional table fp32 [P, C], row index int32 [M])."""
    layers = [enc for block in blocks for enc in block.encoder_list]
    x, xp = _CastIn.apply(feats, pos_specs[0][0], pos_specs[0][1])
    for li, enc in enumerate(layers):
        attn = enc.win_attn.self_attn
        pos_next = pos_specs[(li + 1) % 2] if li + 1 < len(layers) else None
        out = EncoderLayerBF16Fn.apply(
            x, xp, plans[li % 2], enc.win_attn.nhead, enc.act_name, enc.norm1.eps, pos_next, attn.in_proj_weight,
            attn.in_proj_bias, attn.out_proj.weight, attn.out_proj.bias, enc.linear1.weight, enc.linear1.bias,
            enc.linear2.weight, enc.linear2.bias, enc.norm1.weight, enc.norm1.bias, enc.norm2.weight, enc.norm2.bias)
        x, xp = out if pos_next is not None else (out, None)
    return x.float()
