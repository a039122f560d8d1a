"""Sparse Regional Attention block: WindowAttention -> EncoderLayer -> BasicShiftBlockV2.

Mirror of mmdet3d/models/sst/sst_basic_block_v2.py:14-178 (constructor kwargs, forward signatures and
state_dict keys: ``win_attn.self_attn.{in_proj_weight,in_proj_bias,out_proj.weight,out_proj.bias[,tau]}``,
``linear1``, ``linear2``, ``norm1``, ``norm2``) and of the cosine variant's core
(mmdet3d/models/sst/cosine_msa.py:123-185, 449-466).

The reference scatters the flat features into padded per-level [W,T,C] tensors, calls
nn.MultiheadAttention (3 in-proj GEMMs on padded tokens, bmm / masked softmax / bmm, averaged attention
map discarded) and gathers back.  Here the projections run on the M real tokens only and the attention
core is one HIP launch sequence over the window CSR (sra_attn.hip); nothing is padded or masked.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.utils.checkpoint import checkpoint

from . import _lib
from . import kernels as K
from .sra_composed import sra_attention_composed
from .dense import (encoder_tail_ok, encoder_tail_pack, encoder_tail_fwd, encoder_tail_bwd, inproj_pos_ok, inproj_pos,
                    EPI_ADD, EPI_BIAS, EPI_GELU, EPI_MUL_GELU_GRAD, EPI_MUL_RELU_GRAD, EPI_RELU, lds_linear, lds_linear_ok,
                    lds_linear_add_ln, lds_linear_add_ln_ok, lds_linear_dqkv_ok, lds_linear_qkv, lds_linear_qkv_ok,
                    weight_bias_grad_group,
                    tall_gemm, add_layer_norm, add_ln_bwd, add_ln_fwd, dgrad_gelu, linear_gelu, tall_linear,
                    weight_bias_grad)
from .norm import build_norm_layer


class CosineMultiheadAttention(nn.MultiheadAttention):
    """Parameter container with the reference's extra ``tau`` (cosine_msa.py:449-466)."""

    def __init__(self, embed_dim, num_heads, dropout=0.0, batch_first=False, tau_min=0.01, cosine=True,
                 non_shared_tau=False):
        super().__init__(embed_dim, num_heads, dropout=dropout)
        self.tau_min = tau_min
        self.cosine = cosine
        if cosine:
            if non_shared_tau:
                self.tau = nn.Parameter(torch.ones(1, num_heads, 1, 1))
            else:
                self.tau = nn.Parameter(torch.ones(1, 1, 1))


def plan_from_reference_dicts(ind_dict, num_voxels, device):
    """Build the window CSR from a reference-style flat2win dict {level: (flat2win, (flat_pos,)), ...}
    (the output of get_flat2win_inds_v2), so a voxel_info produced by reference code can drive this block."""
    info = ind_dict['batching_info']
    wkeys, inner, toks = [], [], []
    wbase, tmax = 0, 1
    for dl in info:
        if dl not in ind_dict:
            continue
        f2w, (flat_pos,) = ind_dict[dl][0].long(), ind_dict[dl][1]
        if f2w.numel() == 0:
            continue
        t = int(info[dl]['max_tokens'])
        tmax = max(tmax, t)
        w = torch.div(f2w, t, rounding_mode='floor')
        wkeys.append(w + wbase)       # windows of different levels get disjoint id ranges
        inner.append(f2w - w * t)
        toks.append(flat_pos.long())
        wbase += int(w.max().item()) + 1
    wkeys, inner, toks = torch.cat(wkeys), torch.cat(inner), torch.cat(toks)
    order = torch.argsort(wkeys * tmax + inner)
    _, counts = torch.unique_consecutive(wkeys[order], return_counts=True)
    winoff = torch.zeros(counts.numel() + 1, dtype=torch.int32, device=device)
    winoff[1:] = torch.cumsum(counts, 0).to(torch.int32)
    return K.WindowPlan(toks[order].to(torch.int32).contiguous(), winoff, counts.numel(), num_voxels, tmax)


def flat_pos_from_reference_dict(pos_dict, ind_dict, num_voxels):
    from .sst_ops import window2flat_v2
    if pos_dict is None or any(v is None for v in pos_dict.values()):
        return None
    return window2flat_v2(pos_dict, ind_dict)


class WindowAttention(nn.Module):

    def __init__(self, d_model, nhead, dropout, batch_first=False, layer_id=None, layer_cfg=dict()):
        super().__init__()
        self.nhead = nhead
        self.d_model = d_model
        if d_model % nhead != 0:
            raise ValueError('embed_dim must be divisible by num_heads')     # nn.MultiheadAttention's own check
        # head_dim 16 (every SST config: 128/8, 192/12) runs on the SRA kernels; any other one, and attention-weight
        # dropout in training, on the composed path of sst_amd/sra_composed.py
        self.head_dim = d_model // nhead
        self.cosine = layer_cfg.get('cosine', False)
        if self.cosine:
            tau_min = layer_cfg.get('tau_min', 0.01)
            self.self_attn = CosineMultiheadAttention(
                d_model, nhead, dropout=dropout, batch_first=False, tau_min=tau_min, cosine=True,
                non_shared_tau=layer_cfg.get('non_shared_tau', False))
        elif layer_cfg.get('linear', False):
            raise NotImplementedError
        else:
            self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.attn_dropout = dropout
        self.exe_counter = 0
        self.layer_id = layer_id
        self.impl = 0  # 0: MFMA kernels, 1: generic VALU kernels (validation)

    def head_scale(self):
        """cosine attention: the per-head score scale 1 / clamp(tau, tau_min) as a [nhead] device tensor (a shared tau expanded),
        differentiable in tau (cosine_msa.py:162-168); None for standard attention"""
        if not self.cosine:
            return None
        attn = self.self_attn
        return (1.0 / attn.tau.float().clamp(min=attn.tau_min)).reshape(-1).expand(self.nhead).contiguous()

    def forward(self, feat_2d, pos_dict, ind_dict, key_padding_dict=None):
        '''
        Args:
            feat_2d: [M, C] flat voxel features.
            pos_dict: flat positional embedding [M, C] (or the reference's per-level dict, or None).
            ind_dict: kernels.WindowPlan (or the reference's flat2win dict).
            key_padding_dict: unused (there is no padding); accepted for signature parity.
        '''
        composed = self.head_dim != 16 or (self.attn_dropout > 0 and self.training)
        if isinstance(ind_dict, K.WindowPlan):
            plan, pos = ind_dict, pos_dict
        else:
            plan = plan_from_reference_dicts(ind_dict, feat_2d.size(0), feat_2d.device)
            pos = flat_pos_from_reference_dict(pos_dict, ind_dict, feat_2d.size(0)) if isinstance(pos_dict, dict) \
                else pos_dict
        attn = self.self_attn
        c = self.d_model
        x = feat_2d.float()
        w, b = attn.in_proj_weight, attn.in_proj_bias
        xp = x + pos if pos is not None else x          # q = k = feat + pos ; v = feat
        qk = tall_linear(xp, w[:2 * c], b[:2 * c])
        v = tall_linear(x, w[2 * c:], b[2 * c:])
        if self.cosine and not composed and K.cosine_kernels_ok(plan, self.nhead, self.impl) and qk.is_cuda:
            # normalisation and 1 / clamp(tau) inside the attention kernels (csrc/sra_attn.hip, COS variants)
            o = K.sra_cosine_attention_qk_v(qk, v, self.head_scale(), plan, self.nhead)
        elif self.cosine:
            h = self.nhead
            q = F.normalize(qk[:, :c].reshape(-1, h, self.head_dim), dim=2)
            k = F.normalize(qk[:, c:].reshape(-1, h, self.head_dim), dim=2)
            tau = attn.tau.clamp(min=attn.tau_min).reshape(1, -1, 1)  # [1,1,1] or [1,h,1]
            q = (q / tau).reshape(-1, c)
            if composed:
                o = sra_attention_composed(q, k.reshape(-1, c), v, plan, h, 1.0, self.attn_dropout, self.training)
            else:
                o = K.sra_attention(q, k.reshape(-1, c), v, plan, h, scale=1.0, impl=self.impl)
        elif composed:
            o = sra_attention_composed(qk[:, :c], qk[:, c:], v, plan, self.nhead, 1.0 / math.sqrt(self.head_dim),
                                       self.attn_dropout, self.training)
        else:
            o = K.sra_attention_qk_v(qk, v, plan, self.nhead, scale=1.0 / math.sqrt(16.0), impl=self.impl)
        return tall_linear(o, attn.out_proj.weight, attn.out_proj.bias)


# Square (128 -> 128) projections of an encoder layer go through the LDS-resident-weight kernel (csrc/tall_gemm.hip):
# 41 us against 47 us for the best library solution at 90 k tokens.  The K = 256 shapes stay with the library
# (62 us vs 54 us); SST_AMD_TALL_GEMM=0 switches the kernel off, =2 also routes the K = 256 shapes through it.
import os as _os
_TALL_GEMM = int(_os.environ.get('SST_AMD_TALL_GEMM', '1'))
# GELU of the FFN in the epilogues of linear1 / of linear2's data gradient (csrc/tall_gemm.hip, EPI 2 / 3).  Measured and
# rejected as the default (profiles/r02/f_*): the erf / exp arithmetic of 64 elements per lane sits in the store phase of a
# hand-pipelined MFMA kernel where nothing overlaps it - 57 us (forward) and 104 us (backward) per 128-column half against
# 38 us for the plain product, i.e. 16.2 ms per step instead of 14.6 with the library GEMM + torch's HBM-bound GELU kernels.
_FUSED_GELU = int(_os.environ.get('SST_AMD_FUSED_GELU', '0'))
# Exact-fp32 linears with the weight matrix resident in LDS and the activation / residual arithmetic in their epilogues
# (csrc/dense_f32.hip; the bf16 mode's kernels are built the same way).  Default since the 8-wave variant: 35 / 62 / 65 us
# for (K, N) = (128,128) / (128,256) / (256,128) against 37 us (csrc/tall_gemm.hip) and 60-63 us (hipBLASLt, TunableOp) -
# the products themselves are at the sustained fp32 MFMA rate either way (47 us floor at 2.07 GHz for the 256-wide ones) -
# and torch's GELU / GELU-backward passes over [M, 256] (0.85 ms per step) are gone: 14.1 ms per step against 14.2-14.4.
# SST_AMD_LDS_LINEAR=0 restores the library GEMMs + csrc/tall_gemm.hip + torch's GELU kernels.
_LDS_LINEAR = int(_os.environ.get('SST_AMD_LDS_LINEAR', '1'))


def _linear_fwd(x, w, b):
    """x @ w.t() + b"""
    if _LDS_LINEAR and lds_linear_ok(x, w):
        return lds_linear(x, w, b)
    if _TALL_GEMM and w.size(0) == 128 and (w.size(1) == 128 or _TALL_GEMM > 1):
        y = tall_gemm(x, w, b)
        if y is not None:
            return y
    return torch.addmm(b, x, w.t())


def _linear_dgrad(dy, w, out=None, add=None):
    """dy @ w (w: [out_features, in_features]); ``out`` given: out += dy @ w in place; ``add`` given as well: out = add + dy @ w
    (``add`` untouched)."""
    if _LDS_LINEAR and lds_linear_ok(dy, w, trans_w=True):
        if out is not None:
            return lds_linear(dy, w, None, EPI_ADD, trans_w=True, aux_in=out if add is None else add, out=out)
        return lds_linear(dy, w, None, trans_w=True)
    if out is not None and add is not None:
        out.copy_(add)
    if _TALL_GEMM and w.size(1) == 128 and (w.size(0) == 128 or _TALL_GEMM > 1):
        y = tall_gemm(dy, w, None, trans_w=True, out=out, accumulate=out is not None)
        if y is not None:
            return y
    if out is not None:
        return out.addmm_(dy, w)
    return dy @ w


# The whole layer as ONE library call per direction (csrc/layer_exec.hip: the launch sequence below issued from C) in the
# exact-split mode at d_model 128 / feed-forward 256.  SST_AMD_LAYER_EXEC=0: the Python sequence (same kernels, same order, same
# bits: tests/test_gpu_layer_exec.py); the module attribute can be flipped at run time.
_LAYER_EXEC = int(_os.environ.get('SST_AMD_LAYER_EXEC', '1'))
# SST_AMD_POS_FOLD=0: the encoder chain hands x + positional rows from layer to layer as a tensor (round 5) instead of adding the
# rows on load in the in-projection and the W_q | W_k weight gradient (round 6)
_POS_FOLD = int(_os.environ.get('SST_AMD_POS_FOLD', '1'))


def _layer_exec_ok(x, xp, plan, nhead, act, params):
    from . import dense
    if not (_LAYER_EXEC and _LDS_LINEAR and dense.matmul_mode() == 'f32x6' and act in ('gelu', 'relu') and nhead == 8):
        return False
    w_in, b_in, w_out, b_out, w1, b1, w2, b2, n1w, n1b, n2w, n2b = params
    if any(p is None for p in params):
        return False
    m = x.size(0)
    shapes_ok = (x.shape == (m, 128) and (xp is None or xp.shape == (m, 128)) and w_in.shape == (384, 128) and w_out.shape == (128, 128)
                 and w1.shape == (256, 128) and w2.shape == (128, 256) and plan.n_tokens == m
                 and m >= 4096)     # below: the Python sequence takes the library's split-K weight gradients (dense.py)
    if not shapes_ok:
        return False
    for t in (x,) + ((xp,) if xp is not None else ()) + tuple(params):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.data_ptr() % 16 == 0):
            return False
    # the backward call must be possible too: once the forward has taken this path there is no other one to fall back to
    from . import _lib
    return int(_lib.load().sst_encoder_layer_bwd_workspace_bytes(m, nhead)) >= 0


# columns (fp32 words per token) of the tensors a layer keeps for its backward pass, in the order they sit in ONE allocation
# (every piece padded to 256 bytes): ~20 tensor allocations per layer and step were the largest single item of the host time
_SLAB = (('qkv', 384), ('o', 128), ('lse', 8), ('s1', 128), ('st1', 2), ('y1', 128), ('pre', 256), ('h', 256), ('s2', 128),
         ('st2', 2))


def _slab_offsets(m):
    off, out = 0, {}
    for name, cols in _SLAB:
        out[name] = off
        off += (m * cols * 4 + 255) // 256 * 256
    return out, off


def _layer_exec_fwd(x, xp, plan, nhead, impl, act, params, eps, scale, pos_next, need_bwd, head_scale=None, pos_spec=None,
                    wpack=None):
    """-> (slab (uint8: the kept tensors at _slab_offsets), y2, y2p, wpack); head_scale ([nhead], device): cosine attention;
    xp None + pos_spec = (table, row index): x + positional rows formed on load"""
    from . import _lib
    import ctypes
    w_in, b_in, w_out, b_out, w1, b1, w2, b2, n1w, n1b, n2w, n2b = params
    m = x.size(0)
    dev = x.device
    offs, total = _slab_offsets(m)
    slab = torch.empty(total, dtype=torch.uint8, device=dev)
    base = slab.data_ptr()
    y2 = torch.empty((m, 128), dtype=torch.float32, device=dev)
    y2p = torch.empty((m, 128), dtype=torch.float32, device=dev) if pos_next is not None else None
    lib = _lib.load()
    packed = wpack is not None     # the stack has formed this layer's weight images already (stack_tail_images: one launch for all)
    if not packed:
        wpack = torch.empty(int(lib.sst_encoder_layer_wpack_bytes()), dtype=torch.uint8, device=dev)   # weight images of the tail kernel
    order = plan.order
    P = lambda t: None if t is None else t.data_ptr()   # noqa: E731
    S = lambda name: base + offs[name]                   # noqa: E731
    args = _lib.EncoderLayerFwdArgs(
        m, plan.n_windows, nhead, 1 if act == 'gelu' else 2, plan.max_tokens, impl, float(eps), float(scale),
        P(x), P(xp), P(w_in), P(b_in), None if packed else P(w_out), P(b_out), None if packed else P(w1), P(b1),
        None if packed else P(w2), P(b2), P(n1w), P(n1b), P(n2w), P(n2b),
        None if plan.tok_ptr(impl) is None else plan.tok.data_ptr(), P(plan.winoff), P(order),
        P(pos_next[0]) if pos_next is not None else None, P(pos_next[1]) if pos_next is not None else None,
        S('qkv'), S('o'), S('lse'), S('y1'), S('s1') if need_bwd else None, S('st1'), S('pre'), S('h'), S('s2'), P(y2), S('st2'),
        P(y2p), P(head_scale), P(wpack), P(pos_spec[0]) if pos_spec is not None else None,
        P(pos_spec[1]) if pos_spec is not None else None)
    rc = K._bracket('sra_fwd', plan.n_tokens, lambda: lib.sst_encoder_layer_fwd_f32x6(ctypes.byref(args), _lib.stream_ptr()))
    if rc == _lib.SST_ERR_UNSUPPORTED:     # a layout one of the entry points does not take: the Python sequence has the retries
        return None
    _lib.check(rc, 'sst_encoder_layer_fwd_f32x6')
    return slab, y2, y2p, wpack


def _layer_exec_bwd(ctx, dy2, dy2p, saved):
    from . import _lib
    import ctypes
    x, xp, slab, w_in, w_out, w1, w2, n1w, n2w, wpack = saved[:10]
    rest = list(saved[10:])
    pos_spec = (rest.pop(0), rest.pop(0)) if ctx.pos_fold else None
    if ctx.pos_fold:
        xp = None
    head_scale = rest.pop(0) if rest else None
    m = x.size(0)
    dev = x.device
    plan, nhead, impl = ctx.plan, ctx.nhead, ctx.impl
    offs, _ = _slab_offsets(m)
    base = slab.data_ptr()

    def e(*shape):
        return torch.empty(shape, dtype=torch.float32, device=dev)
    dy2 = dy2.contiguous()
    dy2p = dy2p.contiguous() if dy2p is not None else None
    ds1 = e(m, 128)                                    # leaves as d(x)
    scratch = e(m, 128 + 256 + 128 + 384)              # ds2 | dpre | d_o | dqkv, one after the other (not interleaved)
    sb = scratch.data_ptr()
    p_ds2, p_dpre, p_do, p_dqkv = sb, sb + 4 * m * 128, sb + 4 * m * 384, sb + 4 * m * 512
    dw_in, db_in, dwo, dbo = e(384, 128), e(384), e(128, 128), e(128)
    dw1, db1, dw2, db2 = e(256, 128), e(256), e(128, 256), e(128)
    dn = e(4, 128)
    cos_r = e(m, nhead) if head_scale is not None else None
    lib = _lib.load()
    nbytes = lib.sst_encoder_layer_bwd_workspace_bytes(m, nhead)
    if nbytes < 0:
        _lib.check(int(nbytes), 'sst_encoder_layer_bwd_workspace_bytes')
    ws = _lib.workspace(nbytes, dev)
    order = plan.order
    P = lambda t: None if t is None else t.data_ptr()   # noqa: E731
    S = lambda name: base + offs[name]                   # noqa: E731
    dnp = dn.data_ptr()
    args = _lib.EncoderLayerBwdArgs(
        m, plan.n_windows, nhead, 1 if ctx.act == 'gelu' else 2, plan.max_tokens, impl, 0.0, float(ctx.scale),
        P(dy2), P(dy2p), P(x), P(xp), S('qkv'), S('o'), S('lse'), S('s1'), S('st1'), S('y1'), S('pre'), S('h'), S('s2'), S('st2'),
        P(w_in), P(w_out), P(w1), P(w2), P(n1w), P(n2w),
        None if plan.tok_ptr(impl) is None else plan.tok.data_ptr(), P(plan.winoff), P(order),
        p_ds2, p_dpre, P(ds1), p_do, p_dqkv,
        P(dw_in), P(db_in), P(dwo), P(dbo), P(dw1), P(db1), P(dw2), P(db2), dnp, dnp + 512, dnp + 1024, dnp + 1536, P(ws),
        P(head_scale), P(cos_r), None, P(wpack), P(pos_spec[0]) if pos_spec is not None else None,
        P(pos_spec[1]) if pos_spec is not None else None)
    rc = K._bracket('sra_bwd', plan.n_tokens, lambda: lib.sst_encoder_layer_bwd_f32x6(ctypes.byref(args), _lib.stream_ptr()))
    _lib.check(rc, 'sst_encoder_layer_bwd_f32x6')
    d_scale = K.head_scale_grad(cos_r, head_scale) if head_scale is not None else None
    return (ds1, None, None, None, None, None, dw_in, db_in, dwo, dbo, dw1, db1, dw2, db2, dn[0], dn[1], dn[2], dn[3], None, None,
            None, d_scale, None, None, None, None)


def _bn_rows_fwd(bn, s):
    """y = bn(s) for a BatchNorm1d-family module on the rows of s [M, C] (layer_cfg use_bn=True: norm1 / norm2 of an encoder layer
    are naiveSyncBN1d modules, sst_basic_block_v2.py:92-99): the statistics pass with the module's bookkeeping (running statistics,
    the cross-rank average of naiveSyncBN: sst_amd/norm.py bn_prepare) and the apply pass of csrc/bn.hip.
    -> (y, prep [4, C] = mean, invstd, scale, shift, (batch_stats, count, sync))"""
    from .norm import bn_prepare
    prep, batch_stats, count, sync = bn_prepare(bn, s)
    m, c = s.shape
    y = torch.empty((m, c), dtype=torch.float32, device=s.device)
    _lib.check(_lib.load().sst_bn_act_res_fwd_f32(_lib.ptr(s), m, c, s.stride(0), None, 0, _lib.ptr(prep[2]), _lib.ptr(prep[3]), 0,
                                                  _lib.ptr(y), y.stride(0), _lib.stream_ptr()), 'sst_bn_act_res_fwd_f32')
    return y, prep, (bool(batch_stats), float(count), bool(sync))


def _bn_rows_bwd(dy, s, prep, cfg):
    """gradient of _bn_rows_fwd: -> (ds, d weight, d bias).  Batch statistics: ds = scale * (g - (G1 + xhat G2) / count) with
    G1 = sum g, G2 = sum g xhat over every rank when the statistics were averaged across ranks (ops/norm.py:20-24, 53-58: one
    all-reduce of 2 C floats between the two passes); the parameter gradients stay the local sums."""
    from torch import distributed as dist
    batch_stats, count, sync = cfg
    m, c = s.shape
    dy = dy.contiguous()
    lib = _lib.load()
    sums = torch.empty((2, c), dtype=torch.float32, device=s.device)
    ws = _lib.workspace(lib.sst_bn_workspace_bytes(m, c), s.device)
    _lib.check(lib.sst_bn_act_res_bwd_reduce_f32(_lib.ptr(dy), _lib.ptr(s), None, m, c, dy.stride(0), s.stride(0), 0, _lib.ptr(prep[0]),
                                                 _lib.ptr(prep[1]), _lib.ptr(prep[2]), _lib.ptr(prep[3]), 0, _lib.ptr(sums[0]),
                                                 _lib.ptr(sums[1]), _lib.ptr(ws), _lib.stream_ptr()), 'sst_bn_act_res_bwd_reduce_f32')
    total = sums
    if sync and batch_stats:
        total = sums.clone()
        dist.all_reduce(total, async_op=False)
    ds = torch.empty((m, c), dtype=torch.float32, device=s.device)
    _lib.check(lib.sst_bn_act_res_bwd_apply_f32(_lib.ptr(dy), _lib.ptr(s), None, m, c, dy.stride(0), s.stride(0), 0, _lib.ptr(prep[0]),
                                                _lib.ptr(prep[1]), _lib.ptr(prep[2]), _lib.ptr(prep[3]), _lib.ptr(total[0]),
                                                _lib.ptr(total[1]), (1.0 / count) if batch_stats else 0.0, 0, None, 0, _lib.ptr(ds),
                                                ds.stride(0), _lib.stream_ptr()), 'sst_bn_act_res_bwd_apply_f32')
    return ds, sums[1], sums[0]


def _bn_layer_ok(norm, c):
    """a batch-norm module the chain's row kernels take (csrc/bn.hip): affine BatchNorm1d family over c channels"""
    return (isinstance(norm, nn.BatchNorm1d) and norm.weight is not None and norm.bias is not None and norm.num_features == c
            and c % 4 == 0 and c <= 1024)


class FusedEncoderLayerFn(torch.autograd.Function):
    """One post-norm SRA encoder layer (sst_basic_block_v2.py:104-119) as a single autograd node.

    Same arithmetic as the modular path (WindowAttention -> add+LN -> FFN -> add+LN), but the backward is
    written out so that every gradient accumulation rides on a GEMM's beta = 1 (``addmm``) instead of separate
    [M, C] add kernels, and weight + bias gradients come from the split-K MFMA kernel in one pass.
    """

    @staticmethod
    def forward(ctx, x, pos, plan, nhead, impl, act, w_in, b_in, w_out, b_out, w1, b1, w2, b2, n1w, n1b, n2w, n2b, eps,
                xp=None, pos_next=None, head_scale=None, xp_shares_x=False, pos_spec=None, bn=None, wpack=None):
        """wpack (optional): this layer's tail weight images, already formed (stack_tail_images); bn = (norm1, norm2) modules (optional): batch-norm layers (layer_cfg use_bn=True, sst_basic_block_v2.py:92-99,
        configs/fsd/fsd_waymoD1_1x_sst_encoder.py:70) - the same node with the two LayerNorm passes replaced by statistics + apply
        passes over the sums the projections' epilogues leave (n1w .. n2b are then the modules' weight / bias).
        pos_spec = (table fp32 [P, C], row index int32 [M]) (optional, instead of pos / xp): the positional embedding of every
        token is a row of a small table; in the exact-split mode x + table[index] is then formed ON LOAD by the in-projection and
        by the weight gradient of W_q | W_k and never exists as a tensor (round 6: 3 of a layer's 56 [M, 128] passes).
        xp (optional): x + positional embedding, already formed (the previous layer's second output) - ``pos`` is then
        ignored; pos_next = (table, row index): also return y + table[index], the next layer's xp; head_scale ([nhead] fp32 on
        the device, differentiable): scaled cosine attention with the per-head scale 1 / clamp(tau) (cosine_msa.py:123-170),
        normalisation inside the attention kernels; xp_shares_x: the caller states that the ``xp`` it hands in is x + a constant
        (the encoder chain: x + positional rows), so d(xp) may be folded into d(x) - one K = 384 product instead of two, and the
        producer's LayerNorm backward reads one upstream gradient.  An independent ``xp`` without the flag gets its own gradient
        in every mode."""
        c = x.size(1)
        x = x.contiguous()
        if head_scale is not None:
            head_scale = head_scale.contiguous()
        ctx.cosine = head_scale is not None
        from . import dense as _dense
        ctx.matmul = _dense.matmul_mode()    # the backward pass multiplies the way the forward pass did, whatever the mode is by then
        ctx.set_materialize_grads(False)     # an output nobody differentiates (y2p when its gradient was folded into y2's) stays None
        ctx.pos_fold = False
        if pos_spec is not None:
            assert xp is None and pos is None, 'pos_spec replaces pos / xp'
            if _LDS_LINEAR and inproj_pos_ok(x, pos_spec, w_in):
                ctx.pos_fold = True       # x + table[index] on load: no xp tensor
            else:                         # the modes / shapes without that kernel: the sum as a tensor, a constant offset of x
                xp = K.add_table_rows(x, pos_spec[0], pos_spec[1])
                xp_shares_x = True
        ctx.split_input = xp is not None
        ctx.fold_xp = (xp is None) or bool(xp_shares_x)     # d(xp) -> d(x): only when xp = x + constant
        if xp is None and not ctx.pos_fold:
            xp = x + pos if pos is not None else x
        params = (w_in, b_in, w_out, b_out, w1, b1, w2, b2, n1w, n1b, n2w, n2b)
        ctx.exec = False
        ctx.bn = bn is not None
        if bn is None and c == 128 and ctx.fold_xp and _layer_exec_ok(x, xp, plan, nhead, act, params):
            # the launch sequence below as ONE library call (csrc/layer_exec.hip)
            need_bwd = any(ctx.needs_input_grad)
            scale = 1.0 / math.sqrt(16.0)
            done = _layer_exec_fwd(x, xp, plan, nhead, impl, act, params, eps, scale, pos_next, need_bwd, head_scale,
                                   pos_spec if ctx.pos_fold else None, wpack)
            if done is not None:
                slab, y2, y2p, wpack = done
                if need_bwd:
                    ctx.save_for_backward(x, xp if xp is not None else x, slab, w_in, w_out, w1, w2, n1w, n2w, wpack,
                                          *(pos_spec if ctx.pos_fold else ()), *((head_scale,) if ctx.cosine else ()))
                    ctx.plan, ctx.nhead, ctx.impl, ctx.act, ctx.scale = plan, nhead, impl, act, scale
                    ctx.exec = True
                ctx.two = pos_next is not None
                return (y2, y2p) if ctx.two else y2
        if ctx.pos_fold:                                                    # x + positional rows on load (csrc/dense_f32x6.hip)
            qkv = inproj_pos(x, pos_spec, w_in, b_in)
            qk, v = qkv[:, :2 * c], qkv[:, 2 * c:]
            xp = x         # placeholder in the saved list (never read: the weight gradient adds the rows on load again)
        elif _LDS_LINEAR and c == 128 and lds_linear_qkv_ok(xp, x, w_in):   # one launch, two inputs (csrc/dense_f32x6.hip)
            qkv = lds_linear_qkv(xp, x, w_in, b_in)
            qk, v = qkv[:, :2 * c], qkv[:, 2 * c:]
        else:
            qk = _linear_fwd(xp, w_in[:2 * c], b_in[:2 * c])
            v = _linear_fwd(x, w_in[2 * c:], b_in[2 * c:])
        scale = 1.0 / math.sqrt(16.0)
        if ctx.cosine:
            res = K._sra_cos_fwd(qk[:, :c], qk[:, c:], v, plan, nhead, head_scale)
            if res is None:
                raise RuntimeError('sst_amd: cosine attention kernels refused a layout _can_fuse admitted')
            o, lse = res
        else:
            o, lse = K._sra_fwd(qk[:, :c], qk[:, c:], v, plan, nhead, scale, impl)
        need_bwd = any(ctx.needs_input_grad)  # False under torch.no_grad(): nothing is kept for a backward pass
        ctx.tail = False
        if bn is not None:
            y1, s1, st1, pre, h, s2, st2, y2, y2p, ctx.bn_cfg = FusedEncoderLayerFn._tail_batch_norm(
                o, x, w_out, b_out, w1, b1, w2, b2, bn, act, pos_next)
        elif (_LDS_LINEAR and c == 128 and act in ('gelu', 'relu') and o.is_contiguous()
                and encoder_tail_ok(o, x, w_out, w1, w2) and all(t is not None for t in (b_out, b1, b2))):
            # everything behind the attention core as ONE kernel (csrc/layer_tail_x6.hip), as csrc/layer_exec.hip issues it
            if wpack is None:
                wpack = encoder_tail_pack(w_out, w1, w2)
            t = encoder_tail_fwd(o, x, wpack, b_out, b1, b2, n1w, n1b, n2w, n2b, eps, act, save=need_bwd, pos=pos_next)
            s1, st1, y1, pre, h, s2, st2, y2, y2p = (t[k] for k in ('s1', 'st1', 'y1', 'pre', 'h', 's2', 'st2', 'y2', 'y2p'))
            ctx.tail = True
        else:
            y1, s1, st1, pre, h, s2, st2, y2, y2p = FusedEncoderLayerFn._tail_by_products(
                o, x, w_out, b_out, w1, b1, w2, b2, n1w, n1b, n2w, n2b, eps, act, need_bwd, pos_next, c)
        if need_bwd:
            ctx.save_for_backward(x, xp, qk, v, o, lse, s1, st1, y1, pre, h, s2, st2, w_in, w_out, w1, w2, n1w, n2w,
                                  *((wpack,) if ctx.tail else ()), *(pos_spec if ctx.pos_fold else ()),
                                  *((head_scale,) if ctx.cosine else ()))
            ctx.plan, ctx.nhead, ctx.impl, ctx.act, ctx.scale = plan, nhead, impl, act, scale
        ctx.two = pos_next is not None
        return (y2, y2p) if ctx.two else y2

    @staticmethod
    def _tail_batch_norm(o, x, w_out, b_out, w1, b1, w2, b2, bn, act, pos_next):
        """out-projection + residual -> bn1 -> linear1 + activation -> linear2 + residual -> bn2: the residual sums leave the
        projections' epilogues, each batch norm is its statistics pass + its apply pass (training mode needs the whole batch
        before a value can be normalised; naiveSyncBN's collective sits between the two).  st1 / st2 of the saved list = the
        [4, C] statistic rows."""
        if _LDS_LINEAR and lds_linear_ok(o, w_out) and x.is_contiguous():
            s1 = lds_linear(o, w_out, b_out, EPI_ADD, aux_in=x)
        else:
            s1 = torch.addmm(b_out, o, w_out.t()).add_(x)
        y1, prep1, cfg1 = _bn_rows_fwd(bn[0], s1)
        if _LDS_LINEAR and lds_linear_ok(y1, w1) and act in ('gelu', 'relu'):
            h, pre = lds_linear(y1, w1, b1, EPI_GELU if act == 'gelu' else EPI_RELU, want_pre=True)
        else:
            pre = torch.addmm(b1, y1, w1.t())
            h = F.gelu(pre) if act == 'gelu' else F.relu(pre)
        if _LDS_LINEAR and lds_linear_ok(h, w2):
            s2 = lds_linear(h, w2, b2, EPI_ADD, aux_in=y1)
        else:
            s2 = torch.addmm(b2, h, w2.t()).add_(y1)
        y2, prep2, cfg2 = _bn_rows_fwd(bn[1], s2)
        y2p = y2 + pos_next[0].index_select(0, pos_next[1].long()) if pos_next is not None else None
        return y1, s1, prep1, pre, h, s2, prep2, y2, y2p, (cfg1, cfg2)

    @staticmethod
    def _tail_by_products(o, x, w_out, b_out, w1, b1, w2, b2, n1w, n1b, n2w, n2b, eps, act, need_bwd, pos_next, c):
        """out-projection -> norm1 -> feed-forward -> norm2, one launch per product / LayerNorm (the modes and shapes the
        one-kernel tail is not built for)"""
        fuse_ln = _LDS_LINEAR and lds_linear_add_ln_ok(o, w_out, x, c)
        if fuse_ln:    # out-projection + residual + LayerNorm in one kernel (csrc/dense_f32.hip)
            y1, s1, st1, _ = lds_linear_add_ln(o, w_out, b_out, x, n1w, n1b, eps, save_sum=need_bwd)
        else:
            a = _linear_fwd(o, w_out, b_out)
            y1, s1, st1 = add_ln_fwd(x, a, n1w, n1b, eps, save_sum=need_bwd)
        fused_act = linear_gelu(y1, w1, b1) if (act == 'gelu' and _FUSED_GELU) else None
        if _LDS_LINEAR and lds_linear_ok(y1, w1):   # bias + activation in the epilogue of linear1 (csrc/dense_f32.hip)
            h, pre = lds_linear(y1, w1, b1, EPI_GELU if act == 'gelu' else EPI_RELU, want_pre=True)
        elif fused_act is not None:    # csrc/tall_gemm.hip
            pre, h = fused_act
        else:
            pre = torch.addmm(b1, y1, w1.t())
            h = F.gelu(pre) if act == 'gelu' else F.relu(pre)
        y2p = None
        if fuse_ln and lds_linear_add_ln_ok(h, w2, y1, c):
            y2, s2, st2, y2p = lds_linear_add_ln(h, w2, b2, y1, n2w, n2b, eps, save_sum=need_bwd, pos=pos_next)
        elif _LDS_LINEAR and lds_linear_ok(h, w2):
            # residual in the projection's epilogue, LayerNorm as its own pass over the sum (f32x6 at K = 256: see dense.py)
            s2 = lds_linear(h, w2, b2, EPI_ADD, aux_in=y1)
            y2, _, st2 = add_ln_fwd(s2, None, n2w, n2b, eps)
        else:
            f = _linear_fwd(h, w2, b2)
            y2, s2, st2 = add_ln_fwd(y1, f, n2w, n2b, eps, save_sum=need_bwd)
        if pos_next is not None and y2p is None:
            y2p = y2 + pos_next[0].index_select(0, pos_next[1].long())
        return y1, s1, st1, pre, h, s2, st2, y2, y2p

    @staticmethod
    def backward(ctx, dy2, dy2p=None):
        from . import dense as _dense
        with _dense.matmul_mode_scope(ctx.matmul):
            return FusedEncoderLayerFn._backward(ctx, dy2, dy2p)

    @staticmethod
    def _backward(ctx, dy2, dy2p=None):
        if ctx.exec:
            if dy2 is None:       # only the second output was differentiated
                dy2, dy2p = dy2p, None
            return _layer_exec_bwd(ctx, dy2, dy2p if ctx.two else None, ctx.saved_tensors)
        saved = ctx.saved_tensors     # ONE access (a second one under torch.utils.checkpoint is an error)
        x, xp, qk, v, o, lse, s1, st1, y1, pre, h, s2, st2, w_in, w_out, w1, w2, n1w, n2w = saved[:19]
        rest = list(saved[19:])
        wpack = rest.pop(0) if ctx.tail else None
        pos_spec = (rest.pop(0), rest.pop(0)) if ctx.pos_fold else None
        head_scale = rest.pop(0) if ctx.cosine else None
        c = x.size(1)
        if dy2 is None:           # only the second output was differentiated
            dy2, dy2p = dy2p, None
        f32 = dict(dtype=torch.float32, device=x.device)
        dw2, db2 = torch.empty_like(w2), torch.empty(w2.size(0), **f32)
        dw1, db1 = torch.empty_like(w1), torch.empty(w1.size(0), **f32)
        if ctx.tail:
            # norm2' -> linear2' * act' -> linear1' + residual -> norm1' -> out-projection' as ONE kernel (csrc/layer_tail_x6.hip)
            ds2, dpre, ds1, do, dn = encoder_tail_bwd(dy2.contiguous(), dy2p.contiguous() if (ctx.two and dy2p is not None) else None,
                                                      s2, st2, pre, s1, st1, wpack, n1w, n2w, ctx.act)
            ds2_for_w2 = ds2
            dn2w, dn2b, dn1w, dn1b = dn[0], dn[1], dn[2], dn[3]
        else:
            if ctx.bn:
                if ctx.two and dy2p is not None:
                    dy2 = dy2 + dy2p
                ds2, dn2w, dn2b = _bn_rows_bwd(dy2, s2, st2, ctx.bn_cfg[1])
            else:
                ds2, dn2w, dn2b = add_ln_bwd(dy2, s2, st2, n2w, dy2=dy2p if ctx.two else None)   # = d(y1 residual) = d(f)
            ds2_for_w2 = ds2
            dpre = dgrad_gelu(ds2, w2, pre) if (ctx.act == 'gelu' and _FUSED_GELU) else None
            if _LDS_LINEAR and lds_linear_ok(ds2, w2, trans_w=True) and pre.is_contiguous():
                # the activation's derivative in the epilogue of linear2's data gradient
                dpre = lds_linear(ds2, w2, None, EPI_MUL_GELU_GRAD if ctx.act == 'gelu' else EPI_MUL_RELU_GRAD, trans_w=True,
                                  aux_in=pre)
            if dpre is None:
                dh = ds2 @ w2
                if ctx.act == 'gelu':
                    dpre = torch.ops.aten.gelu_backward(dh, pre)
                else:
                    dpre = dh * (pre > 0).to(dh.dtype)
            # residual + FFN branch: GEMM with beta = 1 into a buffer of its OWN - ds2 stays what dW2 needs, and all five parameter
            # gradients of the layer leave in one grouped launch at the end
            dy1 = _linear_dgrad(dpre, w1, out=torch.empty_like(ds2), add=ds2)
            if ctx.bn:
                ds1, dn1w, dn1b = _bn_rows_bwd(dy1, s1, st1, ctx.bn_cfg[0])
            else:
                ds1, dn1w, dn1b = add_ln_bwd(dy1, s1, st1, n1w)               # = d(x residual) = d(attention output)
            do = _linear_dgrad(ds1, w_out)
        # dq | dk | dv in ONE [M, 3C] buffer: d(x) of the whole in-projection is then a single GEMM
        dqkv = torch.empty((x.size(0), 3 * c), dtype=torch.float32, device=x.device)
        dqk, dv = dqkv[:, :2 * c], dqkv[:, 2 * c:]
        d_scale = None
        if ctx.cosine:
            r = K._sra_cos_bwd(qk[:, :c], qk[:, c:], v, o, lse, do, ctx.plan, ctx.nhead, head_scale, dqkv[:, :c],
                               dqkv[:, c:2 * c], dv)
            d_scale = K.head_scale_grad(r, head_scale)
        else:
            K._sra_bwd(qk[:, :c], qk[:, c:], v, o, lse, do, ctx.plan, ctx.nhead, ctx.scale, ctx.impl, dqkv[:, :c],
                       dqkv[:, c:2 * c], dv)
        dw_in = torch.empty_like(w_in)
        db_in = torch.empty(3 * c, **f32)
        dwo, dbo = torch.empty_like(w_out), torch.empty(w_out.size(0), **f32)
        # all five, before ds1 is accumulated into in place
        weight_bias_grad_group([(ds2_for_w2, h, dw2, db2), (dpre, y1, dw1, db1), (ds1, o, dwo, dbo),
                                (dqk, x, dw_in[:2 * c], db_in[:2 * c], pos_spec) if ctx.pos_fold else
                                (dqk, xp, dw_in[:2 * c], db_in[:2 * c]), (dv, x, dw_in[2 * c:], db_in[2 * c:])])
        dxp = None
        if ctx.fold_xp and _LDS_LINEAR and c == 128 and lds_linear_dqkv_ok(dqkv, w_in):
            # xp = x + pos with a constant pos: d(x) += d(xp), so the residual branch and all three projections leave as ONE
            # product over K = 384 with the residual in the epilogue; the gradient of xp is folded in (None): the producer's
            # LayerNorm backward then reads one upstream gradient instead of two
            dx = lds_linear(dqkv, w_in, None, EPI_ADD, trans_w=True, aux_in=ds1, out=ds1)
        elif ctx.split_input:      # x and xp are separate inputs: their gradients leave separately
            dxp = _linear_dgrad(dqk, w_in[:2 * c])
            dx = _linear_dgrad(dv, w_in[2 * c:], out=ds1)
        elif _LDS_LINEAR and lds_linear_ok(dqk, w_in[:2 * c], trans_w=True) and lds_linear_ok(dv, w_in[2 * c:], trans_w=True):
            dx = _linear_dgrad(dv, w_in[2 * c:], out=_linear_dgrad(dqk, w_in[:2 * c], out=ds1))
        else:
            dx = ds1.addmm_(dqkv, w_in)                               # residual + q,k,v branches (in place)
        return (dx, None, None, None, None, None, dw_in, db_in, dwo, dbo, dw1, db1, dw2, db2, dn1w, dn1b, dn2w, dn2b,
                None, dxp, None, d_scale, None, None, None, None)


class _StackHeadScales(torch.autograd.Function):
    """1 / clamp(tau, tau_min) of EVERY cosine layer of a stack in one pass (cosine_msa.py:162-168): -> [L, nhead], row l = the
    per-head score scale of layer l (a shared tau expanded).  Three small launches for the whole stack instead of three per
    layer, and the backward pass - d tau = -d scale * scale^2 where tau >= tau_min, summed over the heads for a shared tau -
    once for all layers: the per-layer version was ~100 tiny launches per training step."""

    @staticmethod
    def forward(ctx, nhead, tau_mins, *taus):
        flat = torch.cat([t.reshape(-1).expand(nhead) if t.numel() == 1 else t.reshape(-1) for t in taus]).view(len(taus), nhead)
        mins = K.const_tensor(tau_mins, flat.device).view(-1, 1)
        scales = 1.0 / torch.maximum(flat.float(), mins)
        ctx.save_for_backward(flat, scales, mins)
        ctx.shapes = [tuple(t.shape) for t in taus]
        return scales

    @staticmethod
    def backward(ctx, g):
        flat, scales, mins = ctx.saved_tensors
        d = torch.where(flat >= mins, -(g * scales * scales), torch.zeros((), dtype=g.dtype, device=g.device))   # [L, H]
        per_layer_sum = None
        out = []
        for l, shape in enumerate(ctx.shapes):
            n = 1
            for v in shape:
                n *= v
            if n == 1:
                if per_layer_sum is None:
                    per_layer_sum = d.sum(1)
                out.append(per_layer_sum[l].reshape(shape))
            else:
                out.append(d[l].reshape(shape))
        return (None, None) + tuple(out)


def stack_head_scales(layers):
    """per-layer head scales of a stack: list aligned with ``layers`` (None for standard attention); cosine layers get rows of
    ONE [L, nhead] tensor (row views: contiguous, 32-byte aligned for 8 heads)"""
    cos = [(i, enc.win_attn) for i, enc in enumerate(layers) if enc.win_attn.cosine]
    out = [None] * len(layers)
    if not cos:
        return out
    heads = {wa.nhead for _, wa in cos}
    if len(heads) != 1:              # mixed widths: every layer on its own
        for i, wa in cos:
            out[i] = wa.head_scale()
        return out
    nhead = heads.pop()
    rows = _StackHeadScales.apply(nhead, tuple(float(wa.self_attn.tau_min) for _, wa in cos),
                                  *[wa.self_attn.tau for _, wa in cos]).unbind(0)
    for (i, _), row in zip(cos, rows):
        out[i] = row
    return out


def stack_tail_images(layers, like):
    """The tail weight images (csrc/layer_tail_x6.hip) of every LayerNorm layer of a stack in ONE launch
    (sst_encoder_tail_pack_f32x6_many) - a list aligned with ``layers``: a byte tensor per layer the one-kernel tail can serve,
    None for the others (batch-norm layers, other widths).  Fresh memory per forward pass (the images are kept for the backward
    pass of THIS forward; ~1 MB per layer)."""
    import ctypes
    out = [None] * len(layers)
    take = []
    for i, enc in enumerate(layers):
        attn = enc.win_attn.self_attn
        wo, w1, w2 = attn.out_proj.weight, enc.linear1.weight, enc.linear2.weight
        if (enc.bn_modules() is None and wo.is_cuda and wo.shape == (128, 128) and w1.shape == (256, 128) and w2.shape == (128, 256)
                and all(w.dtype == torch.float32 and w.is_contiguous() and w.data_ptr() % 16 == 0 for w in (wo, w1, w2))
                and enc.act_name in ('gelu', 'relu') and all(b is not None for b in (attn.out_proj.bias, enc.linear1.bias,
                                                                                     enc.linear2.bias))):
            take.append((i, wo, w1, w2))
    if not take or not like.is_cuda:
        return out
    lib = _lib.load()
    nbytes = (int(lib.sst_encoder_tail_pack_bytes()) + 255) // 256 * 256
    buf = torch.empty(len(take) * nbytes, dtype=torch.uint8, device=like.device)
    n = len(take)
    P = ctypes.c_void_p * n
    d = [buf[j * nbytes:(j + 1) * nbytes] for j in range(n)]
    with torch.no_grad():
        _lib.check(lib.sst_encoder_tail_pack_f32x6_many(P(*[t[1].data_ptr() for t in take]), P(*[t[2].data_ptr() for t in take]),
                                                        P(*[t[3].data_ptr() for t in take]), P(*[v.data_ptr() for v in d]), n,
                                                        _lib.stream_ptr()), 'sst_encoder_tail_pack_f32x6_many')
    for (i, _, _, _), v in zip(take, d):
        out[i] = v
    return out


def run_encoder_stack_fp32(blocks, feats, plans, pos_specs, checkpoint_blocks=()):
    """All encoder layers of the shift blocks as a chain of FusedEncoderLayerFn nodes that hand (x, x + positional embedding)
    to each other: "+ positional embedding" of layer i + 1 is the second output of layer i's last kernel, so no add pass and
    no [M, C] positional tensor exist after the first layer.  pos_specs: per partition (table fp32 [P, C], row index int32).
    checkpoint_blocks: indices of the shift blocks whose two layers keep no activations and are recomputed in the backward pass
    (the reference's torch.utils.checkpoint per block, sst_v2.py:131-133 / sst_basic_block_v2.py:164-165)."""
    n_layers = 2 * len(blocks)

    scales = stack_head_scales([enc for block in blocks for enc in block.encoder_list])

    from . import dense as _dense
    mode = _dense.matmul_mode()
    # exact-split mode: every layer adds its positional rows to x on load (pos_spec) - no "x + pos" tensor between the layers;
    # the other modes keep the round-5 chain (x + pos of layer i + 1 is the second output of layer i's last kernel)
    fold = _POS_FOLD and mode == 'f32x6' and _LDS_LINEAR and all(inproj_pos_ok(feats.contiguous(), spec, blocks[0].encoder_list[0].win_attn.
                                                                 self_attn.in_proj_weight) for spec in pos_specs[:2])

    layers = [enc for block in blocks for enc in block.encoder_list]
    wpacks = stack_tail_images(layers, feats) if (mode == 'f32x6' and _LDS_LINEAR) else [None] * n_layers

    def layer(enc, li, x, xp):
        attn = enc.win_attn.self_attn
        if fold:
            out = FusedEncoderLayerFn.apply(
                x, None, plans[li % 2], enc.win_attn.nhead, enc.win_attn.impl, enc.act_name, attn.in_proj_weight,
                attn.in_proj_bias, attn.out_proj.weight, attn.out_proj.bias, enc.linear1.weight, enc.linear1.bias,
                enc.linear2.weight, enc.linear2.bias, enc.norm1.weight, enc.norm1.bias, enc.norm2.weight, enc.norm2.bias,
                enc.norm1.eps, None, None, scales[li], False, pos_specs[li % 2], enc.bn_modules(), wpacks[li])
            return out, None
        pos_next = pos_specs[(li + 1) % 2] if li + 1 < n_layers else None
        out = FusedEncoderLayerFn.apply(
            x, None, plans[li % 2], enc.win_attn.nhead, enc.win_attn.impl, enc.act_name, attn.in_proj_weight,
            attn.in_proj_bias, attn.out_proj.weight, attn.out_proj.bias, enc.linear1.weight, enc.linear1.bias,
            enc.linear2.weight, enc.linear2.bias, enc.norm1.weight, enc.norm1.bias, enc.norm2.weight, enc.norm2.bias,
            enc.norm1.eps, xp, pos_next, scales[li], True, None, enc.bn_modules(), wpacks[li])   # xp = x + positional rows: a constant offset
        return out if pos_next is not None else (out, None)

    def block_fn(bi):
        def run(x, xp):
            with _dense.matmul_mode_scope(mode):     # also when a checkpointed block is recomputed during the backward pass
                for j, enc in enumerate(blocks[bi].encoder_list):
                    x, xp = layer(enc, 2 * bi + j, x, xp)
            return (x, xp) if xp is not None else (x,)
        return run

    x = feats.contiguous()
    # x + positional rows of the first layer: one pass (csrc/scatter.hip add_table_rows_k) - unless every layer adds them on load
    xp = None if fold else K.add_table_rows(x, pos_specs[0][0], pos_specs[0][1])
    for bi in range(len(blocks)):
        if bi in checkpoint_blocks and torch.is_grad_enabled():
            out = checkpoint(block_fn(bi), x, xp, use_reentrant=False)
        else:
            out = block_fn(bi)(x, xp)
        x, xp = (out[0], out[1]) if len(out) > 1 else (out[0], None)
    return x


class EncoderLayer(nn.Module):

    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation="relu", batch_first=False,
                 layer_id=None, mlp_dropout=0, layer_cfg=dict()):
        super().__init__()
        assert not batch_first
        self.batch_first = batch_first
        self.win_attn = WindowAttention(d_model, nhead, dropout, layer_id=layer_id, layer_cfg=layer_cfg)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(mlp_dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        use_bn = layer_cfg.get('use_bn', False)
        if use_bn:
            self.norm1 = build_norm_layer(dict(type='naiveSyncBN1d', momentum=layer_cfg.get('mom', 0.1)), d_model)[1]
            self.norm2 = build_norm_layer(dict(type='naiveSyncBN1d', momentum=layer_cfg.get('mom', 0.1)), d_model)[1]
        else:
            self.norm1 = nn.LayerNorm(d_model)
            self.norm2 = nn.LayerNorm(d_model)
        self.dropout1 = nn.Dropout(mlp_dropout)
        self.dropout2 = nn.Dropout(mlp_dropout)
        self.activation = _get_activation_fn(activation)
        self.act_name = activation
        self.fused = True   # single-node forward/backward (FusedEncoderLayerFn) when the configuration allows
        self.post_norm = layer_cfg.get('post_norm', True)
        self.fp16_enabled = False

    def bn_modules(self):
        """(norm1, norm2) when they are batch-norm modules (layer_cfg use_bn=True), else None: the argument FusedEncoderLayerFn
        takes for its batch-norm tail"""
        return (self.norm1, self.norm2) if isinstance(self.norm1, nn.BatchNorm1d) else None

    def _ffn(self, x):
        h = self.activation(tall_linear(x, self.linear1.weight, self.linear1.bias))
        return tall_linear(self.dropout(h), self.linear2.weight, self.linear2.bias)

    def _can_fuse(self, src, pos_dict, ind_dict):
        wa = self.win_attn
        return (self.fused and self.post_norm and isinstance(ind_dict, K.WindowPlan)
                and ind_dict.n_tokens == src.size(0)
                and (pos_dict is None or torch.is_tensor(pos_dict)) and wa.head_dim == 16
                and (not wa.cosine or K.cosine_kernels_ok(ind_dict, wa.nhead, wa.impl))
                and ((isinstance(self.norm1, nn.LayerNorm) and isinstance(self.norm2, nn.LayerNorm))
                     or (_bn_layer_ok(self.norm1, src.size(1)) and _bn_layer_ok(self.norm2, src.size(1))))
                and self.act_name in ('gelu', 'relu') and src.dtype == torch.float32 and src.is_cuda
                and src.size(1) % 32 == 0 and self.linear1.out_features % 32 == 0
                and not (self.training and (wa.attn_dropout > 0 or self.dropout.p > 0 or self.dropout1.p > 0)))

    def forward(self, src, pos_dict, ind_dict, key_padding_mask_dict=None):
        if self._can_fuse(src, pos_dict, ind_dict):
            attn = self.win_attn.self_attn
            return FusedEncoderLayerFn.apply(
                src, pos_dict, ind_dict, self.win_attn.nhead, self.win_attn.impl, self.act_name, attn.in_proj_weight,
                attn.in_proj_bias, attn.out_proj.weight, attn.out_proj.bias, self.linear1.weight, self.linear1.bias,
                self.linear2.weight, self.linear2.bias, self.norm1.weight, self.norm1.bias, self.norm2.weight,
                self.norm2.bias, self.norm1.eps, None, None, self.win_attn.head_scale(), False, None, self.bn_modules())
        if self.post_norm:
            src2 = self.win_attn(src, pos_dict, ind_dict, key_padding_mask_dict)  # [N, d_model]
            src = add_layer_norm(src, self.dropout1(src2), self.norm1)     # norm1(src + src2), one kernel
            src2 = self._ffn(src)
            src = add_layer_norm(src, self.dropout2(src2), self.norm2)
        else:
            src2 = add_layer_norm(src, None, self.norm1)
            src2 = self.win_attn(src2, pos_dict, ind_dict, key_padding_mask_dict)
            src = src + self.dropout1(src2)
            src2 = add_layer_norm(src, None, self.norm2)
            src2 = self._ffn(src2)
            src = src + self.dropout2(src2)
        return src


class BasicShiftBlockV2(nn.Module):
    '''Two encoder layers: regular windows, then shifted windows.'''

    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation="relu", batch_first=False,
                 block_id=-100, layer_cfg=dict()):
        super().__init__()
        encoder_1 = EncoderLayer(d_model, nhead, dim_feedforward, dropout, activation, batch_first,
                                 layer_id=block_id * 2 + 0, layer_cfg=layer_cfg)
        encoder_2 = EncoderLayer(d_model, nhead, dim_feedforward, dropout, activation, batch_first,
                                 layer_id=block_id * 2 + 1, layer_cfg=layer_cfg)
        self.encoder_list = nn.ModuleList([encoder_1, encoder_2])

    def forward(self, src, pos_dict_list, ind_dict_list, key_mask_dict_list=None, using_checkpoint=False):
        num_shifts = len(pos_dict_list)
        assert num_shifts in (1, 2)
        output = src
        for i in range(2):
            this_id = i % num_shifts
            pos_dict = pos_dict_list[this_id]
            ind_dict = ind_dict_list[this_id]
            key_mask_dict = key_mask_dict_list[this_id] if key_mask_dict_list is not None else None
            layer = self.encoder_list[i]
            if using_checkpoint and self.training:
                output = checkpoint(layer, output, pos_dict, ind_dict, key_mask_dict, use_reentrant=False)
            else:
                output = layer(output, pos_dict, ind_dict, key_mask_dict)
        return output


def _get_activation_fn(activation):
    if activation == "relu":
        return F.relu
    if activation == "gelu":
        return F.gelu
    if activation == "glu":
        return F.glu
    raise RuntimeError(F"activation should be relu/gelu, not {activation}.")
