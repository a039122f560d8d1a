"""TEST INFRASTRUCTURE ONLY — CPU restatement (numpy for the integer work, float64 torch/numpy for the
floating-point work) of the SST hot path downstream of the voxel encoder.

Each function cites the reference lines it follows (paths relative to the reference tree):
  window_coors        mmdet3d/ops/sst/sst_ops.py:266-314 (get_window_coors)
  ingroup_rank        contract of TorchEx ingroup_indices, call site sst_ops.py:244-264; order fixed to
                      "ascending element index" (the reference's own fallback sst_ops.py:194-242 allows any)
  drop_single_shift   mmdet3d/models/middle_encoders/sst_input_layer_v2.py:128-148
  region_batching     sst_input_layer_v2.py:150-226 (drop_voxel) + sst_ops.py:26-64 (get_flat2win_inds),
                      :316-331 (make_continuous_inds)
  pos_embed           sst_input_layer_v2.py:238-305
  sra_core            the attention inside nn.MultiheadAttention as called by
                      mmdet3d/models/sst/sst_basic_block_v2.py:41-75 (scores / sqrt(16), masked softmax, bmm)
  encoder_layer       sst_basic_block_v2.py:104-126 (post-norm / pre-norm EncoderLayer)
  sir_layer           mmdet3d/models/voxel_encoders/voxel_encoder.py:696-764
"""
import math

import numpy as np


# ------------------------------------------------------------------------------------------------
# integer work
# ------------------------------------------------------------------------------------------------
def window_coors(coors, sparse_shape, window_shape, do_shift):
    """coors [M,4] (b,z,y,x) int -> (batch_win_inds [M] int64, coors_in_win [M,3] int64 (z,y,x))."""
    coors = np.asarray(coors, dtype=np.int64)
    if len(window_shape) == 2:
        wx, wy = window_shape
        wz = sparse_shape[-1]
    else:
        wx, wy, wz = window_shape
    sx, sy, sz = sparse_shape
    nwx = int(np.ceil(sx / wx) + 1)
    nwy = int(np.ceil(sy / wy) + 1)
    nwz = int(np.ceil(sz / wz) + 1)
    per_sample = nwx * nwy * nwz
    if do_shift:
        shx, shy, shz = wx // 2, wy // 2, wz // 2
    else:
        shx, shy, shz = wx, wy, wz
    if sz == wz:
        shz = 0
    xs = coors[:, 3] + shx
    ys = coors[:, 2] + shy
    zs = coors[:, 1] + shz
    wxi, wyi, wzi = xs // wx, ys // wy, zs // wz
    win = coors[:, 0] * per_sample + wxi * nwy * nwz + wyi * nwz + wzi
    ciw = np.stack([zs % wz, ys % wy, xs % wx], axis=-1)
    return win, ciw


def ingroup_rank(ids):
    """rank of each element among the elements with the same id, ascending element index."""
    ids = np.asarray(ids, dtype=np.int64)
    n = ids.shape[0]
    if n == 0:
        return np.zeros(0, dtype=np.int64)
    order = np.argsort(ids, kind='stable')
    s = ids[order]
    head = np.ones(n, dtype=bool)
    head[1:] = s[1:] != s[:-1]
    start = np.maximum.accumulate(np.where(head, np.arange(n), 0))
    rank = np.empty(n, dtype=np.int64)
    rank[order] = np.arange(n) - start
    return rank


def drop_single_shift(win, drop_info):
    """-> (keep_mask, drop_level_per_voxel).  drop_info: {level: {'max_tokens': T, 'drop_range': (lo, hi)}}."""
    win = np.asarray(win, dtype=np.int64)
    inner = ingroup_rank(win)
    counts = np.bincount(win) if win.size else np.zeros(0, dtype=np.int64)
    n_per_voxel = counts[win]
    target = np.zeros_like(win)
    level = -np.ones_like(win)
    for dl in drop_info:
        lo, hi = drop_info[dl]['drop_range']
        m = (n_per_voxel >= lo) & (n_per_voxel < hi)
        target[m] = drop_info[dl]['max_tokens']
        level[m] = dl
    return inner < target, level


def make_continuous(ids):
    uniq, inv = np.unique(ids, return_inverse=True)
    return inv.astype(np.int64)


def region_batching(win0, win1, drop_info):
    """drop_voxel for two shifts + flat2win indices + the window CSR used by the SRA kernels.

    Returns a dict with (all for the SURVIVING voxels, in their original relative order):
      keep_idx, win{0,1}, level{0,1}, flat2win{0,1} (per voxel), inner{0,1},
      tok{0,1} / winoff{0,1}: tokens grouped by ascending window id, ascending inner index.
    """
    win0 = np.asarray(win0, dtype=np.int64)
    win1 = np.asarray(win1, dtype=np.int64)
    m = win0.shape[0]
    keep_idx = np.arange(m)
    k0, l0 = drop_single_shift(win0, drop_info)
    l0, keep_idx, w0, w1 = l0[k0], keep_idx[k0], win0[k0], win1[k0]
    k1, l1 = drop_single_shift(w1, drop_info)
    # levels of shift 0 are NOT recomputed after the second filter (sst_input_layer_v2.py:186-194)
    l0, keep_idx, w0 = l0[k1], keep_idx[k1], w0[k1]
    l1, w1 = l1[k1], w1[k1]
    out = dict(keep_idx=keep_idx, win0=w0, win1=w1, level0=l0, level1=l1)
    for s, (w, lv) in enumerate(((w0, l0), (w1, l1))):
        f2w = -np.ones_like(w)
        inner_all = -np.ones_like(w)
        for dl in drop_info:
            mask = lv == dl
            if not mask.any():
                continue
            cw = make_continuous(w[mask])
            inner = ingroup_rank(cw)
            f2w[mask] = cw * drop_info[dl]['max_tokens'] + inner
            inner_all[mask] = inner
        out[f'flat2win{s}'] = f2w
        out[f'inner{s}'] = inner_all
        order = np.lexsort((inner_all, w))
        uniq, counts = np.unique(w, return_counts=True)
        out[f'tok{s}'] = order.astype(np.int64)
        out[f'winoff{s}'] = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    return out


# ------------------------------------------------------------------------------------------------
# floating-point work (float64 accumulate unless dtype is given)
# ------------------------------------------------------------------------------------------------
def pos_embed(coors_in_win, window_shape, feat_dim, pos_temperature=10000, normalize_pos=False, dtype=np.float32):
    """[M,3] (z,y,x) in-window coordinates -> [M, feat_dim] sin/cos embedding; float32 arithmetic like torch."""
    ciw = np.asarray(coors_in_win)
    if len(window_shape) == 2:
        ndim, (wx, wy), wz = 2, window_shape, 0
    elif window_shape[-1] == 1:
        ndim, (wx, wy), wz = 2, window_shape[:2], 0
    else:
        (wx, wy, wz), ndim = window_shape, 3
    z = ciw[:, 0].astype(np.float32) - np.float32(wz / 2)
    y = ciw[:, 1].astype(np.float32) - np.float32(wy / 2)
    x = ciw[:, 2].astype(np.float32) - np.float32(wx / 2)
    if normalize_pos:
        x = x / wx * 2 * 3.1415
        y = y / wy * 2 * 3.1415
        z = z / wz * 2 * 3.1415
    pos_length = feat_dim // ndim
    i = np.arange(pos_length, dtype=np.float32)
    inv_freq = np.power(np.float32(pos_temperature), (2 * np.floor(i / 2) / pos_length).astype(np.float32))
    inv_freq = inv_freq.astype(np.float32)

    def emb(a):
        e = a.astype(np.float32)[:, None] / inv_freq[None, :]
        out = np.empty_like(e)
        out[:, 0::2] = np.sin(e[:, 0::2])
        out[:, 1::2] = np.cos(e[:, 1::2])
        return out

    parts = [emb(x), emb(y)] + ([emb(z)] if ndim == 3 else [])
    pos = np.concatenate(parts, axis=-1)
    gap = feat_dim - pos.shape[1]
    if gap > 0:
        pos = np.concatenate([pos, np.zeros((pos.shape[0], gap), dtype=pos.dtype)], axis=1)
    return pos.astype(dtype)


def _softmax(s):
    s = s - s.max(axis=-1, keepdims=True)
    e = np.exp(s)
    return e / e.sum(axis=-1, keepdims=True)


def sra_core(q, k, v, tok, winoff, n_heads, scale=None, return_lse=False):
    """Per window, per head softmax(q k^T * scale) v, float64.  q,k,v: [M, n_heads*hd]."""
    q = np.asarray(q, dtype=np.float64)
    k = np.asarray(k, dtype=np.float64)
    v = np.asarray(v, dtype=np.float64)
    m, c = q.shape
    hd = c // n_heads
    if scale is None:
        scale = 1.0 / math.sqrt(hd)
    o = np.zeros_like(q)
    lse = np.zeros((m, n_heads))
    for w in range(len(winoff) - 1):
        idx = np.asarray(tok[winoff[w]:winoff[w + 1]], dtype=np.int64)
        for h in range(n_heads):
            sl = slice(h * hd, (h + 1) * hd)
            s = (q[idx, sl] * scale) @ k[idx, sl].T
            mx = s.max(axis=-1, keepdims=True)
            e = np.exp(s - mx)
            den = e.sum(axis=-1, keepdims=True)
            o[np.ix_(idx, np.arange(h * hd, (h + 1) * hd))] = (e / den) @ v[idx, sl]
            lse[idx, h] = (mx + np.log(den))[:, 0]
    if return_lse:
        return o, lse
    return o


def sra_core_backward(q, k, v, do, tok, winoff, n_heads, scale=None):
    """Gradients (dq, dk, dv) of sum(o * do) for sra_core, float64."""
    q = np.asarray(q, dtype=np.float64)
    k = np.asarray(k, dtype=np.float64)
    v = np.asarray(v, dtype=np.float64)
    do = np.asarray(do, dtype=np.float64)
    m, c = q.shape
    hd = c // n_heads
    if scale is None:
        scale = 1.0 / math.sqrt(hd)
    dq, dk, dv = np.zeros_like(q), np.zeros_like(k), np.zeros_like(v)
    for w in range(len(winoff) - 1):
        idx = np.asarray(tok[winoff[w]:winoff[w + 1]], dtype=np.int64)
        for h in range(n_heads):
            cols = np.arange(h * hd, (h + 1) * hd)
            qi, ki, vi, gi = q[np.ix_(idx, cols)], k[np.ix_(idx, cols)], v[np.ix_(idx, cols)], do[np.ix_(idx, cols)]
            p = _softmax(qi @ ki.T * scale)
            dv[np.ix_(idx, cols)] = p.T @ gi
            dp = gi @ vi.T
            ds = p * (dp - (dp * p).sum(axis=-1, keepdims=True)) * scale
            dq[np.ix_(idx, cols)] = ds @ ki
            dk[np.ix_(idx, cols)] = ds.T @ qi
    return dq, dk, dv


def _layer_norm(x, w, b, eps=1e-5):
    mu = x.mean(axis=-1, keepdims=True)
    var = ((x - mu) ** 2).mean(axis=-1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * w + b


def _gelu(x):
    from scipy.special import erf
    return 0.5 * x * (1.0 + erf(x / math.sqrt(2.0)))


def encoder_layer(x, pos, tok, winoff, params, n_heads, post_norm=True, activation='gelu'):
    """One SRA EncoderLayer in float64.  params: dict of numpy arrays with the reference's state_dict names
    relative to the layer: win_attn.self_attn.in_proj_weight / in_proj_bias / out_proj.weight / out_proj.bias,
    linear1.weight/bias, linear2.weight/bias, norm1.weight/bias, norm2.weight/bias."""
    P = {k_: np.asarray(v_, dtype=np.float64) for k_, v_ in params.items()}
    x = np.asarray(x, dtype=np.float64)
    c = x.shape[1]
    act = _gelu if activation == 'gelu' else (lambda t: np.maximum(t, 0))

    def attn(inp):
        xp = inp + (np.asarray(pos, dtype=np.float64) if pos is not None else 0.0)
        w, b = P['win_attn.self_attn.in_proj_weight'], P['win_attn.self_attn.in_proj_bias']
        q = xp @ w[:c].T + b[:c]
        k = xp @ w[c:2 * c].T + b[c:2 * c]
        v = inp @ w[2 * c:].T + b[2 * c:]
        o = sra_core(q, k, v, tok, winoff, n_heads)
        return o @ P['win_attn.self_attn.out_proj.weight'].T + P['win_attn.self_attn.out_proj.bias']

    def ffn(inp):
        return act(inp @ P['linear1.weight'].T + P['linear1.bias']) @ P['linear2.weight'].T + P['linear2.bias']

    if post_norm:
        x = _layer_norm(x + attn(x), P['norm1.weight'], P['norm1.bias'])
        x = _layer_norm(x + ffn(x), P['norm2.weight'], P['norm2.bias'])
    else:
        x = x + attn(_layer_norm(x, P['norm1.weight'], P['norm1.bias']))
        x = x + ffn(_layer_norm(x, P['norm2.weight'], P['norm2.bias']))
    return x


def segment_reduce(feats, group, n_groups, mode):
    """torch_scatter.scatter_max / scatter(mean|sum) contract: reduce rows of feats by group id (float64)."""
    feats = np.asarray(feats, dtype=np.float64)
    group = np.asarray(group, dtype=np.int64)
    c = feats.shape[1]
    if mode == 'max':
        out = np.full((n_groups, c), -np.inf)
        np.maximum.at(out, group, feats)
        return out
    out = np.zeros((n_groups, c))
    np.add.at(out, group, feats)
    if mode in ('mean', 'avg'):
        cnt = np.bincount(group, minlength=n_groups).astype(np.float64)
        out = out / np.maximum(cnt, 1)[:, None]
    return out


def unique_rows(coors):
    """torch.unique(coors, dim=0, return_inverse=True, return_counts=True): lexicographic order."""
    coors = np.asarray(coors)
    uniq, inv, cnt = np.unique(coors, axis=0, return_inverse=True, return_counts=True)
    return uniq, inv.reshape(-1).astype(np.int64), cnt.astype(np.int64)
