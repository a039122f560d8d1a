"""Attention core over a window plan COMPOSED of library operations (padded windows, batched products) - for the two
configurations the SRA kernels are not built for: attention-weight dropout > 0 in training and head_dim != 16
(nn.MultiheadAttention accepts both: sst_basic_block_v2.py:35; no SST / FSD config uses either).  Same semantics as the
reference's flat2window -> nn.MultiheadAttention core -> window2flat (sst_basic_block_v2.py:41-75) with the padded keys
masked; the dropout acts on the attention weights, as in torch.nn.functional.multi_head_attention_forward.  Runs on the
GPU; it is the slow path (it materialises [W, T, ...] tensors) and says so."""
import torch
import torch.nn.functional as F


def padded_index(plan, m):
    """-> (rows [W, T] int64: token row per window slot, -1 = padding).  One host read of nothing: W and T come from the plan."""
    w, t = plan.n_windows, plan.max_tokens
    off = plan.winoff[:w + 1].long()
    tok = plan.tok[:plan.n_tokens].long()
    win_of = torch.repeat_interleave(torch.arange(w, device=tok.device), off[1:] - off[:-1], output_size=tok.numel())
    slot = torch.arange(tok.numel(), device=tok.device) - off[win_of]
    rows = torch.full((w, t), -1, dtype=torch.long, device=tok.device)
    rows[win_of, slot] = tok
    return rows


def sra_attention_composed(q, k, v, plan, n_heads, scale, dropout_p=0.0, training=False):
    """softmax(q k^T * scale) [dropout] v inside each window; q, k, v [M, C] fp32 -> [M, C]; differentiable (autograd)"""
    m, c = q.shape
    d = c // n_heads
    rows = padded_index(plan, m)                       # [W, T]
    w, t = rows.shape
    pad = rows < 0
    safe = rows.clamp(min=0).reshape(-1)
    qp, kp, vp = (x[safe].reshape(w, t, n_heads, d).permute(0, 2, 1, 3) for x in (q, k, v))     # [W, H, T, D]
    s = torch.matmul(qp, kp.transpose(-1, -2)) * scale                                         # [W, H, T, T]
    s = s.masked_fill(pad[:, None, None, :], float('-inf'))
    p = torch.softmax(s, dim=-1)
    if dropout_p > 0 and training:
        p = F.dropout(p, dropout_p, True)
    o = torch.matmul(p, vp).permute(0, 2, 1, 3).reshape(w * t, c)                              # padded query rows dropped next
    out = q.new_zeros((m, c))
    keep = (~pad).reshape(-1)
    out.index_copy_(0, safe[keep], o[keep])
    return out
