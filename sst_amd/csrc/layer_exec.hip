// One post-norm SRA encoder layer (models/sst/sst_basic_block_v2.py:41-126: WindowAttention -> norm1(x + .) -> FFN ->
// norm2(y1 + .)) as ONE call forward and ONE call backward: the launch sequence of sst_amd/sst_basic_block.py
// FusedEncoderLayerFn in its exact-split mode (d_model 128, feed-forward 256, head dim 16), issued from C.
// Round 6: everything behind the attention core is ONE kernel per direction (csrc/layer_tail_x6.hip) - forward 4 launches
// (in-projection, attention, weight images, tail), backward 5 (tail, attention, weight gradients + their reduction, in-projection's
// data gradient) instead of 6 and 10, and 56 instead of 69 [M, 128] tensors through HBM per layer.
//
// Why: nothing here computes - every step is an entry point of this library - but the Python side of that sequence is
// 6 + 9 foreign calls per layer with their argument marshalling, 2.6 + 2.7 ms of host time per training step on a fast host and
// 3.8 + 5.4 ms on a slow one, against 11.4 ms (fp32) / 6.1 ms (bf16 storage) of kernels: frames with fewer voxels than the bench
// frame (a real sweep keeps 20-50 k) were bound by the host.  One call per layer and direction leaves the allocations of
// the tensors autograd keeps, and nothing else, to the interpreter.  Same kernels, same order, same bits as the Python sequence
// (tests/test_gpu_layer_exec.py).
#include <stdint.h>
#include "common.h"

namespace {

enum { kEpiBias = 0, kEpiGelu = 1, kEpiRelu = 2, kEpiMulGeluGrad = 3, kEpiMulReluGrad = 4, kEpiAdd = 5 };
constexpr int kC = 128, kFF = 256;

int64_t wg_ws(int64_t m, int which) {   // workspace of the two weight-gradient groups of a layer (which = 2: all five in one)
  sst_wgrad_problem_f32 p[5];
  float* const any = (float*)(uintptr_t)256;   // the size query looks at shapes and alignment only; it wants non-null operands
  for (auto& q : p) {
    q.dy = q.x = any;
    q.dw = q.db = any;
    q.m = m;
    q.x_add_rows = nullptr, q.x_add_index = nullptr;
  }
  if (which == 0 || which == 2) {
    p[0].ld_dy = kC, p[0].ld_x = kFF, p[0].out = kC, p[0].in = kFF;    // dW2 = ds2^T h
    p[1].ld_dy = kFF, p[1].ld_x = kC, p[1].out = kFF, p[1].in = kC;    // dW1 = dpre^T y1
    if (which == 0) return sst_weight_grad_group_f32x6_workspace_bytes(p, 2);
    p[2].ld_dy = kC, p[2].ld_x = kC, p[2].out = kC, p[2].in = kC;
    p[3].ld_dy = 3 * kC, p[3].ld_x = kC, p[3].out = 2 * kC, p[3].in = kC;
    p[4].ld_dy = 3 * kC, p[4].ld_x = kC, p[4].out = kC, p[4].in = kC;
    return sst_weight_grad_group_f32x6_workspace_bytes(p, 5);
  }
  p[0].ld_dy = kC, p[0].ld_x = kC, p[0].out = kC, p[0].in = kC;              // dWo = ds1^T o
  p[1].ld_dy = 3 * kC, p[1].ld_x = kC, p[1].out = 2 * kC, p[1].in = kC;      // dWq | dWk = [dq | dk]^T xp
  p[2].ld_dy = 3 * kC, p[2].ld_x = kC, p[2].out = kC, p[2].in = kC;          // dWv = dv^T x
  return sst_weight_grad_group_f32x6_workspace_bytes(p, 3);
}

int64_t wg_ws_bf16(int64_t m) {
  sst_wgrad_problem_bf16 p[5];
  for (auto& q : p) q.m = m;
  p[0].p = 2 * kC, p[1].p = kC, p[2].p = kC, p[3].p = kFF, p[4].p = kFF;
  return sst_wgrad_group_workspace_bytes(p, 5);
}

}  // namespace

extern "C" {

int64_t sst_encoder_layer_bwd_workspace_bytes(int64_t m, int n_heads) {
  if (m < 0 || n_heads < 1) return SST_ERR_ARG;
  const int64_t a = sst_encoder_tail_bwd_workspace_bytes(m), d = sst_sra_attn_bwd_workspace_bytes(m, n_heads), e = wg_ws(m, 2);
  if (a < 0 || d < 0 || e < 0) return SST_ERR_UNSUPPORTED;
  // the users never overlap in time on the stream, but a kernel of one may still run when the next is queued: own pieces.
  // LayerNorm partials of the tail (they wait for the weight gradients' reduction launch) | the five weight gradients | attention
  return sst_align_up(a, 256) + sst_align_up(e, 256) + sst_align_up(d, 256) + 256;
}

int64_t sst_encoder_layer_wpack_bytes(void) { return sst_encoder_tail_pack_bytes(); }

int sst_encoder_layer_fwd_f32x6(const sst_encoder_layer_fwd_args* a, void* stream) {
  if (!a || a->m < 0 || a->n_heads * 16 != kC || (a->act != 1 && a->act != 2)) return SST_ERR_ARG;
  if (a->m == 0) return SST_OK;
  if (!a->x || !a->qkv || !a->o || !a->lse || !a->y1 || !a->st1 || !a->pre || !a->h || !a->s2 || !a->y2 || !a->st2)
    return SST_ERR_ARG;
  if (!a->xp && (!a->xpos_table || !a->xpos_idx)) return SST_ERR_ARG;   // x + pos: either as a tensor or as (table, row index)
  const int64_t m = a->m;
  int rc;
  // q | k = (x + pos) W_qk^T, v = x W_v^T: one launch over 384 columns (sst_basic_block_v2.py:56-62); without an xp tensor the
  // positional rows are added to x on the way into the product (table [P][128] + row index per token)
  if (a->xp == nullptr)
    rc = sst_inproj_pos_f32x6(a->x, kC, a->xpos_table, a->xpos_idx, a->w_in, kC, a->b_in, m, a->qkv, 3 * kC, stream);
  else
    rc = sst_tall_linear_epi2_f32x6(a->xp, a->x, 2 * kC, kC, a->w_in, kC, 0, a->b_in, m, kC, 3 * kC, kEpiBias, nullptr, nullptr, 0,
                                    a->qkv, 3 * kC, stream);
  if (rc) return rc;
  if (a->head_scale != nullptr)   // cosine attention: normalisation and 1 / clamp(tau) inside the kernel (cosine_msa.py:159-170)
    rc = sst_sra_attn_cos_fwd_f32(a->qkv, a->qkv + kC, a->qkv + 2 * kC, 3 * kC, 3 * kC, 3 * kC, a->tok, a->winoff, a->order,
                                  a->n_windows, a->n_heads, a->head_scale, a->max_tokens, a->o, kC, a->lse, stream);
  else
    rc = sst_sra_attn_fwd_ord_f32(a->qkv, a->qkv + kC, a->qkv + 2 * kC, 3 * kC, 3 * kC, 3 * kC, a->tok, a->winoff, a->order,
                                  a->n_windows, a->n_heads, a->scale, a->max_tokens, a->impl, a->o, kC, a->lse, stream);
  if (rc) return rc;
  // everything behind the attention core as one kernel (:113-118): the weight images of this call first - unless the caller has
  // formed them already (w_out = w1 = w2 = NULL: sst_encoder_tail_pack_f32x6_many, one launch for the whole stack)
  if (!a->wpack) return SST_ERR_ARG;
  if (a->w_out || a->w1 || a->w2) {
    rc = sst_encoder_tail_pack_f32x6(a->w_out, a->w1, a->w2, a->wpack, stream);
    if (rc) return rc;
  }
  sst_encoder_tail_fwd_args t;
  t.m = m, t.act = a->act, t.reserved = 0, t.eps = a->eps, t.reserved_f = 0.f;
  t.o = a->o, t.x = a->x, t.packed = a->wpack;
  t.b_out = a->b_out, t.b1 = a->b1, t.b2 = a->b2, t.n1w = a->n1w, t.n1b = a->n1b, t.n2w = a->n2w, t.n2b = a->n2b;
  t.pos_table = a->pos_table, t.pos_idx = a->pos_idx;
  t.s1 = a->s1, t.st1 = a->st1, t.y1 = a->y1, t.pre = a->pre, t.h = a->h, t.s2 = a->s2, t.st2 = a->st2, t.y2 = a->y2;
  t.y2p = a->pos_table != nullptr ? a->y2p : nullptr;
  return sst_encoder_tail_fwd_f32x6(&t, stream);
}

int sst_encoder_layer_bwd_f32x6(const sst_encoder_layer_bwd_args* a, void* stream) {
  if (!a || a->m < 0 || a->n_heads * 16 != kC || (a->act != 1 && a->act != 2)) return SST_ERR_ARG;
  if (a->m == 0) return SST_OK;
  if (!a->dy2 || !a->workspace || !a->ds2 || !a->dpre || !a->ds1 || !a->d_o || !a->dqkv || !a->wpack) return SST_ERR_ARG;
  if (!a->dn2w || !a->dn2b || !a->dn1w || !a->dn1b) return SST_ERR_ARG;
  if (!a->xp && (!a->xpos_table || !a->xpos_idx)) return SST_ERR_ARG;
  const int64_t m = a->m;
  char* ws = (char*)a->workspace;
  void* ws_tail = ws;
  ws += sst_align_up(sst_encoder_tail_bwd_workspace_bytes(m), 256);
  void* ws_wg = ws;
  ws += sst_align_up(wg_ws(m, 2), 256);
  void* ws_sra = ws;
  int rc;
  // norm2' -> linear2' * act' -> linear1' + residual -> norm1' -> out-projection' in one kernel; d(gamma) | d(beta) of both
  // LayerNorms stay as per-workgroup partials for the reduction launch of the weight gradients below
  sst_encoder_tail_bwd_args t;
  t.m = m, t.act = a->act, t.reserved = 0;
  t.dy2 = a->dy2, t.dy2p = a->dy2p, t.s2 = a->s2, t.st2 = a->st2, t.pre = a->pre, t.s1 = a->s1, t.st1 = a->st1;
  t.packed = a->wpack, t.n1w = a->n1w, t.n2w = a->n2w;
  t.ds2 = a->ds2, t.dpre = a->dpre, t.ds1 = a->ds1, t.d_o = a->d_o;
  t.dn2w = a->dn2w, t.dn2b = a->dn2b, t.dn1w = a->dn1w, t.dn1b = a->dn1b;
  t.workspace = ws_tail;
  float *part2 = nullptr, *part1 = nullptr;
  int rows = 0;
  rc = sst_internal_encoder_tail_bwd_f32x6(&t, &part2, &part1, &rows, stream);
  if (rc) return rc;
  sst_colsum_rider riders[2];
  riders[0].partials = part2, riders[0].nb = rows, riders[0].width = 2 * kC, riders[0].split = kC;
  riders[0].out0 = a->dn2w, riders[0].out1 = a->dn2b;
  riders[1].partials = part1, riders[1].nb = rows, riders[1].width = 2 * kC, riders[1].split = kC;
  riders[1].out0 = a->dn1w, riders[1].out1 = a->dn1b;
  if (a->head_scale != nullptr)
    rc = sst_sra_attn_cos_bwd_f32(a->qkv, a->qkv + kC, a->qkv + 2 * kC, a->o, a->d_o, a->lse, 3 * kC, 3 * kC, 3 * kC, kC, kC, a->tok,
                                  a->winoff, a->order, a->n_windows, m, a->n_heads, a->head_scale, a->max_tokens, a->dqkv,
                                  a->dqkv + kC, a->dqkv + 2 * kC, 3 * kC, 3 * kC, 3 * kC, a->cos_r, stream);
  else
    rc = sst_sra_attn_bwd_ord_f32(a->qkv, a->qkv + kC, a->qkv + 2 * kC, a->o, a->d_o, a->lse, 3 * kC, 3 * kC, 3 * kC, kC, kC, a->tok,
                                  a->winoff, a->order, a->n_windows, m, a->n_heads, a->scale, a->max_tokens, a->impl, a->dqkv,
                                  a->dqkv + kC, a->dqkv + 2 * kC, 3 * kC, 3 * kC, 3 * kC, ws_sra, stream);
  if (rc) return rc;
  // all five parameter gradients of the layer in one grouped launch (+ its reduction, which also finishes the LayerNorm
  // parameter gradients), before ds1 is accumulated into
  sst_wgrad_problem_f32 g[5];
  for (auto& q : g) q.x_add_rows = nullptr, q.x_add_index = nullptr;
  g[0].dy = a->ds2, g[0].x = a->h, g[0].m = m, g[0].ld_dy = kC, g[0].ld_x = kFF, g[0].dw = a->dw2, g[0].db = a->db2;
  g[0].out = kC, g[0].in = kFF;
  g[1].dy = a->dpre, g[1].x = a->y1, g[1].m = m, g[1].ld_dy = kFF, g[1].ld_x = kC, g[1].dw = a->dw1, g[1].db = a->db1;
  g[1].out = kFF, g[1].in = kC;
  g[2].dy = a->ds1, g[2].x = a->o, g[2].m = m, g[2].ld_dy = kC, g[2].ld_x = kC, g[2].dw = a->dwo, g[2].db = a->dbo;
  g[2].out = kC, g[2].in = kC;
  g[3].dy = a->dqkv, g[3].x = a->xp ? a->xp : a->x, g[3].m = m, g[3].ld_dy = 3 * kC, g[3].ld_x = kC, g[3].dw = a->dw_in;
  g[3].db = a->db_in, g[3].out = 2 * kC, g[3].in = kC;
  if (!a->xp) g[3].x_add_rows = a->xpos_table, g[3].x_add_index = a->xpos_idx;   // dW_q | dW_k = [dq | dk]^T (x + pos rows)
  g[4].dy = a->dqkv + 2 * kC, g[4].x = a->x, g[4].m = m, g[4].ld_dy = 3 * kC, g[4].ld_x = kC;
  g[4].dw = a->dw_in + 2 * kC * kC, g[4].db = a->db_in + 2 * kC, g[4].out = kC, g[4].in = kC;
  rc = sst_internal_weight_grad_group_f32x6(g, 5, ws_wg, riders, 2, stream);
  if (rc) return rc;
  // d(x) of the residual branch and of all three projections: one product over K = 384 (xp = x + constant: d(x) += d(xp))
  return sst_tall_linear_epi_f32x6(a->dqkv, 3 * kC, a->w_in, kC, 1, nullptr, m, 3 * kC, kC, kEpiAdd, a->ds1, nullptr, kC, a->ds1, kC,
                                   stream);
}

// ---- the reduced-precision layer (sst_amd/bf16.py EncoderLayerBF16Fn, its fused-LayerNorm sequence) ------------------------
int64_t sst_encoder_layer_bwd_bf16_workspace_bytes(int64_t m) {
  if (m < 0) return SST_ERR_ARG;
  const int64_t a = sst_encoder_tail_bwd_bf16_workspace_bytes(m), b = wg_ws_bf16(m);
  if (a < 0 || b < 0) return SST_ERR_UNSUPPORTED;
  return sst_align_up(a, 256) + sst_align_up(b, 256) + 256;    // the tail's LayerNorm partials (both norms) live until the reduction
}

int sst_encoder_layer_fwd_bf16(const sst_encoder_layer_fwd_bf16_args* a, void* stream) {
  if (!a || a->m < 0 || a->n_heads * 16 != kC || (a->act != 1 && a->act != 2)) return SST_ERR_ARG;
  if (a->m == 0) return SST_OK;
  if (!a->x || !a->xp || !a->wqk || !a->wv || !a->wout || !a->w1 || !a->w2 || !a->qk || !a->v || !a->o || !a->lse || !a->y1 ||
      !a->st1 || !a->pre || !a->h || !a->y2 || !a->st2)
    return SST_ERR_ARG;
  const int64_t m = a->m;
  const unsigned short* qk = (const unsigned short*)a->qk;
  int rc;
  // q | k from x + pos, v from x (sst_basic_block_v2.py:58-63)
  rc = sst_tall_linear_bf16(a->xp, kC, a->wqk, a->b_in, m, kC, 2 * kC, kEpiBias, nullptr, nullptr, 0, a->qk, 2 * kC, stream);
  if (rc) return rc;
  rc = sst_tall_linear_bf16(a->x, kC, a->wv, a->b_in ? a->b_in + 2 * kC : nullptr, m, kC, kC, kEpiBias, nullptr, nullptr, 0, a->v,
                            kC, stream);
  if (rc) return rc;
  if (a->head_scale != nullptr)
    rc = sst_sra_attn_cos_fwd_bf16(qk, qk + kC, a->v, 2 * kC, 2 * kC, kC, a->tok, a->winoff, a->order, a->n_windows, a->n_heads,
                                   a->head_scale, a->max_tokens, a->o, kC, a->lse, stream);
  else
    rc = sst_sra_attn_fwd_ord_bf16(qk, qk + kC, a->v, 2 * kC, 2 * kC, kC, a->tok, a->winoff, a->order, a->n_windows, a->n_heads,
                                   a->scale, a->max_tokens, a->o, kC, a->lse, stream);
  if (rc) return rc;
  // everything behind the attention core as ONE kernel (csrc/layer_tail_bf16.hip; sst_basic_block_v2.py:113-118)
  if (!a->wpack) return SST_ERR_ARG;
  sst_encoder_tail_fwd_bf16_args t;
  t.m = m, t.act = a->act, t.reserved = 0, t.eps = a->eps, t.reserved_f = 0.f;
  t.o = a->o, t.x = a->x, t.packed = a->wpack;
  t.b_out = a->b_out, t.b1 = a->b1, t.b2 = a->b2, t.n1w = a->n1w, t.n1b = a->n1b, t.n2w = a->n2w, t.n2b = a->n2b;
  t.pos_table = a->pos_table, t.pos_idx = a->pos_idx;
  t.s1 = a->s1, t.st1 = a->st1, t.y1 = a->y1, t.pre = a->pre, t.h = a->h, t.s2 = a->s2, t.st2 = a->st2, t.y2 = a->y2;
  t.y2p = a->pos_table ? a->y2p : nullptr;
  return sst_encoder_tail_fwd_bf16(&t, stream);
}

int sst_encoder_layer_bwd_bf16(const sst_encoder_layer_bwd_bf16_args* a, void* stream) {
  if (!a || a->m < 0 || a->n_heads * 16 != kC || (a->act != 1 && a->act != 2)) return SST_ERR_ARG;
  if (a->m == 0) return SST_OK;
  if (!a->dy2 || !a->workspace || !a->ds2 || !a->dpre || !a->ds1 || !a->d_o || !a->dqkv || !a->dxp || !a->dx ||
      !a->s1 || !a->s2 || !a->wpack)
    return SST_ERR_ARG;
  if (!a->dn2w || !a->dn2b || !a->dn1w || !a->dn1b) return SST_ERR_ARG;
  const int64_t m = a->m;
  char* ws = (char*)a->workspace;
  void* ws_tail = ws;
  ws += sst_align_up(sst_encoder_tail_bwd_bf16_workspace_bytes(m), 256);
  void* ws_wg = ws;
  const unsigned short* qk = (const unsigned short*)a->qk;
  unsigned short* dqkv = (unsigned short*)a->dqkv;
  int rc;
  // norm2' -> linear2' * act' -> linear1' + residual -> norm1' -> out-projection' in one kernel; d(gamma) | d(beta) of both
  // LayerNorms stay as per-workgroup partials for the reduction launch of the parameter-gradient group
  sst_encoder_tail_bwd_bf16_args t;
  t.m = m, t.act = a->act, t.reserved = 0;
  t.dy2 = a->dy2, t.dy2p = a->dy2p, t.s2 = a->s2, t.st2 = a->st2, t.pre = a->pre, t.s1 = a->s1, t.st1 = a->st1;
  t.packed = a->wpack, t.n1w = a->n1w, t.n2w = a->n2w;
  t.ds2 = a->ds2, t.dpre = a->dpre, t.ds1 = a->ds1, t.d_o = a->d_o;
  t.dn2w = a->dn2w, t.dn2b = a->dn2b, t.dn1w = a->dn1w, t.dn1b = a->dn1b;
  t.workspace = ws_tail;
  float *part2 = nullptr, *part1 = nullptr;
  int rows = 0;
  rc = sst_internal_encoder_tail_bwd_bf16(&t, &part2, &part1, &rows, stream);
  if (rc) return rc;
  sst_colsum_rider riders[2];
  riders[0].partials = part2, riders[0].nb = rows, riders[0].width = 2 * kC, riders[0].split = kC;
  riders[0].out0 = a->dn2w, riders[0].out1 = a->dn2b;
  riders[1].partials = part1, riders[1].nb = rows, riders[1].width = 2 * kC, riders[1].split = kC;
  riders[1].out0 = a->dn1w, riders[1].out1 = a->dn1b;
  if (a->head_scale != nullptr)
    rc = sst_sra_attn_cos_bwd_bf16(qk, qk + kC, a->v, a->o, a->d_o, a->lse, 2 * kC, 2 * kC, kC, kC, kC, a->tok, a->winoff,
                                   a->order, a->n_windows, a->n_heads, a->head_scale, a->max_tokens, dqkv, dqkv + kC,
                                   dqkv + 2 * kC, 3 * kC, 3 * kC, 3 * kC, a->cos_r, stream);
  else
    rc = sst_sra_attn_bwd_ord_bf16(qk, qk + kC, a->v, a->o, a->d_o, a->lse, 2 * kC, 2 * kC, kC, kC, kC, a->tok, a->winoff,
                                   a->order, a->n_windows, a->n_heads, a->scale, a->max_tokens, dqkv, dqkv + kC, dqkv + 2 * kC,
                                   3 * kC, 3 * kC, 3 * kC, stream);
  if (rc) return rc;
  rc = sst_tall_linear_bf16(dqkv, 3 * kC, a->wqk_t, nullptr, m, 2 * kC, kC, kEpiBias, nullptr, nullptr, 0, a->dxp, kC, stream);
  if (rc) return rc;
  rc = sst_tall_linear_bf16(dqkv + 2 * kC, 3 * kC, a->wv_t, nullptr, m, kC, kC, kEpiAdd, a->ds1, nullptr, kC, a->dx, kC, stream);
  if (rc) return rc;
  // every parameter gradient of the layer in one launch (+ its reduction)
  sst_wgrad_problem_bf16 g[5];
  for (auto& q : g) {
    q.m = m;
    q.ldb = kC;
    q.bias_side = 1;
    q.transpose_out = 0;
    q.reserved = 0;
  }
  g[0].a = dqkv, g[0].lda = 3 * kC, g[0].b = a->xp, g[0].out_w = a->dw_in, g[0].out_b = a->db_in, g[0].p = 2 * kC;
  g[1].a = dqkv + 2 * kC, g[1].lda = 3 * kC, g[1].b = a->x, g[1].out_w = a->dw_in + 2 * kC * kC, g[1].out_b = a->db_in + 2 * kC;
  g[1].p = kC;
  g[2].a = a->ds1, g[2].lda = kC, g[2].b = a->o, g[2].out_w = a->dwo, g[2].out_b = a->dbo, g[2].p = kC;
  g[3].a = a->dpre, g[3].lda = kFF, g[3].b = a->y1, g[3].out_w = a->dw1, g[3].out_b = a->db1, g[3].p = kFF;
  // dW2 [128][256] = ds2^T h: operands swapped, stored transposed; its bias gradient = column sums of the b side (ds2)
  g[4].a = a->h, g[4].lda = kFF, g[4].b = a->ds2, g[4].out_w = a->dw2, g[4].out_b = a->db2, g[4].p = kFF;
  g[4].bias_side = 2, g[4].transpose_out = 1;
  return sst_internal_wgrad_group_bf16(g, 5, ws_wg, riders, 2, stream);
}

}  // extern "C"
