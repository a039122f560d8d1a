// Sparse Regional Attention core, bf16 storage (gfx950): Q, K, V, O (and dO, dQ, dK, dV) are bf16 in HBM, the softmax
// and every accumulation are fp32.  1 032 B per token and layer forward instead of 2 056, half of the backward's traffic.
//
// Reference: the reference trains SST under fp16 (configs/sst_refactor/sst_waymoD5_1x_3class_8heads_v2.py:82,
// Fp16OptimizerHook; the attention of WindowAttention.forward, models/sst/sst_basic_block_v2.py:41-75, then runs in half
// precision with an fp32 softmax inside nn.MultiheadAttention).  bf16 is the MI355X counterpart with the fp32 exponent range.
//
// Same mapping as the fp32 kernels of sra_attn.hip - workgroup = window x 4 heads, ONE WAVE = ONE HEAD, the wave's K / V
// fragments register-resident for every query tile, no barrier - with v_mfma_f32_16x16x16_bf16: head_dim is 16, so
// S^T = K Q^T of a (16 key x 16 query) tile pair is ONE instruction (four at fp32) and so is P V.  Fragment vocabulary
// (lane = 16 g + c):
//   row-frag  X[token c][16 h + 4g .. 4g+3]            one 8-byte load; A or B operand of a product contracted over d
//   col-frag  X[token 4g + r][16 h + c], r = 0..3      operand of a product contracted over tokens
//   D layout  value r of lane (g, c) = D[row 4g + r][col c]: packed to bf16 it is directly a B operand (k = 4g + r)
// At bf16 the matrix pipe is 8x faster than at fp32 and no longer the limiter: the kernels are bound by the softmax
// arithmetic (VALU) and HBM, which is why the simple one-accumulator chains below are enough.
// Backward: ONE pass per (window, head) wave, as sra_bwd_fused_k: S and dP in the orientation row = query, col = key;
// dV^T += dO^T P and dK^T += Q^T dS consume the D layout directly; the operands contracted over tokens and the dS^T tile
// of dQ^T += K^T dS^T are transposed by a product with the identity on the matrix core (no LDS at all).  6 MFMAs per
// tile pair.
#include <math.h>
#include <stdlib.h>
#include <hip/hip_ext.h>
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int kHD = 16;
constexpr int kWH = 4;      // heads (= waves) per workgroup
constexpr int kTS = 20;     // row stride (floats) of the LDS tiles
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;
#define SST_SRA_BLOCK(b, n) ((int)(n) - 1 - (int)(b))  // largest windows first (region batching lists them last)

__device__ __forceinline__ float bf2f(unsigned short v) { return __uint_as_float((unsigned)v << 16); }
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
// round-to-nearest-even pair: the conversion compiles to ONE v_cvt_pk_bf16_f32.  It must be the compiler's own
// instruction, not inline asm: an MFMA that reads a register written by the VALU two instructions earlier needs wait
// states on gfx950, and the hazard pass cannot see inside an asm statement (measured: stale operands, NaN).
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ s16x4 pack4(float a, float b, float c, float d) {
  u32x2 p = {pack2(a, b), pack2(c, d)};
  return __builtin_bit_cast(s16x4, p);
}
__device__ __forceinline__ s16x4 ld_row(const unsigned short* __restrict__ base, uint32_t off) {
  return __builtin_bit_cast(s16x4, *(const u32x2*)(base + off));
}
__device__ __forceinline__ f32x4 unpack4(s16x4 v) {
  const u32x2 p = __builtin_bit_cast(u32x2, v);
  f32x4 r = {__uint_as_float(p[0] << 16), __uint_as_float(p[0] & 0xffff0000u), __uint_as_float(p[1] << 16),
             __uint_as_float(p[1] & 0xffff0000u)};
  return r;
}
__device__ __forceinline__ f32x4 mma(s16x4 a, s16x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float rows4_max(float v) {
  u32x2 a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  const float m = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  u32x2 b = __builtin_amdgcn_permlane16_swap(__float_as_uint(m), __float_as_uint(m), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float rows4_sum(float v) {
  u32x2 a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  const float m = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  u32x2 b = __builtin_amdgcn_permlane16_swap(__float_as_uint(m), __float_as_uint(m), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// scaled cosine attention (cosine_msa.py:123-170; see csrc/sra_attn.hip): a bf16 row fragment (token c, channels 4g .. 4g+3 of the
// head) normalised over the head's 16 channels - the four lanes c, c+16, c+32, c+48 hold one row - and rounded back to bf16
__device__ __forceinline__ float rowfrag_inv_norm(const s16x4 x) {
  const f32x4 u = unpack4(x);
  const float n2 = rows4_sum(fmaf(u[0], u[0], fmaf(u[1], u[1], fmaf(u[2], u[2], u[3] * u[3]))));
  return __builtin_amdgcn_rsqf(fmaxf(n2, 1e-24f));      // 1 / max(|x|, 1e-12)
}
__device__ __forceinline__ s16x4 scaled(const s16x4 x, const float sc) {
  const f32x4 u = unpack4(x);
  return pack4(u[0] * sc, u[1] * sc, u[2] * sc, u[3] * sc);
}
__device__ __forceinline__ float dot4(const f32x4 a, const f32x4 b) { return fmaf(a[0], b[0], fmaf(a[1], b[1], fmaf(a[2], b[2], a[3] * b[3]))); }

// ------------------------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------------------------
template <int NT, bool COS>
__device__ __forceinline__ void fwd_body(const unsigned short* __restrict__ Q, const unsigned short* __restrict__ K,
                                         const unsigned short* __restrict__ V, uint32_t ldq, uint32_t ldk, uint32_t ldv,
                                         const int32_t* __restrict__ tok, int beg, int t, int nt, int hg, int H,
                                         float scale, unsigned short* __restrict__ O, uint32_t ldo,
                                         float* __restrict__ LSE, const float* __restrict__ hscale) {
  const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  const int head = hg * kWH + (threadIdx.x >> 6);
  const uint32_t hoff = head * kHD;
  if (COS) scale = hscale[head];
  constexpr int NTK = (NT * 16 + 63) / 64;
  int tk[NTK];
#pragma unroll
  for (int i = 0; i < NTK; ++i) {
    const int p = i * 64 + lane;
    // padded positions repeat the last token: loads are unconditional; no list (tok == nullptr): the window's rows are
    // beg .. beg + t - 1 themselves (feature rows in window order), one dependent load less at the head of the wave
    tk[i] = tok != nullptr ? tok[beg + (p < t ? p : t - 1)] : beg + (p < t ? p : t - 1);
  }
  auto tok_at = [&](int i, int within) -> uint32_t {
    int sel = tk[0];
#pragma unroll
    for (int u = 1; u < NTK; ++u) sel = ((i >> 2) == u) ? tk[u] : sel;
    return (uint32_t)__shfl(sel, (i & 3) * 16 + within, 64);
  };
  s16x4 kf[NT], vf[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    if (j < nt) {
      const uint32_t krow = (uint32_t)__shfl(tk[j >> 2], (j & 3) * 16 + c, 64);
      kf[j] = ld_row(K, krow * ldk + hoff + 4 * g);
      if (COS) kf[j] = scaled(kf[j], rowfrag_inv_norm(kf[j]));
      unsigned short v4[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const uint32_t vrow = (uint32_t)__shfl(tk[j >> 2], (j & 3) * 16 + 4 * g + r, 64);
        v4[r] = V[vrow * ldv + hoff + c];
      }
      u32x2 p = {(unsigned)v4[0] | ((unsigned)v4[1] << 16), (unsigned)v4[2] | ((unsigned)v4[3] << 16)};
      vf[j] = __builtin_bit_cast(s16x4, p);
    } else {
      kf[j] = vf[j] = (s16x4){0, 0, 0, 0};
    }
  }
  const float s2 = scale * kLog2e;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  uint32_t qrow = tok_at(0, c);
  s16x4 qf = ld_row(Q, qrow * ldq + hoff + 4 * g);
  for (int i = 0; i < nt; ++i) {
    const uint32_t qrow_next = tok_at(i + 1 < nt ? i + 1 : i, c);
    const s16x4 qf_next = ld_row(Q, qrow_next * ldq + hoff + 4 * g);  // prefetch
    if (COS) qf = scaled(qf, rowfrag_inv_norm(qf));
    f32x4 st[NT];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      if (j < nt) {
        st[j] = mma(kf[j], qf, zero4);  // S^T[key 16j + 4g + r][query 16i + c]
        if (j == nt - 1) {
#pragma unroll
          for (int r = 0; r < 4; ++r) st[j][r] = (j * 16 + 4 * g + r) < t ? st[j][r] : -INFINITY;
        }
        mx = fmaxf(mx, fmaxf(fmaxf(st[j][0], st[j][1]), fmaxf(st[j][2], st[j][3])));
      }
    }
    mx = rows4_max(mx);
    const float off = mx * s2;
    float sm = 0.f;
    f32x4 o = zero4;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      if (j < nt) {
        float p[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          p[r] = __builtin_amdgcn_exp2f(fmaf(st[j][r], s2, -off));
          sm += p[r];
        }
        o = mma(vf[j], pack4(p[0], p[1], p[2], p[3]), o);  // O^T[d 4g + r][query c] += V^T P^T
      }
    }
    const float sum = rows4_sum(sm);
    if (i * 16 + c < t) {
      const float inv = __builtin_amdgcn_rcpf(sum);
      const u32x2 ov = {pack2(o[0] * inv, o[1] * inv), pack2(o[2] * inv, o[3] * inv)};
      *(u32x2*)(O + (qrow * ldo + hoff + 4 * g)) = ov;
      if (g == 0) LSE[qrow * (uint32_t)H + head] = (off + __builtin_amdgcn_logf(sum)) * kLn2;
    }
    qrow = qrow_next;
    qf = qf_next;
  }
}

template <int NTMAX, bool COS>
__global__ __launch_bounds__(64 * kWH) void sra_fwd_bf16_k(const unsigned short* __restrict__ Q,
                                                           const unsigned short* __restrict__ K,
                                                           const unsigned short* __restrict__ V, int64_t ldq, int64_t ldk,
                                                           int64_t ldv, const int32_t* __restrict__ tok,
                                                           const int32_t* __restrict__ winoff, int n_groups, int H,
                                                           float scale, unsigned short* __restrict__ O, int64_t ldo,
                                                           float* __restrict__ LSE, const int32_t* __restrict__ order,
                                                           const float* __restrict__ hscale) {
  const int bid = SST_SRA_BLOCK(blockIdx.x, gridDim.x);
  const int wpos = bid / n_groups;
  const int hg = bid - wpos * n_groups;
  const int w = order != nullptr ? order[wpos] : wpos;   // launch order of the windows (sst_sra_attn_*_ord_bf16)
  const int beg = winoff[w];
  const int t = winoff[w + 1] - beg;
  const int nt = (t + 15) >> 4;
  if (nt < 1 || nt > NTMAX) return;
#define SST_F_ARGS Q, K, V, (uint32_t)ldq, (uint32_t)ldk, (uint32_t)ldv, tok, beg, t, nt, hg, H, scale, O, (uint32_t)ldo, LSE, hscale
  if (nt <= 2)
    fwd_body<2, COS>(SST_F_ARGS);
  else if (nt <= 4)
    fwd_body<4, COS>(SST_F_ARGS);
  else
    fwd_body<NTMAX, COS>(SST_F_ARGS);
#undef SST_F_ARGS
}

// ------------------------------------------------------------------------------------------------------------------
// backward, one pass
// ------------------------------------------------------------------------------------------------------------------
// Operands contracted over tokens (K / Q / dO as [token][d] column fragments, dS^T) are produced from the row fragments BY
// THE MATRIX CORE: X I with the fragment as A operand (lane = token, registers = 4 consecutive d) lands in the D layout
// (lane = d, registers = 4 consecutive tokens), which packed to bf16 is the operand wanted - exact (a selection of bf16
// values accumulated in fp32).  The same product turns the dS tile (D layout: lane = key, registers = queries, read as
// the A operand of dS^T) into the B operand of dQ^T += K^T dS^T.  No LDS, no barrier, no wave-private tiles.
// COS: q / k normalised as they are loaded, score scale hscale[head], the gradients taken through the normalisation where they are
// stored (d x = (d x^ - x^ (x^ . d x^)) / |x|: dQ / dK leave in the row-fragment layout Q / K were loaded in), q^ . dq^ -> R
template <int NT, bool EXACT, bool COS>
__device__ __forceinline__ void bwd_body(const unsigned short* __restrict__ Q, const unsigned short* __restrict__ K,
                                         const unsigned short* __restrict__ V, const unsigned short* __restrict__ O,
                                         const unsigned short* __restrict__ dO, const float* __restrict__ LSE,
                                         uint32_t ldq, uint32_t ldk, uint32_t ldv, uint32_t ldo, uint32_t lddo,
                                         const int32_t* __restrict__ tok, int beg, int t, int nt, int hg, int H,
                                         float scale, unsigned short* __restrict__ dQ, unsigned short* __restrict__ dK,
                                         unsigned short* __restrict__ dV, uint32_t lddq, uint32_t lddk, uint32_t lddv,
                                         const float* __restrict__ hscale, float* __restrict__ R) {
  const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  const int head = hg * kWH + (threadIdx.x >> 6);
  const uint32_t hoff = head * kHD;
  if (COS) scale = hscale[head];
  constexpr int NTK = (NT * 16 + 63) / 64;
  int tk[NTK];
#pragma unroll
  for (int i = 0; i < NTK; ++i) {
    const int p = i * 64 + lane;
    tk[i] = tok != nullptr ? tok[beg + (p < t ? p : t - 1)] : beg + (p < t ? p : t - 1);
  }
  auto tok_at = [&](int i, int within) -> uint32_t {
    int sel = tk[0];
#pragma unroll
    for (int u = 1; u < NTK; ++u) sel = ((i >> 2) == u) ? tk[u] : sel;
    return (uint32_t)__shfl(sel, (i & 3) * 16 + within, 64);
  };
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  // identity as B operand: B[k = 4 g + x][j = c] = 1 iff 4 g + x == c
  s16x4 ident = {0, 0, 0, 0};
  if (g == (c >> 2)) ident[c & 3] = (short)0x3f80;
  auto transposed = [&](s16x4 frag) -> s16x4 {
    const f32x4 d = mma(frag, ident, zero4);
    return pack4(d[0], d[1], d[2], d[3]);
  };
  s16x4 kf[NT], vf[NT], kcf[NT];
  f32x4 dk[NT], dv[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    dk[j] = zero4;
    dv[j] = zero4;
    if (EXACT || j < nt) {
      const uint32_t krow = (uint32_t)__shfl(tk[j >> 2], (j & 3) * 16 + c, 64);
      kf[j] = ld_row(K, krow * ldk + hoff + 4 * g);
      if (COS) kf[j] = scaled(kf[j], rowfrag_inv_norm(kf[j]));   // k^ (bf16) from here on
      vf[j] = ld_row(V, krow * ldv + hoff + 4 * g);
    } else {
      kf[j] = vf[j] = (s16x4){0, 0, 0, 0};
    }
  }
  struct qtile {
    s16x4 qf, gf, of;
    float lse;
    uint32_t row;
  };
  auto load_tile = [&](int i) -> qtile {
    qtile q;
    const uint32_t row = tok_at(i, c);
    q.row = row;
    q.qf = ld_row(Q, row * ldq + hoff + 4 * g);
    q.gf = ld_row(dO, row * lddo + hoff + 4 * g);
    q.of = ld_row(O, row * ldo + hoff + 4 * g);
    q.lse = LSE[row * (uint32_t)H + head];
    return q;
  };
  qtile cur = load_tile(0);
#pragma unroll
  for (int j = 0; j < NT; ++j) kcf[j] = (EXACT || j < nt) ? transposed(kf[j]) : (s16x4){0, 0, 0, 0};  // K [key 4g + r][d = c]
  const float s2 = scale * kLog2e;

  for (int i = 0; i < nt; ++i) {
    const float q_inv = COS ? rowfrag_inv_norm(cur.qf) : 1.f;
    const s16x4 qf = COS ? scaled(cur.qf, q_inv) : cur.qf, gf = cur.gf;   // COS: q^
    const f32x4 gq = unpack4(gf), oq = unpack4(cur.of);
    float dd = gq[0] * oq[0] + gq[1] * oq[1] + gq[2] * oq[2] + gq[3] * oq[3];
    dd = rows4_sum(dd);
    const float lse_c = cur.lse * kLog2e;
    const uint32_t qrow = cur.row;
    cur = load_tile(i + 1 < nt ? i + 1 : i);
    const s16x4 qcp = transposed(qf);  // Q [query 4g + r][d = c]: A operand of dK^T += Q^T dS
    const s16x4 gcp = transposed(gf);  // dO[query 4g + r][d = c]: A operand of dV^T += dO^T P
    float lse2[4], dd4[4], rmask[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      lse2[r] = __shfl(lse_c, 4 * g + r, 64);
      dd4[r] = __shfl(dd, 4 * g + r, 64);
      rmask[r] = (i * 16 + 4 * g + r) < t ? 1.f : 0.f;
    }
    f32x4 dq = zero4;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      if (EXACT || j < nt) {
        const f32x4 s = mma(qf, kf[j], zero4);   // S [query 4g + r][key 16j + c]
        const f32x4 dp = mma(gf, vf[j], zero4);  // dP
        const bool last_j = EXACT ? (j == NT - 1) : (j == nt - 1);
        const float cmask = (last_j && (j * 16 + c) >= t) ? 0.f : 1.f;
        f32x4 pe, ds;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float p = __builtin_amdgcn_exp2f(fmaf(s[r], s2, -lse2[r])) * rmask[r];
          if (!EXACT || j == NT - 1) p *= cmask;
          pe[r] = p;
          ds[r] = p * (dp[r] - dd4[r]) * scale;
        }
        const s16x4 dsp = pack4(ds[0], ds[1], ds[2], ds[3]);          // dS [query 4g + r][key c]: B operand over queries
        dv[j] = mma(gcp, pack4(pe[0], pe[1], pe[2], pe[3]), dv[j]);  // dV^T[d][key] += dO^T P
        dk[j] = mma(qcp, dsp, dk[j]);                                 // dK^T[d][key] += Q^T dS
        dq = mma(kcf[j], transposed(dsp), dq);                        // dQ^T[d][query] += K^T dS^T
      }
    }
    if (COS) {   // lane (g, c) holds dq^[query c][4g .. 4g+3] and q^ of the same row fragment; every lane takes part in the sum
      const f32x4 qh = unpack4(qf);
      const float rq = rows4_sum(dot4(dq, qh));
#pragma unroll
      for (int r = 0; r < 4; ++r) dq[r] = (dq[r] - qh[r] * rq) * q_inv;
      if (g == 0 && i * 16 + c < t) R[qrow * (uint32_t)H + head] = rq;
    }
    if (i * 16 + c < t) {
      const u32x2 o = {pack2(dq[0], dq[1]), pack2(dq[2], dq[3])};  // lane (g, c): dQ[query c][4g .. 4g+3]
      *(u32x2*)(dQ + (qrow * lddq + hoff + 4 * g)) = o;
    }
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    if (EXACT || j < nt) {
      const uint32_t krow = (uint32_t)__shfl(tk[j >> 2], (j & 3) * 16 + c, 64);
      if (COS) {   // kf[j] is k^; |k| from the row itself once more (an L2 hit)
        const float k_inv = rowfrag_inv_norm(ld_row(K, krow * ldk + hoff + 4 * g));
        const f32x4 kh = unpack4(kf[j]);
        const float rk = rows4_sum(dot4(dk[j], kh));
#pragma unroll
        for (int r = 0; r < 4; ++r) dk[j][r] = (dk[j][r] - kh[r] * rk) * k_inv;
      }
      if (j * 16 + c < t) {
        const u32x2 a = {pack2(dk[j][0], dk[j][1]), pack2(dk[j][2], dk[j][3])};
        const u32x2 b = {pack2(dv[j][0], dv[j][1]), pack2(dv[j][2], dv[j][3])};
        *(u32x2*)(dK + (krow * lddk + hoff + 4 * g)) = a;
        *(u32x2*)(dV + (krow * lddv + hoff + 4 * g)) = b;
      }
    }
  }
}

template <int NTMAX, bool COS>
__global__ __launch_bounds__(64 * kWH, (NTMAX > 7 ? 2 : 3)) void sra_bwd_bf16_k(
    const unsigned short* __restrict__ Q, const unsigned short* __restrict__ K, const unsigned short* __restrict__ V,
    const unsigned short* __restrict__ O, const unsigned short* __restrict__ dO, const float* __restrict__ LSE,
    int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t lddo, const int32_t* __restrict__ tok,
    const int32_t* __restrict__ winoff, int n_groups, int H, float scale, unsigned short* __restrict__ dQ,
    unsigned short* __restrict__ dK, unsigned short* __restrict__ dV, int64_t lddq, int64_t lddk, int64_t lddv,
    const int32_t* __restrict__ order, const float* __restrict__ hscale, float* __restrict__ R) {
  const int bid = SST_SRA_BLOCK(blockIdx.x, gridDim.x);
  const int wpos = bid / n_groups;
  const int hg = bid - wpos * n_groups;
  const int w = order != nullptr ? order[wpos] : wpos;
  const int beg = winoff[w];
  const int t = winoff[w + 1] - beg;
  const int nt = (t + 15) >> 4;
  if (nt < 1 || nt > NTMAX) return;
#define SST_B_ARGS Q, K, V, O, dO, LSE, (uint32_t)ldq, (uint32_t)ldk, (uint32_t)ldv, (uint32_t)ldo, (uint32_t)lddo, tok, beg, t, nt, hg, H, scale, dQ, dK, dV, (uint32_t)lddq, (uint32_t)lddk, (uint32_t)lddv, hscale, R
  switch (nt) {
    case 1: bwd_body<1, true, COS>(SST_B_ARGS); break;
    case 2: bwd_body<2, true, COS>(SST_B_ARGS); break;
    case 3: bwd_body<3, true, COS>(SST_B_ARGS); break;
    case 4: bwd_body<4, true, COS>(SST_B_ARGS); break;
    case 5: if constexpr (NTMAX >= 5) bwd_body<5, true, COS>(SST_B_ARGS); break;
    case 6: if constexpr (NTMAX >= 6) bwd_body<6, true, COS>(SST_B_ARGS); break;
    case 7: if constexpr (NTMAX >= 7) bwd_body<7, true, COS>(SST_B_ARGS); break;
    default:
      if constexpr (NTMAX > 7) bwd_body<NTMAX, false, COS>(SST_B_ARGS);
      break;
  }
#undef SST_B_ARGS
}

thread_local hipEvent_t g_ev[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};  // one-shot hooks: [fwd|bwd][start|stop]
thread_local const int32_t* g_win_order = nullptr;   // launch order of the current call (the *_ord_* entries set it)
thread_local const float* g_head_scale = nullptr;    // cosine attention of the current call (sst_sra_attn_cos_*_bf16 set them)
thread_local float* g_cos_r = nullptr;

bool aligned8(const void* p) { return ((uintptr_t)p & 7) == 0; }

template <int NTMAX>
int launch_fwd(const unsigned short* q, const unsigned short* k, const unsigned short* v, int64_t ldq, int64_t ldk,
               int64_t ldv, const int32_t* tok, const int32_t* winoff, int64_t n_windows, int H, float scale,
               unsigned short* o, int64_t ldo, float* lse, hipStream_t st) {
  const int n_groups = H / kWH;
  const dim3 grid((unsigned)(n_windows * n_groups));
  hipEvent_t e0 = g_ev[0][0], e1 = g_ev[0][1];
  g_ev[0][0] = g_ev[0][1] = nullptr;
  const float* hs = g_head_scale;
  auto kern = hs != nullptr ? sra_fwd_bf16_k<NTMAX, true> : sra_fwd_bf16_k<NTMAX, false>;
  if (e0 != nullptr && e1 != nullptr)
    hipExtLaunchKernelGGL(kern, grid, dim3(64 * kWH), 0, st, e0, e1, 0, q, k, v, ldq, ldk, ldv, tok, winoff,
                          n_groups, H, scale, o, ldo, lse, g_win_order, hs);
  else
    hipLaunchKernelGGL(kern, grid, dim3(64 * kWH), 0, st, q, k, v, ldq, ldk, ldv, tok, winoff, n_groups, H,
                       scale, o, ldo, lse, g_win_order, hs);
  return SST_OK;
}

template <int NTMAX>
int launch_bwd(const unsigned short* q, const unsigned short* k, const unsigned short* v, const unsigned short* o,
               const unsigned short* g, const float* lse, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t ldg,
               const int32_t* tok, const int32_t* winoff, int64_t n_windows, int H, float scale, unsigned short* dq,
               unsigned short* dk, unsigned short* dv, int64_t lddq, int64_t lddk, int64_t lddv, hipStream_t st) {
  const int n_groups = H / kWH;
  const size_t lds = 0;
  const dim3 grid((unsigned)(n_windows * n_groups));
  hipEvent_t e0 = g_ev[1][0], e1 = g_ev[1][1];
  g_ev[1][0] = g_ev[1][1] = nullptr;
  const float* hs = g_head_scale;
  float* rbuf = g_cos_r;
  if (hs != nullptr && rbuf == nullptr) return SST_ERR_ARG;
  auto kern = hs != nullptr ? sra_bwd_bf16_k<NTMAX, true> : sra_bwd_bf16_k<NTMAX, false>;
  if (e0 != nullptr && e1 != nullptr)
    hipExtLaunchKernelGGL(kern, grid, dim3(64 * kWH), lds, st, e0, e1, 0, q, k, v, o, g, lse, ldq, ldk, ldv,
                          ldo, ldg, tok, winoff, n_groups, H, scale, dq, dk, dv, lddq, lddk, lddv, g_win_order, hs, rbuf);
  else
    hipLaunchKernelGGL(kern, grid, dim3(64 * kWH), lds, st, q, k, v, o, g, lse, ldq, ldk, ldv, ldo, ldg, tok,
                       winoff, n_groups, H, scale, dq, dk, dv, lddq, lddk, lddv, g_win_order, hs, rbuf);
  return SST_OK;
}

}  // namespace

extern "C" {

int sst_sra_attn_fwd_bf16(const void* d_q, const void* d_k, const void* d_v, int64_t ldq, int64_t ldk, int64_t ldv,
                          const int32_t* d_tok, const int32_t* d_winoff, int64_t n_windows, int n_heads, float scale,
                          int max_tokens, void* d_o, int64_t ldo, float* d_lse, void* stream) {
  if (n_windows < 0 || n_heads < 1 || (n_heads % kWH) != 0) return SST_ERR_ARG;
  if (n_windows == 0) return SST_OK;
  if (!d_q || !d_k || !d_v || !d_winoff || !d_o || !d_lse) return SST_ERR_ARG;   // d_tok == NULL: rows in window order
  if (((ldq | ldk | ldv | ldo) & 3) || !aligned8(d_q) || !aligned8(d_k) || !aligned8(d_v) || !aligned8(d_o)) return SST_ERR_ARG;
  const int cap_tiles = max_tokens > 0 ? (max_tokens + 15) / 16 : 1 << 30;
  if (cap_tiles > 9) return SST_ERR_UNSUPPORTED;  // windows above 144 tokens: no SST configuration has them
  hipStream_t st = (hipStream_t)stream;
  const unsigned short *q = (const unsigned short*)d_q, *k = (const unsigned short*)d_k, *v = (const unsigned short*)d_v;
  int rc;
  if (cap_tiles <= 5)
    rc = launch_fwd<5>(q, k, v, ldq, ldk, ldv, d_tok, d_winoff, n_windows, n_heads, scale, (unsigned short*)d_o, ldo, d_lse, st);
  else if (cap_tiles <= 7)
    rc = launch_fwd<7>(q, k, v, ldq, ldk, ldv, d_tok, d_winoff, n_windows, n_heads, scale, (unsigned short*)d_o, ldo, d_lse, st);
  else
    rc = launch_fwd<9>(q, k, v, ldq, ldk, ldv, d_tok, d_winoff, n_windows, n_heads, scale, (unsigned short*)d_o, ldo, d_lse, st);
  if (rc) return rc;
  SST_LAUNCH_CHECK();
  return SST_OK;
}

int sst_sra_attn_bwd_bf16(const void* d_q, const void* d_k, const void* d_v, const void* d_o, const void* d_do,
                          const float* d_lse, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t lddo,
                          const int32_t* d_tok, const int32_t* d_winoff, int64_t n_windows, int n_heads, float scale,
                          int max_tokens, void* d_dq, void* d_dk, void* d_dv, int64_t lddq, int64_t lddk, int64_t lddv,
                          void* stream) {
  if (n_windows < 0 || n_heads < 1 || (n_heads % kWH) != 0) return SST_ERR_ARG;
  if (n_windows == 0) return SST_OK;
  if (!d_q || !d_k || !d_v || !d_o || !d_do || !d_lse || !d_winoff || !d_dq || !d_dk || !d_dv) return SST_ERR_ARG;
  if (((ldq | ldk | ldv | ldo | lddo | lddq | lddk | lddv) & 3) || !aligned8(d_q) || !aligned8(d_k) || !aligned8(d_v) ||
      !aligned8(d_o) || !aligned8(d_do) || !aligned8(d_dq) || !aligned8(d_dk) || !aligned8(d_dv))
    return SST_ERR_ARG;
  const int cap_tiles = max_tokens > 0 ? (max_tokens + 15) / 16 : 1 << 30;
  if (cap_tiles > 9) return SST_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
#define SST_BW(NTM)                                                                                                       \
  launch_bwd<NTM>((const unsigned short*)d_q, (const unsigned short*)d_k, (const unsigned short*)d_v,                    \
                  (const unsigned short*)d_o, (const unsigned short*)d_do, d_lse, ldq, ldk, ldv, ldo, lddo, d_tok, d_winoff, \
                  n_windows, n_heads, scale, (unsigned short*)d_dq, (unsigned short*)d_dk, (unsigned short*)d_dv, lddq,  \
                  lddk, lddv, st)
  int rc;
  if (cap_tiles <= 5)
    rc = SST_BW(5);
  else if (cap_tiles <= 7)
    rc = SST_BW(7);
  else
    rc = SST_BW(9);
#undef SST_BW
  if (rc) return rc;
  SST_LAUNCH_CHECK();
  return SST_OK;
}

int sst_sra_attn_fwd_ord_bf16(const void* d_q, const void* d_k, const void* d_v, int64_t ldq, int64_t ldk, int64_t ldv,
                              const int32_t* d_tok, const int32_t* d_winoff, const int32_t* d_win_order, int64_t n_windows,
                              int n_heads, float scale, int max_tokens, void* d_o, int64_t ldo, float* d_lse,
                              void* stream) {
  g_win_order = d_win_order;
  const int rc = sst_sra_attn_fwd_bf16(d_q, d_k, d_v, ldq, ldk, ldv, d_tok, d_winoff, n_windows, n_heads, scale, max_tokens,
                                       d_o, ldo, d_lse, stream);
  g_win_order = nullptr;
  return rc;
}

int sst_sra_attn_bwd_ord_bf16(const void* d_q, const void* d_k, const void* d_v, const void* d_o, const void* d_do,
                              const float* d_lse, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t lddo,
                              const int32_t* d_tok, const int32_t* d_winoff, const int32_t* d_win_order, int64_t n_windows,
                              int n_heads, float scale, int max_tokens, void* d_dq, void* d_dk, void* d_dv, int64_t lddq,
                              int64_t lddk, int64_t lddv, void* stream) {
  g_win_order = d_win_order;
  const int rc = sst_sra_attn_bwd_bf16(d_q, d_k, d_v, d_o, d_do, d_lse, ldq, ldk, ldv, ldo, lddo, d_tok, d_winoff,
                                       n_windows, n_heads, scale, max_tokens, d_dq, d_dk, d_dv, lddq, lddk, lddv, stream);
  g_win_order = nullptr;
  return rc;
}

// Scaled cosine attention with bf16 storage (see sst_sra_attn_cos_{fwd,bwd}_f32): normalize(q) normalize(k)^T * head_scale[h],
// head_scale [n_heads] fp32 in DEVICE memory; the normalised rows are rounded to bf16 before the products (bf16 resolution, as
// every operand of this mode).  Backward: d_dq / d_dk = gradients of the UN-normalised rows, d_r [n_tokens, n_heads] fp32 =
// normalize(q) . d normalize(q) (column sums / head_scale = gradient of the scale).
int sst_sra_attn_cos_fwd_bf16(const void* d_q, const void* d_k, const void* d_v, int64_t ldq, int64_t ldk, int64_t ldv,
                              const int32_t* d_tok, const int32_t* d_winoff, const int32_t* d_win_order, int64_t n_windows,
                              int n_heads, const float* d_head_scale, int max_tokens, void* d_o, int64_t ldo, float* d_lse,
                              void* stream) {
  if (!d_head_scale) return SST_ERR_ARG;
  g_win_order = d_win_order;
  g_head_scale = d_head_scale;
  const int rc = sst_sra_attn_fwd_bf16(d_q, d_k, d_v, ldq, ldk, ldv, d_tok, d_winoff, n_windows, n_heads, 1.0f, max_tokens, d_o,
                                       ldo, d_lse, stream);
  g_head_scale = nullptr;
  g_win_order = nullptr;
  return rc;
}

int sst_sra_attn_cos_bwd_bf16(const void* d_q, const void* d_k, const void* d_v, const void* d_o, const void* d_do,
                              const float* d_lse, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t lddo,
                              const int32_t* d_tok, const int32_t* d_winoff, const int32_t* d_win_order, int64_t n_windows,
                              int n_heads, const float* d_head_scale, int max_tokens, void* d_dq, void* d_dk, void* d_dv,
                              int64_t lddq, int64_t lddk, int64_t lddv, float* d_r, void* stream) {
  if (!d_head_scale || !d_r) return SST_ERR_ARG;
  g_win_order = d_win_order;
  g_head_scale = d_head_scale;
  g_cos_r = d_r;
  const int rc = sst_sra_attn_bwd_bf16(d_q, d_k, d_v, d_o, d_do, d_lse, ldq, ldk, ldv, ldo, lddo, d_tok, d_winoff, n_windows,
                                       n_heads, 1.0f, max_tokens, d_dq, d_dk, d_dv, lddq, lddk, lddv, stream);
  g_cos_r = nullptr;
  g_head_scale = nullptr;
  g_win_order = nullptr;
  return rc;
}

int sst_sra_attn_bf16_profile_next(int backward, void* start, void* stop) {
  g_ev[backward ? 1 : 0][0] = (hipEvent_t)start;
  g_ev[backward ? 1 : 0][1] = (hipEvent_t)stop;
  return SST_OK;
}

}  // extern "C"
