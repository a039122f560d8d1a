// Sparse Regional Attention (SRA) core for gfx950: softmax(Q K^T * scale) V inside every window of
// non-empty voxels, variable window length, no padding and no key mask.
//
// Reference semantics: WindowAttention.forward, mmdet3d/models/sst/sst_basic_block_v2.py:41-75 —
// per drop level: flat2window (scatter to a padded [W,T,C] tensor) -> nn.MultiheadAttention
// (q = k = x + pos, v = x; scores/4; key_padding_mask -> -inf; softmax; bmm) -> window2flat.
// The padded tensors, the mask and the head-averaged attention map the reference materialises and
// throws away do not exist here: the kernel walks the window CSR (tok, winoff) built by window.hip and
// reads / writes token rows in their flat [M, C] layout.
//
// Mapping to the hardware (MI355X, CDNA4) — three implementations behind one entry point (`impl`):
//   impl 0 (default) register-resident: workgroup = window x 4 heads, ONE WAVE = ONE HEAD.  The wave loads its
//     head's K row-fragments and V column-fragments straight from HBM/L2 into the MFMA operand layout and keeps
//     them in VGPRs for every query tile of the window (<= 72 VGPRs at 144 tokens): no LDS, no barrier.
//     S^T = K Q^T and O^T = V^T P^T run on the exact-fp32 MFMA v_mfma_f32_16x16x4_f32: the D layout of S^T
//     (row = key, col = query) is directly the B-operand layout of the second product, so P never leaves
//     registers; softmax in the log2 domain over the whole row, row reductions with v_permlane32/16_swap.
//     Keys are stored transposed inside their 16-key tile so that the padded k-steps of the last tile are skipped.
//     The tile class (2 / 4 / 7 / 9 tiles = the region-batching levels 30 / 60 / 100 / 144 tokens) is chosen per
//     workgroup inside ONE launch; the largest windows are dispatched first.  Backward: sra_bwd_dq_k (dQ, also
//     emits rowsum(dO * O)) and sra_bwd_dkv_k (dK, dV for <= 4 key tiles per workgroup).
//     Measured: 53-55 us per launch at 90 k tokens = 43 % of the 8 TB/s HBM roof; MFMA pipe 36 % busy, 2.5 of 4
//     waves resident per SIMD on average: bound by instruction issue / the dependent chain of a wave.
//   impl 2 LDS-staged (the first implementation, kept for comparison): K and V rows of a 4-head group gathered
//     into LDS (row stride 68 floats: conflict-free b32 column reads, <= 2-way b128 row reads), a wave owns
//     (head, 16-query tile) tasks; one launch per tile class.  113 us per call.
//   impl 1 plain VALU kernel (one thread per (query, head), online softmax): windows above 144 tokens and the
//     in-library cross-check.
// Arithmetic intensity ~0.25*T flop/B puts the fp32 kernel at the HBM / fp32-MFMA ridge.
#include <math.h>
#include <stdlib.h>
#include <hip/hip_ext.h>
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kHD = 16;     // head dim
constexpr int kGH = 4;      // heads per workgroup
constexpr int kGC = 64;     // channels per workgroup (kGH * kHD)
#ifndef SST_WAVE_HEADS
#define SST_WAVE_HEADS 4
#endif
// Block -> window order of the register-resident kernels.  Region batching lays the windows out level by level,
// smallest token class first; reversed, the largest windows are dispatched first (longest-processing-time
// first), which shortens the tail of the launch.
#ifndef SST_SRA_REVERSE
#define SST_SRA_REVERSE 1
#endif
#if SST_SRA_REVERSE
#define SST_SRA_BLOCK(b, n) ((int)(n) - 1 - (int)(b))
#else
#define SST_SRA_BLOCK(b, n) ((int)(b))
#endif
// one-shot measurement hooks (sst_sra_attn_profile_next_fwd / _bwd): per calling thread, consumed by the next launch
thread_local hipEvent_t g_prof_start = nullptr, g_prof_stop = nullptr;
thread_local hipEvent_t g_prof_bwd_start = nullptr, g_prof_bwd_stop = nullptr;
// launch order of the windows of the CURRENT call (sst_sra_attn_{fwd,bwd}_ord_f32 set it around the plain entry points):
// workgroup position p handles window order[p]; nullptr = window p
thread_local const int32_t* g_win_order = nullptr;
// cosine attention of the CURRENT call (sst_sra_attn_cos_{fwd,bwd}_f32 set them around the plain entry points): per-head score
// scale 1 / clamp(tau, tau_min) in device memory (the parameter never visits the host), and the backward's per-(token, head)
// output q^ . dq^ (the gradient of the scale is its column sum); nullptr = standard attention
thread_local const float* g_head_scale = nullptr;
thread_local float* g_cos_r = nullptr;
constexpr int kWH = SST_WAVE_HEADS;  // heads (= waves) per workgroup of the register-resident kernels
constexpr int kRS = 68;     // LDS row stride (floats)
constexpr int kMaxTilesMfma = 9;

__device__ __forceinline__ f32x4 mfma4(const float4 a, const float4 b, f32x4 acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
  return acc;
}

// ------------------------------------------------------------------------------------------------
// generic VALU kernels
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sra_fwd_generic_k(const float* __restrict__ Q, const float* __restrict__ K,
                                                         const float* __restrict__ V, int64_t ldq, int64_t ldk,
                                                         int64_t ldv, const int32_t* __restrict__ tok,
                                                         const int32_t* __restrict__ winoff, int H, float scale,
                                                         int min_tokens_excl, float* __restrict__ O, int64_t ldo,
                                                         float* __restrict__ LSE) {
  const int w = blockIdx.x;
  const int beg = winoff[w];
  const int t = winoff[w + 1] - beg;
  if (t <= min_tokens_excl) return;
  for (int p = threadIdx.x; p < t * H; p += blockDim.x) {
    const int qi = p / H, h = p - qi * H;
    const int64_t row = tok[beg + qi];
    float q[kHD], acc[kHD];
#pragma unroll
    for (int d = 0; d < kHD; ++d) {
      q[d] = Q[row * ldq + h * kHD + d] * scale;
      acc[d] = 0.f;
    }
    float m = -INFINITY, l = 0.f;
    for (int kk = 0; kk < t; ++kk) {
      const int64_t kr = tok[beg + kk];
      const float* kp = K + kr * ldk + h * kHD;
      const float* vp = V + kr * ldv + h * kHD;
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < kHD; ++d) s = fmaf(q[d], kp[d], s);
      const float mn = fmaxf(m, s);
      const float a = expf(m - mn);
      const float pe = expf(s - mn);
      l = l * a + pe;
#pragma unroll
      for (int d = 0; d < kHD; ++d) acc[d] = acc[d] * a + pe * vp[d];
      m = mn;
    }
    const float inv = 1.f / l;
#pragma unroll
    for (int d = 0; d < kHD; ++d) O[row * ldo + h * kHD + d] = acc[d] * inv;
    LSE[row * H + h] = m + logf(l);
  }
}

// dK / dV rows of the windows this kernel touches must be zero on entry (float atomics).
__global__ __launch_bounds__(256) void sra_bwd_generic_k(
    const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
    const float* __restrict__ O, const float* __restrict__ dO, const float* __restrict__ LSE, int64_t ldq,
    int64_t ldk, int64_t ldv, int64_t ldo, int64_t lddo, const int32_t* __restrict__ tok,
    const int32_t* __restrict__ winoff, int H, float scale, int min_tokens_excl, float* __restrict__ dQ,
    float* __restrict__ dK, float* __restrict__ dV, int64_t lddq, int64_t lddk, int64_t lddv) {
  const int w = blockIdx.x;
  const int beg = winoff[w];
  const int t = winoff[w + 1] - beg;
  if (t <= min_tokens_excl) return;
  for (int p = threadIdx.x; p < t * H; p += blockDim.x) {
    const int qi = p / H, h = p - qi * H;
    const int64_t row = tok[beg + qi];
    float q[kHD], go[kHD], dq[kHD];
    float D = 0.f;
#pragma unroll
    for (int d = 0; d < kHD; ++d) {
      q[d] = Q[row * ldq + h * kHD + d];
      go[d] = dO[row * lddo + h * kHD + d];
      D = fmaf(go[d], O[row * ldo + h * kHD + d], D);
      dq[d] = 0.f;
    }
    const float lse = LSE[row * H + h];
    for (int kk = 0; kk < t; ++kk) {
      const int64_t kr = tok[beg + kk];
      const float* kp = K + kr * ldk + h * kHD;
      const float* vp = V + kr * ldv + h * kHD;
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < kHD; ++d) {
        s = fmaf(q[d], kp[d], s);
        dp = fmaf(go[d], vp[d], dp);
      }
      const float pe = expf(s * scale - lse);
      const float ds = pe * (dp - D) * scale;
#pragma unroll
      for (int d = 0; d < kHD; ++d) {
        dq[d] = fmaf(ds, kp[d], dq[d]);
        atomicAdd(dK + kr * lddk + h * kHD + d, ds * q[d]);
        atomicAdd(dV + kr * lddv + h * kHD + d, pe * go[d]);
      }
    }
#pragma unroll
    for (int d = 0; d < kHD; ++d) dQ[row * lddq + h * kHD + d] = dq[d];
  }
}

// ------------------------------------------------------------------------------------------------
// MFMA forward.  Fragment vocabulary (lane = 16*g + c, g = 0..3, c = 0..15):
//   row-frag  X[base + c][h*16 + 4g .. 4g+3]  (float4)  -> A or B operand of a product contracted over d
//   col-frag  X[base + 4g + r][h*16 + c], r = 0..3      -> operand of a product contracted over tokens
//   D layout  value r of lane (g,c) = D[row 4g + r][col c]
// ------------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(256) void sra_fwd_mfma_k(const float* __restrict__ Q, const float* __restrict__ K,
                                                      const float* __restrict__ V, int64_t ldq, int64_t ldk,
                                                      int64_t ldv, const int32_t* __restrict__ tok,
                                                      const int32_t* __restrict__ winoff, int n_groups, int H,
                                                      float scale, int nt_lo, float* __restrict__ O, int64_t ldo,
                                                      float* __restrict__ LSE) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int w = blockIdx.x / n_groups;
  const int hg = blockIdx.x - w * n_groups;
  const int beg = winoff[w];
  const int t = winoff[w + 1] - beg;
  const int nt = (t + 15) >> 4;
  if (nt <= nt_lo || nt > NT) return;  // another variant owns this window

  float* Ks = smem;
  float* Vs = Ks + NT * 16 * kRS;
  int* toks = (int*)(Vs + NT * 16 * kRS);
  const int tid = threadIdx.x;
  {
    const int c4 = tid & 15;
    for (int r = tid >> 4; r < nt * 16; r += 16) {
      float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
      int row = -1;
      if (r < t) {
        row = tok[beg + r];
        kv = *(const float4*)(K + (int64_t)row * ldk + hg * kGC + c4 * 4);
        vv = *(const float4*)(V + (int64_t)row * ldv + hg * kGC + c4 * 4);
      }
      *(float4*)(Ks + r * kRS + c4 * 4) = kv;
      *(float4*)(Vs + r * kRS + c4 * 4) = vv;
      if (c4 == 0) toks[r] = row;
    }
  }
  __syncthreads();

  const int wave = tid >> 6, lane = tid & 63, g = lane >> 4, c = lane & 15;
  for (int task = wave; task < kGH * nt; task += 4) {
    const int h = task & 3, i = task >> 2;
    const int qrow = toks[i * 16 + c];
    float4 qf = make_float4(0.f, 0.f, 0.f, 0.f);
    if (qrow >= 0) {
      qf = *(const float4*)(Q + (int64_t)qrow * ldq + hg * kGC + h * kHD + 4 * g);
      qf.x *= scale;
      qf.y *= scale;
      qf.z *= scale;
      qf.w *= scale;
    }
    // S^T tiles: st[j][r] = S[query i*16+c][key j*16+4g+r]
    f32x4 st[NT];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      if (j < nt) {
        const float4 kf = *(const float4*)(Ks + (j * 16 + c) * kRS + h * kHD + 4 * g);
        acc = mfma4(kf, qf, acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = (j * 16 + 4 * g + r) < t ? acc[r] : -INFINITY;
          acc[r] = v;
          mx = fmaxf(mx, v);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = -INFINITY;
      }
      st[j] = acc;
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pe = __expf(st[j][r] - mx);
        st[j][r] = pe;
        sum += pe;
      }
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    // O^T[d][query] += V^T[d][key] P^T[key][query]
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      if (j < nt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float a = Vs[(j * 16 + 4 * g + r) * kRS + h * kHD + c];
          o = __builtin_amdgcn_mfma_f32_16x16x4f32(a, st[j][r], o, 0, 0, 0);
        }
      }
    }
    if (qrow >= 0) {
      const float inv = 1.f / sum;
      const float4 ov = make_float4(o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv);
      *(float4*)(O + (int64_t)qrow * ldo + hg * kGC + h * kHD + 4 * g) = ov;
      if (g == 0) LSE[(int64_t)qrow * H + hg * kGH + h] = mx + __logf(sum);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// MFMA forward, register-resident variant ("wave = window x head"): no LDS, no barrier.
// A workgroup is still (window, 4-head group) but each of its 4 waves owns ONE head: it loads that head's
// K row-fragments (one float4 per 16-key tile) and V column-fragments (4 dwords per tile) straight from
// HBM/L2 into the MFMA operand layout and keeps them in VGPRs for all query tiles of the window
// (<= 36 + 36 VGPRs at 144 tokens).  The 4 waves of a block touch adjacent 64 B segments of the same token
// rows at the same time, so every 128 B line is fetched once per CU.  Token row ids are held 64 per
// register and broadcast with ds_bpermute (__shfl).  Q tiles are prefetched one iteration ahead.
// ------------------------------------------------------------------------------------------------
// 32-bit element offsets off a uniform base pointer (saddr + voffset addressing, no 64-bit VALU math);
// the host checks that every tensor spans < 2^31 elements before choosing these kernels.
__device__ __forceinline__ float4 ldg4(const float* __restrict__ base, uint32_t off) {
  return *(const float4*)(base + off);
}
__device__ __forceinline__ float ldg1(const float* __restrict__ base, uint32_t off) { return base[off]; }

// Reductions across the four 16-lane rows of a wave (lanes c, c+16, c+32, c+48 hold partial results of the same
// query) with the gfx950 VALU lane swaps v_permlane32_swap / v_permlane16_swap instead of LDS-pipe ds_bpermute.
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
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

// 1 / max(|x|_2, 1e-12) of the 16-channel head row whose four float4 pieces sit on lanes c, c+16, c+32, c+48
// (torch.nn.functional.normalize(x, dim=-1), cosine_msa.py:159-160), on every lane of the column
__device__ __forceinline__ float rowfrag_inv_norm(const float4 x) {
  const float n2 = rows4_sum(fmaf(x.x, x.x, fmaf(x.y, x.y, fmaf(x.z, x.z, x.w * x.w))));
  // 1 / max(sqrt(n2), 1e-12) = rsq(max(n2, 1e-24)): one v_rsq_f32 (1 ulp) instead of an IEEE square root and a division -
  // ~25 VALU instructions per row fragment in a kernel that is bound by instruction issue
  return __builtin_amdgcn_rsqf(fmaxf(n2, 1e-24f));
}
__device__ __forceinline__ float4 scale4(const float4 x, const float s) {
  return make_float4(x.x * s, x.y * s, x.z * s, x.w * s);
}
__device__ __forceinline__ float dot4(const float4 a, const float4 b) {
  return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w)));
}

constexpr float kLog2e = 1.4426950408889634f;
constexpr int kSplitTiles = 4;    // windows of this many 16-token tiles and more are dealt out over several workgroups in small launches
constexpr float kLn2 = 0.6931471805599453f;

// COS: scaled cosine attention (cosine_msa.py:123-170) - q and k rows are normalised per head as they are loaded (one
// 4-lane reduction each) and the scores are scaled by hscale[head] = 1 / clamp(tau, tau_min) instead of `scale`
template <int NT, bool COS>
__device__ __forceinline__ void sra_fwd_wave_body(const float* __restrict__ Q, const float* __restrict__ K,
                                                  const float* __restrict__ V, uint32_t ldq, uint32_t ldk,
                                                  uint32_t ldv, const int32_t* __restrict__ tok, int beg, int t, int nt,
                                                  int hg, int H, float scale, float* __restrict__ O, uint32_t ldo,
                                                  float* __restrict__ LSE, const float* __restrict__ hscale,
                                                  int part = 0, int parts = 1) {
  // part / parts: the query tiles part, part + parts, ... of the window (a large window of a SMALL launch is dealt out over
  // several workgroups: its one wave per head is otherwise the critical path of the whole launch - see sra_fwd_wave_k)
  const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  const int head = hg * kWH + (threadIdx.x >> 6);
  if (COS) scale = hscale[head];
  const uint32_t hoff = head * kHD;
  constexpr int NTK = (NT * 16 + 63) / 64;
  // token ids, 64 window positions per register; positions past the window repeat its last token so that
  // every load below is unconditional (padded keys are masked, padded queries are never stored)
  int tk[NTK];
#pragma unroll
  for (int i = 0; i < NTK; ++i) {
    const int p = i * 64 + lane;
    tk[i] = tok != nullptr ? tok[beg + (p < t ? p : t - 1)] : beg + (p < t ? p : t - 1);   // no list: the window's rows are beg .. beg + t - 1
  }
  float4 kf[NT];
  float vf[NT][4];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    if (j < nt) {
      // Keys sit in their 16-key tile TRANSPOSED as a 4 x 4 index grid: tile position p holds key (p >> 2) + 4 * (p & 3).
      // MFMA k-step r of P V then covers the four CONSECUTIVE keys 4r..4r+3 (position 4g + r <-> key g + 4r), so in
      // the last, partially filled tile the steps that would only see padded keys are skipped altogether.
      const uint32_t krow = (uint32_t)__shfl(tk[j >> 2], (j & 3) * 16 + (c >> 2) + 4 * (c & 3), 64);
      kf[j] = ldg4(K, krow * ldk + hoff + 4 * g);
      if (COS) kf[j] = scale4(kf[j], rowfrag_inv_norm(kf[j]));
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const uint32_t vrow = (uint32_t)__shfl(tk[j >> 2], (j & 3) * 16 + g + 4 * r, 64);
        vf[j][r] = ldg1(V, vrow * ldv + hoff + c);
      }
    } else {
      kf[j] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int r = 0; r < 4; ++r) vf[j][r] = 0.f;
    }
  }
  auto q_row = [&](int i) -> uint32_t {
    int sel = tk[0];
#pragma unroll
    for (int u = 1; u < NTK; ++u) sel = ((i >> 2) == u) ? tk[u] : sel;
    return (uint32_t)__shfl(sel, (i & 3) * 16 + c, 64);
  };
  const float qscale = scale * kLog2e;  // scores in the log2 domain: softmax via v_exp_f32 directly
  const int last_steps = (t - 16 * (nt - 1) + 3) >> 2;  // k-steps of the last key tile that contain a real key (1..4)
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  // S^T tiles of one query tile: st[j][r] = S[query 16i+c][key 16j+4g+r]; k-step major so that consecutive
  // MFMAs are independent
  auto qk_tiles = [&](const float4 q, f32x4 (&st)[NT]) {
    const float qs = COS ? qscale * rowfrag_inv_norm(q) : qscale;
    const float qx = q.x * qs, qy = q.y * qs, qz = q.z * qs, qw = q.w * qs;
#pragma unroll
    for (int j = 0; j < NT; ++j)
      if (j < nt) st[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[j].x, qx, zero4, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NT; ++j)
      if (j < nt) st[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[j].y, qy, st[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NT; ++j)
      if (j < nt) st[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[j].z, qz, st[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NT; ++j)
      if (j < nt) st[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[j].w, qw, st[j], 0, 0, 0);
  };
  // softmax of the tile in registers + O^T = V^T P^T + store
  auto finish_tile = [&](f32x4 (&st)[NT], int i, uint32_t qrow) {
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      if (j < nt) {
        if (j == nt - 1) {  // only the last tile can hold padded keys
#pragma unroll
          for (int r = 0; r < 4; ++r) st[j][r] = (j * 16 + g + 4 * r) < t ? st[j][r] : -INFINITY;
        }
        mx = fmaxf(mx, fmaxf(fmaxf(st[j][0], st[j][1]), fmaxf(st[j][2], st[j][3])));
      }
    }
    mx = rows4_max(mx);
    float sm[4] = {0.f, 0.f, 0.f, 0.f};  // four independent partial sums (no serial add chain)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      if (j < nt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pe = __builtin_amdgcn_exp2f(st[j][r] - mx);
          st[j][r] = pe;
          sm[r] += pe;
        }
      }
    }
    const float sum = rows4_sum((sm[0] + sm[1]) + (sm[2] + sm[3]));
    // one accumulator per k-step -> no dependent back-to-back MFMAs
    f32x4 o0 = zero4, o1 = zero4, o2 = zero4, o3 = zero4;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      if (j < nt) {
        const int steps = (j == nt - 1) ? last_steps : 4;  // wave-uniform
        o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[j][0], st[j][0], o0, 0, 0, 0);
        if (steps > 1) o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[j][1], st[j][1], o1, 0, 0, 0);
        if (steps > 2) o2 = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[j][2], st[j][2], o2, 0, 0, 0);
        if (steps > 3) o3 = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[j][3], st[j][3], o3, 0, 0, 0);
      }
    }
    if (i * 16 + c < t) {
      const float inv = __builtin_amdgcn_rcpf(sum);
      const float4 ov = make_float4((o0[0] + o1[0] + o2[0] + o3[0]) * inv, (o0[1] + o1[1] + o2[1] + o3[1]) * inv,
                                    (o0[2] + o1[2] + o2[2] + o3[2]) * inv, (o0[3] + o1[3] + o2[3] + o3[3]) * inv);
      *(float4*)(O + (qrow * ldo + hoff + 4 * g)) = ov;
      if (g == 0) LSE[qrow * (uint32_t)H + head] = (mx + __builtin_amdgcn_logf(sum)) * kLn2;
    }
  };
  // (A two-deep software pipeline over query tiles — QK^T of tile i+1 issued ahead of the softmax of tile i —
  // was measured slower: +34 VGPRs drop the occupancy from 4 to 3 waves/SIMD, 61.8 vs 56.2 us.)
  f32x4 st[NT];
  if (part >= nt) return;
  uint32_t qrow = q_row(part);
  float4 qf = ldg4(Q, qrow * ldq + hoff + 4 * g);
  for (int i = part; i < nt; i += parts) {
    const uint32_t qrow_next = q_row(i + parts < nt ? i + parts : i);
    const float4 qf_next = ldg4(Q, qrow_next * ldq + hoff + 4 * g);  // prefetch
    qk_tiles(qf, st);
    finish_tile(st, i, qrow);
    qrow = qrow_next;
    qf = qf_next;
  }
}

// One launch for every window: the tile class is picked per workgroup, so all classes run concurrently
// (separate per-class launches serialise on the stream and the sparse classes run at very low occupancy).
template <int NTMAX, bool COS>
// (second launch bound = waves per SIMD the register allocation aims at: the 5-tile classes and the standard 4-tile class fit 95-96
// VGPRs without a spill when asked to - five waves per SIMD instead of four at 100; same-box A/B on the bench frame: 50.9 -> 50.2 us
// per launch.  The cosine 4-tile class would spill 12 bytes per lane: left alone.)
__global__ __launch_bounds__(64 * kWH, ((NTMAX == 5 || (NTMAX <= 4 && !COS)) ? 5 : 1)) void sra_fwd_wave_k(const float* __restrict__ Q, const float* __restrict__ K,
                                                      const float* __restrict__ V, int64_t ldq, int64_t ldk,
                                                      int64_t ldv, const int32_t* __restrict__ tok,
                                                      const int32_t* __restrict__ winoff, int n_groups, int H,
                                                      float scale, float* __restrict__ O, int64_t ldo,
                                                      float* __restrict__ LSE, const int32_t* __restrict__ order,
                                                      const float* __restrict__ hscale, int n_win, int qs) {
  // Small launches (qs > 1): the grid carries (qs - 1) * n_groups further workgroups per
  // window IN FRONT of the regular ones; those of a window of kSplitTiles tiles and more take the query tiles
  // part, part + qs, ... of it, the others leave at once.  Why: a 100-token window is 7 dependent query tiles on ONE wave per
  // head, ~11 us - on a LiDAR sweep (18 k voxels, 23 tokens per window on average, a few at the cap) the whole launch waited
  // for those waves (19 us for 38 MB; tools/sra_sizes.py).
  const int n_main = n_win * n_groups;
  const int n_extra = (int)gridDim.x - n_main;
  int part = 0, wpos, hg;
  if ((int)blockIdx.x < n_extra) {
    const int per = n_groups * (qs - 1);
    const int from_end = (int)blockIdx.x / per, r = (int)blockIdx.x - from_end * per;
    part = 1 + r / n_groups;
    hg = r % n_groups;
    wpos = n_win - 1 - from_end;      // with a launch order (ascending) the largest windows sit at the end: first to go
  } else {
    const int bid = SST_SRA_BLOCK((int)blockIdx.x - n_extra, n_main);
    wpos = bid / n_groups;
    hg = bid - wpos * n_groups;
  }
  const int w = order != nullptr ? order[wpos] : wpos;
  const int beg = winoff[w];
  const int t = winoff[w + 1] - beg;
  const int nt = (t + 15) >> 4;
  if (nt < 1 || nt > NTMAX) return;  // > NTMAX: the generic kernel owns this window
  int parts = 1;
  if (qs > 1) {
    if (nt >= kSplitTiles) parts = qs;
    else if (part > 0) return;
  }
  const uint32_t q_ld = (uint32_t)ldq, k_ld = (uint32_t)ldk, v_ld = (uint32_t)ldv, o_ld = (uint32_t)ldo;
  if (nt <= 2)
    sra_fwd_wave_body<2, COS>(Q, K, V, q_ld, k_ld, v_ld, tok, beg, t, nt, hg, H, scale, O, o_ld, LSE, hscale);
  else if (nt <= 4)
    sra_fwd_wave_body<4, COS>(Q, K, V, q_ld, k_ld, v_ld, tok, beg, t, nt, hg, H, scale, O, o_ld, LSE, hscale, part, parts);
  else if (nt <= 7 || NTMAX <= 7)
    sra_fwd_wave_body<(NTMAX < 7 ? NTMAX : 7), COS>(Q, K, V, q_ld, k_ld, v_ld, tok, beg, t, nt, hg, H, scale, O, o_ld, LSE,
                                                    hscale, part, parts);
  else
    sra_fwd_wave_body<NTMAX, COS>(Q, K, V, q_ld, k_ld, v_ld, tok, beg, t, nt, hg, H, scale, O, o_ld, LSE, hscale, part, parts);
}

// ------------------------------------------------------------------------------------------------
// MFMA backward.  Phase A: a wave owns (head, key tile) and accumulates dK, dV over query tiles.
// Phase B: a wave owns (head, query tile) and accumulates dQ over key tiles (S is recomputed in the
// transposed orientation so that dS lands in the A-operand layout of dS K).
// ------------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(256) void sra_bwd_mfma_k(
    const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
    const float* __restrict__ O, const float* __restrict__ dO, const float* __restrict__ LSE, int64_t ldq,
    int64_t ldk, int64_t ldv, int64_t ldo, int64_t lddo, const int32_t* __restrict__ tok,
    const int32_t* __restrict__ winoff, int n_groups, int H, float scale, int nt_lo, float* __restrict__ dQ,
    float* __restrict__ dK, float* __restrict__ dV, int64_t lddq, int64_t lddk, int64_t lddv) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int w = blockIdx.x / n_groups;
  const int hg = blockIdx.x - w * n_groups;
  const int beg = winoff[w];
  const int t = winoff[w + 1] - beg;
  const int nt = (t + 15) >> 4;
  if (nt <= nt_lo || nt > NT) return;

  constexpr int kTile = NT * 16 * kRS;
  float* Qs = smem;
  float* Ks = Qs + kTile;
  float* Vs = Ks + kTile;
  float* Gs = Vs + kTile;              // dO
  float* Ls = Gs + kTile;              // [NT*16][4] log-sum-exp
  float* Ds = Ls + NT * 16 * kGH;      // [NT*16][4] rowsum(dO * O)
  int* toks = (int*)(Ds + NT * 16 * kGH);
  const int tid = threadIdx.x;
  {
    const int c4 = tid & 15;
    const int hh = c4 >> 2;  // head of this float4
    for (int r = tid >> 4; r < nt * 16; r += 16) {
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      float4 qv = z, kv = z, vv = z, gv = z, ov = z;
      int row = -1;
      if (r < t) {
        row = tok[beg + r];
        const int co = hg * kGC + c4 * 4;
        qv = *(const float4*)(Q + (int64_t)row * ldq + co);
        kv = *(const float4*)(K + (int64_t)row * ldk + co);
        vv = *(const float4*)(V + (int64_t)row * ldv + co);
        gv = *(const float4*)(dO + (int64_t)row * lddo + co);
        ov = *(const float4*)(O + (int64_t)row * ldo + co);
      }
      *(float4*)(Qs + r * kRS + c4 * 4) = qv;
      *(float4*)(Ks + r * kRS + c4 * 4) = kv;
      *(float4*)(Vs + r * kRS + c4 * 4) = vv;
      *(float4*)(Gs + r * kRS + c4 * 4) = gv;
      float dpart = gv.x * ov.x + gv.y * ov.y + gv.z * ov.z + gv.w * ov.w;
      dpart += __shfl_xor(dpart, 1, 64);
      dpart += __shfl_xor(dpart, 2, 64);
      if ((c4 & 3) == 0) {
        Ds[r * kGH + hh] = dpart;
        Ls[r * kGH + hh] = (row >= 0) ? LSE[(int64_t)row * H + hg * kGH + hh] : 0.f;
      }
      if (c4 == 0) toks[r] = row;
    }
  }
  __syncthreads();

  const int wave = tid >> 6, lane = tid & 63, g = lane >> 4, c = lane & 15;

  // ---- phase A: dK, dV ----
  for (int task = wave; task < kGH * nt; task += 4) {
    const int h = task & 3, j = task >> 2;
    const float4 kf = *(const float4*)(Ks + (j * 16 + c) * kRS + h * kHD + 4 * g);
    const float4 vf = *(const float4*)(Vs + (j * 16 + c) * kRS + h * kHD + 4 * g);
    const bool key_ok = (j * 16 + c) < t;
    f32x4 dk = {0.f, 0.f, 0.f, 0.f}, dv = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < nt; ++i) {
      const float4 qf = *(const float4*)(Qs + (i * 16 + c) * kRS + h * kHD + 4 * g);
      const float4 gf = *(const float4*)(Gs + (i * 16 + c) * kRS + h * kHD + 4 * g);
      f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
      s = mfma4(qf, kf, s);    // S[query i*16+4g+r][key j*16+c]
      dp = mfma4(gf, vf, dp);  // dP same layout
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qi = i * 16 + 4 * g + r;
        const bool ok = key_ok && (qi < t);
        const float pe = ok ? __expf(s[r] * scale - Ls[qi * kGH + h]) : 0.f;
        const float ds = pe * (dp[r] - Ds[qi * kGH + h]) * scale;
        const float gcol = Gs[qi * kRS + h * kHD + c];  // dO[query qi][d = c]
        const float qcol = Qs[qi * kRS + h * kHD + c];  // Q[query qi][d = c]
        dv = __builtin_amdgcn_mfma_f32_16x16x4f32(pe, gcol, dv, 0, 0, 0);  // dV[key c][d] += P^T dO
        dk = __builtin_amdgcn_mfma_f32_16x16x4f32(ds, qcol, dk, 0, 0, 0);  // dK[key c][d] += dS^T Q
      }
    }
    // D layout: value r = d{K,V}[key j*16 + 4g + r][d = c]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int krow = toks[j * 16 + 4 * g + r];
      if (krow >= 0) {
        dK[(int64_t)krow * lddk + hg * kGC + h * kHD + c] = dk[r];
        dV[(int64_t)krow * lddv + hg * kGC + h * kHD + c] = dv[r];
      }
    }
  }

  // ---- phase B: dQ ----
  for (int task = wave; task < kGH * nt; task += 4) {
    const int h = task & 3, i = task >> 2;
    const float4 qf = *(const float4*)(Qs + (i * 16 + c) * kRS + h * kHD + 4 * g);
    const float4 gf = *(const float4*)(Gs + (i * 16 + c) * kRS + h * kHD + 4 * g);
    const int qi = i * 16 + c;
    const bool q_ok = qi < t;
    const float lse = Ls[qi * kGH + h];
    const float dd = Ds[qi * kGH + h];
    f32x4 dq = {0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < nt; ++j) {
      const float4 kf = *(const float4*)(Ks + (j * 16 + c) * kRS + h * kHD + 4 * g);
      const float4 vf = *(const float4*)(Vs + (j * 16 + c) * kRS + h * kHD + 4 * g);
      f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
      s = mfma4(kf, qf, s);    // S^T[key j*16+4g+r][query i*16+c]
      dp = mfma4(vf, gf, dp);  // dP^T same layout
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kj = j * 16 + 4 * g + r;
        const bool ok = q_ok && (kj < t);
        const float pe = ok ? __expf(s[r] * scale - lse) : 0.f;
        const float ds = pe * (dp[r] - dd) * scale;
        const float kcol = Ks[kj * kRS + h * kHD + c];  // K[key kj][d = c]
        dq = __builtin_amdgcn_mfma_f32_16x16x4f32(ds, kcol, dq, 0, 0, 0);  // dQ[query c][d] += dS K
      }
    }
    // D layout: value r = dQ[query i*16 + 4g + r][d = c]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qrow = toks[i * 16 + 4 * g + r];
      if (qrow >= 0) dQ[(int64_t)qrow * lddq + hg * kGC + h * kHD + c] = dq[r];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// MFMA backward, register-resident ("wave = window x head"), two launches:
//   sra_bwd_dq_k   : per query tile, S^T = K Q^T and dP^T = V dO^T (row = key, col = query), dS^T, then
//                    dQ += dS K with the K column-fragments held in registers; also emits
//                    D[token, head] = rowsum(dO * O) for the second kernel.
//   sra_bwd_dkv_k  : per query tile, S = Q K^T and dP = dO V^T (row = query, col = key); P and dS are then
//                    directly the A operands of dV += P^T dO and dK += dS^T Q, accumulated in registers
//                    over all query tiles of the window.
// ------------------------------------------------------------------------------------------------
// two independent 4-step MFMA chains interleaved (S and dP of one tile pair)
__device__ __forceinline__ void mfma4x2(const float4 a0, const float4 b0, f32x4& d0, const float4 a1, const float4 b1,
                                        f32x4& d1) {
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b0.x, z, 0, 0, 0);
  d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b1.x, z, 0, 0, 0);
  d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b0.y, d0, 0, 0, 0);
  d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b1.y, d1, 0, 0, 0);
  d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b0.z, d0, 0, 0, 0);
  d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b1.z, d1, 0, 0, 0);
  d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b0.w, d0, 0, 0, 0);
  d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b1.w, d1, 0, 0, 0);
}

template <int NT>
__device__ __forceinline__ void sra_bwd_dq_body(const float* __restrict__ Q, const float* __restrict__ K,
                                                const float* __restrict__ V, const float* __restrict__ O,
                                                const float* __restrict__ dO, const float* __restrict__ LSE,
                                                uint32_t ldq, uint32_t ldk, uint32_t ldv, uint32_t ldo, uint32_t lddo,
                                                const int32_t* __restrict__ tok, int beg, int t, int nt, int hg, int H,
                                                float scale, float* __restrict__ dQ, uint32_t lddq,
                                                float* __restrict__ Dbuf) {
  const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  const int head = hg * kWH + (threadIdx.x >> 6);
  const uint32_t hoff = head * kHD;
  constexpr int NTK = (NT * 16 + 63) / 64;
  int tk[NTK];
#pragma unroll
  for (int i = 0; i < NTK; ++i) {
    const int p = i * 64 + lane;
    tk[i] = tok != nullptr ? tok[beg + (p < t ? p : t - 1)] : beg + (p < t ? p : t - 1);   // no list: the window's rows are beg .. beg + t - 1
  }
  float4 kf[NT], vf[NT];
  float kc[NT][4];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    if (j < nt) {
      // keys transposed inside their tile (position p <-> key (p >> 2) + 4 * (p & 3)) as in the forward kernel: k-step r
      // of dQ += dS K covers the consecutive keys 4r..4r+3, padded steps of the last tile are skipped
      const uint32_t krow = (uint32_t)__shfl(tk[j >> 2], (j & 3) * 16 + (c >> 2) + 4 * (c & 3), 64);
      kf[j] = ldg4(K, krow * ldk + hoff + 4 * g);
      vf[j] = ldg4(V, krow * ldv + hoff + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const uint32_t crow = (uint32_t)__shfl(tk[j >> 2], (j & 3) * 16 + g + 4 * r, 64);
        kc[j][r] = ldg1(K, crow * ldk + hoff + c);
      }
    } else {
      kf[j] = vf[j] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int r = 0; r < 4; ++r) kc[j][r] = 0.f;
    }
  }
  auto tok_at = [&](int i, int within) -> uint32_t {  // token id of window position 16*i + within
    int sel = tk[0];
#pragma unroll
    for (int u = 1; u < NTK; ++u) sel = ((i >> 2) == u) ? tk[u] : sel;
    return (uint32_t)__shfl(sel, (i & 3) * 16 + within, 64);
  };
  const float s2 = scale * kLog2e;
  const int last_steps = (t - 16 * (nt - 1) + 3) >> 2;  // k-steps of the last key tile that hold a real key
  for (int i = 0; i < nt; ++i) {
    const uint32_t qrow = tok_at(i, c);
    const float4 qf = ldg4(Q, qrow * ldq + hoff + 4 * g);
    const float4 gf = ldg4(dO, qrow * lddo + hoff + 4 * g);
    const float4 of = ldg4(O, qrow * ldo + hoff + 4 * g);
    const float lse2 = LSE[qrow * (uint32_t)H + head] * kLog2e;
    const bool q_ok = (i * 16 + c) < t;
    float dd = gf.x * of.x + gf.y * of.y + gf.z * of.z + gf.w * of.w;
    dd = rows4_sum(dd);
    if (q_ok && g == 0) Dbuf[qrow * (uint32_t)H + head] = dd;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 dq0 = zero4, dq1 = zero4, dq2 = zero4, dq3 = zero4;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      if (j < nt) {
        f32x4 s, dp;
        mfma4x2(kf[j], qf, s, vf[j], gf, dp);  // S^T / dP^T [key 16j+4g+r][query 16i+c]
        float ds[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float pe = __builtin_amdgcn_exp2f(fmaf(s[r], s2, -lse2));
          if (j == nt - 1) pe = (j * 16 + g + 4 * r) < t ? pe : 0.f;  // padded keys
          ds[r] = pe * (dp[r] - dd) * scale;
        }
        const int steps = (j == nt - 1) ? last_steps : 4;  // wave-uniform
        dq0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ds[0], kc[j][0], dq0, 0, 0, 0);
        if (steps > 1) dq1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ds[1], kc[j][1], dq1, 0, 0, 0);
        if (steps > 2) dq2 = __builtin_amdgcn_mfma_f32_16x16x4f32(ds[2], kc[j][2], dq2, 0, 0, 0);
        if (steps > 3) dq3 = __builtin_amdgcn_mfma_f32_16x16x4f32(ds[3], kc[j][3], dq3, 0, 0, 0);
      }
    }
    // D layout: value r = dQ[query 16i + 4g + r][d = c]   (rows of padded queries are never stored)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint32_t orow = tok_at(i, 4 * g + r);  // cross-lane read: must run with all lanes active
      if (i * 16 + 4 * g + r < t) dQ[orow * lddq + hoff + c] = (dq0[r] + dq1[r]) + (dq2[r] + dq3[r]);
    }
  }
}

// dK/dV of up to 4 key tiles (j0 .. j0+3) of one head, accumulated over ALL query tiles of the window.
// Windows with more than 4 key tiles are covered by several workgroups (key tiles are independent), which
// keeps the accumulators + fragments at 64 VGPRs and the occupancy at 4 waves/SIMD.
template <int NTW>  // NTW: compile-time bound on the window's tile count (sizes the token-id registers)
__device__ __forceinline__ void sra_bwd_dkv_body(const float* __restrict__ Q, const float* __restrict__ K,
                                                 const float* __restrict__ V, const float* __restrict__ dO,
                                                 const float* __restrict__ LSE, const float* __restrict__ Dbuf,
                                                 uint32_t ldq, uint32_t ldk, uint32_t ldv, uint32_t lddo,
                                                 const int32_t* __restrict__ tok, int beg, int t, int nt, int j0,
                                                 int hg, int H, float scale, float* __restrict__ dK,
                                                 float* __restrict__ dV, uint32_t lddk, uint32_t lddv) {
  constexpr int NJ = 4;
  const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  const int head = hg * kWH + (threadIdx.x >> 6);
  const uint32_t hoff = head * kHD;
  constexpr int NTK = (NTW * 16 + 63) / 64;
  int tk[NTK];
#pragma unroll
  for (int i = 0; i < NTK; ++i) {
    const int p = i * 64 + lane;
    tk[i] = tok != nullptr ? tok[beg + (p < t ? p : t - 1)] : beg + (p < t ? p : t - 1);   // no list: the window's rows are beg .. beg + t - 1
  }
  auto tok_at = [&](int i, int within) -> uint32_t {
    int sel = tk[0];
#pragma unroll
    for (int u = 1; u < NTK; ++u) sel = ((i >> 2) == u) ? tk[u] : sel;
    return (uint32_t)__shfl(sel, (i & 3) * 16 + within, 64);
  };
  const int nj = (nt - j0) < NJ ? (nt - j0) : NJ;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  float4 kf[NJ], vf[NJ];
  f32x4 dk[NJ], dv[NJ];
#pragma unroll
  for (int jj = 0; jj < NJ; ++jj) {
    dk[jj] = zero4;
    dv[jj] = zero4;
    if (jj < nj) {
      const uint32_t krow = tok_at(j0 + jj, c);
      kf[jj] = ldg4(K, krow * ldk + hoff + 4 * g);
      vf[jj] = ldg4(V, krow * ldv + hoff + 4 * g);
    } else {
      kf[jj] = vf[jj] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  const float s2 = scale * kLog2e;
  // operands of one query tile: row fragments for query 16i+c, column fragments for queries 16i+4g+r;
  // LSE / D are loaded once per query (lane c) and redistributed with cross-lane reads
  struct qtile {
    float4 qf, gf;
    float qc[4], gc[4];
    float lse_c, dd_c;
  };
  auto load_tile = [&](int i) -> qtile {
    qtile q;
    const uint32_t arow = tok_at(i, c);
    q.qf = ldg4(Q, arow * ldq + hoff + 4 * g);
    q.gf = ldg4(dO, arow * lddo + hoff + 4 * g);
    q.lse_c = LSE[arow * (uint32_t)H + head];
    q.dd_c = Dbuf[arow * (uint32_t)H + head];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint32_t crow = tok_at(i, 4 * g + r);
      q.qc[r] = ldg1(Q, crow * ldq + hoff + c);
      q.gc[r] = ldg1(dO, crow * lddo + hoff + c);
    }
    return q;
  };
  qtile cur = load_tile(0);
  for (int i = 0; i < nt; ++i) {
    const qtile nxt = load_tile(i + 1 < nt ? i + 1 : i);  // software prefetch of the next query tile
    const float4 qf = cur.qf, gf = cur.gf;
    float qc[4], gc[4], lse2[4], dd4[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      qc[r] = cur.qc[r];
      gc[r] = cur.gc[r];
      lse2[r] = __shfl(cur.lse_c, 4 * g + r, 64) * kLog2e;  // lane 4g+r holds query 16i+4g+r
      dd4[r] = __shfl(cur.dd_c, 4 * g + r, 64);
    }
    const bool last_q = (i == nt - 1);
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
      if (jj < nj) {
        f32x4 s, dp;
        mfma4x2(qf, kf[jj], s, gf, vf[jj], dp);  // S / dP [query 16i+4g+r][key 16(j0+jj)+c]
        float pe[4], ds[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          pe[r] = __builtin_amdgcn_exp2f(fmaf(s[r], s2, -lse2[r]));
          if (last_q) pe[r] = (i * 16 + 4 * g + r) < t ? pe[r] : 0.f;           // padded queries
          if (j0 + jj == nt - 1) pe[r] = ((j0 + jj) * 16 + c) < t ? pe[r] : 0.f;  // padded keys
          ds[r] = pe[r] * (dp[r] - dd4[r]) * scale;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          dv[jj] = __builtin_amdgcn_mfma_f32_16x16x4f32(pe[r], gc[r], dv[jj], 0, 0, 0);  // dV += P^T dO
          dk[jj] = __builtin_amdgcn_mfma_f32_16x16x4f32(ds[r], qc[r], dk[jj], 0, 0, 0);  // dK += dS^T Q
        }
      }
    }
    cur = nxt;
  }
  // D layout: value r = d{K,V}[key 16(j0+jj) + 4g + r][d = c]
#pragma unroll
  for (int jj = 0; jj < NJ; ++jj) {
    if (jj < nj) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const uint32_t krow = tok_at(j0 + jj, 4 * g + r);  // cross-lane read: all lanes active
        if ((j0 + jj) * 16 + 4 * g + r < t) {
          dK[krow * lddk + hoff + c] = dk[jj][r];
          dV[krow * lddv + hoff + c] = dv[jj][r];
        }
      }
    }
  }
}

template <int NTMAX>
__global__ __launch_bounds__(64 * kWH) void sra_bwd_dq_k(const float* __restrict__ Q, const float* __restrict__ K,
                                                    const float* __restrict__ V, const float* __restrict__ O,
                                                    const float* __restrict__ dO, const float* __restrict__ LSE,
                                                    int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t lddo,
                                                    const int32_t* __restrict__ tok,
                                                    const int32_t* __restrict__ winoff, int n_groups, int H,
                                                    float scale, float* __restrict__ dQ, int64_t lddq,
                                                    float* __restrict__ Dbuf) {
  const int bid = SST_SRA_BLOCK(blockIdx.x, gridDim.x);
  const int w = bid / n_groups;
  const int hg = bid - w * n_groups;
  const int beg = winoff[w];
  const int t = winoff[w + 1] - beg;
  const int nt = (t + 15) >> 4;
  if (nt < 1 || nt > NTMAX) return;
#define SST_DQ_ARGS Q, K, V, O, dO, LSE, (uint32_t)ldq, (uint32_t)ldk, (uint32_t)ldv, (uint32_t)ldo, (uint32_t)lddo, tok, beg, t, nt, hg, H, scale, dQ, (uint32_t)lddq, Dbuf
  if (nt <= 2)
    sra_bwd_dq_body<2>(SST_DQ_ARGS);
  else if (nt <= 4)
    sra_bwd_dq_body<4>(SST_DQ_ARGS);
  else if (nt <= 7 || NTMAX <= 7)
    sra_bwd_dq_body<(NTMAX < 7 ? NTMAX : 7)>(SST_DQ_ARGS);
  else
    sra_bwd_dq_body<NTMAX>(SST_DQ_ARGS);
#undef SST_DQ_ARGS
}

template <int NTMAX>
__global__ __launch_bounds__(64 * kWH) void sra_bwd_dkv_k(const float* __restrict__ Q, const float* __restrict__ K,
                                                     const float* __restrict__ V, const float* __restrict__ dO,
                                                     const float* __restrict__ LSE, const float* __restrict__ Dbuf,
                                                     int64_t ldq, int64_t ldk, int64_t ldv, int64_t lddo,
                                                     const int32_t* __restrict__ tok,
                                                     const int32_t* __restrict__ winoff, int n_groups, int H,
                                                     float scale, float* __restrict__ dK, float* __restrict__ dV,
                                                     int64_t lddk, int64_t lddv) {
  // workgroups per (window, head group): 4 key tiles each.  The key-group index is the SLOWEST block
  // coordinate: block b runs on XCD b % 8, and most windows only have key group 0, so a fastest-varying ks
  // would park all the real work on the even XCDs (measured: 1.36 waves/SIMD average, 169 us).
  const int per_ks = gridDim.x / ((NTMAX + 3) / 4);
  const int ks = blockIdx.x / per_ks;
  const int rest = SST_SRA_BLOCK(blockIdx.x - ks * per_ks, per_ks);
  const int w = rest / n_groups;
  const int hg = rest - w * n_groups;
  const int beg = winoff[w];
  const int t = winoff[w + 1] - beg;
  const int nt = (t + 15) >> 4;
  if (nt < 1 || nt > NTMAX || ks * 4 >= nt) return;
#define SST_DKV_ARGS Q, K, V, dO, LSE, Dbuf, (uint32_t)ldq, (uint32_t)ldk, (uint32_t)ldv, (uint32_t)lddo, tok, beg, t, nt, ks * 4, hg, H, scale, dK, dV, (uint32_t)lddk, (uint32_t)lddv
  if (nt <= 4)
    sra_bwd_dkv_body<4>(SST_DKV_ARGS);
  else
    sra_bwd_dkv_body<NTMAX>(SST_DKV_ARGS);
#undef SST_DKV_ARGS
}

// ------------------------------------------------------------------------------------------------
// MFMA backward, register-resident, ONE pass ("wave = window x head"): dQ, dK and dV from one read of Q, K, V, O, dO.
//   S = Q K^T and dP = dO V^T are computed once per (query tile, key tile) pair in the orientation row = query,
//   col = key.  P and dS in that D layout are directly the A operands of dV += P^T dO and dK += dS^T Q (contraction
//   over the rows); dQ += dS K contracts over the columns and needs the tile transposed: it makes a round trip
//   through a wave-private 16 x 16 LDS tile (one ds_write_b128, four ds_read_b32, row stride 20 floats: both
//   conflict-free), hidden behind the eight dV / dK MFMAs of the pair.  The operands that are contracted over tokens
//   (K for dQ; Q and dO for dK / dV) are needed as column fragments: they are read from wave-private LDS images of
//   the row fragments (K once per window, Q / dO once per query tile) instead of being fetched a second time.
//   20 MFMAs per tile pair (the two-kernel form: 12 + 16), every input row is read once, nothing is recomputed.
//   K / V row fragments and the dK / dV accumulators of ALL key tiles stay in VGPRs (16 per tile); the kernel is
//   built for 2 waves per SIMD.  The tile count is a template parameter of the body (exact for 1..7 tiles: the
//   key-tile loop is straight-line code without guards, so the scheduler can overlap neighbouring pairs).
//   Tokens sit transposed inside their 16-token tile (slot p <-> token (p >> 2) + 4 (p & 3)) as in the forward kernel:
//   MFMA k-step r then covers the consecutive tokens 4r..4r+3 and the padded steps of the last key tile are skipped.
// ------------------------------------------------------------------------------------------------
constexpr int kTS = 20;  // row stride (floats) of the LDS tiles

__host__ __device__ constexpr int sra_fused_lds_floats_per_wave(int nt) { return (nt * 16 + 3 * 16) * kTS; }

// COS (scaled cosine attention, cosine_msa.py:123-170): q and k rows are normalised per head as they are loaded, the score scale
// is hscale[head] = 1 / clamp(tau, tau_min), and the gradients of the normalised rows are taken through the normalisation where
// they are stored - d x = (d x^ - x^ (x^ . d x^)) / |x| - with the row fragment the lane already holds (dQ / dK leave in the
// layout Q / K were loaded in).  q^ . dq^ = sum_k dS_qk S_qk is also what the gradient of the scale needs: written to
// R[token][head]; d hscale[head] = colsum(R)[head] / hscale[head] (taken by the caller).
template <int NT, bool EXACT, bool NOHOIST, bool COS>
__device__ __forceinline__ void sra_bwd_fused_body(
    const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V, const float* __restrict__ O,
    const float* __restrict__ dO, const float* __restrict__ LSE, uint32_t ldq, uint32_t ldk, uint32_t ldv, uint32_t ldo,
    uint32_t lddo, const int32_t* __restrict__ tok, int beg, int t, int nt, int head, int H, float scale,
    float* __restrict__ dQ, float* __restrict__ dK, float* __restrict__ dV, uint32_t lddq, uint32_t lddk, uint32_t lddv,
    float* __restrict__ lds, const float* __restrict__ hscale, float* __restrict__ R, int part = 0, int parts = 1,
    float* __restrict__ partner_lds = nullptr) {
  // part / parts (small launches, sra_bwd_fused_k): this wave takes the query tiles part, part + 2, ... of the window; the wave
  // next to it in the workgroup (same head, partner_lds = its LDS block) takes the others.  dQ rows belong to one of the two;
  // dK / dV are partial sums over each wave's query tiles: wave 1 hands its partials to wave 0 through its LDS block at the end.
  const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  const int sig = (c >> 2) + 4 * (c & 3);  // token held by tile slot c
  const uint32_t hoff = head * kHD;
  if (COS) scale = hscale[head];
  float* Kimg = lds;                 // [NT * 16][kTS]  K rows, tile slot order
  float* Qimg = Kimg + NT * 16 * kTS;  // [16][kTS]   Q rows of the current query tile
  float* Gimg = Qimg + 16 * kTS;       // [16][kTS]   dO rows of the current query tile
  float* Dimg = Gimg + 16 * kTS;       // [16][kTS]   dS tile, [key slot][query slot]
  constexpr int NTK = (NT * 16 + 63) / 64;
  int tk[NTK];
#pragma unroll
  for (int i = 0; i < NTK; ++i) {
    const int p = i * 64 + lane;
    tk[i] = tok != nullptr ? tok[beg + (p < t ? p : t - 1)] : beg + (p < t ? p : t - 1);   // no list: the window's rows are beg .. beg + t - 1
  }
  auto tok_at = [&](int i, int within) -> uint32_t {  // token id of window position 16 i + within
    int sel = tk[0];
#pragma unroll
    for (int u = 1; u < NTK; ++u) sel = ((i >> 2) == u) ? tk[u] : sel;
    return (uint32_t)__shfl(sel, (i & 3) * 16 + within, 64);
  };
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  float4 kf[NT], vf[NT];
  f32x4 dk[NT], dv[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    dk[j] = zero4;
    dv[j] = zero4;
    if (EXACT || j < nt) {
      const uint32_t krow = (uint32_t)__shfl(tk[j >> 2], (j & 3) * 16 + sig, 64);
      kf[j] = ldg4(K, krow * ldk + hoff + 4 * g);
      if (COS) kf[j] = scale4(kf[j], rowfrag_inv_norm(kf[j]));   // k^ from here on (the K image below included)
      vf[j] = ldg4(V, krow * ldv + hoff + 4 * g);
    } else {
      kf[j] = vf[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  // first query tile
  struct qtile {
    float4 qf, gf, of;
    float lse;
    uint32_t row;
  };
  auto load_tile = [&](int i) -> qtile {
    qtile q;
    const uint32_t row = tok_at(i, sig);
    q.row = row;
    q.qf = ldg4(Q, row * ldq + hoff + 4 * g);
    q.gf = ldg4(dO, row * lddo + hoff + 4 * g);
    q.of = ldg4(O, row * ldo + hoff + 4 * g);
    q.lse = LSE[row * (uint32_t)H + head];
    return q;
  };
  qtile cur = load_tile(part < nt ? part : 0);
  // K image (column fragments of K for dQ += dS K)
#pragma unroll
  for (int j = 0; j < NT; ++j)
    if (EXACT || j < nt) *(float4*)(Kimg + (j * 16 + c) * kTS + 4 * g) = kf[j];
  const float s2 = scale * kLog2e;
  const int last_steps = (t - 16 * (nt - 1) + 3) >> 2;  // k-steps of the last tile that hold a real token (1..4)
  const float* kcol = Kimg + (4 * g) * kTS + c;          // + (16 j + r) * kTS : K[key slot 4g + r of tile j][d = c]
  const float* qcol = Qimg + (4 * g) * kTS + c;
  const float* gcol = Gimg + (4 * g) * kTS + c;
  float* drow = Dimg + c * kTS + 4 * g;                  // write: [key slot c][query slots 4g .. 4g+3]
  const float* dcol = Dimg + (4 * g) * kTS + c;          // read:  [key slot 4g + r][query slot c]

  for (int i = part; i < nt; i += parts) {
    // NOHOIST: the K column fragments are re-read from the LDS image for every tile pair (4 ds_read_b32) instead of
    // being kept in 4 VGPRs per key tile across the whole loop - the price of a third wave per SIMD
    if (NOHOIST) asm volatile("" ::: "memory");
    const float q_inv = COS ? rowfrag_inv_norm(cur.qf) : 1.f;
    const float4 qf = COS ? scale4(cur.qf, q_inv) : cur.qf, gf = cur.gf;   // COS: q^
    float dd = gf.x * cur.of.x + gf.y * cur.of.y + gf.z * cur.of.z + gf.w * cur.of.w;
    dd = rows4_sum(dd);                       // rowsum(dO * O) of query slot c, on every lane of the column
    const float lse_c = cur.lse * kLog2e;
    const uint32_t qrow = cur.row;
    *(float4*)(Qimg + c * kTS + 4 * g) = qf;
    *(float4*)(Gimg + c * kTS + 4 * g) = gf;
    cur = load_tile(i + parts < nt ? i + parts : i);  // prefetch of this wave's next query tile
    float qc[4], gc[4], lse2[4], dd4[4], rmask[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      qc[r] = qcol[r * kTS];
      gc[r] = gcol[r * kTS];
      lse2[r] = __shfl(lse_c, 4 * g + r, 64);  // lane (0, 4g + r) holds query slot 4g + r
      dd4[r] = __shfl(dd, 4 * g + r, 64);
      rmask[r] = (i * 16 + g + 4 * r) < t ? 1.f : 0.f;  // padded queries (last tile only)
    }
    f32x4 dq0 = zero4, dq1 = zero4, dq2 = zero4, dq3 = zero4;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      if (EXACT || j < nt) {
        f32x4 s, dp;
        mfma4x2(qf, kf[j], s, gf, vf[j], dp);  // S / dP [query slot 4g+r][key slot c]
        f32x4 pe, ds;
        const bool last_j = EXACT ? (j == NT - 1) : (j == nt - 1);
        const float cmask = (last_j && (j * 16 + sig) >= t) ? 0.f : 1.f;  // padded keys (last tile only)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float p = __builtin_amdgcn_exp2f(fmaf(s[r], s2, -lse2[r])) * rmask[r];
          if (!EXACT || j == NT - 1) p *= cmask;
          pe[r] = p;
          ds[r] = p * (dp[r] - dd4[r]) * scale;
        }
        *(f32x4*)drow = ds;  // transposed hand-over of the dS tile
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          dv[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(gc[r], pe[r], dv[j], 0, 0, 0);  // dV^T += dO^T P
          dk[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(qc[r], ds[r], dk[j], 0, 0, 0);  // dK^T += Q^T dS
        }
        float dst[4], kc[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          dst[r] = dcol[r * kTS];                  // dS[query slot c][key slot 4g + r]
          kc[r] = kcol[(j * 16 + r) * kTS];        // K [key slot 4g + r][d = c]
        }
        const int steps = last_j ? last_steps : 4;  // wave-uniform; compile-time 4 except in the last tile
        dq0 = __builtin_amdgcn_mfma_f32_16x16x4f32(kc[0], dst[0], dq0, 0, 0, 0);  // dQ^T += K^T dS^T
        if (steps > 1) dq1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kc[1], dst[1], dq1, 0, 0, 0);
        if (steps > 2) dq2 = __builtin_amdgcn_mfma_f32_16x16x4f32(kc[2], dst[2], dq2, 0, 0, 0);
        if (steps > 3) dq3 = __builtin_amdgcn_mfma_f32_16x16x4f32(kc[3], dst[3], dq3, 0, 0, 0);
      }
    }
    // The products are formed transposed (dQ^T = K^T dS^T): in the D layout value r of lane (g, c) is
    // dQ^T[d = 4g + r][query slot c], i.e. the lane holds the ROW fragment dQ[token of slot c][4g .. 4g+3] - one
    // 16-byte store at the address pattern of the Q loads, no lane exchange.
    float4 o = make_float4((dq0[0] + dq1[0]) + (dq2[0] + dq3[0]), (dq0[1] + dq1[1]) + (dq2[1] + dq3[1]),
                           (dq0[2] + dq1[2]) + (dq2[2] + dq3[2]), (dq0[3] + dq1[3]) + (dq2[3] + dq3[3]));
    if (COS) {  // through the normalisation: the lane holds q^[4g..4g+3] of the same token (every lane takes part in the sum)
      const float rq = rows4_sum(dot4(o, qf));
      o = make_float4((o.x - qf.x * rq) * q_inv, (o.y - qf.y * rq) * q_inv, (o.z - qf.z * rq) * q_inv,
                      (o.w - qf.w * rq) * q_inv);
      if (g == 0 && i * 16 + sig < t) R[qrow * (uint32_t)H + head] = rq;
    }
    if (i * 16 + sig < t) *(float4*)(dQ + (qrow * lddq + hoff + 4 * g)) = o;
  }
  if (parts == 2) {
    // the two waves of a head meet: wave 1 leaves its dK (then its dV) partial tiles in ITS LDS block (its K image is dead: its
    // query loop is over), wave 0 adds them to its own in a fixed order (own + partner: deterministic) and stores.  Workgroup
    // barriers: all four waves of a split workgroup belong to the same window and run the same number of rounds.
    f32x4* mine = (f32x4*)lds;
    const f32x4* theirs = (const f32x4*)partner_lds;
#pragma unroll
    for (int round = 0; round < 2; ++round) {
      if (part == 1) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
          if (EXACT || j < nt) mine[j * 64 + lane] = round == 0 ? dk[j] : dv[j];
      }
      __syncthreads();
      if (part == 0) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
          if (EXACT || j < nt) {
            if (round == 0) dk[j] += theirs[j * 64 + lane];
            else dv[j] += theirs[j * 64 + lane];
          }
      }
      __syncthreads();
    }
    if (part == 1) return;
  }
  // dK^T / dV^T likewise: lane (g, c) holds d{K,V}[token of key slot c][4g .. 4g+3]
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    if (EXACT || j < nt) {
      const uint32_t krow = (uint32_t)__shfl(tk[j >> 2], (j & 3) * 16 + sig, 64);
      float4 dkv = make_float4(dk[j][0], dk[j][1], dk[j][2], dk[j][3]);
      if (COS) {  // kf[j] is k^; |k| comes from the row itself once more (an L2 hit) instead of a register per tile all along
        const float k_inv = rowfrag_inv_norm(ldg4(K, krow * ldk + hoff + 4 * g));
        const float rk = rows4_sum(dot4(dkv, kf[j]));
        dkv = make_float4((dkv.x - kf[j].x * rk) * k_inv, (dkv.y - kf[j].y * rk) * k_inv, (dkv.z - kf[j].z * rk) * k_inv,
                          (dkv.w - kf[j].w * rk) * k_inv);
      }
      if (j * 16 + sig < t) {
        *(float4*)(dK + (krow * lddk + hoff + 4 * g)) = dkv;
        *(float4*)(dV + (krow * lddv + hoff + 4 * g)) = make_float4(dv[j][0], dv[j][1], dv[j][2], dv[j][3]);
      }
    }
  }
}

// NTMAX: largest tile count the launch has to handle (the caller knows the largest window); WPS: waves per SIMD the
// kernel is built for (2: K column fragments stay in registers, <= 256 VGPRs; 3: they are re-read from LDS, <= 168).
template <int NTMAX, int WPS, bool COS>
__global__ __launch_bounds__(64 * kWH, WPS) void sra_bwd_fused_k(
    const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V, const float* __restrict__ O,
    const float* __restrict__ dO, const float* __restrict__ LSE, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
    int64_t lddo, const int32_t* __restrict__ tok, const int32_t* __restrict__ winoff, int n_groups, int H, float scale,
    float* __restrict__ dQ, float* __restrict__ dK, float* __restrict__ dV, int64_t lddq, int64_t lddk, int64_t lddv,
    const int32_t* __restrict__ order, const float* __restrict__ hscale, float* __restrict__ R, int n_win, int qs) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr bool NH = WPS >= 3;
  // Small launches (qs == 2): the grid carries one further workgroup per (window, head group) IN FRONT of the regular ones.  A
  // window of kSplitTiles tiles and more is then served by both: each takes two of the group's four heads, and the two waves of a
  // head share the window's query tiles (part 0: tiles 0, 2, .., part 1: tiles 1, 3, ..) - the 7 x 7 dependent tile pairs of a
  // 100-token window on ONE wave per head (27.6 us alone) were the whole launch on a sweep-sized frame (34 us for 75 MB).
  const int n_main = n_win * n_groups;
  const int n_extra = (int)gridDim.x - n_main;
  int sub = 0, wpos, hg;
  if ((int)blockIdx.x < n_extra) {
    sub = 1;
    const int from_end = (int)blockIdx.x / n_groups;
    hg = (int)blockIdx.x - from_end * n_groups;
    wpos = n_win - 1 - from_end;
  } else {
    const int bid = SST_SRA_BLOCK((int)blockIdx.x - n_extra, n_main);
    wpos = bid / n_groups;
    hg = bid - wpos * n_groups;
  }
  const int w = order != nullptr ? order[wpos] : wpos;
  const int beg = winoff[w];
  const int t = winoff[w + 1] - beg;
  const int nt = (t + 15) >> 4;
  if (nt < 1 || nt > NTMAX) return;  // > NTMAX: the generic kernel owns this window
  const int wave = threadIdx.x >> 6;
  int head = hg * kWH + wave, part = 0, parts = 1;
  if (qs == 2) {
    if (nt >= kSplitTiles) {
      static_assert(kWH % 2 == 0, "the split pairs the waves of a workgroup");
      head = hg * kWH + sub * (kWH / 2) + (wave >> 1);
      part = wave & 1;
      parts = 2;
    } else if (sub) {
      return;
    }
  }
  float* lds = smem + wave * sra_fused_lds_floats_per_wave(NTMAX);
  float* plds = smem + (wave ^ 1) * sra_fused_lds_floats_per_wave(NTMAX);
#define SST_FUSED_ARGS Q, K, V, O, dO, LSE, (uint32_t)ldq, (uint32_t)ldk, (uint32_t)ldv, (uint32_t)ldo, (uint32_t)lddo, tok, beg, t, nt, head, H, scale, dQ, dK, dV, (uint32_t)lddq, (uint32_t)lddk, (uint32_t)lddv, lds, hscale, R, part, parts, plds
  switch (nt) {
    case 1: sra_bwd_fused_body<1, true, NH, COS>(SST_FUSED_ARGS); break;
    case 2: sra_bwd_fused_body<2, true, NH, COS>(SST_FUSED_ARGS); break;
    case 3: sra_bwd_fused_body<3, true, NH, COS>(SST_FUSED_ARGS); break;
    case 4: sra_bwd_fused_body<4, true, NH, COS>(SST_FUSED_ARGS); break;
    case 5: if constexpr (NTMAX >= 5) sra_bwd_fused_body<5, true, NH, COS>(SST_FUSED_ARGS); break;
    case 6: if constexpr (NTMAX >= 6) sra_bwd_fused_body<6, true, NH, COS>(SST_FUSED_ARGS); break;
    case 7: if constexpr (NTMAX >= 7) sra_bwd_fused_body<7, true, NH, COS>(SST_FUSED_ARGS); break;
    default:
      if constexpr (NTMAX > 7) sra_bwd_fused_body<NTMAX, false, NH, COS>(SST_FUSED_ARGS);
      break;
  }
#undef SST_FUSED_ARGS
}

// parts a large window is dealt out over, by the size of the launch (workgroups before the split).  A launch that fills the chip
// several times over hides a long wave behind the others; below that the long waves ARE the launch.
// SST_SRA_SPLIT=<parts> / SST_SRA_SPLIT_MAX_WG=<workgroups> override (A/B runs).
int sra_split_parts(int64_t n_workgroups) {
  static int parts_env = -1, max_wg = 0;
  if (parts_env < 0) {
    const char* e = getenv("SST_SRA_SPLIT");
    const char* m = getenv("SST_SRA_SPLIT_MAX_WG");
    max_wg = m ? atoi(m) : 2048;     // measured (tools/sra_sizes.py, kernel-exact events): 10 x 100 tokens 12.8 -> 8.5 us, LiDAR-like
    parts_env = e ? atoi(e) : 2;     // frame (1 620 workgroups) 18.0 -> 17.4 us with 2 parts; 4 parts and larger launches lose to the empty workgroups
  }
  if (parts_env <= 1 || n_workgroups > max_wg) return 1;
  return parts_env;
}

template <int NTMAX, int WPS>
int launch_bwd_fused(const float* Q, const float* K, const float* V, const float* O, const float* dO, const float* LSE,
                     int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t lddo, const int32_t* tok,
                     const int32_t* winoff, int64_t n_windows, int H, float scale, float* dQ, float* dK, float* dV,
                     int64_t lddq, int64_t lddk, int64_t lddv, hipStream_t st) {
  const int n_groups = H / kWH;
  const size_t lds = (size_t)kWH * sra_fused_lds_floats_per_wave(NTMAX) * sizeof(float);
  const float* hs = g_head_scale;
  float* rbuf = g_cos_r;
  if (hs != nullptr && rbuf == nullptr) return SST_ERR_ARG;
  auto kern = hs != nullptr ? sra_bwd_fused_k<NTMAX, WPS, true> : sra_bwd_fused_k<NTMAX, WPS, false>;
  // the attribute is per device and per kernel: set on first use of each (device, variant)
  static unsigned long long configured[2] = {0ull, 0ull};   // bit d of [cos]: device d done (per instantiation)
  int dev_id = 0;
  SST_HIP(hipGetDevice(&dev_id));
  const unsigned long long bit = 1ull << (dev_id & 63);
  const int slot = hs != nullptr ? 1 : 0;
  if (!(__atomic_load_n(&configured[slot], __ATOMIC_ACQUIRE) & bit)) {
    SST_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    __atomic_fetch_or(&configured[slot], bit, __ATOMIC_RELEASE);
  }
  const int qs = (NTMAX >= kSplitTiles && sra_split_parts(n_windows * n_groups) > 1) ? 2 : 1;
  const dim3 grid((unsigned)(n_windows * n_groups * qs));
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (g_prof_bwd_start != nullptr && g_prof_bwd_stop != nullptr) {
    e0 = g_prof_bwd_start;
    e1 = g_prof_bwd_stop;
    g_prof_bwd_start = g_prof_bwd_stop = nullptr;  // one-shot, as for the forward kernel
  }
  if (e0 != nullptr)  // kernel-exact start / stop timestamps on the launch stream
    hipExtLaunchKernelGGL(kern, grid, dim3(64 * kWH), lds, st, e0, e1, 0, Q, K, V, O, dO, LSE, ldq,
                          ldk, ldv, ldo, lddo, tok, winoff, n_groups, H, scale, dQ, dK, dV, lddq, lddk, lddv, g_win_order, hs, rbuf,
                          (int)n_windows, qs);
  else
    hipLaunchKernelGGL(kern, grid, dim3(64 * kWH), lds, st, Q, K, V, O, dO, LSE, ldq, ldk, ldv, ldo,
                       lddo, tok, winoff, n_groups, H, scale, dQ, dK, dV, lddq, lddk, lddv, g_win_order, hs, rbuf, (int)n_windows, qs);
  return SST_OK;
}

template <int NTMAX>
int launch_bwd_wave(const float* Q, const float* K, const float* V, const float* O, const float* dO, const float* LSE,
                    int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t lddo, const int32_t* tok,
                    const int32_t* winoff, int64_t n_windows, int H, float scale, float* dQ, float* dK, float* dV,
                    int64_t lddq, int64_t lddk, int64_t lddv, float* Dbuf, hipStream_t st) {
  const int n_groups = H / kWH;
  const dim3 grid((unsigned)(n_windows * n_groups));
  hipLaunchKernelGGL(sra_bwd_dq_k<NTMAX>, grid, dim3(64 * kWH), 0, st, Q, K, V, O, dO, LSE, ldq, ldk, ldv, ldo, lddo, tok,
                     winoff, n_groups, H, scale, dQ, lddq, Dbuf);
  const dim3 grid_kv((unsigned)(n_windows * n_groups * ((NTMAX + 3) / 4)));
  hipLaunchKernelGGL(sra_bwd_dkv_k<NTMAX>, grid_kv, dim3(64 * kWH), 0, st, Q, K, V, dO, LSE, Dbuf, ldq, ldk, ldv, lddo, tok,
                     winoff, n_groups, H, scale, dK, dV, lddk, lddv);
  return SST_OK;
}

template <int NT>
int launch_fwd_variant(const float* Q, const float* K, const float* V, int64_t ldq, int64_t ldk, int64_t ldv,
                       const int32_t* tok, const int32_t* winoff, int64_t n_windows, int H, float scale, int nt_lo,
                       float* O, int64_t ldo, float* LSE, hipStream_t st) {
  const int n_groups = H / kGH;
  const size_t lds = (size_t)(2 * NT * 16 * kRS) * sizeof(float) + (size_t)NT * 16 * sizeof(int);
  SST_HIP(hipFuncSetAttribute((const void*)sra_fwd_mfma_k<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(sra_fwd_mfma_k<NT>, dim3((unsigned)(n_windows * n_groups)), dim3(256), lds, st, Q, K, V, ldq, ldk,
                     ldv, tok, winoff, n_groups, H, scale, nt_lo, O, ldo, LSE);
  return SST_OK;
}

template <int NTMAX>
int launch_fwd_wave(const float* Q, const float* K, const float* V, int64_t ldq, int64_t ldk, int64_t ldv,
                    const int32_t* tok, const int32_t* winoff, int64_t n_windows, int H, float scale, float* O,
                    int64_t ldo, float* LSE, hipStream_t st) {
  const int n_groups = H / kWH;
  const float* hs = g_head_scale;
  auto kern = hs != nullptr ? sra_fwd_wave_k<NTMAX, true> : sra_fwd_wave_k<NTMAX, false>;
  const int qs = NTMAX >= kSplitTiles ? sra_split_parts(n_windows * n_groups) : 1;
  const dim3 grid((unsigned)(n_windows * n_groups * qs));
  if (g_prof_start != nullptr && g_prof_stop != nullptr) {
    // one-shot: kernel-exact start / stop timestamps on the launch stream (no barrier packets, no cache flush
    // between the marks and the kernel, unlike a pair of hipEventRecord calls around the launch)
    hipExtLaunchKernelGGL(kern, grid, dim3(64 * kWH), 0, st,
                          g_prof_start, g_prof_stop, 0, Q, K, V, ldq, ldk, ldv, tok, winoff, n_groups, H, scale, O, ldo,
                          LSE, g_win_order, hs, (int)n_windows, qs);
    g_prof_start = g_prof_stop = nullptr;
    return SST_OK;
  }
  hipLaunchKernelGGL(kern, grid, dim3(64 * kWH), 0, st, Q, K, V, ldq, ldk,
                     ldv, tok, winoff, n_groups, H, scale, O, ldo, LSE, g_win_order, hs, (int)n_windows, qs);
  return SST_OK;
}

template <int NT>
int launch_bwd_variant(const float* Q, const float* K, const float* V, const float* O, const float* dO,
                       const float* LSE, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t lddo,
                       const int32_t* tok, const int32_t* winoff, int64_t n_windows, int H, float scale, int nt_lo,
                       float* dQ, float* dK, float* dV, int64_t lddq, int64_t lddk, int64_t lddv, hipStream_t st) {
  const int n_groups = H / kGH;
  const size_t lds =
      (size_t)(4 * NT * 16 * kRS + 2 * NT * 16 * kGH) * sizeof(float) + (size_t)NT * 16 * sizeof(int);
  SST_HIP(hipFuncSetAttribute((const void*)sra_bwd_mfma_k<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(sra_bwd_mfma_k<NT>, dim3((unsigned)(n_windows * n_groups)), dim3(256), lds, st, Q, K, V, O, dO, LSE,
                     ldq, ldk, ldv, ldo, lddo, tok, winoff, n_groups, H, scale, nt_lo, dQ, dK, dV, lddq, lddk, lddv);
  return SST_OK;
}

bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

extern "C" {

int sst_sra_attn_fwd_f32(const float* d_q, const float* d_k, const float* d_v, int64_t ldq, int64_t ldk,
                         int64_t ldv, const int32_t* d_tok, const int32_t* d_winoff, int64_t n_windows, int n_heads,
                         float scale, int max_tokens, int impl, float* d_o, int64_t ldo, float* d_lse,
                         void* stream) {
  if (n_windows < 0 || n_heads < 1 || impl < 0 || impl > 3) return SST_ERR_ARG;
  if (n_windows == 0) return SST_OK;
  if (!d_q || !d_k || !d_v || !d_winoff || !d_o || !d_lse) return SST_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const bool mfma_ok = (n_heads % kGH == 0) && ((ldq | ldk | ldv | ldo) % 4 == 0) && aligned16(d_q) &&
                       aligned16(d_k) && aligned16(d_v) && aligned16(d_o);
  // d_tok == NULL: the rows of window w are d_winoff[w] .. d_winoff[w + 1] - 1 themselves (window-major feature rows: one
  // dependent load less per wave); only the register-resident kernels take it
  if (!d_tok && (!mfma_ok || (impl != 0 && impl != 3) || max_tokens <= 0 || max_tokens > kMaxTilesMfma * 16))
    return SST_ERR_UNSUPPORTED;
  // (the register-resident kernels use 32-bit element offsets: callers keep n_tokens * ld < 2^31)
  if (impl == 1 || !mfma_ok) {
    hipLaunchKernelGGL(sra_fwd_generic_k, dim3((unsigned)n_windows), dim3(256), 0, st, d_q, d_k, d_v, ldq, ldk, ldv,
                       d_tok, d_winoff, n_heads, scale, 0, d_o, ldo, d_lse);
    SST_LAUNCH_CHECK();
    return SST_OK;
  }
  int rc;
  const int cap_tiles = max_tokens > 0 ? (max_tokens + 15) / 16 : 1 << 30;
#define SST_FWD_ARGS(lo) d_q, d_k, d_v, ldq, ldk, ldv, d_tok, d_winoff, n_windows, n_heads, scale, lo, d_o, ldo, d_lse, st
  if (impl == 2) {  // LDS-staged kernels
    rc = launch_fwd_variant<2>(SST_FWD_ARGS(0));
    if (rc) return rc;
    if (cap_tiles > 2 && (rc = launch_fwd_variant<4>(SST_FWD_ARGS(2)))) return rc;
    if (cap_tiles > 4 && (rc = launch_fwd_variant<7>(SST_FWD_ARGS(4)))) return rc;
    if (cap_tiles > 7 && (rc = launch_fwd_variant<9>(SST_FWD_ARGS(7)))) return rc;
  } else {          // register-resident kernels (default): ONE launch, tile class chosen per workgroup
    if (cap_tiles <= 4)   // the largest window the caller announces picks the register class of the launch
      rc = launch_fwd_wave<4>(d_q, d_k, d_v, ldq, ldk, ldv, d_tok, d_winoff, n_windows, n_heads, scale, d_o, ldo,
                              d_lse, st);
    else if (cap_tiles <= 5)
      rc = launch_fwd_wave<5>(d_q, d_k, d_v, ldq, ldk, ldv, d_tok, d_winoff, n_windows, n_heads, scale, d_o, ldo,
                              d_lse, st);
    else if (cap_tiles <= 7)
      rc = launch_fwd_wave<7>(d_q, d_k, d_v, ldq, ldk, ldv, d_tok, d_winoff, n_windows, n_heads, scale, d_o, ldo,
                              d_lse, st);
    else
      rc = launch_fwd_wave<9>(d_q, d_k, d_v, ldq, ldk, ldv, d_tok, d_winoff, n_windows, n_heads, scale, d_o, ldo,
                              d_lse, st);
    if (rc) return rc;
  }
#undef SST_FWD_ARGS
  if (cap_tiles > kMaxTilesMfma) {
    hipLaunchKernelGGL(sra_fwd_generic_k, dim3((unsigned)n_windows), dim3(256), 0, st, d_q, d_k, d_v, ldq, ldk, ldv,
                       d_tok, d_winoff, n_heads, scale, kMaxTilesMfma * 16, d_o, ldo, d_lse);
  }
  SST_LAUNCH_CHECK();
  return SST_OK;
}

int sst_sra_attn_fwd_ord_f32(const float* d_q, const float* d_k, const float* d_v, int64_t ldq, int64_t ldk, int64_t ldv,
                             const int32_t* d_tok, const int32_t* d_winoff, const int32_t* d_win_order, int64_t n_windows,
                             int n_heads, float scale, int max_tokens, int impl, float* d_o, int64_t ldo, float* d_lse,
                             void* stream) {
  g_win_order = d_win_order;
  const int rc = sst_sra_attn_fwd_f32(d_q, d_k, d_v, ldq, ldk, ldv, d_tok, d_winoff, n_windows, n_heads, scale,
                                      max_tokens, impl, d_o, ldo, d_lse, stream);
  g_win_order = nullptr;
  return rc;
}

int64_t sst_sra_attn_bwd_workspace_bytes(int64_t n_tokens, int n_heads) {
  return sst_align_up((n_tokens > 0 ? n_tokens : 1) * (int64_t)n_heads * sizeof(float), 256);
}

int sst_sra_attn_bwd_f32(const float* d_q, const float* d_k, const float* d_v, const float* d_o, const float* d_do,
                         const float* d_lse, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t lddo,
                         const int32_t* d_tok, const int32_t* d_winoff, int64_t n_windows, int64_t n_tokens,
                         int n_heads, float scale, int max_tokens, int impl, float* d_dq, float* d_dk, float* d_dv,
                         int64_t lddq, int64_t lddk, int64_t lddv, void* d_workspace, void* stream) {
  if (n_windows < 0 || n_tokens < 0 || n_heads < 1 || impl < 0 || impl > 3) return SST_ERR_ARG;
  if (n_windows == 0) return SST_OK;
  if (!d_q || !d_k || !d_v || !d_o || !d_do || !d_lse || !d_winoff || !d_dq || !d_dk || !d_dv)
    return SST_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const bool mfma_ok = (n_heads % kGH == 0) && ((ldq | ldk | ldv | ldo | lddo) % 4 == 0) && aligned16(d_q) &&
                       aligned16(d_k) && aligned16(d_v) && aligned16(d_o) && aligned16(d_do);
  if (!d_tok && (!mfma_ok || (impl != 0 && impl != 3) || max_tokens <= 0 || max_tokens > kMaxTilesMfma * 16))
    return SST_ERR_UNSUPPORTED;   // see sst_sra_attn_fwd_f32
  const bool out_vec_ok = ((lddq | lddk | lddv) % 4 == 0) && aligned16(d_dq) && aligned16(d_dk) && aligned16(d_dv);
  const int cap_tiles = max_tokens > 0 ? (max_tokens + 15) / 16 : 1 << 30;
  const bool all_generic = (impl == 1) || !mfma_ok;
  const bool need_generic = all_generic || cap_tiles > kMaxTilesMfma;
  if (need_generic) {
    // the generic kernel accumulates dK/dV with float atomics
    const size_t width = (size_t)n_heads * kHD * sizeof(float);
    SST_HIP(hipMemset2DAsync(d_dk, (size_t)lddk * sizeof(float), 0, width, (size_t)n_tokens, st));
    SST_HIP(hipMemset2DAsync(d_dv, (size_t)lddv * sizeof(float), 0, width, (size_t)n_tokens, st));
  }
  if (!all_generic) {
    int rc = SST_OK;
#define SST_BWD_ARGS(lo) d_q, d_k, d_v, d_o, d_do, d_lse, ldq, ldk, ldv, ldo, lddo, d_tok, d_winoff, n_windows, n_heads, scale, lo, d_dq, d_dk, d_dv, lddq, lddk, lddv, st
    if (impl == 2) {  // LDS-staged kernels
      rc = launch_bwd_variant<2>(SST_BWD_ARGS(0));
      if (rc) return rc;
      if (cap_tiles > 2 && (rc = launch_bwd_variant<4>(SST_BWD_ARGS(2)))) return rc;
      if (cap_tiles > 4 && (rc = launch_bwd_variant<7>(SST_BWD_ARGS(4)))) return rc;
      if (cap_tiles > 7 && (rc = launch_bwd_variant<9>(SST_BWD_ARGS(7)))) return rc;
    } else if (impl == 3 || !out_vec_ok) {  // register-resident kernels, two launches (dQ, then dK / dV): kept for
                                            // comparison, and for gradient buffers that are not 16-byte aligned
      if (!d_workspace) return SST_ERR_ARG;
      if (cap_tiles <= 7)
        rc = launch_bwd_wave<7>(d_q, d_k, d_v, d_o, d_do, d_lse, ldq, ldk, ldv, ldo, lddo, d_tok, d_winoff, n_windows,
                                n_heads, scale, d_dq, d_dk, d_dv, lddq, lddk, lddv, (float*)d_workspace, st);
      else
        rc = launch_bwd_wave<9>(d_q, d_k, d_v, d_o, d_do, d_lse, ldq, ldk, ldv, ldo, lddo, d_tok, d_winoff, n_windows,
                                n_heads, scale, d_dq, d_dk, d_dv, lddq, lddk, lddv, (float*)d_workspace, st);
      if (rc) return rc;
    } else {          // register-resident one-pass kernel (default): one launch, tile count chosen per workgroup
#define SST_FUSED_CALL(NTM, WPS)                                                                                      \
  launch_bwd_fused<NTM, WPS>(d_q, d_k, d_v, d_o, d_do, d_lse, ldq, ldk, ldv, ldo, lddo, d_tok, d_winoff, n_windows, \
                             n_heads, scale, d_dq, d_dk, d_dv, lddq, lddk, lddv, st)
      // the kernel class follows the largest window the caller announces (max_tokens): up to 5 tiles the three-wave
      // build fits (SST_SRA_BWD_WPS=2 forces the two-wave build for A/B runs)
      static int wps_env = -1;
      if (wps_env < 0) {
        const char* e = getenv("SST_SRA_BWD_WPS");
        wps_env = e ? atoi(e) : 0;
      }
      if (cap_tiles <= 4)
        rc = wps_env == 2 ? SST_FUSED_CALL(4, 2) : SST_FUSED_CALL(4, 3);
      else if (cap_tiles <= 5)
        rc = wps_env == 2 ? SST_FUSED_CALL(5, 2) : SST_FUSED_CALL(5, 3);
      else if (cap_tiles <= 7)
        rc = SST_FUSED_CALL(7, 2);
      else
        rc = SST_FUSED_CALL(9, 2);
#undef SST_FUSED_CALL
      if (rc) return rc;
    }
#undef SST_BWD_ARGS
  }
  if (need_generic) {
    hipLaunchKernelGGL(sra_bwd_generic_k, dim3((unsigned)n_windows), dim3(256), 0, st, d_q, d_k, d_v, d_o, d_do,
                       d_lse, ldq, ldk, ldv, ldo, lddo, d_tok, d_winoff, n_heads, scale,
                       all_generic ? 0 : kMaxTilesMfma * 16, d_dq, d_dk, d_dv, lddq, lddk, lddv);
  }
  SST_LAUNCH_CHECK();
  return SST_OK;
}

// ---- measurement hooks (bench.py): HIP events bound to ONE kernel launch ------------------------------------
void* sst_event_create(void) {
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return (void*)e;
}

void sst_event_destroy(void* ev) {
  if (ev) (void)hipEventDestroy((hipEvent_t)ev);
}

float sst_event_elapsed_ms(void* start, void* stop) {
  float ms = -1.f;
  if (!start || !stop) return ms;
  if (hipEventSynchronize((hipEvent_t)stop) != hipSuccess) return -1.f;
  if (hipEventElapsedTime(&ms, (hipEvent_t)start, (hipEvent_t)stop) != hipSuccess) return -1.f;
  return ms;
}

int sst_sra_attn_bwd_ord_f32(const float* d_q, const float* d_k, const float* d_v, const float* d_o, const float* d_do,
                             const float* d_lse, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t lddo,
                             const int32_t* d_tok, const int32_t* d_winoff, const int32_t* d_win_order, int64_t n_windows,
                             int64_t n_tokens, int n_heads, float scale, int max_tokens, int impl, float* d_dq,
                             float* d_dk, float* d_dv, int64_t lddq, int64_t lddk, int64_t lddv, void* d_workspace,
                             void* stream) {
  g_win_order = d_win_order;
  const int rc = sst_sra_attn_bwd_f32(d_q, d_k, d_v, d_o, d_do, d_lse, ldq, ldk, ldv, ldo, lddo, d_tok, d_winoff,
                                      n_windows, n_tokens, n_heads, scale, max_tokens, impl, d_dq, d_dk, d_dv, lddq,
                                      lddk, lddv, d_workspace, stream);
  g_win_order = nullptr;
  return rc;
}

// Scaled cosine attention (cosine_msa.py:123-185): softmax(normalize(q) normalize(k)^T * head_scale[h]) v inside every window,
// head_scale[h] = 1 / clamp(tau, tau_min) in DEVICE memory ([n_heads] floats; a shared tau is passed expanded).  Only the
// register-resident kernels carry it (impl 0, windows of <= 144 tokens, 16-byte aligned rows): anything else is
// SST_ERR_UNSUPPORTED and the caller normalises outside (sst_amd/sst_basic_block.py WindowAttention).
int sst_sra_attn_cos_fwd_f32(const float* d_q, const float* d_k, const float* d_v, int64_t ldq, int64_t ldk, int64_t ldv,
                             const int32_t* d_tok, const int32_t* d_winoff, const int32_t* d_win_order, int64_t n_windows,
                             int n_heads, const float* d_head_scale, int max_tokens, float* d_o, int64_t ldo, float* d_lse,
                             void* stream) {
  if (!d_head_scale) return SST_ERR_ARG;
  if (n_heads < 1 || n_heads % kGH != 0 || max_tokens <= 0 || max_tokens > kMaxTilesMfma * 16 || ((ldq | ldk | ldv | ldo) % 4) != 0 ||
      !aligned16(d_q) || !aligned16(d_k) || !aligned16(d_v) || !aligned16(d_o))
    return SST_ERR_UNSUPPORTED;
  g_win_order = d_win_order;
  g_head_scale = d_head_scale;
  const int rc = sst_sra_attn_fwd_f32(d_q, d_k, d_v, ldq, ldk, ldv, d_tok, d_winoff, n_windows, n_heads, 1.0f, max_tokens, 0, d_o,
                                      ldo, d_lse, stream);
  g_head_scale = nullptr;
  g_win_order = nullptr;
  return rc;
}

// Backward of the above.  d_dq / d_dk are the gradients of the UN-normalised q / k; d_r [n_tokens, n_heads] receives
// normalize(q) . d normalize(q) per (token, head): d head_scale[h] = sum_tokens d_r[:, h] / head_scale[h].
int sst_sra_attn_cos_bwd_f32(const float* d_q, const float* d_k, const float* d_v, const float* d_o, const float* d_do,
                             const float* d_lse, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t lddo,
                             const int32_t* d_tok, const int32_t* d_winoff, const int32_t* d_win_order, int64_t n_windows,
                             int64_t n_tokens, int n_heads, const float* d_head_scale, int max_tokens, float* d_dq, float* d_dk,
                             float* d_dv, int64_t lddq, int64_t lddk, int64_t lddv, float* d_r, void* stream) {
  if (!d_head_scale || !d_r) return SST_ERR_ARG;
  if (n_heads < 1 || n_heads % kGH != 0 || max_tokens <= 0 || max_tokens > kMaxTilesMfma * 16 ||
      ((ldq | ldk | ldv | ldo | lddo | lddq | lddk | lddv) % 4) != 0 || !aligned16(d_q) || !aligned16(d_k) || !aligned16(d_v) ||
      !aligned16(d_o) || !aligned16(d_do) || !aligned16(d_dq) || !aligned16(d_dk) || !aligned16(d_dv))
    return SST_ERR_UNSUPPORTED;
  g_win_order = d_win_order;
  g_head_scale = d_head_scale;
  g_cos_r = d_r;
  const int rc = sst_sra_attn_bwd_f32(d_q, d_k, d_v, d_o, d_do, d_lse, ldq, ldk, ldv, ldo, lddo, d_tok, d_winoff, n_windows,
                                      n_tokens, n_heads, 1.0f, max_tokens, 0, d_dq, d_dk, d_dv, lddq, lddk, lddv, nullptr, stream);
  g_cos_r = nullptr;
  g_head_scale = nullptr;
  g_win_order = nullptr;
  return rc;
}

int sst_sra_attn_profile_next_fwd(void* start, void* stop) {
  g_prof_start = (hipEvent_t)start;
  g_prof_stop = (hipEvent_t)stop;
  return SST_OK;
}

int sst_sra_attn_profile_next_bwd(void* start, void* stop) {
  g_prof_bwd_start = (hipEvent_t)start;
  g_prof_bwd_stop = (hipEvent_t)stop;
  return SST_OK;
}

}  // extern "C"
