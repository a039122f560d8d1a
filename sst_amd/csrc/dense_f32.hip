// Tall fp32 linear layers with the weight matrix resident in LDS (gfx950):
//   Y[M, N] = epilogue(X[M, K] W^T + b),  W given as [N][K] rows (trans_w = 0) or as [K][N] rows (trans_w = 1, the data
//   gradient dY[M, out] w[out, in] of a layer whose parameter is w), (K, N) in {(128,128), (128,256), (256,128)}.
// The projections and the FFN of an SRA encoder layer, forward and data gradient: sst_basic_block_v2.py:41-75, 104-126.
// Exact fp32: v_mfma_f32_16x16x4_f32 (the same arithmetic as an fmaf chain), fp32 operands in HBM.
//
// Bound: the fp32 matrix pipe AND HBM at once - 32 flop per byte at K = N = 128 is exactly 157 TFLOP/s over 4.9 TB/s - so
// the kernel has to keep the pipe issuing while a whole row tile is in flight:
//   * W lives in LDS for the whole kernel (<= 136 KB, rows permuted as in csrc/dense_bf16.hip so that the TRANSPOSED
//     product Y^T = W X^T leaves a lane with 8 consecutive output columns of one row: two 16-byte stores);
//   * one wave per SIMD, each with a contiguous range of rows; the X fragments are the MFMA B operand straight from
//     global memory - a 16-byte load gives a lane 4 consecutive k of its row, which are then 4 consecutive MFMA k-steps
//     (the k order inside a product is free as long as both operands agree) - and the next 32-row tile is prefetched
//     during the ~14 us of MFMAs of the current one; no LDS round trip for X, no barrier after the weight fill;
//   * bias, GELU (+ stored pre-activation), multiplication by the activation's derivative, and the residual sum of a
//     data gradient ride on the epilogue, where they overlap the other waves' MFMAs - this is what removes the
//     separate GELU / GELU-backward / add passes over [M, 256] (the hand-pipelined tall_gemm.hip could not hide them).
#include <math.h>
#include <stdlib.h>
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float erf_as(float z, float& e) {  // Abramowitz & Stegun 7.1.26, |error| <= 1.5e-7
  const float az = fabsf(z);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, az, 1.f));
  e = __expf(-az * az);
  const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
  return copysignf(fmaf(-poly, e, 1.f), z);
}
__device__ __forceinline__ float gelu_f(float x) {
  float e;
  return 0.5f * x * (1.f + erf_as(x * 0.70710678118654752f, e));
}
__device__ __forceinline__ float gelu_grad_f(float x) {
  float e;
  const float phi = 0.5f * (1.f + erf_as(x * 0.70710678118654752f, e));
  return fmaf(x * 0.3989422804014327f, e, phi);
}

enum { kEpiBias = 0, kEpiGelu = 1, kEpiRelu = 2, kEpiMulGeluGrad = 3, kEpiMulReluGrad = 4, kEpiAdd = 5, kEpiAddLN = 6 };

// kEpiAddLN (N = 128, 8-wave kernel): y = LayerNorm(x W^T + bias + residual) - `norm(src + src2)` of
// sst_basic_block_v2.py:113-118 in the epilogue of the projection that produces src2.  A row's 128 columns sit in the four
// lanes (g = 0..3) of its MFMA column: two shuffle reductions give the statistics.  Also written: the sum (for the
// backward pass), (mean, rstd), and optionally y + pos_table[pos_idx[row]] (the next layer's q / k input).
struct ln_epi {
  const float* w;
  const float* b;
  float eps;
  float2* stats;
  const float* pos_table;
  const int32_t* pos_idx;
  float* yp;
};

__device__ __forceinline__ int w_lds_row(int n) {  // see csrc/dense_bf16.hip
  const int tp = n >> 5, within = n & 31;
  return 16 * (2 * tp + ((within >> 2) & 1)) + ((within >> 3) << 2) + (within & 3);
}

// 128 output columns x one 32-row (TWO) / 16-row step: W fragments are read one group ahead of the MFMAs that consume
// them; the scheduling barriers keep the compiler from sinking the LDS read next to its first use, where its latency
// would be exposed every 8 MFMAs (measured: 42 us -> see profiles for K = N = 128).
template <int K, bool TWO, typename EMIT>
__device__ __forceinline__ void mfma_phase(const float* __restrict__ wbase, const f32x4 (&xb)[2][K / 16], f32x4 (&acc)[2][8],
                                           EMIT&& emit) {
  constexpr int RS = K + 4, KJ = K / 16, HT = 8, G = KJ * HT;
  f32x4 wf = *(const f32x4*)(wbase);
#pragma clang loop unroll(full)
  for (int q = 0; q < G; ++q) {
    const int j = q / HT, T = q - j * HT;
    const int qn = q + 1 < G ? q + 1 : q;
    const f32x4 wn = *(const f32x4*)(wbase + (qn % HT) * 16 * RS + (qn / HT) * 16);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      acc[0][T] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[x], xb[0][j][x], acc[0][T], 0, 0, 0);
      if (TWO) acc[1][T] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[x], xb[1][j][x], acc[1][T], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    // one eighth of the PREVIOUS phase's epilogue (bias / activation / stores of one 16-row x 32-column unit) behind
    // every G / 8 groups: its arithmetic and its store issue overlap this phase's MFMAs instead of stalling the wave
    // (one wave per SIMD: nobody else would use the pipe meanwhile)
    if ((q + 1) % (G / 8) == 0) emit((q + 1) / (G / 8) - 1);
    wf = wn;
  }
}

// The same for ONE 16-row tile (the 8-wave variant below): two column tiles per group, their MFMAs interleaved, so that
// consecutive MFMAs never depend on each other; 4 emission slots (one per 32-column unit of the previous phase).
template <int K, typename EMIT>
__device__ __forceinline__ void mfma_phase_single(const float* __restrict__ wbase, const f32x4 (&xb)[K / 16], f32x4 (&acc)[8],
                                                  EMIT&& emit) {
  constexpr int RS = K + 4, KJ = K / 16, HP = 4, G = KJ * HP;
  f32x4 wa = *(const f32x4*)(wbase), wb = *(const f32x4*)(wbase + 16 * RS);
#pragma clang loop unroll(full)  // the X fragments are indexed by the group: anything less than a full unroll sends them to scratch
  for (int q = 0; q < G; ++q) {
    const int j = q / HP, P = q - j * HP;
    const int qn = q + 1 < G ? q + 1 : q;
    const float* pn = wbase + (2 * (qn % HP)) * 16 * RS + (qn / HP) * 16;
    const f32x4 na = *(const f32x4*)(pn), nb = *(const f32x4*)(pn + 16 * RS);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      acc[2 * P] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[x], xb[j][x], acc[2 * P], 0, 0, 0);
      acc[2 * P + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[x], xb[j][x], acc[2 * P + 1], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if ((q + 1) % (G / 4) == 0) emit((q + 1) / (G / 4) - 1);
    wa = na;
    wb = nb;
  }
}

template <int K, int N, int EPI>
__global__ __launch_bounds__(256, 1) void tall_linear_lds_f32_k(
    const float* __restrict__ X, int64_t ldx, const float* __restrict__ W, int64_t ldw, int trans_w,
    const float* __restrict__ bias, int64_t M, int rows_per_wave, float* __restrict__ Y, int64_t ldy,
    const float* __restrict__ aux_in, float* __restrict__ aux_out, int64_t ldaux) {
  constexpr int RS = K + 4;  // LDS row stride in floats: the 16 lanes of a ds_read_b128 phase on distinct banks
  constexpr int KJ = K / 16;
  extern __shared__ __attribute__((aligned(16))) float smem_f[];
  float* wimg = smem_f;
  float* bimg = smem_f + N * RS;
  const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  int64_t r0 = wave * rows_per_wave;
  const int64_t r1 = r0 + rows_per_wave < M ? r0 + rows_per_wave : M;
  f32x4 xb[2][KJ], xn[2][KJ];
  auto load_x = [&](int64_t r, f32x4 (&dst)[2][KJ]) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      int64_t row = r + 16 * t + c;
      row = row < M ? row : M - 1;
      const float* p = X + row * ldx + 4 * g;
#pragma unroll
      for (int j = 0; j < KJ; ++j) dst[t][j] = *(const f32x4*)(p + 16 * j);
    }
  };
  load_x(r0 < M ? r0 : M - 1, xb);  // in flight while the weights are copied to LDS
  // weight fill: batches of 8 independent 16-byte loads per thread before the LDS writes (a load -> write loop is paced
  // by one L2 round trip per iteration: 32 iterations for 128 KB, ~17 us measured as idle matrix pipe)
  constexpr int CHUNKS = N * K / 4, BATCH = 8;
  static_assert(CHUNKS % (256 * BATCH) == 0, "fill loop assumes a whole number of batches");
  for (int base = threadIdx.x; base < CHUNKS; base += 256 * BATCH) {
    f32x4 v[BATCH];
    if (!trans_w) {
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        const int idx = base + u * 256, n = idx / (K / 4), ch = idx - n * (K / 4);
        v[u] = *(const f32x4*)(W + (size_t)n * ldw + ch * 4);
      }
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        const int idx = base + u * 256, n = idx / (K / 4), ch = idx - n * (K / 4);
        *(f32x4*)(wimg + w_lds_row(n) * RS + ch * 4) = v[u];
      }
    } else {  // W[n][k] = w[k][n]: 16-byte reads along n, scattered to four LDS rows
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        const int idx = base + u * 256, k = idx / (N / 4), n4 = (idx - k * (N / 4)) * 4;
        v[u] = *(const f32x4*)(W + (size_t)k * ldw + n4);
      }
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        const int idx = base + u * 256, k = idx / (N / 4), n4 = (idx - k * (N / 4)) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) wimg[w_lds_row(n4 + e) * RS + k] = v[u][e];
      }
    }
  }
  for (int n = threadIdx.x; n < N; n += 256) bimg[n] = bias != nullptr ? bias[n] : 0.f;
  __syncthreads();
  if (r0 >= r1) return;
  const float* wlane = wimg + c * RS + 4 * g;

  // the finished accumulators of one phase wait in `pend` and are written out during the next phase
  f32x4 pend[2][8];
  int64_t pend_r0 = 0;
  int pend_nh = 0;
  bool pend_valid = false;
  auto emit = [&](int slot) {
    if (!pend_valid) return;
    const int t = slot >> 2, tp = slot & 3;
    const int64_t row = pend_r0 + 16 * t + c;
    if (row >= r1) return;
    const int n0 = 128 * pend_nh + 32 * tp + 8 * g;
    const f32x4 b0 = *(const f32x4*)(bimg + n0), b1 = *(const f32x4*)(bimg + n0 + 4);
    f32x4 v0 = pend[t][2 * tp] + b0, v1 = pend[t][2 * tp + 1] + b1;
    if (EPI == kEpiGelu || EPI == kEpiRelu) {
      if (aux_out != nullptr) {
        *(f32x4*)(aux_out + row * ldaux + n0) = v0;
        *(f32x4*)(aux_out + row * ldaux + n0 + 4) = v1;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v0[r] = EPI == kEpiGelu ? gelu_f(v0[r]) : fmaxf(v0[r], 0.f);
        v1[r] = EPI == kEpiGelu ? gelu_f(v1[r]) : fmaxf(v1[r], 0.f);
      }
    }
    if (EPI == kEpiMulGeluGrad || EPI == kEpiMulReluGrad) {
      const f32x4 p0 = *(const f32x4*)(aux_in + row * ldaux + n0), p1 = *(const f32x4*)(aux_in + row * ldaux + n0 + 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v0[r] *= EPI == kEpiMulGeluGrad ? gelu_grad_f(p0[r]) : (p0[r] > 0.f ? 1.f : 0.f);
        v1[r] *= EPI == kEpiMulGeluGrad ? gelu_grad_f(p1[r]) : (p1[r] > 0.f ? 1.f : 0.f);
      }
    }
    if (EPI == kEpiAdd) {  // + a second [M, N] term; aux_in may be Y itself (in-place accumulation)
      v0 += *(const f32x4*)(aux_in + row * ldaux + n0);
      v1 += *(const f32x4*)(aux_in + row * ldaux + n0 + 4);
    }
    *(f32x4*)(Y + row * ldy + n0) = v0;
    *(f32x4*)(Y + row * ldy + n0 + 4) = v1;
  };

  for (; r0 < r1; r0 += 32) {
    asm volatile("" ::: "memory");  // W fragments are re-read from LDS per row tile (never hoisted into registers)
    const bool more = r0 + 32 < r1;
    const bool two = r0 + 16 < r1;  // the second 16-row tile of this step exists
    // everything issued so far has landed (this step's X tile was requested a whole step ago) BEFORE the next tile is
    // requested: otherwise the wait the compiler places in front of the first MFMA - vmcnt(15), sized for the path on
    // which no prefetch was issued - makes the phase wait for the loads just issued, and nothing overlaps (measured:
    // kernel time = load time + MFMA time + store time)
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), expcnt / lgkmcnt untouched
    if (more) load_x(r0 + 32, xn);
#pragma unroll
    for (int nh = 0; nh < N / 128; ++nh) {  // 128 output columns at a time
      f32x4 acc[2][8];
#pragma unroll
      for (int T = 0; T < 8; ++T) {
        acc[0][T] = (f32x4){0.f, 0.f, 0.f, 0.f};
        acc[1][T] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      if (two)
        mfma_phase<K, true>(wlane + nh * 8 * 16 * (K + 4), xb, acc, emit);
      else
        mfma_phase<K, false>(wlane + nh * 8 * 16 * (K + 4), xb, acc, emit);
#pragma unroll
      for (int T = 0; T < 8; ++T) {
        pend[0][T] = acc[0][T];
        pend[1][T] = acc[1][T];
      }
      pend_r0 = r0;
      pend_nh = nh;
      pend_valid = true;
    }
    if (more) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < KJ; ++j) xb[t][j] = xn[t][j];
    }
  }
#pragma unroll
  for (int slot = 0; slot < 8; ++slot) emit(slot);
}

// 8-wave variant: two waves per SIMD over the same LDS image, 16-row steps.  While one wave of a SIMD is between phases
// (waiting for its X tile, issuing stores, copying registers) the other one owns the matrix pipe: the load / MFMA /
// store phases that are strictly additive with one wave per SIMD overlap across the pair.
template <int K, int N, int EPI>
__global__ __launch_bounds__(512, 2) void tall_linear_lds8_f32_k(
    const float* __restrict__ X, int64_t ldx, const float* __restrict__ W, int64_t ldw, int trans_w,
    const float* __restrict__ bias, int64_t M, int rows_per_wave, float* __restrict__ Y, int64_t ldy,
    const float* __restrict__ aux_in, float* __restrict__ aux_out, int64_t ldaux, const ln_epi ln) {
  static_assert(EPI != kEpiAddLN || N == 128, "the LayerNorm epilogue needs a whole row in one accumulator set");
  constexpr int RS = K + 4, KJ = K / 16, NTH = 512;
  extern __shared__ __attribute__((aligned(16))) float smem_f[];
  float* wimg = smem_f;
  float* bimg = smem_f + N * RS;
  const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  const int64_t wave = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 6);
  int64_t r0 = wave * rows_per_wave;
  const int64_t r1 = r0 + rows_per_wave < M ? r0 + rows_per_wave : M;
  f32x4 xb[KJ], xn[KJ];
  auto load_x = [&](int64_t r, f32x4 (&dst)[KJ]) {
    int64_t row = r + c;
    row = row < M ? row : M - 1;
    const float* p = X + row * ldx + 4 * g;
#pragma unroll
    for (int j = 0; j < KJ; ++j) dst[j] = *(const f32x4*)(p + 16 * j);
  };
  load_x(r0 < M ? r0 : M - 1, xb);
  constexpr int CHUNKS = N * K / 4, BATCH = 8;
  static_assert(CHUNKS % (NTH * BATCH) == 0, "fill loop assumes a whole number of batches");
  for (int base = threadIdx.x; base < CHUNKS; base += NTH * BATCH) {
    f32x4 v[BATCH];
    if (!trans_w) {
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        const int idx = base + u * NTH, n = idx / (K / 4), ch = idx - n * (K / 4);
        v[u] = *(const f32x4*)(W + (size_t)n * ldw + ch * 4);
      }
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        const int idx = base + u * NTH, n = idx / (K / 4), ch = idx - n * (K / 4);
        *(f32x4*)(wimg + w_lds_row(n) * RS + ch * 4) = v[u];
      }
    } else {
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        const int idx = base + u * NTH, k = idx / (N / 4), n4 = (idx - k * (N / 4)) * 4;
        v[u] = *(const f32x4*)(W + (size_t)k * ldw + n4);
      }
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        const int idx = base + u * NTH, k = idx / (N / 4), n4 = (idx - k * (N / 4)) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) wimg[w_lds_row(n4 + e) * RS + k] = v[u][e];
      }
    }
  }
  for (int n = threadIdx.x; n < N; n += NTH) {
    bimg[n] = bias != nullptr ? bias[n] : 0.f;
    if (EPI == kEpiAddLN) {
      bimg[N + n] = ln.w[n];
      bimg[2 * N + n] = ln.b[n];
    }
  }
  __syncthreads();
  if (r0 >= r1) return;
  const float* wlane = wimg + c * RS + 4 * g;

  f32x4 pend[8];
  int64_t pend_r0 = 0;
  int pend_nh = 0;
  bool pend_valid = false;
  float ln_rstd = 0.f;
  auto emit = [&](int tp) {
    if (!pend_valid) return;
    const int64_t row = pend_r0 + c;
    if (row >= r1) return;  // uniform over the four lanes (g) that share the row
    if (EPI == kEpiAddLN) {
      if (tp == 0) {  // the whole row: sum = product + bias + residual, statistics; pend <- sum - mean
        float sum = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int n0 = 32 * u + 8 * g;
          pend[2 * u] += *(const f32x4*)(bimg + n0) + *(const f32x4*)(aux_in + row * ldaux + n0);
          pend[2 * u + 1] += *(const f32x4*)(bimg + n0 + 4) + *(const f32x4*)(aux_in + row * ldaux + n0 + 4);
          if (aux_out != nullptr) {
            *(f32x4*)(aux_out + row * ldaux + n0) = pend[2 * u];
            *(f32x4*)(aux_out + row * ldaux + n0 + 4) = pend[2 * u + 1];
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) sum += pend[2 * u][r] + pend[2 * u + 1][r];
        }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float mean = sum * (1.f / 128.f);
        float sq = 0.f;
#pragma unroll
        for (int T = 0; T < 8; ++T)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            pend[T][r] -= mean;
            sq = fmaf(pend[T][r], pend[T][r], sq);
          }
        sq += __shfl_xor(sq, 16, 64);
        sq += __shfl_xor(sq, 32, 64);
        ln_rstd = rsqrtf(sq * (1.f / 128.f) + ln.eps);
        if (g == 0) ln.stats[row] = make_float2(mean, ln_rstd);
      }
      const int n0 = 32 * tp + 8 * g;
      const f32x4 y0 = pend[2 * tp] * ln_rstd * *(const f32x4*)(bimg + N + n0) + *(const f32x4*)(bimg + 2 * N + n0);
      const f32x4 y1 = pend[2 * tp + 1] * ln_rstd * *(const f32x4*)(bimg + N + n0 + 4) + *(const f32x4*)(bimg + 2 * N + n0 + 4);
      *(f32x4*)(Y + row * ldy + n0) = y0;
      *(f32x4*)(Y + row * ldy + n0 + 4) = y1;
      if (ln.yp != nullptr) {
        const float* prow = ln.pos_table + (size_t)ln.pos_idx[row] * 128 + n0;
        *(f32x4*)(ln.yp + row * 128 + n0) = y0 + *(const f32x4*)(prow);
        *(f32x4*)(ln.yp + row * 128 + n0 + 4) = y1 + *(const f32x4*)(prow + 4);
      }
      return;
    }
    const int n0 = 128 * pend_nh + 32 * tp + 8 * g;
    const f32x4 b0 = *(const f32x4*)(bimg + n0), b1 = *(const f32x4*)(bimg + n0 + 4);
    f32x4 v0 = pend[2 * tp] + b0, v1 = pend[2 * tp + 1] + b1;
    if (EPI == kEpiGelu || EPI == kEpiRelu) {
      if (aux_out != nullptr) {
        *(f32x4*)(aux_out + row * ldaux + n0) = v0;
        *(f32x4*)(aux_out + row * ldaux + n0 + 4) = v1;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v0[r] = EPI == kEpiGelu ? gelu_f(v0[r]) : fmaxf(v0[r], 0.f);
        v1[r] = EPI == kEpiGelu ? gelu_f(v1[r]) : fmaxf(v1[r], 0.f);
      }
    }
    if (EPI == kEpiMulGeluGrad || EPI == kEpiMulReluGrad) {
      const f32x4 p0 = *(const f32x4*)(aux_in + row * ldaux + n0), p1 = *(const f32x4*)(aux_in + row * ldaux + n0 + 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v0[r] *= EPI == kEpiMulGeluGrad ? gelu_grad_f(p0[r]) : (p0[r] > 0.f ? 1.f : 0.f);
        v1[r] *= EPI == kEpiMulGeluGrad ? gelu_grad_f(p1[r]) : (p1[r] > 0.f ? 1.f : 0.f);
      }
    }
    if (EPI == kEpiAdd) {
      v0 += *(const f32x4*)(aux_in + row * ldaux + n0);
      v1 += *(const f32x4*)(aux_in + row * ldaux + n0 + 4);
    }
    *(f32x4*)(Y + row * ldy + n0) = v0;
    *(f32x4*)(Y + row * ldy + n0 + 4) = v1;
  };

  // K = 256 with the LayerNorm epilogue does not fit a second X tile in 256 registers (it went to scratch: 239 us): the
  // next tile is then loaded into the same registers after the phase, the partner wave of the SIMD covers the latency
  constexpr bool PREFETCH = !(EPI == kEpiAddLN && K == 256);
  for (; r0 < r1; r0 += 16) {
    asm volatile("" ::: "memory");
    const bool more = r0 + 16 < r1;
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): see tall_linear_lds_f32_k
    if (PREFETCH && more) load_x(r0 + 16, xn);
#pragma unroll
    for (int nh = 0; nh < N / 128; ++nh) {
      f32x4 acc[8];
#pragma unroll
      for (int T = 0; T < 8; ++T) acc[T] = (f32x4){0.f, 0.f, 0.f, 0.f};
      mfma_phase_single<K>(wlane + nh * 8 * 16 * RS, xb, acc, emit);
#pragma unroll
      for (int T = 0; T < 8; ++T) pend[T] = acc[T];
      pend_r0 = r0;
      pend_nh = nh;
      pend_valid = true;
    }
    if (more) {
      if (PREFETCH) {
#pragma unroll
        for (int j = 0; j < KJ; ++j) xb[j] = xn[j];
      } else {
        load_x(r0 + 16, xb);
      }
    }
  }
#pragma unroll
  for (int tp = 0; tp < 4; ++tp) emit(tp);
}

template <int K, int N, int EPI>
int launch_linear(const float* x, int64_t ldx, const float* w, int64_t ldw, int trans_w, const float* bias, int64_t m,
                  float* y, int64_t ldy, const float* aux_in, float* aux_out, int64_t ldaux, hipStream_t st,
                  const ln_epi ln = ln_epi()) {
  constexpr int lds = (N * (K + 4) + N * (EPI == kEpiAddLN ? 3 : 1)) * 4;
  static int variant = -1;  // SST_AMD_LDS_LINEAR_WAVES = 4: one wave per SIMD, 32-row steps; 8 (default): two, 16-row steps
  if (variant < 0) {
    const char* e = getenv("SST_AMD_LDS_LINEAR_WAVES");
    variant = (e != nullptr && atoi(e) == 4 && EPI != kEpiAddLN) ? 4 : 8;
    if constexpr (EPI != kEpiAddLN)
      SST_HIP(hipFuncSetAttribute((const void*)tall_linear_lds_f32_k<K, N, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    SST_HIP(hipFuncSetAttribute((const void*)tall_linear_lds8_f32_k<K, N, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  }
  // one workgroup per CU (up to 136 KB of LDS); every wave a contiguous row range, a multiple of 8
  int64_t blocks = 256;
  int64_t rpw = sst_align_up(sst_div_up(m, blocks * variant), 8);
  blocks = sst_div_up(m, rpw * variant);
  if constexpr (EPI != kEpiAddLN) {
    if (variant == 4) {
      hipLaunchKernelGGL((tall_linear_lds_f32_k<K, N, EPI>), dim3((unsigned)blocks), dim3(256), lds, st, x, ldx, w, ldw,
                         trans_w, bias, m, (int)rpw, y, ldy, aux_in, aux_out, ldaux);
      return SST_OK;
    }
  }
  hipLaunchKernelGGL((tall_linear_lds8_f32_k<K, N, EPI>), dim3((unsigned)blocks), dim3(512), lds, st, x, ldx, w, ldw, trans_w,
                     bias, m, (int)rpw, y, ldy, aux_in, aux_out, ldaux, ln);
  return SST_OK;
}

template <int K, int N>
int dispatch_epi(int epi, const float* x, int64_t ldx, const float* w, int64_t ldw, int trans_w, const float* bias, int64_t m,
                 float* y, int64_t ldy, const float* aux_in, float* aux_out, int64_t ldaux, hipStream_t st) {
#define SST_CASE(E) \
  case E: return launch_linear<K, N, E>(x, ldx, w, ldw, trans_w, bias, m, y, ldy, aux_in, aux_out, ldaux, st)
  switch (epi) {
    SST_CASE(kEpiBias);
    SST_CASE(kEpiGelu);
    SST_CASE(kEpiRelu);
    SST_CASE(kEpiMulGeluGrad);
    SST_CASE(kEpiMulReluGrad);
    SST_CASE(kEpiAdd);
  }
#undef SST_CASE
  return SST_ERR_ARG;
}

bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

extern "C" {

int sst_tall_linear_epi_f32(const float* d_x, int64_t ldx, const float* d_w, int64_t ldw, int trans_w, const float* d_bias,
                            int64_t m, int k, int n, int epilogue, const float* d_aux_in, float* d_aux_out, int64_t ldaux,
                            float* d_y, int64_t ldy, void* stream) {
  if (m < 0 || !d_w || epilogue < 0 || epilogue > kEpiAdd) return SST_ERR_ARG;
  if (m == 0) return SST_OK;
  if (!d_x || !d_y || (ldx & 3) || (ldy & 3) || (ldw & 3) || !aligned16(d_x) || !aligned16(d_y) || !aligned16(d_w))
    return SST_ERR_ARG;
  if (epilogue >= kEpiMulGeluGrad && (!d_aux_in || (ldaux & 3) || !aligned16(d_aux_in))) return SST_ERR_ARG;
  if (d_aux_out && ((ldaux & 3) || !aligned16(d_aux_out))) return SST_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (k == 128 && n == 128)
    rc = dispatch_epi<128, 128>(epilogue, d_x, ldx, d_w, ldw, trans_w, d_bias, m, d_y, ldy, d_aux_in, d_aux_out, ldaux, st);
  else if (k == 128 && n == 256)
    rc = dispatch_epi<128, 256>(epilogue, d_x, ldx, d_w, ldw, trans_w, d_bias, m, d_y, ldy, d_aux_in, d_aux_out, ldaux, st);
  else if (k == 256 && n == 128)
    rc = dispatch_epi<256, 128>(epilogue, d_x, ldx, d_w, ldw, trans_w, d_bias, m, d_y, ldy, d_aux_in, d_aux_out, ldaux, st);
  else
    return SST_ERR_UNSUPPORTED;
  if (rc) return rc;
  SST_LAUNCH_CHECK();
  return SST_OK;
}

int sst_tall_linear_ln_f32(const float* d_x, int64_t ldx, const float* d_w, int64_t ldw, const float* d_bias, int64_t m, int k,
                           const float* d_res, int64_t ldres, const float* d_ln_weight, const float* d_ln_bias, float eps,
                           float* d_y, float* d_sum, float* d_stats, const float* d_pos_table, const int32_t* d_pos_idx,
                           float* d_y_plus_pos, void* stream) {
  if (m < 0 || !d_w || !d_ln_weight || !d_ln_bias || !d_stats) return SST_ERR_ARG;
  if (m == 0) return SST_OK;
  if (!d_x || !d_y || !d_res || (ldx & 3) || (ldw & 3) || (ldres & 3) || !aligned16(d_x) || !aligned16(d_y) ||
      !aligned16(d_w) || !aligned16(d_res) || (d_sum && !aligned16(d_sum)))
    return SST_ERR_ARG;
  if ((d_pos_table != nullptr) != (d_pos_idx != nullptr) || (d_pos_table != nullptr) != (d_y_plus_pos != nullptr))
    return SST_ERR_ARG;
  ln_epi ln;
  ln.w = d_ln_weight;
  ln.b = d_ln_bias;
  ln.eps = eps;
  ln.stats = (float2*)d_stats;
  ln.pos_table = d_pos_table;
  ln.pos_idx = d_pos_idx;
  ln.yp = d_y_plus_pos;
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (k == 128)
    rc = launch_linear<128, 128, kEpiAddLN>(d_x, ldx, d_w, ldw, 0, d_bias, m, d_y, 128, d_res, d_sum, ldres, st, ln);
  else if (k == 256)
    rc = launch_linear<256, 128, kEpiAddLN>(d_x, ldx, d_w, ldw, 0, d_bias, m, d_y, 128, d_res, d_sum, ldres, st, ln);
  else
    return SST_ERR_UNSUPPORTED;
  if (rc) return rc;
  SST_LAUNCH_CHECK();
  return SST_OK;
}

}  // extern "C"
