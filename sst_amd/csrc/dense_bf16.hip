// Tall linear layers with bf16 storage on gfx950 (the reduced-precision mode of the SRA encoder layers):
//   Y[M, N] = epilogue(X[M, K] W[N, K]^T + b)       projections / FFN, forward and data gradient
//   dW[N, K] = dY[M, N]^T X[M, K],  db = colsum(dY)  weight gradients of a whole encoder layer in ONE launch
// M ~ 1e5 tokens, K and N in {128, 256}: sst_basic_block_v2.py:41-75 (in_proj / out_proj of nn.MultiheadAttention) and
// :104-126 (linear1, activation, linear2).  bf16 operands in HBM, fp32 accumulation, fp32 bias / parameter gradients.
//
// Bound: HBM.  One [M, 128] -> [M, 256] product moves 69 MB and needs 5.9 GFLOP = 2.4 us of the bf16 matrix pipe, 14 us
// of HBM at 5 TB/s; the library's generic tiles take 36-48 us on these shapes (profiles/r02).  So the design spends the
// matrix pipe freely to keep the memory side simple:
//   * forward: the whole weight matrix lives in LDS (<= 68 KB, two workgroups per CU), rows permuted so that the
//     TRANSPOSED product Y^T = W X^T leaves every lane with 8 consecutive output columns of one row (one 16-byte
//     store).  A wave owns a contiguous range of rows; its X fragments are the MFMA B operand straight from global
//     memory (16-byte loads, a whole 256 B row consumed by the same wave), no LDS round trip, no barrier after the fill.
//   * weight gradient: the contraction runs over tokens, i.e. both operands are needed token-major per lane while HBM
//     holds them row (= token) major.  The transposition is done BY THE MATRIX CORE: a product with a 0/1 selection
//     matrix turns a row-major 16-token fragment into the D layout (lane = column, registers = tokens), which packed to
//     bf16 is exactly an operand of the next MFMA.  No LDS, no barrier; exact (selection of bf16 values in fp32).
#include <math.h>
#include <stdlib.h>
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short bf16_t;

__device__ __forceinline__ unsigned pack2(float lo, float hi) {  // one v_cvt_pk_bf16_f32 (RNE); compiler-visible
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ bf16_t f2bf(float v) { return (bf16_t)(pack2(v, 0.f) & 0xffffu); }
__device__ __forceinline__ float lo_f(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float hi_f(unsigned p) { return __uint_as_float(p & 0xffff0000u); }
__device__ __forceinline__ f32x4 mma32(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below the bf16 resolution of every consumer): one
// reciprocal, one exponential and five FMAs instead of the ~40 instructions of the library's erff.
// e = exp(-z^2) is returned too: the GELU derivative needs exp(-x^2 / 2) = e at z = x / sqrt(2).
__device__ __forceinline__ float erf_as(float z, float& e) {
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

// ------------------------------------------------------------------------------------------------------------------
// Y = epi(X W^T + b)
// ------------------------------------------------------------------------------------------------------------------
enum { kEpiBias = 0, kEpiGelu = 1, kEpiRelu = 2, kEpiMulGeluGrad = 3, kEpiMulReluGrad = 4, kEpiAdd = 5, kEpiAddLN = 6 };

// kEpiAddLN (N = 128): y = LayerNorm(x W^T + bias + residual) - `norm(src + src2)` of sst_basic_block_v2.py:113-118 in the
// epilogue of the projection that produces src2 (out_proj / linear2).  A row's 128 columns sit in the four lanes (g = 0..3)
// of its MFMA column, 32 values each: two shuffle reductions give the statistics.  Also written: the sum (bf16, for the
// backward pass), (mean, rstd), and optionally y + pos_table[pos_idx[row]] (the next layer's q / k input).
struct ln_epi {
  const float* w;
  const float* b;
  float eps;
  float2* stats;
  const float* pos_table;
  const int32_t* pos_idx;
  bf16_t* yp;
};

// LDS image of W: row rho = 16 T + i holds W[n(T, i)][0..K), n(T, i) = 32 (T >> 1) + 8 (i >> 2) + 4 (T & 1) + (i & 3):
// tile pair (2 tp, 2 tp + 1) then leaves lane (g, c) with columns 32 tp + 8 g .. + 7 of row c.
__device__ __forceinline__ int w_lds_row(int n) {
  const int tp = n >> 5, within = n & 31;
  return 16 * (2 * tp + ((within >> 2) & 1)) + ((within >> 3) << 2) + (within & 3);
}

template <int K, int N, int EPI, int NTH>
__global__ __launch_bounds__(NTH, (NTH == 512 ? 2 : (K * N <= 128 * 128 ? 3 : 2))) void tall_linear_bf16_k(const bf16_t* __restrict__ X, int64_t ldx,
                                                             const bf16_t* __restrict__ W, const float* __restrict__ bias,
                                                             int64_t M, int rows_per_wave, bf16_t* __restrict__ Y,
                                                             int64_t ldy, const bf16_t* __restrict__ aux_in,
                                                             bf16_t* __restrict__ aux_out, int64_t ldaux, const ln_epi ln) {
  static_assert(EPI != kEpiAddLN || N == 128, "the LayerNorm epilogue needs a whole row in one accumulator set");
  // the LDS image is FRAGMENT-contiguous (round 6, see csrc/dense_f32x6.hip): the 16 rows x 32 k of an MFMA A operand are one
  // 1 KiB block in lane order, a fragment read is base + 16 * lane - no bank conflicts under the real ds_read_b128 lane groups
  // (the row-major image with a K * 2 + 16 byte row stride had two lanes of every group on the same banks), no padding
  constexpr int KS = K / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* wimg = smem;
  float* bimg = (float*)(smem + N * K * 2);  // [N] bias, then (kEpiAddLN) [N] LayerNorm weight, [N] LayerNorm bias
  const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  const int64_t wave = (int64_t)blockIdx.x * (NTH / 64) + (threadIdx.x >> 6);
  int64_t r0 = wave * rows_per_wave;
  const int64_t r1 = r0 + rows_per_wave < M ? r0 + rows_per_wave : M;
  u32x4 xb[2][KS], xn[2][KS];
  auto load_x = [&](int64_t r, u32x4 (&dst)[2][KS]) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      int64_t row = r + 16 * t + c;
      row = row < M ? row : M - 1;
      const bf16_t* p = X + row * ldx + 8 * g;
#pragma unroll
      for (int s = 0; s < KS; ++s) dst[t][s] = *(const u32x4*)(p + 32 * s);
    }
  };
  load_x(r0 < M ? r0 : M - 1, xb);  // the first row tile is in flight while the weights are copied to LDS
  // weight fill: batches of 8 independent 16-byte loads per thread, then the LDS writes (a load -> write loop would be
  // paced by one L2 round trip per iteration)
  constexpr int CHUNKS = N * K / 8, BATCH = (CHUNKS / NTH >= 8 ? 8 : CHUNKS / NTH);
  static_assert(CHUNKS % (NTH * BATCH) == 0, "fill loop assumes a whole number of batches");
  for (int base = threadIdx.x; base < CHUNKS; base += NTH * BATCH) {
    u32x4 v[BATCH];
#pragma unroll
    for (int u = 0; u < BATCH; ++u) {
      const int idx = base + u * NTH, n = idx / (K / 8), ch = idx - n * (K / 8);
      v[u] = *(const u32x4*)(W + (size_t)n * K + ch * 8);
    }
#pragma unroll
    for (int u = 0; u < BATCH; ++u) {
      const int idx = base + u * NTH, n = idx / (K / 8), ch = idx - n * (K / 8);
      const int lr = w_lds_row(n);    // ch = 8-k chunk: k-step ch >> 2, k group ch & 3
      *(u32x4*)(wimg + ((lr >> 4) * KS + (ch >> 2)) * 1024 + (16 * (ch & 3) + (lr & 15)) * 16) = v[u];
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
  const unsigned char* wlane = wimg + lane * 16;
  for (; r0 < r1; r0 += 32) {
    asm volatile("" ::: "memory");  // W fragments are re-read from LDS per row tile (never hoisted into registers)
    const bool more = r0 + 32 < r1;
    // this step's X tile (requested a whole step ago) has landed BEFORE the next one is requested: the compiler sizes the
    // wait in front of the first MFMA for the path without a prefetch, which would otherwise stall on the new loads
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    if (more) load_x(r0 + 32, xn);
#pragma unroll
    for (int nh = 0; nh < N / 128; ++nh) {  // 128 output columns at a time: 64 accumulator registers
    constexpr int HT = 8;
    f32x4 acc[2][HT];
#pragma unroll
    for (int T = 0; T < HT; ++T) {
      acc[0][T] = (f32x4){0.f, 0.f, 0.f, 0.f};
      acc[1][T] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
#pragma unroll
      for (int T = 0; T < HT; ++T) {
        const u32x4 wf = *(const u32x4*)(wlane + ((nh * HT + T) * KS + s) * 1024);
        acc[0][T] = mma32(wf, xb[0][s], acc[0][T]);
        acc[1][T] = mma32(wf, xb[1][s], acc[1][T]);
      }
    }
    if (EPI == kEpiAddLN) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int64_t row = r0 + 16 * t + c;
        if (row < r1) {  // uniform over the four lanes (g) that share the row
          float v[4][8];
          float sum = 0.f;
#pragma unroll
          for (int tp = 0; tp < 4; ++tp) {
            const int n0 = 32 * tp + 8 * g;
            const f32x4 b0 = *(const f32x4*)(bimg + n0), b1 = *(const f32x4*)(bimg + n0 + 4);
            const u32x4 res = *(const u32x4*)(aux_in + row * ldaux + n0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              v[tp][r] = acc[t][2 * tp][r] + b0[r];
              v[tp][4 + r] = acc[t][2 * tp + 1][r] + b1[r];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              v[tp][2 * q] += lo_f(res[q]);
              v[tp][2 * q + 1] += hi_f(res[q]);
            }
            if (aux_out != nullptr)
              *(u32x4*)(aux_out + row * ldaux + n0) =
                  (u32x4){pack2(v[tp][0], v[tp][1]), pack2(v[tp][2], v[tp][3]), pack2(v[tp][4], v[tp][5]), pack2(v[tp][6], v[tp][7])};
#pragma unroll
            for (int e = 0; e < 8; ++e) sum += v[tp][e];
          }
          sum += __shfl_xor(sum, 16, 64);
          sum += __shfl_xor(sum, 32, 64);
          const float mean = sum * (1.f / 128.f);
          float sq = 0.f;
#pragma unroll
          for (int tp = 0; tp < 4; ++tp)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              v[tp][e] -= mean;
              sq = fmaf(v[tp][e], v[tp][e], sq);
            }
          sq += __shfl_xor(sq, 16, 64);
          sq += __shfl_xor(sq, 32, 64);
          const float rstd = rsqrtf(sq * (1.f / 128.f) + ln.eps);
          if (g == 0) ln.stats[row] = make_float2(mean, rstd);
          const float* prow = ln.yp != nullptr ? ln.pos_table + (size_t)ln.pos_idx[row] * 128 : nullptr;
#pragma unroll
          for (int tp = 0; tp < 4; ++tp) {
            const int n0 = 32 * tp + 8 * g;
            float y[8];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const f32x4 gw = *(const f32x4*)(bimg + N + n0 + 4 * h), gb = *(const f32x4*)(bimg + 2 * N + n0 + 4 * h);
#pragma unroll
              for (int r = 0; r < 4; ++r) y[4 * h + r] = fmaf(v[tp][4 * h + r] * rstd, gw[r], gb[r]);
            }
            *(u32x4*)(Y + row * ldy + n0) = (u32x4){pack2(y[0], y[1]), pack2(y[2], y[3]), pack2(y[4], y[5]), pack2(y[6], y[7])};
            if (ln.yp != nullptr) {
              const f32x4 p0 = *(const f32x4*)(prow + n0), p1 = *(const f32x4*)(prow + n0 + 4);
              *(u32x4*)(ln.yp + row * 128 + n0) = (u32x4){pack2(y[0] + p0[0], y[1] + p0[1]), pack2(y[2] + p0[2], y[3] + p0[3]),
                                                          pack2(y[4] + p1[0], y[5] + p1[1]), pack2(y[6] + p1[2], y[7] + p1[3])};
            }
          }
        }
      }
    } else {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int64_t row = r0 + 16 * t + c;
      if (row < r1) {
#pragma unroll
        for (int tp = 0; tp < HT / 2; ++tp) {
          const int n0 = 128 * nh + 32 * tp + 8 * g;
          const f32x4 b0 = *(const f32x4*)(bimg + n0), b1 = *(const f32x4*)(bimg + n0 + 4);
          float v[8];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[r] = acc[t][2 * tp][r] + b0[r];
            v[4 + r] = acc[t][2 * tp + 1][r] + b1[r];
          }
          if (EPI == kEpiGelu || EPI == kEpiRelu) {
            // the pre-activation is kept (bf16) for the backward pass; the activation acts on the ROUNDED value so
            // that the backward's derivative is taken at the point the forward used
            const u32x4 pre = {pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
            if (aux_out != nullptr) *(u32x4*)(aux_out + row * ldaux + n0) = pre;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float a = lo_f(pre[q]), b = hi_f(pre[q]);
              v[2 * q] = EPI == kEpiGelu ? gelu_f(a) : fmaxf(a, 0.f);
              v[2 * q + 1] = EPI == kEpiGelu ? gelu_f(b) : fmaxf(b, 0.f);
            }
          }
          if (EPI == kEpiMulGeluGrad || EPI == kEpiMulReluGrad) {
            const u32x4 pre = *(const u32x4*)(aux_in + row * ldaux + n0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float a = lo_f(pre[q]), b = hi_f(pre[q]);
              v[2 * q] *= EPI == kEpiMulGeluGrad ? gelu_grad_f(a) : (a > 0.f ? 1.f : 0.f);
              v[2 * q + 1] *= EPI == kEpiMulGeluGrad ? gelu_grad_f(b) : (b > 0.f ? 1.f : 0.f);
            }
          }
          if (EPI == kEpiAdd) {  // + a second bf16 [M, N] term (residual branch of a data gradient)
            const u32x4 add = *(const u32x4*)(aux_in + row * ldaux + n0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              v[2 * q] += lo_f(add[q]);
              v[2 * q + 1] += hi_f(add[q]);
            }
          }
          const u32x4 o = {pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
          *(u32x4*)(Y + row * ldy + n0) = o;
        }
      }
    }
    }
    }
    if (more) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int s = 0; s < KS; ++s) xb[t][s] = xn[t][s];
    }
  }
}

template <int K, int N, int EPI>
int launch_linear(const bf16_t* x, int64_t ldx, const bf16_t* w, const float* bias, int64_t m, bf16_t* y, int64_t ldy,
                  const bf16_t* aux_in, bf16_t* aux_out, int64_t ldaux, hipStream_t st, const ln_epi ln = ln_epi()) {
  constexpr int lds = N * K * 2 + N * 4 * (EPI == kEpiAddLN ? 3 : 1);
  // SST_AMD_BF16_LINEAR_WAVES = 8: one 8-wave workgroup per CU (the weights are copied to LDS once per CU, not 2-3 times)
  static int nth = 0;
  if (nth == 0) {
    const char* e = getenv("SST_AMD_BF16_LINEAR_WAVES");
    nth = (e != nullptr && atoi(e) == 8) ? 512 : 256;
  }
  static unsigned long long configured = 0;     // the attribute belongs to a (kernel, device) pair
  if (sst_first_use_on_device(&configured)) {
    SST_HIP(hipFuncSetAttribute((const void*)tall_linear_bf16_k<K, N, EPI, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    SST_HIP(hipFuncSetAttribute((const void*)tall_linear_bf16_k<K, N, EPI, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    sst_mark_device(&configured);
  }
  // persistent shape: every wave a contiguous row range (multiple of 16 rows)
  const int wpb = nth / 64;
  int64_t blocks = nth == 512 ? 256 : 256 * (lds <= 40 * 1024 ? 3 : 2);
  int64_t rpw = sst_align_up(sst_div_up(m, blocks * wpb), 16);
  blocks = sst_div_up(m, rpw * wpb);
  if (nth == 512)
    hipLaunchKernelGGL((tall_linear_bf16_k<K, N, EPI, 512>), dim3((unsigned)blocks), dim3(512), lds, st, x, ldx, w, bias, m,
                       (int)rpw, y, ldy, aux_in, aux_out, ldaux, ln);
  else
    hipLaunchKernelGGL((tall_linear_bf16_k<K, N, EPI, 256>), dim3((unsigned)blocks), dim3(256), lds, st, x, ldx, w, bias, m,
                       (int)rpw, y, ldy, aux_in, aux_out, ldaux, ln);
  return SST_OK;
}

template <int K, int N>
int dispatch_epi(int epi, const bf16_t* x, int64_t ldx, const bf16_t* w, const float* bias, int64_t m, bf16_t* y,
                 int64_t ldy, const bf16_t* aux_in, bf16_t* aux_out, int64_t ldaux, hipStream_t st) {
  switch (epi) {
    case kEpiBias: return launch_linear<K, N, kEpiBias>(x, ldx, w, bias, m, y, ldy, aux_in, aux_out, ldaux, st);
    case kEpiGelu: return launch_linear<K, N, kEpiGelu>(x, ldx, w, bias, m, y, ldy, aux_in, aux_out, ldaux, st);
    case kEpiRelu: return launch_linear<K, N, kEpiRelu>(x, ldx, w, bias, m, y, ldy, aux_in, aux_out, ldaux, st);
    case kEpiMulGeluGrad: return launch_linear<K, N, kEpiMulGeluGrad>(x, ldx, w, bias, m, y, ldy, aux_in, aux_out, ldaux, st);
    case kEpiMulReluGrad: return launch_linear<K, N, kEpiMulReluGrad>(x, ldx, w, bias, m, y, ldy, aux_in, aux_out, ldaux, st);
    case kEpiAdd: return launch_linear<K, N, kEpiAdd>(x, ldx, w, bias, m, y, ldy, aux_in, aux_out, ldaux, st);
  }
  return SST_ERR_ARG;
}

bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }


// ------------------------------------------------------------------------------------------------------------------
// weight / bias gradients of several tall products in one launch
// ------------------------------------------------------------------------------------------------------------------
// One problem: C[P][128] = A[M, P]^T B[M, 128] (P = 128 or 256) and optionally the column sums of A or of B.  The
// tokens are cut into slices; a workgroup of 8 waves owns one slice (P = 256: wave = 64 x 64 tile of C) or two
// (P = 128).  Partial tiles go to part_w[slice][P][128]; wgrad_reduce_bf16_k sums them (deterministic) and writes the
// parameter-shaped result (optionally transposed: a problem with the operands swapped).
struct wg_problem {
  const bf16_t* a;
  const bf16_t* b;
  int64_t lda, ldb, m, tokens_per_slice;
  float* part_w;
  float* part_b;  // [slices][P] (bias_side 1) / [slices][128] (bias_side 2)
  float* out_w;   // [P][128], or [128][P] when transpose_out
  float* out_b;
  int p, bias_side, transpose_out, first_block, n_blocks, n_slices;
};
constexpr int kWgMaxProblems = 8;
struct wg_args {
  wg_problem pr[kWgMaxProblems];
  int n;
  int n_riders;                 // riders of the reduction launch (csrc/common.h): blockIdx.y < n_riders
  sst_colsum_rider riders[2];
};

// k-slot (g, x) of the packed operand = token 4 g + x (x < 4) / 16 + 4 g + x - 4 (x >= 4) of the 32-token step, on both
// operands alike; column of lane c in tile (cg, sel) = 32 cg + 8 (c >> 2) + 4 sel + (c & 3).
//
// Memory side: the 32-token x (P + 128)-column step tile is copied global -> registers -> LDS once per slice by the
// slice's waves together (16-byte chunks, three steps of loads in flight per thread), double-buffered, one barrier per
// step; every wave then reads the row-major fragments of ITS 64 + 64 columns from LDS.  (Loading them per wave from
// global memory re-reads every row 2-4 times through the L1 and leaves too few distinct bytes in flight: measured
// 1.6 TB/s on the operands.)
constexpr int kWgDepth = 3;  // register sets of global loads in flight ahead of the LDS write

template <int P>
__device__ __forceinline__ void wgrad_body(const wg_problem& pr, int blk, unsigned char* smem) {
  constexpr int COLS = P + 128, CPR = COLS / 8, RSB = COLS * 2 + 16, BUFB = 32 * RSB;
  constexpr int GT = P == 256 ? 512 : 256;       // threads that share one slice
  constexpr int NCH = 32 * CPR / GT;             // 16-byte chunks per thread and step (3 / 4)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  const int grp = P == 256 ? 0 : (wave >> 2);
  const int gtid = threadIdx.x - grp * GT;
  const int slice = P == 256 ? blk : 2 * blk + grp;
  const int wq = wave & 1, wp = P == 256 ? (wave >> 1) : ((wave >> 1) & 1);
  const bool valid = slice < pr.n_slices;
  const int64_t t_beg = valid ? (int64_t)slice * pr.tokens_per_slice : 0;
  const int64_t t_end = valid ? (t_beg + pr.tokens_per_slice < pr.m ? t_beg + pr.tokens_per_slice : pr.m) : 0;
  const int nsteps = (int)((pr.tokens_per_slice + 31) >> 5);  // the same for every slice: barriers stay aligned
  unsigned char* buf = smem + grp * 2 * BUFB;
  const int p0 = 64 * wp, q0 = 64 * wq;

  u32x4 selop[2];
#pragma unroll
  for (int sl = 0; sl < 2; ++sl) {
    selop[sl] = (u32x4){0u, 0u, 0u, 0u};
    if (g == (c >> 2)) {
      const int x = 4 * sl + (c & 3);
      const unsigned one = (x & 1) ? 0x3f800000u : 0x00003f80u;
#pragma unroll
      for (int d = 0; d < 4; ++d) selop[sl][d] = (d == (x >> 1)) ? one : 0u;
    }
  }
  // this thread's chunks of a step tile
  int ch_row[NCH], ch_lds[NCH];
  const bf16_t* ch_src[NCH];
  int64_t ch_ld[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int q = gtid + i * GT;
    const int row = q / CPR, cc = q - row * CPR;
    ch_row[i] = row;
    ch_lds[i] = row * RSB + cc * 16;
    const bool is_a = cc < P / 8;
    ch_src[i] = is_a ? pr.a + cc * 8 : pr.b + (cc - P / 8) * 8;
    ch_ld[i] = is_a ? pr.lda : pr.ldb;
  }
  auto gload = [&](int step, u32x4 (&dst)[NCH]) {
    const int64_t t0 = t_beg + 32 * (int64_t)step;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int64_t tok = t0 + ch_row[i];
      if (tok < t_end)
        dst[i] = *(const u32x4*)(ch_src[i] + tok * ch_ld[i]);
      else
        dst[i] = (u32x4){0u, 0u, 0u, 0u};  // rows past the slice contribute nothing (to the sums either)
    }
  };
  auto lds_put = [&](int b, const u32x4 (&src)[NCH]) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) *(u32x4*)(buf + b * BUFB + ch_lds[i]) = src[i];
  };

  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = zero4;
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};
  const bool sum_a = pr.bias_side == 1 && wq == 0, sum_b = pr.bias_side == 2 && wp == 0;

  u32x4 rg[kWgDepth][NCH];
#pragma unroll
  for (int d = 0; d < kWgDepth; ++d) gload(d, rg[d]);
  lds_put(0, rg[0]);
  gload(kWgDepth, rg[0]);
  __syncthreads();
  const unsigned char* fa = buf + c * RSB + (p0 + 8 * g) * 2;
  const unsigned char* fb = buf + c * RSB + (P + q0 + 8 * g) * 2;
  for (int s0 = 0; s0 < nsteps; s0 += kWgDepth) {
#pragma unroll
    for (int u = 0; u < kWgDepth; ++u) {
      const int s = s0 + u;
      if (s < nsteps) {
        const int bo = (s & 1) * BUFB;
        u32x4 opa[4], opb[4];
#pragma unroll
        for (int cg = 0; cg < 2; ++cg) {
          const u32x4 a_lo = *(const u32x4*)(fa + bo + cg * 64), a_hi = *(const u32x4*)(fa + bo + 16 * RSB + cg * 64);
          const u32x4 b_lo = *(const u32x4*)(fb + bo + cg * 64), b_hi = *(const u32x4*)(fb + bo + 16 * RSB + cg * 64);
#pragma unroll
          for (int sl = 0; sl < 2; ++sl) {
            const f32x4 a0 = mma32(a_lo, selop[sl], zero4), a1 = mma32(a_hi, selop[sl], zero4);
            const f32x4 b0 = mma32(b_lo, selop[sl], zero4), b1 = mma32(b_hi, selop[sl], zero4);
            opa[2 * cg + sl] = (u32x4){pack2(a0[0], a0[1]), pack2(a0[2], a0[3]), pack2(a1[0], a1[1]), pack2(a1[2], a1[3])};
            opb[2 * cg + sl] = (u32x4){pack2(b0[0], b0[1]), pack2(b0[2], b0[3]), pack2(b1[0], b1[1]), pack2(b1[2], b1[3])};
            if (sum_a) bsum[2 * cg + sl] += (a0[0] + a0[1]) + (a0[2] + a0[3]) + (a1[0] + a1[1]) + (a1[2] + a1[3]);
            if (sum_b) bsum[2 * cg + sl] += (b0[0] + b0[1]) + (b0[2] + b0[3]) + (b1[0] + b1[1]) + (b1[2] + b1[3]);
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = mma32(opa[i], opb[j], acc[i][j]);
        // hand the next step's tile to LDS (its loads were issued kWgDepth steps ago) and refill the register set
        if (s + 1 < nsteps) {
          lds_put((s + 1) & 1, rg[(u + 1) % kWgDepth]);
          gload(s + 1 + kWgDepth, rg[(u + 1) % kWgDepth]);
        }
        __syncthreads();
      }
    }
  }
  if (!valid) return;
  // partial tile: acc[i][j][r] = C[p0 + 32 (i >> 1) + 8 g + 4 (i & 1) + r][q0 + 32 (j >> 1) + 8 (c >> 2) + 4 (j & 1) + (c & 3)]
  float* pw = pr.part_w + (size_t)slice * P * 128;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int q = q0 + 32 * (j >> 1) + 8 * (c >> 2) + 4 * (j & 1) + (c & 3);
#pragma unroll
      for (int r = 0; r < 4; ++r) pw[(size_t)(p0 + 32 * (i >> 1) + 8 * g + 4 * (i & 1) + r) * 128 + q] = acc[i][j][r];
    }
  if (sum_a || sum_b) {
    const int width = sum_a ? P : 128, base = sum_a ? p0 : q0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v = bsum[i];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (g == 0) pr.part_b[(size_t)slice * width + base + 32 * (i >> 1) + 8 * (c >> 2) + 4 * (i & 1) + (c & 3)] = v;
    }
  }
}

constexpr int kWgLdsBytes = 2 * 2 * 32 * ((128 + 128) * 2 + 16);  // P = 128: two groups x two buffers (the larger case)
static_assert(kWgLdsBytes >= 2 * 32 * ((256 + 128) * 2 + 16), "LDS size covers P = 256");

__global__ __launch_bounds__(512, 2) void wgrad_group_bf16_k(const wg_args args) {
  extern __shared__ __attribute__((aligned(16))) unsigned char wg_smem[];
  int pi = 0;
#pragma unroll 1
  for (int i = 1; i < args.n; ++i)
    if ((int)blockIdx.x >= args.pr[i].first_block) pi = i;
  const wg_problem& pr = args.pr[pi];
  const int blk = blockIdx.x - pr.first_block;
  if (pr.p == 256)
    wgrad_body<256>(pr, blk, wg_smem);
  else
    wgrad_body<128>(pr, blk, wg_smem);
}

// 64 output elements per workgroup; the four waves take every fourth slice, combined in a fixed order (deterministic)
__global__ __launch_bounds__(256) void wgrad_reduce_bf16_k(const wg_args args) {
  __shared__ float red[4][64];
  if ((int)blockIdx.y < args.n_riders) {
    // a rider (the FIRST rows of the grid: they start with the launch): column sums of another kernel's block partials - the
    // LayerNorm backward's d(gamma) | d(beta) partials - in the arithmetic of colsum_partials_k (csrc/dense.hip: 32 strided
    // partial sums per column, added in order).  Workgroup = 32 columns; thread (cx, gq) forms the sums gy = gq + 8 j.
    // The partials are cold by now: all 64 loads are requested before the first is added.
    __shared__ float fr[32 * 33];
    const sst_colsum_rider& J = args.riders[blockIdx.y];
    if ((int)blockIdx.x * 32 >= J.width) return;
    const int cx = threadIdx.x & 31, gq = threadIdx.x >> 5;
    const int i = (int)blockIdx.x * 32 + cx;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (i < J.width) {
      const float* src = J.partials + i;
      for (int base = 0; base < J.nb; base += 512) {   // 512 partial rows at a time (any nb: csrc/layer_tail_bf16.hip has one per 128 tokens)
        float v[16][4];
#pragma unroll
        for (int it = 0; it < 16; ++it)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int b = base + 32 * it + gq + 8 * j;
            v[it][j] = src[(int64_t)(b < J.nb ? b : J.nb - 1) * J.width];
          }
#pragma unroll
        for (int it = 0; it < 16; ++it)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (base + 32 * it + gq + 8 * j < J.nb) acc[j] += v[it][j];
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) fr[(gq + 8 * j) * 33 + cx] = acc[j];
    __syncthreads();
    if (gq == 0 && i < J.width) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 32; ++k) t += fr[k * 33 + cx];
      if (i < J.split)
        J.out0[i] = t;
      else
        J.out1[i - J.split] = t;
    }
    return;
  }
  const wg_problem& pr = args.pr[blockIdx.y - args.n_riders];
  const int total_w = pr.p * 128;
  const int total_b = pr.bias_side == 1 ? pr.p : (pr.bias_side == 2 ? 128 : 0);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + lane;
  if ((int)blockIdx.x * 64 >= total_w + total_b) return;
  float s0 = 0.f, s1 = 0.f;
  if (e < total_w + total_b) {
    const bool is_w = e < total_w;
    const float* src = is_w ? pr.part_w + e : pr.part_b + (e - total_w);
    const size_t stride = is_w ? total_w : total_b;
    int sl = wave;
    for (; sl + 4 < pr.n_slices; sl += 8) {
      s0 += src[(size_t)sl * stride];
      s1 += src[(size_t)(sl + 4) * stride];
    }
    if (sl < pr.n_slices) s0 += src[(size_t)sl * stride];
  }
  red[wave][lane] = s0 + s1;
  __syncthreads();
  if (wave == 0 && e < total_w + total_b) {
    const float s = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    if (e < total_w) {
      const int p = e >> 7, q = e & 127;
      pr.out_w[pr.transpose_out ? (size_t)q * pr.p + p : (size_t)e] = s;
    } else {
      pr.out_b[e - total_w] = s;
    }
  }
}

// bf16 copies ("shadows") of fp32 parameters for the reduced-precision mode, ALL of an encoder stack in one launch:
// problem i = a rows x cols block of an fp32 matrix (row stride ld_src) -> bf16 [rows][cols], or its transpose
// [cols][rows].  The reference keeps fp16 copies of the fp32 master weights and re-makes them after every optimizer step
// (mmcv Fp16OptimizerHook.copy_params_to_fp16; configs/sst_refactor/sst_waymoD5_1x_3class_8heads_v2.py:82); here they are
// re-made at every forward call, so no cache can go stale whatever wrote the parameters (`.data` writes of an EMA hook do not
// move the version counter a cache could watch).  32 x 32 tiles through LDS: coalesced on both sides in both orientations.
constexpr int kCastMaxProblems = 96;
struct cast_args {
  const float* src[kCastMaxProblems];
  bf16_t* dst[kCastMaxProblems];
  int ld_src[kCastMaxProblems];
  short rows[kCastMaxProblems], cols[kCastMaxProblems];
  unsigned short first_tile[kCastMaxProblems + 1];
  unsigned char transpose[kCastMaxProblems];
  int n;
};
__global__ __launch_bounds__(256) void cast_group_bf16_k(const cast_args a) {
  __shared__ float tile[32][33];
  int i = 0;
  while (i + 1 < a.n && (int)blockIdx.x >= a.first_tile[i + 1]) ++i;  // uniform
  const int t = blockIdx.x - a.first_tile[i];
  const int rows = a.rows[i], cols = a.cols[i], tc = (cols + 31) / 32;
  const int r0 = (t / tc) * 32, c0 = (t % tc) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const float* src = a.src[i];
  for (int y = ty; y < 32; y += 8)
    if (r0 + y < rows && c0 + tx < cols) tile[y][tx] = src[(int64_t)(r0 + y) * a.ld_src[i] + c0 + tx];
  __syncthreads();
  bf16_t* dst = a.dst[i];
  if (!a.transpose[i]) {
    for (int y = ty; y < 32; y += 8)
      if (r0 + y < rows && c0 + tx < cols) dst[(int64_t)(r0 + y) * cols + c0 + tx] = f2bf(tile[y][tx]);
  } else {
    for (int y = ty; y < 32; y += 8)
      if (c0 + y < cols && r0 + tx < rows) dst[(int64_t)(c0 + y) * rows + r0 + tx] = f2bf(tile[tx][y]);
  }
}


}  // namespace

extern "C" {

int sst_tall_linear_bf16(const void* d_x, int64_t ldx, const void* d_w, const float* d_bias, int64_t m, int k, int n,
                         int epilogue, const void* d_aux_in, void* d_aux_out, int64_t ldaux, void* d_y, int64_t ldy,
                         void* stream) {
  if (m < 0 || !d_w || epilogue < 0 || epilogue > kEpiAdd) return SST_ERR_ARG;
  if (m == 0) return SST_OK;
  if (!d_x || !d_y || (ldx & 7) || (ldy & 7) || !aligned16(d_x) || !aligned16(d_y) || !aligned16(d_w)) return SST_ERR_ARG;
  if (epilogue >= kEpiMulGeluGrad && (!d_aux_in || (ldaux & 7) || !aligned16(d_aux_in)))
    return SST_ERR_ARG;
  if (d_aux_out && ((ldaux & 7) || !aligned16(d_aux_out))) return SST_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const bf16_t *x = (const bf16_t*)d_x, *w = (const bf16_t*)d_w, *ai = (const bf16_t*)d_aux_in;
  bf16_t *y = (bf16_t*)d_y, *ao = (bf16_t*)d_aux_out;
  int rc;
  if (k == 128 && n == 128)
    rc = dispatch_epi<128, 128>(epilogue, x, ldx, w, d_bias, m, y, ldy, ai, ao, ldaux, st);
  else if (k == 128 && n == 256)
    rc = dispatch_epi<128, 256>(epilogue, x, ldx, w, d_bias, m, y, ldy, ai, ao, ldaux, st);
  else if (k == 256 && n == 128)
    rc = dispatch_epi<256, 128>(epilogue, x, ldx, w, d_bias, m, y, ldy, ai, ao, ldaux, st);
  else
    return SST_ERR_UNSUPPORTED;
  if (rc) return rc;
  SST_LAUNCH_CHECK();
  return SST_OK;
}


int sst_tall_linear_ln_bf16(const void* d_x, int64_t ldx, const void* d_w, const float* d_bias, int64_t m, int k,
                            const void* d_res, int64_t ldres, const float* d_ln_weight, const float* d_ln_bias, float eps,
                            void* d_y, void* d_sum, float* d_stats, const float* d_pos_table, const int32_t* d_pos_idx,
                            void* d_y_plus_pos, void* stream) {
  if (m < 0 || !d_w || !d_ln_weight || !d_ln_bias || !d_stats) return SST_ERR_ARG;
  if (m == 0) return SST_OK;
  if (!d_x || !d_y || !d_res || (ldx & 7) || (ldres & 7) || !aligned16(d_x) || !aligned16(d_y) || !aligned16(d_w) ||
      !aligned16(d_res) || (d_sum && !aligned16(d_sum)))
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
  ln.yp = (bf16_t*)d_y_plus_pos;
  hipStream_t st = (hipStream_t)stream;
  int rc;
  // the residual and the stored sum share the row stride ldres (both are [m, 128] activations of the layer)
  if (k == 128)
    rc = launch_linear<128, 128, kEpiAddLN>((const bf16_t*)d_x, ldx, (const bf16_t*)d_w, d_bias, m, (bf16_t*)d_y, 128,
                                            (const bf16_t*)d_res, (bf16_t*)d_sum, ldres, st, ln);
  else if (k == 256)
    rc = launch_linear<256, 128, kEpiAddLN>((const bf16_t*)d_x, ldx, (const bf16_t*)d_w, d_bias, m, (bf16_t*)d_y, 128,
                                            (const bf16_t*)d_res, (bf16_t*)d_sum, ldres, st, ln);
  else
    return SST_ERR_UNSUPPORTED;
  if (rc) return rc;
  SST_LAUNCH_CHECK();
  return SST_OK;
}

int sst_cast_group_bf16(const sst_cast_problem_bf16* problems, int n, void* stream) {
  if (n < 0 || (n > 0 && !problems)) return SST_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  for (int base = 0; base < n; base += kCastMaxProblems) {
    cast_args a;
    const int cnt = n - base < kCastMaxProblems ? n - base : kCastMaxProblems;
    int tiles = 0;
    for (int i = 0; i < cnt; ++i) {
      const sst_cast_problem_bf16& q = problems[base + i];
      if (!q.src || !q.dst || q.rows < 1 || q.cols < 1 || q.rows > 32767 || q.cols > 32767 || q.ld_src < q.cols ||
          q.ld_src > 0x7fffffff)
        return SST_ERR_ARG;
      a.src[i] = q.src;
      a.dst[i] = (bf16_t*)q.dst;
      a.ld_src[i] = (int)q.ld_src;
      a.rows[i] = (short)q.rows;
      a.cols[i] = (short)q.cols;
      a.transpose[i] = q.transpose ? 1 : 0;
      a.first_tile[i] = (unsigned short)tiles;
      tiles += (int)(sst_div_up(q.rows, 32) * sst_div_up(q.cols, 32));
      if (tiles > 65535) return SST_ERR_UNSUPPORTED;
    }
    a.first_tile[cnt] = (unsigned short)tiles;
    a.n = cnt;
    hipLaunchKernelGGL(cast_group_bf16_k, dim3((unsigned)tiles), dim3(256), 0, st, a);
  }
  SST_LAUNCH_CHECK();
  return SST_OK;
}

// problems: n x sst_wgrad_problem_bf16 (host array, include/sst_amd.h).  workspace: sst_wgrad_group_workspace_bytes.
int64_t sst_wgrad_group_workspace_bytes(const sst_wgrad_problem_bf16* problems, int n) {
  // upper bound that does not depend on the slicing: 160 slices per problem (the planner below never exceeds it)
  int64_t total = 0;
  for (int i = 0; i < n; ++i) total += sst_align_up((int64_t)160 * (problems[i].p * 128 + 256) * 4, 256);
  return total;
}

int sst_wgrad_group_bf16(const sst_wgrad_problem_bf16* problems, int n, void* d_workspace, void* stream) {
  return sst_internal_wgrad_group_bf16(problems, n, d_workspace, nullptr, 0, stream);
}

}  // extern "C"

int sst_internal_wgrad_group_bf16(const sst_wgrad_problem_bf16* problems, int n, void* d_workspace, const sst_colsum_rider* riders,
                                  int n_riders, void* stream) {
  if (n < 1 || n > kWgMaxProblems || !problems || !d_workspace) return SST_ERR_ARG;
  if (n_riders < 0 || n_riders > 2 || (n_riders > 0 && !riders)) return SST_ERR_ARG;
  wg_args args;
  args.n = n;
  args.n_riders = n_riders;
  for (int i = 0; i < n_riders; ++i) {
    if (!riders[i].partials || riders[i].nb < 1 || riders[i].width < 1 || riders[i].width > 516 * 32 ||
        !riders[i].out0 || !riders[i].out1)
      return SST_ERR_ARG;
    args.riders[i] = riders[i];
  }
  // cost of a problem in wave-steps: tokens x (P / 256); 256 workgroups in total, shared in proportion
  double cost = 0;
  for (int i = 0; i < n; ++i) {
    const sst_wgrad_problem_bf16& q = problems[i];
    if ((q.p != 128 && q.p != 256) || q.m < 1 || !q.a || !q.b || !q.out_w || (q.lda & 7) || (q.ldb & 7) ||
        !aligned16(q.a) || !aligned16(q.b) || q.bias_side < 0 || q.bias_side > 2 || (q.bias_side && !q.out_b))
      return SST_ERR_ARG;
    cost += (double)q.m * q.p / 256.0;
  }
  sst_carver carve(d_workspace);
  int next_block = 0;
  for (int i = 0; i < n; ++i) {
    const sst_wgrad_problem_bf16& q = problems[i];
    wg_problem& w = args.pr[i];
    int blocks = (int)(256.0 * ((double)q.m * q.p / 256.0) / cost + 0.5);
    if (blocks < 1) blocks = 1;
    int slices = q.p == 256 ? blocks : 2 * blocks;
    if (slices > 160) slices = 160;
    int64_t tps = sst_align_up(sst_div_up(q.m, slices), 32);
    slices = (int)sst_div_up(q.m, tps);
    blocks = q.p == 256 ? slices : (slices + 1) / 2;
    w.a = (const bf16_t*)q.a;
    w.b = (const bf16_t*)q.b;
    w.lda = q.lda;
    w.ldb = q.ldb;
    w.m = q.m;
    w.tokens_per_slice = tps;
    w.part_w = carve.take<float>((int64_t)slices * q.p * 128);
    w.part_b = carve.take<float>((int64_t)slices * 256);
    w.out_w = q.out_w;
    w.out_b = q.out_b;
    w.p = q.p;
    w.bias_side = q.bias_side;
    w.transpose_out = q.transpose_out;
    w.first_block = next_block;
    w.n_blocks = blocks;
    w.n_slices = slices;
    next_block += blocks;
  }
  hipStream_t st = (hipStream_t)stream;
  static unsigned long long configured = 0;
  if (sst_first_use_on_device(&configured)) {
    SST_HIP(hipFuncSetAttribute((const void*)wgrad_group_bf16_k, hipFuncAttributeMaxDynamicSharedMemorySize, kWgLdsBytes));
    sst_mark_device(&configured);
  }
  hipLaunchKernelGGL(wgrad_group_bf16_k, dim3((unsigned)next_block), dim3(512), kWgLdsBytes, st, args);
  hipLaunchKernelGGL(wgrad_reduce_bf16_k, dim3((256 * 128 + 256) / 64, (unsigned)(n + n_riders)), dim3(256), 0, st, args);
  SST_LAUNCH_CHECK();
  return SST_OK;
}
