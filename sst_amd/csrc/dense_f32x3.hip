// Tall fp32 linear layers on the bf16 matrix pipe with SPLIT operands ("f32x3"):
//   Y[M, N] = epilogue(X[M, K] W^T + b),  x = x_hi + x_lo, w = w_hi + w_lo (bf16 each),
//   x w ~= x_hi w_hi + x_lo w_hi + x_hi w_lo        (fp32 accumulation; the dropped x_lo w_lo term is 2^-16 of the product)
// for the projections and the FFN of an SRA encoder layer, forward and data gradient (sst_basic_block_v2.py:41-75, 104-126),
// (K, N) in {(128,128), (128,256), (256,128)}, fp32 operands in HBM, the same epilogues as csrc/dense_f32.hip.
//
// Why: the exact-fp32 kernels of dense_f32.hip run at the fp32 matrix rate (v_mfma_f32_16x16x4_f32, 157 TFLOP/s peak,
// ~136 sustained) and are 73 % of the fp32 step.  The reference itself did NOT compute these products in exact fp32 on its
// own hardware: torch 1.8 (docs/overall_instructions.md:30-38) leaves torch.backends.cuda.matmul.allow_tf32 at its default
// True, so nn.Linear / nn.MultiheadAttention ran on TF32 tensor cores (10-bit mantissas, ~5e-4 relative per product) on the
// A100s the numbers of docs/ were produced on.  Three bf16 products recover ~16 mantissa bits (measured against float64 in
// tests/test_gpu_dense_f32x3.py: ~1e-5 relative, two orders inside the 1e-3 parity bar and tighter than TF32) at 3/16 of the
// fp32 pipe time: the kernel becomes HBM-bound on its fp32 operands ((K + N) x 4 B / token).  Reported BESIDE the exact-fp32
// headline (`bench.py`: precision_f32x3), never instead of it.
//
// Structure = tall_linear_lds8_f32_k: W resident in LDS for the whole kernel - here as TWO bf16 images (hi, lo; split once
// per launch while it is copied in, rows permuted so that the transposed product leaves a lane with 8 consecutive output
// columns of one row) -, 8 waves per workgroup, 16-row steps, X fragments straight from global memory (fp32, 2 x 16 bytes per
// lane and k-step, split into hi / lo packs in registers right before use), next tile prefetched, the finished accumulators
// of a phase written out during the next one.
#include <math.h>
#include <stdlib.h>
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack2(float lo, float hi) {  // one v_cvt_pk_bf16_f32 (RNE); compiler-visible
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float lo_f(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float hi_f(unsigned p) { return __uint_as_float(p & 0xffff0000u); }
__device__ __forceinline__ f32x4 mma32(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// 8 floats -> (hi pack, lo pack): hi = bf16(x), lo = bf16(x - hi)
__device__ __forceinline__ void split8(const f32x4& a, const f32x4& b, u32x4& hi, u32x4& lo) {
  hi[0] = pack2(a[0], a[1]);
  hi[1] = pack2(a[2], a[3]);
  hi[2] = pack2(b[0], b[1]);
  hi[3] = pack2(b[2], b[3]);
  lo[0] = pack2(a[0] - lo_f(hi[0]), a[1] - hi_f(hi[0]));
  lo[1] = pack2(a[2] - lo_f(hi[1]), a[3] - hi_f(hi[1]));
  lo[2] = pack2(b[0] - lo_f(hi[2]), b[1] - hi_f(hi[2]));
  lo[3] = pack2(b[2] - lo_f(hi[3]), b[3] - hi_f(hi[3]));
}

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

struct ln_epi {   // see csrc/dense_f32.hip
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

template <int K, int N, int EPI>
__global__ __launch_bounds__(512, 2) void tall_linear_f32x3_k(
    const float* __restrict__ X, int64_t ldx, const float* __restrict__ W, int64_t ldw, int trans_w,
    const float* __restrict__ bias, int64_t M, int rows_per_wave, float* __restrict__ Y, int64_t ldy,
    const float* __restrict__ aux_in, float* __restrict__ aux_out, int64_t ldaux, const ln_epi ln) {
  static_assert(EPI != kEpiAddLN || N == 128, "the LayerNorm epilogue needs a whole row in one accumulator set");
  constexpr int RS = K * 2 + 16;  // LDS row stride in bytes of one bf16 image
  constexpr int KS = K / 32, NTH = 512;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];
  unsigned char* whi = smem3;
  unsigned char* wlo = smem3 + N * RS;
  float* bimg = (float*)(smem3 + 2 * N * RS);
  const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  const int64_t wave = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 6);
  int64_t r0 = wave * rows_per_wave;
  const int64_t r1 = r0 + rows_per_wave < M ? r0 + rows_per_wave : M;
  f32x4 xb[KS][2], xn[KS][2];
  auto load_x = [&](int64_t r, f32x4 (&dst)[KS][2]) {
    int64_t row = r + c;
    row = row < M ? row : M - 1;
    const float* p = X + row * ldx + 8 * g;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      dst[s][0] = *(const f32x4*)(p + 32 * s);
      dst[s][1] = *(const f32x4*)(p + 32 * s + 4);
    }
  };
  load_x(r0 < M ? r0 : M - 1, xb);
  // weight fill: W[n][k] (trans_w = 0: row n of W; trans_w = 1: W is [K][N], the data gradient of a layer whose parameter is
  // W) -> hi / lo bf16 images, row w_lds_row(n).  8 consecutive k per thread and step.
  constexpr int CHUNKS = N * K / 8;
  for (int idx = threadIdx.x; idx < CHUNKS; idx += NTH) {
    const int n = idx / (K / 8), k8 = (idx - n * (K / 8)) * 8;
    f32x4 a, b;
    if (!trans_w) {
      a = *(const f32x4*)(W + (size_t)n * ldw + k8);
      b = *(const f32x4*)(W + (size_t)n * ldw + k8 + 4);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        a[e] = W[(size_t)(k8 + e) * ldw + n];
        b[e] = W[(size_t)(k8 + 4 + e) * ldw + n];
      }
    }
    u32x4 hi, lo;
    split8(a, b, hi, lo);
    *(u32x4*)(whi + w_lds_row(n) * RS + k8 * 2) = hi;
    *(u32x4*)(wlo + w_lds_row(n) * RS + k8 * 2) = lo;
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
  const unsigned char* hlane = whi + c * RS + g * 16;
  const unsigned char* llane = wlo + c * RS + g * 16;

  f32x4 pend[8];
  int64_t pend_r0 = 0;
  int pend_nh = 0;
  bool pend_valid = false;
  float ln_rstd = 0.f;
  auto emit = [&](int tp) {   // identical to tall_linear_lds8_f32_k's (csrc/dense_f32.hip)
    if (!pend_valid) return;
    const int64_t row = pend_r0 + c;
    if (row >= r1) return;
    if (EPI == kEpiAddLN) {
      if (tp == 0) {
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

  constexpr bool PREFETCH = !(EPI == kEpiAddLN && K == 256);   // as in dense_f32.hip: no room for a second X tile there
  for (; r0 < r1; r0 += 16) {
    asm volatile("" ::: "memory");  // W fragments are re-read from LDS per row tile (never hoisted into registers)
    const bool more = r0 + 16 < r1;
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this tile's X has landed before the next one is requested
    if (PREFETCH && more) load_x(r0 + 16, xn);
#pragma unroll
    for (int nh = 0; nh < N / 128; ++nh) {
      f32x4 acc[8];
#pragma unroll
      for (int T = 0; T < 8; ++T) acc[T] = (f32x4){0.f, 0.f, 0.f, 0.f};
      // (k-step s, column-tile group of 4): three products per tile, issued product by product over the four tiles so that a
      // dependent MFMA is four instructions behind its predecessor; one eighth... of the previous phase's epilogue after
      // every quarter of the groups
      constexpr int G = KS * 2;
#pragma clang loop unroll(full)
      for (int q = 0; q < G; ++q) {
        const int s = q >> 1, t0 = (q & 1) * 4;
        u32x4 xh, xl;
        split8(xb[s][0], xb[s][1], xh, xl);
        u32x4 wh[4], wl[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int off = (nh * 8 + t0 + u) * 16 * RS + s * 64;
          wh[u] = *(const u32x4*)(hlane + off);
          wl[u] = *(const u32x4*)(llane + off);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[t0 + u] = mma32(wh[u], xh, acc[t0 + u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[t0 + u] = mma32(wh[u], xl, acc[t0 + u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[t0 + u] = mma32(wl[u], xh, acc[t0 + u]);
        if ((q + 1) % (G / 4) == 0) emit((q + 1) / (G / 4) - 1);
      }
#pragma unroll
      for (int T = 0; T < 8; ++T) pend[T] = acc[T];
      pend_r0 = r0;
      pend_nh = nh;
      pend_valid = true;
    }
    if (more) {
      if (PREFETCH) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          xb[s][0] = xn[s][0];
          xb[s][1] = xn[s][1];
        }
      } else {
        load_x(r0 + 16, xb);
      }
    }
  }
#pragma unroll
  for (int tp = 0; tp < 4; ++tp) emit(tp);
}

template <int K, int N, int EPI>
int launch_x3(const float* x, int64_t ldx, const float* w, int64_t ldw, int trans_w, const float* bias, int64_t m, float* y,
              int64_t ldy, const float* aux_in, float* aux_out, int64_t ldaux, hipStream_t st, const ln_epi ln = ln_epi()) {
  constexpr int lds = 2 * N * (K * 2 + 16) + N * (EPI == kEpiAddLN ? 3 : 1) * 4;
  static unsigned long long configured = 0;
  if (sst_first_use_on_device(&configured)) {
    SST_HIP(hipFuncSetAttribute((const void*)tall_linear_f32x3_k<K, N, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    sst_mark_device(&configured);
  }
  int64_t blocks = 256;   // one 8-wave workgroup per CU
  int64_t rpw = sst_align_up(sst_div_up(m, blocks * 8), 8);
  blocks = sst_div_up(m, rpw * 8);
  hipLaunchKernelGGL((tall_linear_f32x3_k<K, N, EPI>), dim3((unsigned)blocks), dim3(512), lds, st, x, ldx, w, ldw, trans_w, bias,
                     m, (int)rpw, y, ldy, aux_in, aux_out, ldaux, ln);
  return SST_OK;
}

template <int K, int N>
int dispatch_x3(int epi, const float* x, int64_t ldx, const float* w, int64_t ldw, int trans_w, const float* bias, int64_t m,
                float* y, int64_t ldy, const float* aux_in, float* aux_out, int64_t ldaux, hipStream_t st) {
#define SST_CASE(E) \
  case E: return launch_x3<K, N, E>(x, ldx, w, ldw, trans_w, bias, m, y, ldy, aux_in, aux_out, ldaux, st)
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

int sst_tall_linear_epi_f32x3(const float* d_x, int64_t ldx, const float* d_w, int64_t ldw, int trans_w, const float* d_bias,
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
    rc = dispatch_x3<128, 128>(epilogue, d_x, ldx, d_w, ldw, trans_w, d_bias, m, d_y, ldy, d_aux_in, d_aux_out, ldaux, st);
  else if (k == 128 && n == 256)
    rc = dispatch_x3<128, 256>(epilogue, d_x, ldx, d_w, ldw, trans_w, d_bias, m, d_y, ldy, d_aux_in, d_aux_out, ldaux, st);
  else if (k == 256 && n == 128)
    rc = dispatch_x3<256, 128>(epilogue, d_x, ldx, d_w, ldw, trans_w, d_bias, m, d_y, ldy, d_aux_in, d_aux_out, ldaux, st);
  else
    return SST_ERR_UNSUPPORTED;
  if (rc) return rc;
  SST_LAUNCH_CHECK();
  return SST_OK;
}

int sst_tall_linear_ln_f32x3(const float* d_x, int64_t ldx, const float* d_w, int64_t ldw, const float* d_bias, int64_t m, int k,
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
    rc = launch_x3<128, 128, kEpiAddLN>(d_x, ldx, d_w, ldw, 0, d_bias, m, d_y, 128, d_res, d_sum, ldres, st, ln);
  else if (k == 256)
    rc = launch_x3<256, 128, kEpiAddLN>(d_x, ldx, d_w, ldw, 0, d_bias, m, d_y, 128, d_res, d_sum, ldres, st, ln);
  else
    return SST_ERR_UNSUPPORTED;
  if (rc) return rc;
  SST_LAUNCH_CHECK();
  return SST_OK;
}

}  // extern "C"
