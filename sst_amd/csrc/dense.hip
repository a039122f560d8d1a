// Row-wise fp32 kernels either side of the SRA core inside an encoder layer:
//   fused residual-add + LayerNorm forward / backward   (sst_basic_block_v2.py:113-118: src = norm(src + src2))
//   column sum of a tall matrix                          (bias gradients of the projections / FFN)
// All HBM-bound: 1 read + 1 write of 4C B/token forward (+ the residual read), row = one 32-lane group.
#include <math.h>
#include "common.h"

namespace {

constexpr int kLnThreads = 256;
constexpr int kLnRowsPerBlock = kLnThreads / 32;  // one row per 32-lane half-wave
constexpr int kLnMaxVec = 4;                      // up to 4 float4 per lane -> C <= 512

// activation folded into the LayerNorm passes ("Linear -> LN -> GELU" of FSD's SIR layers, voxel_encoder.py:628-650):
// 0 none, 1 GELU (erf form; Abramowitz-Stegun 7.1.26, |error| < 1.5e-7, as in csrc/dense_f32.hip), 2 ReLU
__device__ __forceinline__ float ln_erf(float z, float& e) {
  const float az = fabsf(z);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, az, 1.f));
  e = __expf(-az * az);
  const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
  return copysignf(fmaf(-poly, e, 1.f), z);
}
__device__ __forceinline__ float ln_act(float x, int act) {
  if (act == 1) {
    float e;
    return 0.5f * x * (1.f + ln_erf(x * 0.70710678118654752f, e));
  }
  return act == 2 ? fmaxf(x, 0.f) : x;
}
__device__ __forceinline__ float ln_act_grad(float x, int act) {
  if (act == 1) {
    float e;
    const float phi = 0.5f * (1.f + ln_erf(x * 0.70710678118654752f, e));
    return fmaf(x * 0.3989422804014327f, e, phi);
  }
  return act == 2 ? (x > 0.f ? 1.f : 0.f) : 1.f;
}

__device__ __forceinline__ float group32_sum(float v) {
#pragma unroll
  for (int d = 1; d < 32; d <<= 1) v += __shfl_xor(v, d, 64);
  return v;
}

// y = LN(x + r) * w + b ; stats[row] = (mean, rstd).  r may be null.
__global__ __launch_bounds__(kLnThreads) void add_ln_fwd_k(const float* __restrict__ x, const float* __restrict__ r,
                                                           const float* __restrict__ w, const float* __restrict__ b,
                                                           int64_t m, int c, float eps, int act,
                                                           float* __restrict__ y, float* __restrict__ sum_out,
                                                           float2* __restrict__ stats,
                                                           const float* __restrict__ pos_table = nullptr,
                                                           const int32_t* __restrict__ pos_idx = nullptr,
                                                           float* __restrict__ y_plus_pos = nullptr) {
  const int lane = threadIdx.x & 31;
  const int sub = threadIdx.x >> 5;
  const int nvec = (c + 127) / 128;
  for (int64_t row = (int64_t)blockIdx.x * kLnRowsPerBlock + sub; row < m; row += (int64_t)gridDim.x * kLnRowsPerBlock) {
    float4 v[kLnMaxVec];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < kLnMaxVec; ++k) {
      const int col = k * 128 + lane * 4;
      v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k < nvec && col < c) {
        v[k] = *(const float4*)(x + row * c + col);
        if (r != nullptr) {
          const float4 rv = *(const float4*)(r + row * c + col);
          v[k].x += rv.x;
          v[k].y += rv.y;
          v[k].z += rv.z;
          v[k].w += rv.w;
        }
        if (sum_out != nullptr) *(float4*)(sum_out + row * c + col) = v[k];
        s += v[k].x + v[k].y + v[k].z + v[k].w;
      }
    }
    const float mean = group32_sum(s) / (float)c;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < kLnMaxVec; ++k) {
      const int col = k * 128 + lane * 4;
      if (k < nvec && col < c) {
        const float dx = v[k].x - mean, dy = v[k].y - mean, dz = v[k].z - mean, dw = v[k].w - mean;
        q += dx * dx + dy * dy + dz * dz + dw * dw;
      }
    }
    const float var = group32_sum(q) / (float)c;
    const float rstd = rsqrtf(var + eps);
#pragma unroll
    for (int k = 0; k < kLnMaxVec; ++k) {
      const int col = k * 128 + lane * 4;
      if (k < nvec && col < c) {
        const float4 wv = *(const float4*)(w + col);
        const float4 bv = *(const float4*)(b + col);
        float4 o;
        o.x = (v[k].x - mean) * rstd * wv.x + bv.x;
        o.y = (v[k].y - mean) * rstd * wv.y + bv.y;
        o.z = (v[k].z - mean) * rstd * wv.z + bv.z;
        o.w = (v[k].w - mean) * rstd * wv.w + bv.w;
        if (act) o.x = ln_act(o.x, act), o.y = ln_act(o.y, act), o.z = ln_act(o.z, act), o.w = ln_act(o.w, act);
        *(float4*)(y + row * c + col) = o;
        if (y_plus_pos != nullptr) {   // the next encoder layer's q / k input: y + positional embedding (row of a small table)
          const float4 pv = *(const float4*)(pos_table + (size_t)pos_idx[row] * c + col);
          *(float4*)(y_plus_pos + row * c + col) = make_float4(o.x + pv.x, o.y + pv.y, o.z + pv.z, o.w + pv.w);
        }
      }
    }
    if (lane == 0) stats[row] = make_float2(mean, rstd);
  }
}

// Given s = x + r (saved), stats, dy:  dx = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * w
// dw += sum_rows dy * xhat ; db += sum_rows dy   (block partials in LDS, then one atomic per column per block)
__global__ __launch_bounds__(kLnThreads) void add_ln_bwd_k(const float* __restrict__ dy, const float* __restrict__ s,
                                                           const float2* __restrict__ stats,
                                                           const float* __restrict__ w,
                                                           const float* __restrict__ b, int act, int64_t m, int c,
                                                           float* __restrict__ dx,
                                                           float* __restrict__ partials) {
  extern __shared__ __attribute__((aligned(16))) float part[];  // [2][c]
  for (int i = threadIdx.x; i < 2 * c; i += kLnThreads) part[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 31;
  const int sub = threadIdx.x >> 5;
  const int nvec = (c + 127) / 128;
  float4 aw[kLnMaxVec], ab[kLnMaxVec];
#pragma unroll
  for (int k = 0; k < kLnMaxVec; ++k) aw[k] = ab[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t row = (int64_t)blockIdx.x * kLnRowsPerBlock + sub; row < m; row += (int64_t)gridDim.x * kLnRowsPerBlock) {
    const float2 st = stats[row];
    float4 g[kLnMaxVec], xh[kLnMaxVec];
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int k = 0; k < kLnMaxVec; ++k) {
      const int col = k * 128 + lane * 4;
      g[k] = xh[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k < nvec && col < c) {
        float4 d = *(const float4*)(dy + row * c + col);
        const float4 sv = *(const float4*)(s + row * c + col);
        const float4 wv = *(const float4*)(w + col);
        xh[k] = make_float4((sv.x - st.x) * st.y, (sv.y - st.x) * st.y, (sv.z - st.x) * st.y, (sv.w - st.x) * st.y);
        if (act) {   // the gradient arrives behind the activation: through it first, at the recomputed LayerNorm output
          const float4 bv = *(const float4*)(b + col);
          d.x *= ln_act_grad(fmaf(xh[k].x, wv.x, bv.x), act);
          d.y *= ln_act_grad(fmaf(xh[k].y, wv.y, bv.y), act);
          d.z *= ln_act_grad(fmaf(xh[k].z, wv.z, bv.z), act);
          d.w *= ln_act_grad(fmaf(xh[k].w, wv.w, bv.w), act);
        }
        g[k] = make_float4(d.x * wv.x, d.y * wv.y, d.z * wv.z, d.w * wv.w);
        sg += g[k].x + g[k].y + g[k].z + g[k].w;
        sgx += g[k].x * xh[k].x + g[k].y * xh[k].y + g[k].z * xh[k].z + g[k].w * xh[k].w;
        aw[k].x += d.x * xh[k].x;
        aw[k].y += d.y * xh[k].y;
        aw[k].z += d.z * xh[k].z;
        aw[k].w += d.w * xh[k].w;
        ab[k].x += d.x;
        ab[k].y += d.y;
        ab[k].z += d.z;
        ab[k].w += d.w;
      }
    }
    const float mg = group32_sum(sg) / (float)c;
    const float mgx = group32_sum(sgx) / (float)c;
#pragma unroll
    for (int k = 0; k < kLnMaxVec; ++k) {
      const int col = k * 128 + lane * 4;
      if (k < nvec && col < c) {
        float4 o;
        o.x = st.y * (g[k].x - mg - xh[k].x * mgx);
        o.y = st.y * (g[k].y - mg - xh[k].y * mgx);
        o.z = st.y * (g[k].z - mg - xh[k].z * mgx);
        o.w = st.y * (g[k].w - mg - xh[k].w * mgx);
        *(float4*)(dx + row * c + col) = o;
      }
    }
  }
  // the row groups of the block add their column sums one after the other (a float atomicAdd into LDS made the order - and
  // the last bits of d(gamma), d(beta) - depend on the schedule)
  for (int turn = 0; turn < kLnRowsPerBlock; ++turn) {
    if (sub == turn) {
#pragma unroll
      for (int k = 0; k < kLnMaxVec; ++k) {
        const int col = k * 128 + lane * 4;
        if (k < nvec && col < c) {
          part[col + 0] += aw[k].x, part[col + 1] += aw[k].y, part[col + 2] += aw[k].z, part[col + 3] += aw[k].w;
          part[c + col + 0] += ab[k].x, part[c + col + 1] += ab[k].y, part[c + col + 2] += ab[k].z, part[c + col + 3] += ab[k].w;
        }
      }
    }
    __syncthreads();
  }
  // block partials [gridDim.x][2c]; reduced by colsum_partials_k (no global atomics, deterministic)
  float* dst = partials + (int64_t)blockIdx.x * 2 * c;
  for (int i = threadIdx.x; i < 2 * c; i += kLnThreads) dst[i] = part[i];
}

// Any width 1 <= C <= 512 (FSD's first SIR layer of a stack and its relative-position gate have C = 5 + point features: 133,
// 148 ...; rows are then not 16-byte aligned): the same two kernels with 4-byte accesses, lane l of the row's 32-lane group
// owning columns l, l + 32, ...  The row (<= 2 KB) is read from cache in the later passes instead of being kept in registers.
constexpr int kLnMaxScalar = 16;   // 512 / 32 columns per lane

__global__ __launch_bounds__(kLnThreads) void add_ln_fwd_any_k(const float* __restrict__ x, const float* __restrict__ r,
                                                               const float* __restrict__ w, const float* __restrict__ b,
                                                               int64_t m, int c, float eps, int act,
                                                               float* __restrict__ y, float* __restrict__ sum_out,
                                                               float2* __restrict__ stats) {
  const int lane = threadIdx.x & 31, sub = threadIdx.x >> 5;
  for (int64_t row = (int64_t)blockIdx.x * kLnRowsPerBlock + sub; row < m; row += (int64_t)gridDim.x * kLnRowsPerBlock) {
    float v[kLnMaxScalar];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < kLnMaxScalar; ++k) {
      const int col = k * 32 + lane;
      v[k] = 0.f;
      if (col < c) {
        v[k] = x[row * c + col] + (r != nullptr ? r[row * c + col] : 0.f);
        if (sum_out != nullptr) sum_out[row * c + col] = v[k];
        s += v[k];
      }
    }
    const float mean = group32_sum(s) / (float)c;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < kLnMaxScalar; ++k)
      if (k * 32 + lane < c) q += (v[k] - mean) * (v[k] - mean);
    const float rstd = rsqrtf(group32_sum(q) / (float)c + eps);
#pragma unroll
    for (int k = 0; k < kLnMaxScalar; ++k) {
      const int col = k * 32 + lane;
      if (col < c) y[row * c + col] = ln_act((v[k] - mean) * rstd * w[col] + b[col], act);
    }
    if (lane == 0) stats[row] = make_float2(mean, rstd);
  }
}

__global__ __launch_bounds__(kLnThreads) void add_ln_bwd_any_k(const float* __restrict__ dy, const float* __restrict__ s,
                                                               const float2* __restrict__ stats,
                                                               const float* __restrict__ w,
                                                               const float* __restrict__ b, int act, int64_t m, int c,
                                                               float* __restrict__ dx, float* __restrict__ partials) {
  extern __shared__ __attribute__((aligned(16))) float part[];  // [2][c]
  for (int i = threadIdx.x; i < 2 * c; i += kLnThreads) part[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 31, sub = threadIdx.x >> 5;
  float aw[kLnMaxScalar], ab[kLnMaxScalar];
#pragma unroll
  for (int k = 0; k < kLnMaxScalar; ++k) aw[k] = ab[k] = 0.f;
  for (int64_t row = (int64_t)blockIdx.x * kLnRowsPerBlock + sub; row < m; row += (int64_t)gridDim.x * kLnRowsPerBlock) {
    const float2 st = stats[row];
    float g[kLnMaxScalar], xh[kLnMaxScalar];
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int k = 0; k < kLnMaxScalar; ++k) {
      const int col = k * 32 + lane;
      g[k] = xh[k] = 0.f;
      if (col < c) {
        float d = dy[row * c + col];
        xh[k] = (s[row * c + col] - st.x) * st.y;
        if (act) d *= ln_act_grad(fmaf(xh[k], w[col], b[col]), act);
        g[k] = d * w[col];
        sg += g[k];
        sgx += g[k] * xh[k];
        aw[k] += d * xh[k];
        ab[k] += d;
      }
    }
    const float mg = group32_sum(sg) / (float)c, mgx = group32_sum(sgx) / (float)c;
#pragma unroll
    for (int k = 0; k < kLnMaxScalar; ++k) {
      const int col = k * 32 + lane;
      if (col < c) dx[row * c + col] = st.y * (g[k] - mg - xh[k] * mgx);
    }
  }
  for (int turn = 0; turn < kLnRowsPerBlock; ++turn) {   // row groups in a fixed order: deterministic to the last bit
    if (sub == turn) {
#pragma unroll
      for (int k = 0; k < kLnMaxScalar; ++k) {
        const int col = k * 32 + lane;
        if (col < c) part[col] += aw[k], part[c + col] += ab[k];
      }
    }
    __syncthreads();
  }
  float* dst = partials + (int64_t)blockIdx.x * 2 * c;
  for (int i = threadIdx.x; i < 2 * c; i += kLnThreads) dst[i] = part[i];
}

// C = 128 (every SST config): one float4 per lane, FOUR rows per 32-lane group in flight per iteration (the generic
// kernel above issues the two loads of a single row, then two dependent 5-step shuffle reductions: latency-bound
// at two waves per SIMD, 4.2 TB/s).
__global__ __launch_bounds__(kLnThreads) void add_ln_bwd_c128_k(const float* __restrict__ dy,
                                                                const float* __restrict__ dy2,
                                                                const float* __restrict__ s,
                                                                const float2* __restrict__ stats,
                                                                const float* __restrict__ w, int64_t m,
                                                                float* __restrict__ dx,
                                                                float* __restrict__ partials) {
  constexpr int C = 128, R = 4;
  __shared__ float part[2 * C];
  for (int i = threadIdx.x; i < 2 * C; i += kLnThreads) part[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 31;
  const int sub = threadIdx.x >> 5;
  const int col = lane * 4;
  const float4 wv = *(const float4*)(w + col);
  float4 aw = make_float4(0.f, 0.f, 0.f, 0.f), ab = aw;
  const int64_t stride = (int64_t)gridDim.x * kLnRowsPerBlock;
  for (int64_t row0 = (int64_t)blockIdx.x * kLnRowsPerBlock + sub; row0 < m; row0 += stride * R) {
    float4 d[R], sv[R];
    float2 st[R];
    bool ok[R];
#pragma unroll
    for (int u = 0; u < R; ++u) {
      const int64_t row = row0 + u * stride;
      ok[u] = row < m;
      const int64_t rr = ok[u] ? row : row0;  // clamped: loads are unconditional
      d[u] = *(const float4*)(dy + rr * C + col);
      if (dy2 != nullptr) {  // second gradient arriving at the LayerNorm output (its "+ positional embedding" copy)
        const float4 e = *(const float4*)(dy2 + rr * C + col);
        d[u].x += e.x, d[u].y += e.y, d[u].z += e.z, d[u].w += e.w;
      }
      sv[u] = *(const float4*)(s + rr * C + col);
      st[u] = stats[rr];
    }
    float4 g[R], xh[R];
    float sg[R], sgx[R];
#pragma unroll
    for (int u = 0; u < R; ++u) {
      xh[u] = make_float4((sv[u].x - st[u].x) * st[u].y, (sv[u].y - st[u].x) * st[u].y, (sv[u].z - st[u].x) * st[u].y,
                          (sv[u].w - st[u].x) * st[u].y);
      g[u] = make_float4(d[u].x * wv.x, d[u].y * wv.y, d[u].z * wv.z, d[u].w * wv.w);
      sg[u] = g[u].x + g[u].y + g[u].z + g[u].w;
      sgx[u] = g[u].x * xh[u].x + g[u].y * xh[u].y + g[u].z * xh[u].z + g[u].w * xh[u].w;
      if (ok[u]) {
        aw.x += d[u].x * xh[u].x, aw.y += d[u].y * xh[u].y, aw.z += d[u].z * xh[u].z, aw.w += d[u].w * xh[u].w;
        ab.x += d[u].x, ab.y += d[u].y, ab.z += d[u].z, ab.w += d[u].w;
      }
    }
#pragma unroll
    for (int dlt = 1; dlt < 32; dlt <<= 1) {  // the 2 * R reductions interleaved
#pragma unroll
      for (int u = 0; u < R; ++u) {
        sg[u] += __shfl_xor(sg[u], dlt, 64);
        sgx[u] += __shfl_xor(sgx[u], dlt, 64);
      }
    }
#pragma unroll
    for (int u = 0; u < R; ++u) {
      if (ok[u]) {
        const float mg = sg[u] * (1.f / C), mgx = sgx[u] * (1.f / C);
        float4 o;
        o.x = st[u].y * (g[u].x - mg - xh[u].x * mgx);
        o.y = st[u].y * (g[u].y - mg - xh[u].y * mgx);
        o.z = st[u].y * (g[u].z - mg - xh[u].z * mgx);
        o.w = st[u].y * (g[u].w - mg - xh[u].w * mgx);
        *(float4*)(dx + (row0 + u * stride) * C + col) = o;
      }
    }
  }
  for (int turn = 0; turn < kLnRowsPerBlock; ++turn) {   // row groups in a fixed order: deterministic to the last bit
    if (sub == turn) {
      part[col + 0] += aw.x, part[col + 1] += aw.y, part[col + 2] += aw.z, part[col + 3] += aw.w;
      part[C + col + 0] += ab.x, part[C + col + 1] += ab.y, part[C + col + 2] += ab.z, part[C + col + 3] += ab.w;
    }
    __syncthreads();
  }
  float* dst = partials + (int64_t)blockIdx.x * 2 * C;
  for (int i = threadIdx.x; i < 2 * C; i += kLnThreads) dst[i] = part[i];
}

// out[i] = sum_b partials[b][i], i < width.  Block = 32 columns x 32 slices of the nb partial rows.
__global__ __launch_bounds__(1024) void colsum_partials_k(const float* __restrict__ partials, int nb, int width,
                                                          float* __restrict__ out0, float* __restrict__ out1,
                                                          int split) {
  __shared__ float red[32][33];
  const int cx = threadIdx.x & 31, gy = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + cx;
  float acc = 0.f;
  if (i < width)
    for (int b = gy; b < nb; b += 32) acc += partials[(int64_t)b * width + i];
  red[gy][cx] = acc;
  __syncthreads();
  if (gy == 0 && i < width) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) t += red[k][cx];
    if (i < split)
      out0[i] = t;
    else
      out1[i - split] = t;
  }
}

// out[col] += sum over rows of x[row, col]; out must be zero on entry.  c % 4 == 0, c <= 1024.
__global__ __launch_bounds__(256) void colsum_k(const float* __restrict__ x, int64_t m, int c, int64_t ld,
                                                int64_t rows_per_block, float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) float slots[1024];   // [rows per iteration][c]: (256 / c4) * c <= 1024
  const int c4 = c >> 2;
  const int rpi = 256 / c4 > 0 ? 256 / c4 : 1;  // rows per iteration
  const int ry = threadIdx.x / c4, cx = threadIdx.x - ry * c4;
  const int64_t beg = (int64_t)blockIdx.x * rows_per_block;
  const int64_t end = beg + rows_per_block < m ? beg + rows_per_block : m;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (ry < rpi) {
    for (int64_t row = beg + ry; row < end; row += rpi) {
      for (int cc = cx; cc < c4; cc += 256) {  // c4 <= 256: single trip
        const float4 v = *(const float4*)(x + row * ld + cc * 4);
        acc.x += v.x;
        acc.y += v.y;
        acc.z += v.z;
        acc.w += v.w;
      }
    }
    *(float4*)(slots + ry * c + cx * 4) = acc;
  }
  __syncthreads();
  float* dst = out + (int64_t)blockIdx.x * c;  // block partials, reduced by colsum_partials_k; row groups summed in order
  const int used = rpi < 256 / c4 ? rpi : 256 / c4;
  for (int i = threadIdx.x; i < c; i += 256) {
    float t = 0.f;
    for (int r = 0; r < used; ++r) t += slots[r * c + i];
    dst[i] = t;
  }
}

// ---- bf16 storage variants (the reduced-precision mode of the encoder layers; statistics and arithmetic in fp32) ------
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 bf4_load(const unsigned short* __restrict__ p) {
  const u32x2_t v = *(const u32x2_t*)p;
  return make_float4(__uint_as_float(v[0] << 16), __uint_as_float(v[0] & 0xffff0000u), __uint_as_float(v[1] << 16),
                     __uint_as_float(v[1] & 0xffff0000u));
}
__device__ __forceinline__ void bf4_store(unsigned short* __restrict__ p, float4 v) {
  typedef float f32x4_t __attribute__((ext_vector_type(4)));
  typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
  const f32x4_t f = {v.x, v.y, v.z, v.w};
  *(u32x2_t*)p = __builtin_bit_cast(u32x2_t, __builtin_convertvector(f, bf16x4_t));  // two v_cvt_pk_bf16_f32 (RNE)
}

// y = LN(x + r) * w + b (bf16 in / out, fp32 parameters and statistics); sum_out (bf16, optional) = x + r for the
// backward pass; yp (bf16, optional) = y + pos_table[pos_idx[row]]: the next layer's q / k input (x + positional
// embedding, sst_basic_block_v2.py:58-60), so that no separate add pass and no [M, C] positional tensor exist.
__global__ __launch_bounds__(kLnThreads) void add_ln_fwd_bf16_k(
    const unsigned short* __restrict__ x, const unsigned short* __restrict__ r, const float* __restrict__ w,
    const float* __restrict__ b, int64_t m, int c, float eps, unsigned short* __restrict__ y,
    unsigned short* __restrict__ sum_out, float2* __restrict__ stats, const float* __restrict__ pos_table,
    const int32_t* __restrict__ pos_idx, unsigned short* __restrict__ yp) {
  const int lane = threadIdx.x & 31;
  const int sub = threadIdx.x >> 5;
  const int nvec = (c + 127) / 128;
  for (int64_t row = (int64_t)blockIdx.x * kLnRowsPerBlock + sub; row < m; row += (int64_t)gridDim.x * kLnRowsPerBlock) {
    float4 v[kLnMaxVec];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < kLnMaxVec; ++k) {
      const int col = k * 128 + lane * 4;
      v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k < nvec && col < c) {
        v[k] = bf4_load(x + row * c + col);
        if (r != nullptr) {
          const float4 rv = bf4_load(r + row * c + col);
          v[k].x += rv.x, v[k].y += rv.y, v[k].z += rv.z, v[k].w += rv.w;
        }
        if (sum_out != nullptr) bf4_store(sum_out + row * c + col, v[k]);
        s += v[k].x + v[k].y + v[k].z + v[k].w;
      }
    }
    const float mean = group32_sum(s) / (float)c;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < kLnMaxVec; ++k) {
      const int col = k * 128 + lane * 4;
      if (k < nvec && col < c) {
        const float dx = v[k].x - mean, dy = v[k].y - mean, dz = v[k].z - mean, dw = v[k].w - mean;
        q += dx * dx + dy * dy + dz * dz + dw * dw;
      }
    }
    const float rstd = rsqrtf(group32_sum(q) / (float)c + eps);
    const float* prow = (pos_table != nullptr) ? pos_table + (int64_t)pos_idx[row] * c : nullptr;
#pragma unroll
    for (int k = 0; k < kLnMaxVec; ++k) {
      const int col = k * 128 + lane * 4;
      if (k < nvec && col < c) {
        const float4 wv = *(const float4*)(w + col);
        const float4 bv = *(const float4*)(b + col);
        float4 o;
        o.x = (v[k].x - mean) * rstd * wv.x + bv.x;
        o.y = (v[k].y - mean) * rstd * wv.y + bv.y;
        o.z = (v[k].z - mean) * rstd * wv.z + bv.z;
        o.w = (v[k].w - mean) * rstd * wv.w + bv.w;
        bf4_store(y + row * c + col, o);
        if (prow != nullptr) {
          const float4 pv = *(const float4*)(prow + col);
          bf4_store(yp + row * c + col, make_float4(o.x + pv.x, o.y + pv.y, o.z + pv.z, o.w + pv.w));
        }
      }
    }
    if (lane == 0) stats[row] = make_float2(mean, rstd);
  }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = (dy [+ dy2]) * w; block partials of dw / db as in the fp32 kernel
__global__ __launch_bounds__(kLnThreads) void add_ln_bwd_bf16_k(const unsigned short* __restrict__ dy,
                                                                const unsigned short* __restrict__ dy2,
                                                                const unsigned short* __restrict__ s,
                                                                const float2* __restrict__ stats,
                                                                const float* __restrict__ w, int64_t m, int c,
                                                                unsigned short* __restrict__ dx,
                                                                float* __restrict__ partials) {
  extern __shared__ __attribute__((aligned(16))) float part[];  // [2][c]
  for (int i = threadIdx.x; i < 2 * c; i += kLnThreads) part[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 31;
  const int sub = threadIdx.x >> 5;
  const int nvec = (c + 127) / 128;
  float4 aw[kLnMaxVec], ab[kLnMaxVec];
#pragma unroll
  for (int k = 0; k < kLnMaxVec; ++k) aw[k] = ab[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t row = (int64_t)blockIdx.x * kLnRowsPerBlock + sub; row < m; row += (int64_t)gridDim.x * kLnRowsPerBlock) {
    const float2 st = stats[row];
    float4 g[kLnMaxVec], xh[kLnMaxVec];
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int k = 0; k < kLnMaxVec; ++k) {
      const int col = k * 128 + lane * 4;
      g[k] = xh[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k < nvec && col < c) {
        float4 d = bf4_load(dy + row * c + col);
        if (dy2 != nullptr) {
          const float4 d2 = bf4_load(dy2 + row * c + col);
          d.x += d2.x, d.y += d2.y, d.z += d2.z, d.w += d2.w;
        }
        const float4 sv = bf4_load(s + row * c + col);
        const float4 wv = *(const float4*)(w + col);
        xh[k] = make_float4((sv.x - st.x) * st.y, (sv.y - st.x) * st.y, (sv.z - st.x) * st.y, (sv.w - st.x) * st.y);
        g[k] = make_float4(d.x * wv.x, d.y * wv.y, d.z * wv.z, d.w * wv.w);
        sg += g[k].x + g[k].y + g[k].z + g[k].w;
        sgx += g[k].x * xh[k].x + g[k].y * xh[k].y + g[k].z * xh[k].z + g[k].w * xh[k].w;
        aw[k].x += d.x * xh[k].x, aw[k].y += d.y * xh[k].y, aw[k].z += d.z * xh[k].z, aw[k].w += d.w * xh[k].w;
        ab[k].x += d.x, ab[k].y += d.y, ab[k].z += d.z, ab[k].w += d.w;
      }
    }
    const float mg = group32_sum(sg) / (float)c;
    const float mgx = group32_sum(sgx) / (float)c;
#pragma unroll
    for (int k = 0; k < kLnMaxVec; ++k) {
      const int col = k * 128 + lane * 4;
      if (k < nvec && col < c) {
        float4 o;
        o.x = st.y * (g[k].x - mg - xh[k].x * mgx);
        o.y = st.y * (g[k].y - mg - xh[k].y * mgx);
        o.z = st.y * (g[k].z - mg - xh[k].z * mgx);
        o.w = st.y * (g[k].w - mg - xh[k].w * mgx);
        bf4_store(dx + row * c + col, o);
      }
    }
  }
  // the row groups of the block add their column sums one after the other (a float atomicAdd into LDS made the order - and
  // the last bits of d(gamma), d(beta) - depend on the schedule)
  for (int turn = 0; turn < kLnRowsPerBlock; ++turn) {
    if (sub == turn) {
#pragma unroll
      for (int k = 0; k < kLnMaxVec; ++k) {
        const int col = k * 128 + lane * 4;
        if (k < nvec && col < c) {
          part[col + 0] += aw[k].x, part[col + 1] += aw[k].y, part[col + 2] += aw[k].z, part[col + 3] += aw[k].w;
          part[c + col + 0] += ab[k].x, part[c + col + 1] += ab[k].y, part[c + col + 2] += ab[k].z, part[c + col + 3] += ab[k].w;
        }
      }
    }
    __syncthreads();
  }
  float* dst = partials + (int64_t)blockIdx.x * 2 * c;
  for (int i = threadIdx.x; i < 2 * c; i += kLnThreads) dst[i] = part[i];
}

// C = 128: 8 bytes per lane and row, EIGHT rows per 32-lane group in flight per iteration (the generic kernel above walks
// one row at a time behind two dependent shuffle reductions: 34 us = 2.4 TB/s on 3-4 bf16 streams).
__global__ __launch_bounds__(kLnThreads) void add_ln_bwd_bf16_c128_k(const unsigned short* __restrict__ dy,
                                                                     const unsigned short* __restrict__ dy2,
                                                                     const unsigned short* __restrict__ s,
                                                                     const float2* __restrict__ stats,
                                                                     const float* __restrict__ w, int64_t m,
                                                                     unsigned short* __restrict__ dx,
                                                                     float* __restrict__ partials) {
  constexpr int C = 128, R = 8;
  __shared__ float part[2 * C];
  for (int i = threadIdx.x; i < 2 * C; i += kLnThreads) part[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 31;
  const int sub = threadIdx.x >> 5;
  const int col = lane * 4;
  const float4 wv = *(const float4*)(w + col);
  float4 aw = make_float4(0.f, 0.f, 0.f, 0.f), ab = aw;
  const int64_t stride = (int64_t)gridDim.x * kLnRowsPerBlock;
  for (int64_t row0 = (int64_t)blockIdx.x * kLnRowsPerBlock + sub; row0 < m; row0 += stride * R) {
    u32x2_t dv[R], d2v[R], svv[R];
    float2 st[R];
    bool ok[R];
#pragma unroll
    for (int u = 0; u < R; ++u) {
      const int64_t row = row0 + u * stride;
      ok[u] = row < m;
      const int64_t rr = ok[u] ? row : row0;  // clamped: loads are unconditional
      dv[u] = *(const u32x2_t*)(dy + rr * C + col);
      if (dy2 != nullptr) d2v[u] = *(const u32x2_t*)(dy2 + rr * C + col);
      svv[u] = *(const u32x2_t*)(s + rr * C + col);
      st[u] = stats[rr];
    }
    float4 g[R], xh[R];
    float sg[R], sgx[R];
#pragma unroll
    for (int u = 0; u < R; ++u) {
      float4 d = make_float4(__uint_as_float(dv[u][0] << 16), __uint_as_float(dv[u][0] & 0xffff0000u),
                             __uint_as_float(dv[u][1] << 16), __uint_as_float(dv[u][1] & 0xffff0000u));
      if (dy2 != nullptr) {
        d.x += __uint_as_float(d2v[u][0] << 16), d.y += __uint_as_float(d2v[u][0] & 0xffff0000u);
        d.z += __uint_as_float(d2v[u][1] << 16), d.w += __uint_as_float(d2v[u][1] & 0xffff0000u);
      }
      const float4 sv = make_float4(__uint_as_float(svv[u][0] << 16), __uint_as_float(svv[u][0] & 0xffff0000u),
                                    __uint_as_float(svv[u][1] << 16), __uint_as_float(svv[u][1] & 0xffff0000u));
      xh[u] = make_float4((sv.x - st[u].x) * st[u].y, (sv.y - st[u].x) * st[u].y, (sv.z - st[u].x) * st[u].y,
                          (sv.w - st[u].x) * st[u].y);
      g[u] = make_float4(d.x * wv.x, d.y * wv.y, d.z * wv.z, d.w * wv.w);
      sg[u] = g[u].x + g[u].y + g[u].z + g[u].w;
      sgx[u] = g[u].x * xh[u].x + g[u].y * xh[u].y + g[u].z * xh[u].z + g[u].w * xh[u].w;
      if (ok[u]) {
        aw.x += d.x * xh[u].x, aw.y += d.y * xh[u].y, aw.z += d.z * xh[u].z, aw.w += d.w * xh[u].w;
        ab.x += d.x, ab.y += d.y, ab.z += d.z, ab.w += d.w;
      }
    }
#pragma unroll
    for (int dlt = 1; dlt < 32; dlt <<= 1) {  // the 2 * R reductions interleaved
#pragma unroll
      for (int u = 0; u < R; ++u) {
        sg[u] += __shfl_xor(sg[u], dlt, 64);
        sgx[u] += __shfl_xor(sgx[u], dlt, 64);
      }
    }
#pragma unroll
    for (int u = 0; u < R; ++u) {
      if (ok[u]) {
        const float mg = sg[u] * (1.f / C), mgx = sgx[u] * (1.f / C);
        float4 o;
        o.x = st[u].y * (g[u].x - mg - xh[u].x * mgx);
        o.y = st[u].y * (g[u].y - mg - xh[u].y * mgx);
        o.z = st[u].y * (g[u].z - mg - xh[u].z * mgx);
        o.w = st[u].y * (g[u].w - mg - xh[u].w * mgx);
        bf4_store(dx + (row0 + u * stride) * C + col, o);
      }
    }
  }
  for (int turn = 0; turn < kLnRowsPerBlock; ++turn) {   // row groups in a fixed order: deterministic to the last bit
    if (sub == turn) {
      part[col + 0] += aw.x, part[col + 1] += aw.y, part[col + 2] += aw.z, part[col + 3] += aw.w;
      part[C + col + 0] += ab.x, part[C + col + 1] += ab.y, part[C + col + 2] += ab.z, part[C + col + 3] += ab.w;
    }
    __syncthreads();
  }
  float* dst = partials + (int64_t)blockIdx.x * 2 * C;
  for (int i = threadIdx.x; i < 2 * C; i += kLnThreads) dst[i] = part[i];
}

// out (bf16) = x (fp32 or bf16) [+ pos_table[pos_idx[row]]]: the entry of the bf16 encoder stack
template <typename T>
__global__ __launch_bounds__(256) void cast_add_pos_bf16_k(const T* __restrict__ x, int64_t m, int c,
                                                           const float* __restrict__ pos_table,
                                                           const int32_t* __restrict__ pos_idx,
                                                           unsigned short* __restrict__ out) {
  const int c4 = c >> 2;
  const int64_t total = m * c4;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = e / c4;
    const int col = (int)(e - row * c4) * 4;
    float4 v;
    if (sizeof(T) == 4)
      v = *(const float4*)((const float*)x + row * c + col);
    else
      v = bf4_load((const unsigned short*)x + row * c + col);
    if (pos_table != nullptr) {
      const float4 pv = *(const float4*)(pos_table + (int64_t)pos_idx[row] * c + col);
      v.x += pv.x, v.y += pv.y, v.z += pv.z, v.w += pv.w;
    }
    bf4_store(out + row * c + col, v);
  }
}

}  // namespace

extern "C" {

int sst_add_layernorm_act_fwd_f32(const float* d_x, const float* d_res, const float* d_weight, const float* d_bias,
                                  int64_t m, int c, float eps, int act, float* d_y, float* d_sum, float* d_stats,
                                  void* stream) {
  if (m < 0 || c < 1 || c > 128 * kLnMaxVec || act < 0 || act > 2) return SST_ERR_UNSUPPORTED;
  if (m == 0) return SST_OK;
  if (!d_x || !d_weight || !d_bias || !d_y || !d_stats) return SST_ERR_ARG;
  const int grid = sst_grid_1d(m, kLnRowsPerBlock);
  const bool vec = (c & 3) == 0 && (((uintptr_t)d_x | (uintptr_t)d_res | (uintptr_t)d_y | (uintptr_t)d_sum |
                                      (uintptr_t)d_weight | (uintptr_t)d_bias) & 15) == 0;
  if (vec)
    hipLaunchKernelGGL(add_ln_fwd_k, dim3(grid), dim3(kLnThreads), 0, (hipStream_t)stream, d_x, d_res, d_weight, d_bias,
                       m, c, eps, act, d_y, d_sum, (float2*)d_stats);
  else
    hipLaunchKernelGGL(add_ln_fwd_any_k, dim3(grid), dim3(kLnThreads), 0, (hipStream_t)stream, d_x, d_res, d_weight,
                       d_bias, m, c, eps, act, d_y, d_sum, (float2*)d_stats);
  SST_LAUNCH_CHECK();
  return SST_OK;
}

int sst_add_layernorm_pos_fwd_f32(const float* d_x, const float* d_res, const float* d_weight, const float* d_bias, int64_t m,
                                  int c, float eps, float* d_y, float* d_sum, float* d_stats, const float* d_pos_table,
                                  const int32_t* d_pos_idx, float* d_y_plus_pos, void* stream) {
  if (m < 0 || c < 4 || (c & 3) || c > 128 * kLnMaxVec) return SST_ERR_UNSUPPORTED;
  if (m == 0) return SST_OK;
  if (!d_x || !d_weight || !d_bias || !d_y || !d_stats || !d_pos_table || !d_pos_idx || !d_y_plus_pos) return SST_ERR_ARG;
  if ((((uintptr_t)d_x | (uintptr_t)d_res | (uintptr_t)d_y | (uintptr_t)d_sum | (uintptr_t)d_weight | (uintptr_t)d_bias |
        (uintptr_t)d_pos_table | (uintptr_t)d_y_plus_pos) & 15) != 0)
    return SST_ERR_ARG;
  hipLaunchKernelGGL(add_ln_fwd_k, dim3(sst_grid_1d(m, kLnRowsPerBlock)), dim3(kLnThreads), 0, (hipStream_t)stream, d_x, d_res,
                     d_weight, d_bias, m, c, eps, 0, d_y, d_sum, (float2*)d_stats, d_pos_table, d_pos_idx, d_y_plus_pos);
  SST_LAUNCH_CHECK();
  return SST_OK;
}

int sst_add_layernorm_fwd_f32(const float* d_x, const float* d_res, const float* d_weight, const float* d_bias,
                              int64_t m, int c, float eps, float* d_y, float* d_sum, float* d_stats, void* stream) {
  return sst_add_layernorm_act_fwd_f32(d_x, d_res, d_weight, d_bias, m, c, eps, 0, d_y, d_sum, d_stats, stream);
}

int64_t sst_add_layernorm_bwd_workspace_bytes(int64_t m, int c) {
  (void)m;
  return (int64_t)1024 * 2 * c * sizeof(float) + 256;
}

static int add_layernorm_bwd_any(const float* d_dy, const float* d_dy2, const float* d_sum, const float* d_stats,
                                 const float* d_weight, const float* d_bias, int act, int64_t m, int c, float* d_dx,
                                 float* d_dweight, float* d_dbias, void* d_workspace, void* stream,
                                 int* partial_rows = nullptr) {
  if (m < 0 || c < 1 || c > 128 * kLnMaxVec || act < 0 || act > 2) return SST_ERR_UNSUPPORTED;
  if (act && (!d_bias || d_dy2)) return SST_ERR_ARG;
  if (!d_dweight || !d_dbias) return SST_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (m == 0) {
    SST_HIP(hipMemsetAsync(d_dweight, 0, sizeof(float) * c, st));
    SST_HIP(hipMemsetAsync(d_dbias, 0, sizeof(float) * c, st));
    return SST_OK;
  }
  if (!d_dy || !d_sum || !d_stats || !d_weight || !d_dx || !d_workspace) return SST_ERR_ARG;
  int grid = (int)sst_div_up(m, kLnRowsPerBlock * 4);
  if (grid > 512) grid = 512;
  float* partials = (float*)d_workspace;
  if (d_dy2 != nullptr && c != 128) return SST_ERR_UNSUPPORTED;
  const bool vec = (c & 3) == 0 && (((uintptr_t)d_dy | (uintptr_t)d_dy2 | (uintptr_t)d_sum | (uintptr_t)d_dx |
                                      (uintptr_t)d_weight | (uintptr_t)d_bias) & 15) == 0;
  if (!vec && d_dy2 != nullptr) return SST_ERR_UNSUPPORTED;
  if (!vec)
    hipLaunchKernelGGL(add_ln_bwd_any_k, dim3(grid), dim3(kLnThreads), 2 * c * sizeof(float), st, d_dy, d_sum,
                       (const float2*)d_stats, d_weight, d_bias, act, m, c, d_dx, partials);
  else if (c == 128 && act == 0)
    hipLaunchKernelGGL(add_ln_bwd_c128_k, dim3(grid), dim3(kLnThreads), 0, st, d_dy, d_dy2, d_sum, (const float2*)d_stats,
                       d_weight, m, d_dx, partials);
  else
    hipLaunchKernelGGL(add_ln_bwd_k, dim3(grid), dim3(kLnThreads), 2 * c * sizeof(float), st, d_dy, d_sum,
                       (const float2*)d_stats, d_weight, d_bias, act, m, c, d_dx, partials);
  if (partial_rows != nullptr)     // the caller finishes the column sums itself (a rider on its next reduction launch)
    *partial_rows = grid;
  else
    hipLaunchKernelGGL(colsum_partials_k, dim3((2 * c + 31) / 32), dim3(1024), 0, st, partials, grid, 2 * c, d_dweight,
                       d_dbias, c);
  SST_LAUNCH_CHECK();
  return SST_OK;
}

}  // extern "C"

// sst_add_layernorm_bwd2_f32 without its finishing launch: the block partials of d(gamma) | d(beta) stay in the workspace as
// [*partial_rows][2 c] and the caller sums their columns (csrc/layer_exec.hip hands them to the weight-gradient reduction as a
// rider, csrc/wgrad_x6.hip).  m must be > 0.
int sst_internal_add_layernorm_bwd2_partials_f32(const float* d_dy, const float* d_dy2, const float* d_sum, const float* d_stats,
                                                 const float* d_weight, int64_t m, int c, float* d_dx, void* d_workspace,
                                                 int* partial_rows, void* stream) {
  float dummy;
  if (m <= 0 || !partial_rows) return SST_ERR_ARG;
  return add_layernorm_bwd_any(d_dy, d_dy2, d_sum, d_stats, d_weight, nullptr, 0, m, c, d_dx, &dummy, &dummy, d_workspace, stream,
                               partial_rows);
}

extern "C" {

int sst_add_layernorm_bwd2_f32(const float* d_dy, const float* d_dy2, const float* d_sum, const float* d_stats,
                               const float* d_weight, int64_t m, int c, float* d_dx, float* d_dweight, float* d_dbias,
                               void* d_workspace, void* stream) {
  return add_layernorm_bwd_any(d_dy, d_dy2, d_sum, d_stats, d_weight, nullptr, 0, m, c, d_dx, d_dweight, d_dbias,
                               d_workspace, stream);
}

int sst_add_layernorm_act_bwd_f32(const float* d_dy, const float* d_sum, const float* d_stats, const float* d_weight,
                                  const float* d_bias, int act, int64_t m, int c, float* d_dx, float* d_dweight,
                                  float* d_dbias, void* d_workspace, void* stream) {
  return add_layernorm_bwd_any(d_dy, nullptr, d_sum, d_stats, d_weight, d_bias, act, m, c, d_dx, d_dweight, d_dbias,
                               d_workspace, stream);
}

int sst_add_layernorm_bwd_f32(const float* d_dy, const float* d_sum, const float* d_stats, const float* d_weight,
                              int64_t m, int c, float* d_dx, float* d_dweight, float* d_dbias, void* d_workspace,
                              void* stream) {
  return sst_add_layernorm_bwd2_f32(d_dy, nullptr, d_sum, d_stats, d_weight, m, c, d_dx, d_dweight, d_dbias, d_workspace, stream);
}

int sst_add_layernorm_fwd_bf16(const void* d_x, const void* d_res, const float* d_weight, const float* d_bias, int64_t m,
                               int c, float eps, void* d_y, void* d_sum, float* d_stats, const float* d_pos_table,
                               const int32_t* d_pos_idx, void* d_y_plus_pos, void* stream) {
  if (m < 0 || c < 4 || (c & 3) || c > 128 * kLnMaxVec) return SST_ERR_UNSUPPORTED;
  if (m == 0) return SST_OK;
  if (!d_x || !d_weight || !d_bias || !d_y || !d_stats) return SST_ERR_ARG;
  if ((d_pos_table != nullptr) != (d_pos_idx != nullptr) || (d_pos_table != nullptr) != (d_y_plus_pos != nullptr))
    return SST_ERR_ARG;
  hipLaunchKernelGGL(add_ln_fwd_bf16_k, dim3(sst_grid_1d(m, kLnRowsPerBlock)), dim3(kLnThreads), 0, (hipStream_t)stream,
                     (const unsigned short*)d_x, (const unsigned short*)d_res, d_weight, d_bias, m, c, eps,
                     (unsigned short*)d_y, (unsigned short*)d_sum, (float2*)d_stats, d_pos_table, d_pos_idx,
                     (unsigned short*)d_y_plus_pos);
  SST_LAUNCH_CHECK();
  return SST_OK;
}

static int add_layernorm_bwd_bf16_any(const void* d_dy, const void* d_dy2, const void* d_sum, const float* d_stats,
                                      const float* d_weight, int64_t m, int c, void* d_dx, float* d_dweight, float* d_dbias,
                                      void* d_workspace, void* stream, int* partial_rows) {
  // 4-wide bf16 row accesses (col = k * 128 + lane * 4 under a `col < c` check): c must be a multiple of 4, like the forward
  if (m < 0 || c < 4 || (c & 3) || c > 128 * kLnMaxVec) return SST_ERR_UNSUPPORTED;
  if (!d_dweight || !d_dbias) return SST_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (m == 0) {
    SST_HIP(hipMemsetAsync(d_dweight, 0, sizeof(float) * c, st));
    SST_HIP(hipMemsetAsync(d_dbias, 0, sizeof(float) * c, st));
    return SST_OK;
  }
  if (!d_dy || !d_sum || !d_stats || !d_weight || !d_dx || !d_workspace) return SST_ERR_ARG;
  int grid = (int)sst_div_up(m, kLnRowsPerBlock * (c == 128 ? 8 : 4));
  if (grid > 512) grid = 512;
  float* partials = (float*)d_workspace;
  if (c == 128)
    hipLaunchKernelGGL(add_ln_bwd_bf16_c128_k, dim3(grid), dim3(kLnThreads), 0, st, (const unsigned short*)d_dy,
                       (const unsigned short*)d_dy2, (const unsigned short*)d_sum, (const float2*)d_stats, d_weight, m,
                       (unsigned short*)d_dx, partials);
  else
    hipLaunchKernelGGL(add_ln_bwd_bf16_k, dim3(grid), dim3(kLnThreads), 2 * c * sizeof(float), st,
                       (const unsigned short*)d_dy, (const unsigned short*)d_dy2, (const unsigned short*)d_sum,
                       (const float2*)d_stats, d_weight, m, c, (unsigned short*)d_dx, partials);
  if (partial_rows != nullptr)     // the caller finishes the column sums itself (a rider on its reduction launch)
    *partial_rows = grid;
  else
    hipLaunchKernelGGL(colsum_partials_k, dim3((2 * c + 31) / 32), dim3(1024), 0, st, partials, grid, 2 * c, d_dweight,
                       d_dbias, c);
  SST_LAUNCH_CHECK();
  return SST_OK;
}

int sst_add_layernorm_bwd_bf16(const void* d_dy, const void* d_dy2, const void* d_sum, const float* d_stats,
                               const float* d_weight, int64_t m, int c, void* d_dx, float* d_dweight, float* d_dbias,
                               void* d_workspace, void* stream) {
  return add_layernorm_bwd_bf16_any(d_dy, d_dy2, d_sum, d_stats, d_weight, m, c, d_dx, d_dweight, d_dbias, d_workspace, stream,
                                    nullptr);
}

}  // extern "C"

// sst_add_layernorm_bwd_bf16 without its finishing launch (see sst_internal_add_layernorm_bwd2_partials_f32).  m > 0.
int sst_internal_add_layernorm_bwd_bf16_partials(const void* d_dy, const void* d_dy2, const void* d_sum, const float* d_stats,
                                                 const float* d_weight, int64_t m, int c, void* d_dx, void* d_workspace,
                                                 int* partial_rows, void* stream) {
  float dummy;
  if (m <= 0 || !partial_rows) return SST_ERR_ARG;
  return add_layernorm_bwd_bf16_any(d_dy, d_dy2, d_sum, d_stats, d_weight, m, c, d_dx, &dummy, &dummy, d_workspace, stream,
                                    partial_rows);
}

extern "C" {

int sst_cast_add_pos_bf16(const void* d_x, int x_is_bf16, int64_t m, int c, const float* d_pos_table,
                          const int32_t* d_pos_idx, void* d_out, void* stream) {
  if (m < 0 || c < 4 || (c & 3)) return SST_ERR_ARG;
  if (m == 0) return SST_OK;
  if (!d_x || !d_out || ((d_pos_table != nullptr) != (d_pos_idx != nullptr))) return SST_ERR_ARG;
  const int grid = sst_grid_1d(m * (c >> 2), 256);
  if (x_is_bf16)
    hipLaunchKernelGGL(cast_add_pos_bf16_k<unsigned short>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)d_x, m, c, d_pos_table, d_pos_idx, (unsigned short*)d_out);
  else
    hipLaunchKernelGGL(cast_add_pos_bf16_k<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)d_x, m, c,
                       d_pos_table, d_pos_idx, (unsigned short*)d_out);
  SST_LAUNCH_CHECK();
  return SST_OK;
}

int64_t sst_colsum_workspace_bytes(int64_t m, int c) {
  (void)m;
  return (int64_t)2048 * c * sizeof(float) + 256;
}

int sst_colsum_f32(const float* d_x, int64_t m, int c, int64_t ld, float* d_out, void* d_workspace, void* stream) {
  if (m < 0 || c < 4 || (c & 3) || c > 1024 || ld < c) return SST_ERR_UNSUPPORTED;
  if (!d_out) return SST_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  SST_HIP(hipMemsetAsync(d_out, 0, sizeof(float) * c, st));
  if (m == 0) return SST_OK;
  if (!d_x || !d_workspace) return SST_ERR_ARG;
  int64_t rows_per_block = 64;
  int64_t grid = sst_div_up(m, rows_per_block);
  if (grid > 2048) {
    grid = 2048;
    rows_per_block = sst_div_up(m, grid);
    grid = sst_div_up(m, rows_per_block);
  }
  float* partials = (float*)d_workspace;
  hipLaunchKernelGGL(colsum_k, dim3((unsigned)grid), dim3(256), 0, st, d_x, m, c, ld, rows_per_block, partials);
  // one block per 32 columns walks ALL block partials in a fixed order (the 2-D variant added its 64-row slices into the
  // output with float atomics: bias gradients differed in the last bits from launch to launch)
  hipLaunchKernelGGL(colsum_partials_k, dim3((c + 31) / 32), dim3(1024), 0, st, partials, (int)grid, c, d_out, d_out, c);
  SST_LAUNCH_CHECK();
  return SST_OK;
}

}  // extern "C"
