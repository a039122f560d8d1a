// BatchNorm1d (+ ReLU) over tall point matrices x[N, C] on gfx950: the norm of every DynamicVFE / SIR layer
// (mmdet3d/models/voxel_encoders/utils.py:107-144 "Linear -> norm -> ReLU", norm = BN1d / naiveSyncBN1d,
// mmdet3d/ops/norm.py:28-86).  N ~ 1e5 points, C = 64..256: pure HBM streaming, so the work is organised as
//   forward : bn_moments_k (read x once, fp64 column sums)      -> bn_act_fwd_k (y = act(x * scale + shift))
//   backward: bn_bwd_moments_k (sum g, sum g * xhat, fp64)      -> bn_act_bwd_k (dx = scale * (g - a - xhat * b))
// with the [C]-sized algebra in between (running statistics, cross-rank averaging of naiveSyncBN) left to the
// host.  The library path this replaces runs 107 us (statistics) + 130 us (backward reduce) per layer at
// N = 116 k, C = 128 plus separate ReLU kernels.  Column sums are accumulated in fp64: var = E[x^2] - mean^2 is
// then exact to fp32 rounding even when |mean| >> std (raw coordinates are among the input channels).
#include "common.h"

namespace {

constexpr int kBnThreads = 256;

// Optional SPARSE part of the incoming gradient: the gradient of a max pooling over point groups that sits right behind
// this norm + activation (DynamicVFE: vfe layer -> scatter max, voxel_encoder.py:286-296).  Point `row` receives
// dpool[v][ch] where v = group[row] >= 0 and arg[v][ch] == row (the arg-max rows the pooling recorded), nothing elsewhere;
// the dense [N, C] matrix of that gradient (mostly zeros) is never written.  The dense part (dy) may then be absent.
struct bn_pool_grad {
  const int32_t* group;   // [N]
  const int32_t* arg;     // [G, C]
  const float* dpool;     // [G, ldp]
  int64_t ldp;
};
__device__ __forceinline__ float4 bn_pool_part(const bn_pool_grad& pg, int64_t row, int c, int cc) {
  float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
  const int v = pg.group[row];
  if (v >= 0) {
    const int4 a = *(const int4*)(pg.arg + (int64_t)v * c + cc * 4);
    const float4 d = *(const float4*)(pg.dpool + (int64_t)v * pg.ldp + cc * 4);
    g.x = a.x == (int)row ? d.x : 0.f;
    g.y = a.y == (int)row ? d.y : 0.f;
    g.z = a.z == (int)row ? d.z : 0.f;
    g.w = a.w == (int)row ? d.w : 0.f;
  }
  return g;
}

// Block partials of two column moments.  MODE 0: (sum x, sum x^2).  MODE 1: (sum g, sum g * xhat) with
// g = dy * [act(x*scale+shift (+ res)) > 0 or no act], xhat = (x - mean) * invstd.
// Thread (ry, cx) owns float4 column group cx and rows ry, ry + rpi, ...; partial[block][2c] doubles.
template <int MODE>
__global__ __launch_bounds__(kBnThreads) void bn_moments_k(const float* __restrict__ x, const float* __restrict__ dy,
                                                          int64_t n, int c, int64_t ldx, int64_t lddy,
                                                          const float* __restrict__ mean,
                                                          const float* __restrict__ invstd,
                                                          const float* __restrict__ scale,
                                                          const float* __restrict__ shift, int act,
                                                          const float* __restrict__ res, int64_t ldr,
                                                          int64_t rows_per_block, double* __restrict__ partial,
                                                          const bn_pool_grad pg) {
  extern __shared__ __attribute__((aligned(16))) double red[];  // [rpi][2c]
  const int c4 = c >> 2;
  const int rpi = kBnThreads / c4 > 0 ? kBnThreads / c4 : 1;
  const int ry = threadIdx.x / c4, cx = threadIdx.x - ry * c4;
  const int64_t beg = (int64_t)blockIdx.x * rows_per_block;
  const int64_t end = beg + rows_per_block < n ? beg + rows_per_block : n;
  double s1[4] = {0., 0., 0., 0.}, s2[4] = {0., 0., 0., 0.};
  if (ry < rpi) {
    for (int cc = cx; cc < c4; cc += kBnThreads) {  // c4 <= 256: single trip
      float4 mu = make_float4(0.f, 0.f, 0.f, 0.f), is = mu, sc = mu, sh = mu;
      if (MODE == 1) {
        mu = *(const float4*)(mean + cc * 4);
        is = *(const float4*)(invstd + cc * 4);
        sc = *(const float4*)(scale + cc * 4);
        sh = *(const float4*)(shift + cc * 4);
      }
      for (int64_t row = beg + ry; row < end; row += rpi) {
        const float4 v = *(const float4*)(x + row * ldx + cc * 4);
        if (MODE == 0) {
          s1[0] += v.x, s1[1] += v.y, s1[2] += v.z, s1[3] += v.w;
          s2[0] += (double)v.x * v.x, s2[1] += (double)v.y * v.y;
          s2[2] += (double)v.z * v.z, s2[3] += (double)v.w * v.w;
        } else {
          float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
          if (dy != nullptr) g = *(const float4*)(dy + row * lddy + cc * 4);
          if (pg.group != nullptr) {
            const float4 sp = bn_pool_part(pg, row, c, cc);
            g.x += sp.x, g.y += sp.y, g.z += sp.z, g.w += sp.w;
          }
          if (act) {
            float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
            if (res != nullptr) r = *(const float4*)(res + row * ldr + cc * 4);
            g.x = (v.x * sc.x + sh.x + r.x) > 0.f ? g.x : 0.f;
            g.y = (v.y * sc.y + sh.y + r.y) > 0.f ? g.y : 0.f;
            g.z = (v.z * sc.z + sh.z + r.z) > 0.f ? g.z : 0.f;
            g.w = (v.w * sc.w + sh.w + r.w) > 0.f ? g.w : 0.f;
          }
          s1[0] += g.x, s1[1] += g.y, s1[2] += g.z, s1[3] += g.w;
          s2[0] += (double)(g.x * ((v.x - mu.x) * is.x)), s2[1] += (double)(g.y * ((v.y - mu.y) * is.y));
          s2[2] += (double)(g.z * ((v.z - mu.z) * is.z)), s2[3] += (double)(g.w * ((v.w - mu.w) * is.w));
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      red[(size_t)ry * 2 * c + cx * 4 + k] = s1[k];
      red[(size_t)ry * 2 * c + c + cx * 4 + k] = s2[k];
    }
  }
  __syncthreads();
  double* dst = partial + (int64_t)blockIdx.x * 2 * c;
  for (int i = threadIdx.x; i < 2 * c; i += kBnThreads) {
    double t = 0.;
    for (int r = 0; r < rpi; ++r) t += red[(size_t)r * 2 * c + i];
    dst[i] = t;
  }
}

// Combines the block partials of 32 channels (both moments).  MODE 0: mean = S1/n, var = S2/n - mean^2 (biased).
// MODE 1: out0 = S1, out1 = S2 (plain sums; the host divides by the right count).
template <int MODE>
__global__ __launch_bounds__(1024) void bn_finish_k(const double* __restrict__ partial, int nb, int c, double inv_n,
                                                    float* __restrict__ out0, float* __restrict__ out1) {
  __shared__ double r1[32][33], r2[32][33];
  const int cx = threadIdx.x & 31, gy = threadIdx.x >> 5;  // 32 channels x 32 slices of the block partials
  const int ch = blockIdx.x * 32 + cx;
  double a1 = 0., a2 = 0.;
  if (ch < c)
    for (int b = gy; b < nb; b += 32) {
      a1 += partial[(int64_t)b * 2 * c + ch];
      a2 += partial[(int64_t)b * 2 * c + c + ch];
    }
  r1[gy][cx] = a1;
  r2[gy][cx] = a2;
  __syncthreads();
  if (gy == 0 && ch < c) {
    double t1 = 0., t2 = 0.;
#pragma unroll
    for (int k = 0; k < 32; ++k) t1 += r1[k][cx], t2 += r2[k][cx];
    if (MODE == 0) {
      const double m = t1 * inv_n;
      double v = t2 * inv_n - m * m;
      if (v < 0.) v = 0.;
      out0[ch] = (float)m;
      out1[ch] = (float)v;
    } else {
      out0[ch] = (float)t1;
      out1[ch] = (float)t2;
    }
  }
}

// Training-mode "prepare" of one BatchNorm layer from the block partials: mean / biased variance -> invstd,
// scale = weight * invstd, shift = bias - mean * scale, and the running statistics
// (running = (1 - f) * running + f * {mean, var * unbiased}) in the same launch: the [C]-sized algebra costs a dozen
// tiny element-wise launches when it is left to the host framework.
__global__ __launch_bounds__(1024) void bn_prepare_finish_k(const double* __restrict__ partial, int nb, int c,
                                                            double inv_n, const float* __restrict__ weight,
                                                            const float* __restrict__ bias, float eps,
                                                            float* __restrict__ running_mean,
                                                            float* __restrict__ running_var, float factor,
                                                            float unbiased, int64_t* __restrict__ tracked,
                                                            float* __restrict__ out) {  // out [4][c]
  __shared__ double r1[32][33], r2[32][33];
  const int cx = threadIdx.x & 31, gy = threadIdx.x >> 5;
  const int ch = blockIdx.x * 32 + cx;
  double a1 = 0., a2 = 0.;
  if (ch < c)
    for (int b = gy; b < nb; b += 32) {
      a1 += partial[(int64_t)b * 2 * c + ch];
      a2 += partial[(int64_t)b * 2 * c + c + ch];
    }
  r1[gy][cx] = a1;
  r2[gy][cx] = a2;
  __syncthreads();
  if (gy == 0 && ch < c) {
    double t1 = 0., t2 = 0.;
#pragma unroll
    for (int k = 0; k < 32; ++k) t1 += r1[k][cx], t2 += r2[k][cx];
    const double m = t1 * inv_n;
    double v = t2 * inv_n - m * m;
    if (v < 0.) v = 0.;
    const float mean = (float)m, var = (float)v;
    const float invstd = rsqrtf(var + eps);
    const float sc = weight != nullptr ? weight[ch] * invstd : invstd;
    const float sh = (bias != nullptr ? bias[ch] : 0.f) - mean * sc;
    out[ch] = mean;
    out[c + ch] = invstd;
    out[2 * c + ch] = sc;
    out[3 * c + ch] = sh;
    if (running_mean != nullptr) running_mean[ch] = (1.f - factor) * running_mean[ch] + factor * mean;
    if (running_var != nullptr) running_var[ch] = (1.f - factor) * running_var[ch] + factor * (var * unbiased);
    if (tracked != nullptr && ch == 0) tracked[0] += 1;   // nn.BatchNorm1d.num_batches_tracked
  }
}

// y = act(x * scale + shift (+ res)); one float4 per thread, grid-stride.  res: the identity branch of a residual block
// (sparse_block.py:127-139: bn2 -> += identity -> ReLU), folded into the same pass.
__global__ __launch_bounds__(kBnThreads) void bn_act_fwd_k(const float* __restrict__ x, int64_t n, int c, int64_t ldx,
                                                          const float* __restrict__ scale,
                                                          const float* __restrict__ shift, int act,
                                                          const float* __restrict__ res, int64_t ldr,
                                                          float* __restrict__ y, int64_t ldy) {
  const int c4 = c >> 2;
  const int64_t total = n * c4;
  for (int64_t i = (int64_t)blockIdx.x * kBnThreads + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBnThreads) {
    const int64_t row = i / c4;
    const int cc = (int)(i - row * c4);
    const float4 v = *(const float4*)(x + row * ldx + cc * 4);
    const float4 sc = *(const float4*)(scale + cc * 4);
    const float4 sh = *(const float4*)(shift + cc * 4);
    float4 o = make_float4(v.x * sc.x + sh.x, v.y * sc.y + sh.y, v.z * sc.z + sh.z, v.w * sc.w + sh.w);
    if (res != nullptr) {
      const float4 r = *(const float4*)(res + row * ldr + cc * 4);
      o.x += r.x, o.y += r.y, o.z += r.z, o.w += r.w;
    }
    if (act) {
      o.x = o.x > 0.f ? o.x : 0.f;
      o.y = o.y > 0.f ? o.y : 0.f;
      o.z = o.z > 0.f ? o.z : 0.f;
      o.w = o.w > 0.f ? o.w : 0.f;
    }
    *(float4*)(y + row * ldy + cc * 4) = o;
  }
}

// dx = scale * (g - ca - xhat * cb), g = dy masked by the activation (ca = G1/count, cb = G2/count; zeros in eval mode)
__global__ __launch_bounds__(kBnThreads) void bn_act_bwd_k(const float* __restrict__ dy, const float* __restrict__ x,
                                                          int64_t n, int c, int64_t lddy, int64_t ldx,
                                                          const float* __restrict__ mean,
                                                          const float* __restrict__ invstd,
                                                          const float* __restrict__ scale,
                                                          const float* __restrict__ shift,
                                                          const float* __restrict__ ca, const float* __restrict__ cb,
                                                          float coef_scale, int act, const float* __restrict__ res,
                                                          int64_t ldr, float* __restrict__ dres, int64_t lddres,
                                                          float* __restrict__ dx, int64_t lddx, const bn_pool_grad pg) {
  const int c4 = c >> 2;
  const int64_t total = n * c4;
  for (int64_t i = (int64_t)blockIdx.x * kBnThreads + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBnThreads) {
    const int64_t row = i / c4;
    const int cc = (int)(i - row * c4);
    const float4 v = *(const float4*)(x + row * ldx + cc * 4);
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    if (dy != nullptr) g = *(const float4*)(dy + row * lddy + cc * 4);
    if (pg.group != nullptr) {
      const float4 sp = bn_pool_part(pg, row, c, cc);
      g.x += sp.x, g.y += sp.y, g.z += sp.z, g.w += sp.w;
    }
    const float4 sc = *(const float4*)(scale + cc * 4);
    const float4 sh = *(const float4*)(shift + cc * 4);
    const float4 mu = *(const float4*)(mean + cc * 4);
    const float4 is = *(const float4*)(invstd + cc * 4);
    float4 a = *(const float4*)(ca + cc * 4);  // sum g, sum g * xhat: scaled to means here (0 in eval mode)
    float4 b = *(const float4*)(cb + cc * 4);
    a.x *= coef_scale, a.y *= coef_scale, a.z *= coef_scale, a.w *= coef_scale;
    b.x *= coef_scale, b.y *= coef_scale, b.z *= coef_scale, b.w *= coef_scale;
    if (act) {
      float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
      if (res != nullptr) r = *(const float4*)(res + row * ldr + cc * 4);
      g.x = (v.x * sc.x + sh.x + r.x) > 0.f ? g.x : 0.f;
      g.y = (v.y * sc.y + sh.y + r.y) > 0.f ? g.y : 0.f;
      g.z = (v.z * sc.z + sh.z + r.z) > 0.f ? g.z : 0.f;
      g.w = (v.w * sc.w + sh.w + r.w) > 0.f ? g.w : 0.f;
    }
    if (dres != nullptr) *(float4*)(dres + row * lddres + cc * 4) = g;   // gradient of the identity branch
    float4 o;
    o.x = sc.x * (g.x - a.x - (v.x - mu.x) * is.x * b.x);
    o.y = sc.y * (g.y - a.y - (v.y - mu.y) * is.y * b.y);
    o.z = sc.z * (g.z - a.z - (v.z - mu.z) * is.z * b.z);
    o.w = sc.w * (g.w - a.w - (v.w - mu.w) * is.w * b.w);
    *(float4*)(dx + row * lddx + cc * 4) = o;
  }
}

// y[N, C] = x[N, K] W^T for the FIRST layer of a point encoder (K = 10 / 11 decorated point channels, voxel_encoder.py:
// 258-286; utils.py:107-144 Linear(bias=False) -> norm -> ReLU) together with the block partials of the batch-norm moments
// of y - the matrix is written once and not read again for its statistics.  K <= 16: no matrix instruction pays (a 16 x 16 x 4
// tile would be 60 % padding and the kernel is bound by the 4 C bytes it writes per point); W^T sits in LDS, thread (ry, cx)
// owns float4 column group cx of the rows ry, ry + rpi, ... exactly like bn_moments_k, so the partials have its layout.
__global__ __launch_bounds__(kBnThreads) void vfe_linear_smallk_k(const float* __restrict__ x, int64_t ldx, int64_t n, int k,
                                                                 const float* __restrict__ w, int64_t ldw, int c,
                                                                 float* __restrict__ y, int64_t ldy, int64_t rows_per_block,
                                                                 double* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) double red[];  // [rpi][2c] doubles, then W^T [k][c] floats
  const int c4 = c >> 2;
  const int rpi = kBnThreads / c4;
  float* wt = (float*)(red + (size_t)rpi * 2 * c);
  for (int i = threadIdx.x; i < k * c; i += kBnThreads) {
    const int kk = i / c, ch = i - kk * c;
    wt[i] = w[(int64_t)ch * ldw + kk];
  }
  __syncthreads();
  const int ry = threadIdx.x / c4, cx = threadIdx.x - ry * c4;
  const int64_t beg = (int64_t)blockIdx.x * rows_per_block;
  const int64_t end = beg + rows_per_block < n ? beg + rows_per_block : n;
  double s1[4] = {0., 0., 0., 0.}, s2[4] = {0., 0., 0., 0.};
  if (ry < rpi) {
    for (int64_t row = beg + ry; row < end; row += rpi) {
      const float* xr = x + row * ldx;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int kk = 0; kk < k; ++kk) {
        const float xv = xr[kk];
        const float4 wv = *(const float4*)(wt + kk * c + cx * 4);
        acc.x = fmaf(xv, wv.x, acc.x), acc.y = fmaf(xv, wv.y, acc.y), acc.z = fmaf(xv, wv.z, acc.z), acc.w = fmaf(xv, wv.w, acc.w);
      }
      *(float4*)(y + row * ldy + cx * 4) = acc;
      s1[0] += acc.x, s1[1] += acc.y, s1[2] += acc.z, s1[3] += acc.w;
      s2[0] += (double)acc.x * acc.x, s2[1] += (double)acc.y * acc.y;
      s2[2] += (double)acc.z * acc.z, s2[3] += (double)acc.w * acc.w;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      red[(size_t)ry * 2 * c + cx * 4 + q] = s1[q];
      red[(size_t)ry * 2 * c + c + cx * 4 + q] = s2[q];
    }
  }
  __syncthreads();
  double* dst = partial + (int64_t)blockIdx.x * 2 * c;
  for (int i = threadIdx.x; i < 2 * c; i += kBnThreads) {
    double t = 0.;
    for (int r = 0; r < rpi; ++r) t += red[(size_t)r * 2 * c + i];
    dst[i] = t;
  }
}

int moments_grid(int64_t n, int64_t* rows_per_block) {
  int64_t rpb = 128;
  int64_t grid = sst_div_up(n, rpb);
  if (grid > 512) {
    rpb = sst_div_up(n, 512);
    grid = sst_div_up(n, rpb);
  }
  *rows_per_block = rpb;
  return (int)grid;
}

bool bn_shape_ok(int64_t n, int c) { return n >= 0 && c >= 4 && (c & 3) == 0 && c <= 1024; }

}  // namespace

extern "C" {

int64_t sst_bn_workspace_bytes(int64_t n, int c) {
  (void)n;
  return (int64_t)1024 * 2 * c * sizeof(double) + 256;
}

int sst_bn_stats_f32(const float* d_x, int64_t n, int c, int64_t ld, float* d_mean, float* d_var, void* d_workspace,
                     void* stream) {
  if (!bn_shape_ok(n, c) || ld < c || (ld & 3)) return SST_ERR_UNSUPPORTED;
  if (n == 0) return SST_ERR_ARG;  // batch statistics of an empty batch are undefined (the reference asserts)
  if (!d_x || !d_mean || !d_var || !d_workspace || ((uintptr_t)d_x & 15)) return SST_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  int64_t rpb;
  const int grid = moments_grid(n, &rpb);
  double* partial = (double*)d_workspace;
  const size_t lds = (size_t)kBnThreads * 8 * sizeof(double);
  hipLaunchKernelGGL(bn_moments_k<0>, dim3(grid), dim3(kBnThreads), lds, st, d_x, nullptr, n, c, ld, 0, nullptr,
                     nullptr, nullptr, nullptr, 0, nullptr, 0, rpb, partial, bn_pool_grad{nullptr, nullptr, nullptr, 0});
  hipLaunchKernelGGL(bn_finish_k<0>, dim3((c + 31) / 32), dim3(1024), 0, st, partial, grid, c, 1.0 / (double)n,
                     d_mean, d_var);
  SST_LAUNCH_CHECK();
  return SST_OK;
}

int sst_bn_act_res_fwd_f32(const float* d_x, int64_t n, int c, int64_t ldx, const float* d_res, int64_t ldr,
                           const float* d_scale, const float* d_shift, int act, float* d_y, int64_t ldy, void* stream) {
  if (!bn_shape_ok(n, c) || ldx < c || ldy < c || (ldx & 3) || (ldy & 3) || act < 0 || act > 1)
    return SST_ERR_UNSUPPORTED;
  if (d_res && (ldr < c || (ldr & 3) || ((uintptr_t)d_res & 15))) return SST_ERR_UNSUPPORTED;
  if (n == 0) return SST_OK;
  if (!d_x || !d_scale || !d_shift || !d_y || ((uintptr_t)d_x & 15) || ((uintptr_t)d_y & 15)) return SST_ERR_ARG;
  const int64_t total = n * (c >> 2);
  int64_t grid = sst_div_up(total, kBnThreads);
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(bn_act_fwd_k, dim3((unsigned)grid), dim3(kBnThreads), 0, (hipStream_t)stream, d_x, n, c, ldx,
                     d_scale, d_shift, act, d_res, ldr, d_y, ldy);
  SST_LAUNCH_CHECK();
  return SST_OK;
}

int sst_bn_act_fwd_f32(const float* d_x, int64_t n, int c, int64_t ldx, const float* d_scale, const float* d_shift,
                       int act, float* d_y, int64_t ldy, void* stream) {
  return sst_bn_act_res_fwd_f32(d_x, n, c, ldx, nullptr, 0, d_scale, d_shift, act, d_y, ldy, stream);
}

int sst_bn_act_res_bwd_reduce_f32(const float* d_dy, const float* d_x, const float* d_res, int64_t n, int c,
                                  int64_t lddy, int64_t ldx, int64_t ldr, const float* d_mean, const float* d_invstd,
                                  const float* d_scale, const float* d_shift, int act, float* d_sum_g,
                                  float* d_sum_gxhat, void* d_workspace, void* stream) {
  if (!bn_shape_ok(n, c) || ldx < c || lddy < c || (ldx & 3) || (lddy & 3) || act < 0 || act > 1)
    return SST_ERR_UNSUPPORTED;
  if (d_res && (ldr < c || (ldr & 3) || ((uintptr_t)d_res & 15))) return SST_ERR_UNSUPPORTED;
  if (!d_sum_g || !d_sum_gxhat) return SST_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) {
    SST_HIP(hipMemsetAsync(d_sum_g, 0, sizeof(float) * c, st));
    SST_HIP(hipMemsetAsync(d_sum_gxhat, 0, sizeof(float) * c, st));
    return SST_OK;
  }
  if (!d_dy || !d_x || !d_mean || !d_invstd || !d_scale || !d_shift || !d_workspace || ((uintptr_t)d_x & 15) ||
      ((uintptr_t)d_dy & 15))
    return SST_ERR_ARG;
  int64_t rpb;
  const int grid = moments_grid(n, &rpb);
  double* partial = (double*)d_workspace;
  const size_t lds = (size_t)kBnThreads * 8 * sizeof(double);
  hipLaunchKernelGGL(bn_moments_k<1>, dim3(grid), dim3(kBnThreads), lds, st, d_x, d_dy, n, c, ldx, lddy, d_mean,
                     d_invstd, d_scale, d_shift, act, d_res, ldr, rpb, partial, bn_pool_grad{nullptr, nullptr, nullptr, 0});
  hipLaunchKernelGGL(bn_finish_k<1>, dim3((c + 31) / 32), dim3(1024), 0, st, partial, grid, c, 0.0, d_sum_g,
                     d_sum_gxhat);
  SST_LAUNCH_CHECK();
  return SST_OK;
}

int sst_bn_act_bwd_reduce_f32(const float* d_dy, const float* d_x, int64_t n, int c, int64_t lddy, int64_t ldx,
                              const float* d_mean, const float* d_invstd, const float* d_scale, const float* d_shift,
                              int act, float* d_sum_g, float* d_sum_gxhat, void* d_workspace, void* stream) {
  return sst_bn_act_res_bwd_reduce_f32(d_dy, d_x, nullptr, n, c, lddy, ldx, 0, d_mean, d_invstd, d_scale, d_shift, act,
                                       d_sum_g, d_sum_gxhat, d_workspace, stream);
}

int sst_bn_act_res_bwd_apply_f32(const float* d_dy, const float* d_x, const float* d_res, int64_t n, int c,
                                 int64_t lddy, int64_t ldx, int64_t ldr, const float* d_mean, const float* d_invstd,
                                 const float* d_scale, const float* d_shift, const float* d_coef_a,
                                 const float* d_coef_b, float coef_scale, int act, float* d_dres, int64_t lddres,
                                 float* d_dx, int64_t lddx, void* stream) {
  if (!bn_shape_ok(n, c) || ldx < c || lddy < c || lddx < c || (ldx & 3) || (lddy & 3) || (lddx & 3) || act < 0 ||
      act > 1)
    return SST_ERR_UNSUPPORTED;
  if (d_res && (ldr < c || (ldr & 3) || ((uintptr_t)d_res & 15))) return SST_ERR_UNSUPPORTED;
  if (d_dres && (lddres < c || (lddres & 3) || ((uintptr_t)d_dres & 15))) return SST_ERR_UNSUPPORTED;
  if (n == 0) return SST_OK;
  if (!d_dy || !d_x || !d_mean || !d_invstd || !d_scale || !d_shift || !d_coef_a || !d_coef_b || !d_dx ||
      ((uintptr_t)d_x & 15) || ((uintptr_t)d_dy & 15) || ((uintptr_t)d_dx & 15))
    return SST_ERR_ARG;
  const int64_t total = n * (c >> 2);
  int64_t grid = sst_div_up(total, kBnThreads);
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(bn_act_bwd_k, dim3((unsigned)grid), dim3(kBnThreads), 0, (hipStream_t)stream, d_dy, d_x, n, c,
                     lddy, ldx, d_mean, d_invstd, d_scale, d_shift, d_coef_a, d_coef_b, coef_scale, act, d_res, ldr,
                     d_dres, lddres, d_dx, lddx, bn_pool_grad{nullptr, nullptr, nullptr, 0});
  SST_LAUNCH_CHECK();
  return SST_OK;
}

int sst_bn_act_bwd_apply_f32(const float* d_dy, const float* d_x, int64_t n, int c, int64_t lddy, int64_t ldx,
                             const float* d_mean, const float* d_invstd, const float* d_scale, const float* d_shift,
                             const float* d_coef_a, const float* d_coef_b, float coef_scale, int act, float* d_dx,
                             int64_t lddx, void* stream) {
  return sst_bn_act_res_bwd_apply_f32(d_dy, d_x, nullptr, n, c, lddy, ldx, 0, d_mean, d_invstd, d_scale, d_shift,
                                      d_coef_a, d_coef_b, coef_scale, act, nullptr, 0, d_dx, lddx, stream);
}

int sst_bn_prepare_tracked_f32(const float* d_x, int64_t n, int c, int64_t ld, const float* d_weight,
                               const float* d_bias, float eps, float* d_running_mean, float* d_running_var,
                               float factor, int64_t* d_num_batches_tracked, float* d_out4, void* d_workspace,
                               void* stream) {
  if (!bn_shape_ok(n, c) || ld < c || (ld & 3)) return SST_ERR_UNSUPPORTED;
  if (n == 0) return SST_ERR_ARG;
  if (!d_x || !d_out4 || !d_workspace || ((uintptr_t)d_x & 15)) return SST_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  int64_t rpb;
  const int grid = moments_grid(n, &rpb);
  double* partial = (double*)d_workspace;
  const size_t lds = (size_t)kBnThreads * 8 * sizeof(double);
  hipLaunchKernelGGL(bn_moments_k<0>, dim3(grid), dim3(kBnThreads), lds, st, d_x, nullptr, n, c, ld, 0, nullptr,
                     nullptr, nullptr, nullptr, 0, nullptr, 0, rpb, partial, bn_pool_grad{nullptr, nullptr, nullptr, 0});
  const float unbiased = n > 1 ? (float)((double)n / (double)(n - 1)) : 1.f;
  hipLaunchKernelGGL(bn_prepare_finish_k, dim3((c + 31) / 32), dim3(1024), 0, st, partial, grid, c, 1.0 / (double)n,
                     d_weight, d_bias, eps, d_running_mean, d_running_var, factor, unbiased, d_num_batches_tracked,
                     d_out4);
  SST_LAUNCH_CHECK();
  return SST_OK;
}

int sst_bn_prepare_f32(const float* d_x, int64_t n, int c, int64_t ld, const float* d_weight, const float* d_bias,
                       float eps, float* d_running_mean, float* d_running_var, float factor, float* d_out4,
                       void* d_workspace, void* stream) {
  return sst_bn_prepare_tracked_f32(d_x, n, c, ld, d_weight, d_bias, eps, d_running_mean, d_running_var, factor, nullptr,
                                    d_out4, d_workspace, stream);
}

/* ------------------------------------------------------------------------------------------------
 * Passes of DynamicVFE's layer stack fused with their neighbours (sst_amd/voxel_encoder.py: FusedVFE2).
 * sst_vfe_linear_moments_f32: y = x W^T (K <= 16, C % 4 == 0, C <= 256) + the block partials of the column moments of y in
 *   d_workspace (sst_bn_workspace_bytes(n, c) bytes), to be finished by one of the two entries below.
 * sst_bn_prepare_from_partials_f32 / sst_bn_stats_from_partials_f32: the second halves of sst_bn_prepare_tracked_f32 /
 *   sst_bn_stats_f32 on those partials (n = the row count the partials were taken over).
 * ---------------------------------------------------------------------------------------------- */
int sst_vfe_linear_moments_f32(const float* d_x, int64_t ldx, int64_t n, int k, const float* d_w, int64_t ldw, int c,
                               float* d_y, int64_t ldy, void* d_workspace, void* stream) {
  if (n < 0 || k < 1 || k > 16 || c < 4 || (c & 3) || c > 256 || ldx < k || ldw < k || ldy < c || (ldy & 3))
    return SST_ERR_UNSUPPORTED;
  if (n == 0) return SST_ERR_ARG;
  if (!d_x || !d_w || !d_y || !d_workspace || ((uintptr_t)d_y & 15)) return SST_ERR_ARG;
  int64_t rpb;
  const int grid = moments_grid(n, &rpb);
  const int rpi = kBnThreads / (c >> 2);
  const size_t lds = (size_t)rpi * 2 * c * sizeof(double) + (size_t)k * c * sizeof(float);
  hipLaunchKernelGGL(vfe_linear_smallk_k, dim3(grid), dim3(kBnThreads), lds, (hipStream_t)stream, d_x, ldx, n, k, d_w, ldw, c,
                     d_y, ldy, rpb, (double*)d_workspace);
  SST_LAUNCH_CHECK();
  return SST_OK;
}

int sst_bn_prepare_from_partials_f32(int64_t n, int c, const float* d_weight, const float* d_bias, float eps,
                                     float* d_running_mean, float* d_running_var, float factor,
                                     int64_t* d_num_batches_tracked, float* d_out4, void* d_workspace, void* stream) {
  if (!bn_shape_ok(n, c)) return SST_ERR_UNSUPPORTED;
  if (n == 0 || !d_out4 || !d_workspace) return SST_ERR_ARG;
  int64_t rpb;
  const int grid = moments_grid(n, &rpb);
  const float unbiased = n > 1 ? (float)((double)n / (double)(n - 1)) : 1.f;
  hipLaunchKernelGGL(bn_prepare_finish_k, dim3((c + 31) / 32), dim3(1024), 0, (hipStream_t)stream, (const double*)d_workspace,
                     grid, c, 1.0 / (double)n, d_weight, d_bias, eps, d_running_mean, d_running_var, factor, unbiased,
                     d_num_batches_tracked, d_out4);
  SST_LAUNCH_CHECK();
  return SST_OK;
}

int sst_bn_stats_from_partials_f32(int64_t n, int c, float* d_mean, float* d_var, void* d_workspace, void* stream) {
  if (!bn_shape_ok(n, c)) return SST_ERR_UNSUPPORTED;
  if (n == 0 || !d_mean || !d_var || !d_workspace) return SST_ERR_ARG;
  int64_t rpb;
  const int grid = moments_grid(n, &rpb);
  hipLaunchKernelGGL(bn_finish_k<0>, dim3((c + 31) / 32), dim3(1024), 0, (hipStream_t)stream, (const double*)d_workspace, grid,
                     c, 1.0 / (double)n, d_mean, d_var);
  SST_LAUNCH_CHECK();
  return SST_OK;
}

/* Backward of y = act(bn(x)) whose output was max-pooled over point groups right away: the incoming gradient is
 * d_dy (dense, optional) + the pooled gradient d_dpool [G, ldp] routed to the recorded arg-max rows d_arg [G, c] through the
 * point -> group map d_group [n] (negative: no group) - sst_bn_act_bwd_reduce_f32 / _apply_f32 without the dense [n, c]
 * matrix of the pooling's gradient (scatter_points_cuda.cu:135-179 writes it; voxel_encoder.py:286-296). */
int sst_bn_act_pool_bwd_reduce_f32(const float* d_dy, const float* d_x, int64_t n, int c, int64_t lddy, int64_t ldx,
                                   const float* d_mean, const float* d_invstd, const float* d_scale, const float* d_shift,
                                   int act, const int32_t* d_group, const int32_t* d_arg, const float* d_dpool, int64_t ldp,
                                   float* d_sum_g, float* d_sum_gxhat, void* d_workspace, void* stream) {
  if (!bn_shape_ok(n, c) || ldx < c || (ldx & 3) || act < 0 || act > 1 || ldp < c || (ldp & 3)) return SST_ERR_UNSUPPORTED;
  if (d_dy && (lddy < c || (lddy & 3) || ((uintptr_t)d_dy & 15))) return SST_ERR_UNSUPPORTED;
  if (!d_sum_g || !d_sum_gxhat) return SST_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) {
    SST_HIP(hipMemsetAsync(d_sum_g, 0, sizeof(float) * c, st));
    SST_HIP(hipMemsetAsync(d_sum_gxhat, 0, sizeof(float) * c, st));
    return SST_OK;
  }
  if (!d_x || !d_mean || !d_invstd || !d_scale || !d_shift || !d_workspace || !d_group || !d_arg || !d_dpool ||
      ((uintptr_t)d_x & 15) || ((uintptr_t)d_arg & 15) || ((uintptr_t)d_dpool & 15))
    return SST_ERR_ARG;
  int64_t rpb;
  const int grid = moments_grid(n, &rpb);
  double* partial = (double*)d_workspace;
  const size_t lds = (size_t)kBnThreads * 8 * sizeof(double);
  hipLaunchKernelGGL(bn_moments_k<1>, dim3(grid), dim3(kBnThreads), lds, st, d_x, d_dy, n, c, ldx, lddy, d_mean, d_invstd,
                     d_scale, d_shift, act, nullptr, 0, rpb, partial, bn_pool_grad{d_group, d_arg, d_dpool, ldp});
  hipLaunchKernelGGL(bn_finish_k<1>, dim3((c + 31) / 32), dim3(1024), 0, st, partial, grid, c, 0.0, d_sum_g, d_sum_gxhat);
  SST_LAUNCH_CHECK();
  return SST_OK;
}

int sst_bn_act_pool_bwd_apply_f32(const float* d_dy, const float* d_x, int64_t n, int c, int64_t lddy, int64_t ldx,
                                  const float* d_mean, const float* d_invstd, const float* d_scale, const float* d_shift,
                                  const float* d_coef_a, const float* d_coef_b, float coef_scale, int act,
                                  const int32_t* d_group, const int32_t* d_arg, const float* d_dpool, int64_t ldp, float* d_dx,
                                  int64_t lddx, void* stream) {
  if (!bn_shape_ok(n, c) || ldx < c || lddx < c || (ldx & 3) || (lddx & 3) || act < 0 || act > 1 || ldp < c || (ldp & 3))
    return SST_ERR_UNSUPPORTED;
  if (d_dy && (lddy < c || (lddy & 3) || ((uintptr_t)d_dy & 15))) return SST_ERR_UNSUPPORTED;
  if (n == 0) return SST_OK;
  if (!d_x || !d_mean || !d_invstd || !d_scale || !d_shift || !d_coef_a || !d_coef_b || !d_dx || !d_group || !d_arg ||
      !d_dpool || ((uintptr_t)d_x & 15) || ((uintptr_t)d_dx & 15) || ((uintptr_t)d_arg & 15) || ((uintptr_t)d_dpool & 15))
    return SST_ERR_ARG;
  const int64_t total = n * (c >> 2);
  int64_t grid = sst_div_up(total, kBnThreads);
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(bn_act_bwd_k, dim3((unsigned)grid), dim3(kBnThreads), 0, (hipStream_t)stream, d_dy, d_x, n, c, lddy, ldx,
                     d_mean, d_invstd, d_scale, d_shift, d_coef_a, d_coef_b, coef_scale, act, nullptr, 0, nullptr, 0, d_dx, lddx,
                     bn_pool_grad{d_group, d_arg, d_dpool, ldp});
  SST_LAUNCH_CHECK();
  return SST_OK;
}

}  // extern "C"
