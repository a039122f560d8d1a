// Weight / bias gradient of a tall linear layer on gfx950:  dW[out,in] = dY[M,out]^T X[M,in],  db[out] = colsum(dY)
// with M ~ 1e5 tokens and out,in <= 512 (projections / FFN of an SRA encoder layer,
// mmdet3d/models/sst/sst_basic_block_v2.py:104-126; VFE / SIR linears).
//
// Why a dedicated kernel: the output is tiny and the reduction is M long.  The library's single-pass GEMM uses
// ~48 workgroups (433 us for 384x128 at M = 90k, 20 TF/s), its batched split-K form 105-132 us
// (tools/microbench.py gemm).  Here: split-K over M into S slices so that ~768 workgroups fill the 256 CUs;
// both operands are K-major in memory (row = token), which is exactly the operand layout of
// v_mfma_f32_32x32x2_f32 (lane l: A[i = l&31][k = l>>5]) — fragments are loaded straight from global memory
// as coalesced 128 B row segments, no LDS, no transposes.  A wave owns a 64(out) x 32(in) tile (32 accumulator
// VGPRs), a workgroup 128 x 64.  The bias gradient falls out of the A fragments for free.  Partials are
// reduced by a second tiny kernel (deterministic, no float atomics).
#include <stdlib.h>
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kWgTileO = 128, kWgTileI = 64;

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

template <int U>  // k-steps (of 2 rows) per unrolled group
__global__ __launch_bounds__(256) void wgrad_k(const float* __restrict__ dy, const float* __restrict__ x, int64_t m,
                                               int out, int in, int64_t ld_dy, int64_t ld_x, int64_t rows_per_split,
                                               float* __restrict__ part_w, float* __restrict__ part_b) {
  const int o_tiles = (out + kWgTileO - 1) / kWgTileO;
  const int i_tiles = (in + kWgTileI - 1) / kWgTileI;
  const int tiles = o_tiles * i_tiles;
  const int s = blockIdx.x / tiles;
  const int tt = blockIdx.x - s * tiles;
  const int ot = tt / i_tiles, it = tt - ot * i_tiles;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wo = wave >> 1, wi = wave & 1;
  const int o0 = ot * kWgTileO + wo * 64;
  const int i0 = it * kWgTileI + wi * 32;
  if (o0 >= out || i0 >= in) return;  // no barriers in this kernel: waves may leave independently
  const bool has_o1 = (o0 + 32) < out;
  const int col = lane & 31, kk = lane >> 5;
  const int64_t k0 = (int64_t)s * rows_per_split;
  const int64_t k1 = k0 + rows_per_split < m ? k0 + rows_per_split : m;

  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
  float bs0 = 0.f, bs1 = 0.f;
  const float* pa = dy + (k0 + kk) * ld_dy + o0 + col;
  const float* pb = x + (k0 + kk) * ld_x + i0 + col;
  // software pipeline: the loads of group g+1 are in flight while the MFMAs of group g issue
  auto load = [&](float (&a0)[U], float (&a1)[U], float (&b)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      a0[u] = pa[(int64_t)u * 2 * ld_dy];
      a1[u] = has_o1 ? pa[(int64_t)u * 2 * ld_dy + 32] : 0.f;
      b[u] = pb[(int64_t)u * 2 * ld_x];
    }
    pa += (int64_t)2 * U * ld_dy;
    pb += (int64_t)2 * U * ld_x;
  };
  auto comp = [&](const float (&a0)[U], const float (&a1)[U], const float (&b)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      acc0 = mfma32(a0[u], b[u], acc0);
      acc1 = mfma32(a1[u], b[u], acc1);
      bs0 += a0[u];
      bs1 += a1[u];
    }
  };
  const int64_t ng = (k1 - k0) / (2 * U);
  float A0[U], A1[U], B0[U], C0[U], C1[U], D0[U];
  if (ng > 0) load(A0, A1, B0);
  int64_t gi = 0;
  while (gi < ng) {
    bool more = gi + 1 < ng;
    if (more) load(C0, C1, D0);
    comp(A0, A1, B0);
    ++gi;
    if (!more) break;
    more = gi + 1 < ng;
    if (more) load(A0, A1, B0);
    comp(C0, C1, D0);
    ++gi;
  }
  int64_t k = k0 + ng * 2 * U;
  for (; k < k1; k += 2) {  // ragged tail, row-guarded
    const bool ok = (k + kk) < k1;
    const float a0 = ok ? pa[0] : 0.f;
    const float a1 = (ok && has_o1) ? pa[32] : 0.f;
    const float b = ok ? pb[0] : 0.f;
    acc0 = mfma32(a0, b, acc0);
    acc1 = mfma32(a1, b, acc1);
    bs0 += a0;
    bs1 += a1;
    pa += 2 * ld_dy;
    pb += 2 * ld_x;
  }
  // C/D layout of 32x32: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  const int64_t pstride = (int64_t)out * in + (part_b != nullptr ? out : 0);  // per-split record: [dW | db]
  float* pw = part_w + (int64_t)s * pstride;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * kk;
    pw[(int64_t)(o0 + row) * in + i0 + col] = acc0[r];
    if (has_o1) pw[(int64_t)(o0 + 32 + row) * in + i0 + col] = acc1[r];
  }
  if (part_b != nullptr && it == 0 && wi == 0) {
    bs0 += __shfl_xor(bs0, 32, 64);
    bs1 += __shfl_xor(bs1, 32, 64);
    if (lane < 32) {
      part_b[(int64_t)s * pstride + o0 + lane] = bs0;
      if (has_o1) part_b[(int64_t)s * pstride + o0 + 32 + lane] = bs1;
    }
  }
}

// record e of every split summed: e < n_w -> dw[e], else db[e - n_w]
__global__ __launch_bounds__(256) void wgrad_reduce_k(const float* __restrict__ part, int nsplit, int64_t n_w,
                                                      int64_t n_b, float* __restrict__ dw, float* __restrict__ db) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t n = n_w + n_b;
  if (e >= n) return;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int s = 0;
  for (; s + 4 <= nsplit; s += 4) {
    a0 += part[(int64_t)s * n + e];
    a1 += part[(int64_t)(s + 1) * n + e];
    a2 += part[(int64_t)(s + 2) * n + e];
    a3 += part[(int64_t)(s + 3) * n + e];
  }
  for (; s < nsplit; ++s) a0 += part[(int64_t)s * n + e];
  const float r = (a0 + a1) + (a2 + a3);
  if (e < n_w)
    dw[e] = r;
  else
    db[e - n_w] = r;
}

int pick_splits(int64_t m, int out, int in, int64_t* rows_per_split) {
  const int tiles = ((out + kWgTileO - 1) / kWgTileO) * ((in + kWgTileI - 1) / kWgTileI);
  static int target = 0;
  if (target == 0) {
    const char* e = getenv("SST_WGRAD_WGS");
    target = e ? atoi(e) : 256;
    if (target < 1) target = 256;
  }
  int s = target / tiles;
  if (s < 1) s = 1;
  if (s > 512) s = 512;
  int64_t rps = sst_div_up(m > 0 ? m : 1, s);
  rps = sst_align_up(rps, 16);
  if (rps < 64) rps = 64;
  *rows_per_split = rps;
  return (int)sst_div_up(m > 0 ? m : 1, rps);
}

}  // namespace

extern "C" {

int64_t sst_weight_grad_workspace_bytes(int64_t m, int out, int in) {
  int64_t rps;
  const int s = pick_splits(m, out, in, &rps);
  return sst_align_up((int64_t)s * ((int64_t)out * in + out) * sizeof(float), 256) + 256;
}

int sst_weight_grad_f32(const float* d_dy, const float* d_x, int64_t m, int out, int in, int64_t ld_dy, int64_t ld_x,
                        float* d_dw, float* d_db, void* d_workspace, void* stream) {
  if (m < 0 || out < 32 || in < 32 || (out & 31) || (in & 31) || out > 4096 || in > 4096) return SST_ERR_UNSUPPORTED;
  if (!d_dw || !d_workspace || ld_dy < out || ld_x < in) return SST_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (m == 0) {
    SST_HIP(hipMemsetAsync(d_dw, 0, sizeof(float) * out * in, st));
    if (d_db) SST_HIP(hipMemsetAsync(d_db, 0, sizeof(float) * out, st));
    return SST_OK;
  }
  if (!d_dy || !d_x) return SST_ERR_ARG;
  int64_t rps;
  const int s = pick_splits(m, out, in, &rps);
  const int tiles = ((out + kWgTileO - 1) / kWgTileO) * ((in + kWgTileI - 1) / kWgTileI);
  const int64_t nw = (int64_t)out * in;
  float* part_w = (float*)d_workspace;
  float* part_b = d_db ? part_w + nw : nullptr;  // bias sums sit behind the dW block of each split record
  hipLaunchKernelGGL(wgrad_k<8>, dim3((unsigned)(s * tiles)), dim3(256), 0, st, d_dy, d_x, m, out, in, ld_dy, ld_x, rps,
                     part_w, part_b);
  const int64_t nb = d_db ? out : 0;
  hipLaunchKernelGGL(wgrad_reduce_k, dim3((unsigned)sst_div_up(nw + nb, 256)), dim3(256), 0, st, part_w, s, nw, nb, d_dw,
                     d_db);
  SST_LAUNCH_CHECK();
  return SST_OK;
}

}  // extern "C"
