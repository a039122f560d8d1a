// (§8 f4, first part) sparse 3-D convolution: rulebook construction and the convolution itself.
//
// Replaces, for 3-D int32 indices, the vendored spconv of the reference:
//   * getIndicePair<3> (mmdet3d/ops/spconv/include/spconv/spconv_ops.h:26-150) with its functors
//     CreateSubMIndicePairFunctor / CreateConvIndicePairFunctorP1/P2 (src/indice_cuda.cu, include/spconv/indice.cu.h)
//     and the position / kernel-offset arithmetic of include/spconv/geometry.h:24-151;
//   * indiceConv / indiceConvBackward (spconv_ops.h:256-446): per kernel offset gather -> mm -> scatter-add.
//
// Design.  A sparse convolution maps every output row i and kernel offset k to AT MOST ONE input row (and every
// input row j and offset k to at most one output row).  The rulebook is therefore stored as two dense int32 maps,
//   out2in[k][i] = j or -1        in2out[k][j] = i or -1,
// next to the reference's pair lists (indice_pairs [K, 2, N], indice_num [K]; pairs of an offset ordered by input
// row, the order of the reference's CPU path -- its GPU path leaves it to atomics).  With the maps the convolution is
// ONE output-stationary kernel, Y[r] = sum_k X[map[k][r]] W[k]: a tall GEMM whose A rows are gathered on the fly,
// fp32 MFMA (v_mfma_f32_32x32x2_f32), no atomics, no [nnz, C] gather / scatter buffers, deterministic.  The same
// kernel is the data gradient (map = in2out, W transposed) and the inverse convolution (map = in2out of the coupled
// convolution).  Offsets for which none of a workgroup's 64 rows has a neighbour are skipped.  The weight gradient
// walks the pair lists (exactly the non-zero work), split over the pairs, partial sums reduced in a fixed order.
//
// Output rows of a regular (strided / transposed) convolution are numbered by ascending (batch, z, y, x), the order
// of the reference's GPU path (torch::_unique of the linear indices, spconv_ops.h:128); they come from the
// sorted-unique of sort_scan.hip applied to the candidate positions produced here.
#include <stdlib.h>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct SpGeom {
  int in_shape[3], out_shape[3], ks[3], st[3], pd[3], dl[3];
  int kvol;
  int batch;   // dense-grid builders only: number of samples the cell grid was sized for
};

// kernel offset k -> (kz, ky, kx), row-major as geometry.h:66-70 composes `offset`
__device__ __forceinline__ void sp_koff(const SpGeom& g, int k, int (&ko)[3]) {
  ko[2] = k % g.ks[2];
  k /= g.ks[2];
  ko[1] = k % g.ks[1];
  ko[0] = k / g.ks[1];
}

// Candidate output position of input voxel `in` under offset ko.  Regular: in = out * st - pd + ko * dl
// (geometry.h:41-47 enumerates the same set); transposed: out = in * st - pd + ko * dl (geometry.h:101-104).
__device__ __forceinline__ bool sp_out_pos(const SpGeom& g, const int* in, const int (&ko)[3], bool transpose,
                                           int (&out)[3]) {
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    int v;
    if (transpose) {
      v = in[d] * g.st[d] - g.pd[d] + ko[d] * g.dl[d];
    } else {
      const int num = in[d] + g.pd[d] - ko[d] * g.dl[d];
      if (num < 0 || num % g.st[d] != 0) return false;
      v = num / g.st[d];
    }
    if (v < 0 || v >= g.out_shape[d]) return false;
    out[d] = v;
  }
  return true;
}

// rows [kvol * n + 1, 4]: (b, z, y, x) of the output voxel touched by (k, j), or -1 x 4; the extra last row is
// always invalid so that the sorted-unique's group 0 is always the "invalid" group.
__global__ __launch_bounds__(256) void sp_candidates_k(const int32_t* __restrict__ coors, int64_t n, SpGeom g,
                                                       int transpose, int4* __restrict__ rows) {
  const int64_t total = (int64_t)g.kvol * n;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e <= total; e += (int64_t)gridDim.x * blockDim.x) {
    int4 r = make_int4(-1, -1, -1, -1);
    if (e < total) {
      const int k = (int)(e / n);
      const int64_t j = e - (int64_t)k * n;
      const int4 c = ((const int4*)coors)[j];
      const int in[3] = {c.y, c.z, c.w};
      int ko[3], out[3];
      sp_koff(g, k, ko);
      if (sp_out_pos(g, in, ko, transpose != 0, out)) r = make_int4(c.x, out[0], out[1], out[2]);
    }
    rows[e] = r;
  }
}

// in2out[k][j] = inverse[k * n + j] - 1  (group 0 = invalid)
__global__ __launch_bounds__(256) void sp_inverse_to_map_k(const int32_t* __restrict__ inverse, int64_t total,
                                                           int32_t* __restrict__ in2out) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x)
    in2out[e] = inverse[e] - 1;
}

// Submanifold: output voxels = input voxels (same numbering).  sorted_keys[g] = 1 + linear index of the g-th voxel
// in ascending order, perm[g] = its row; binary search for the neighbour position.
__global__ __launch_bounds__(256) void sp_subm_map_k(const int32_t* __restrict__ coors, int64_t n, SpGeom g,
                                                     const uint64_t* __restrict__ sorted_keys,
                                                     const uint32_t* __restrict__ perm, int32_t* __restrict__ in2out) {
  const int64_t total = (int64_t)g.kvol * n;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(e / n);
    const int64_t j = e - (int64_t)k * n;
    const int4 c = ((const int4*)coors)[j];
    const int in[3] = {c.y, c.z, c.w};
    int ko[3], out[3];
    sp_koff(g, k, ko);
    int res = -1;
    if (sp_out_pos(g, in, ko, false, out)) {
      const uint64_t key =
          1ull + (((uint64_t)c.x * g.out_shape[0] + out[0]) * g.out_shape[1] + out[1]) * g.out_shape[2] + out[2];
      int64_t lo = 0, hi = n;
      while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (sorted_keys[mid] < key)
          lo = mid + 1;
        else
          hi = mid;
      }
      if (lo < n && sorted_keys[lo] == key) res = (int)perm[lo];
    }
    in2out[e] = res;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Rulebook through a DENSE CELL GRID (second builder; the sort-based one above stays for grids too large to hold).
// The spatial shape of a SparseConvTensor is known and small next to 288 GB of HBM (FSD: 2 x 32 x 640 x 640 cells = 105 MB
// of int32), so "which row sits at position p" is one load from a cell -> row grid instead of a binary search in sorted
// keys, and the output voxels of a strided / transposed convolution are the marked cells of the OUTPUT grid compacted by a
// scan over the cells - which numbers them by ascending (b, z, y, x), exactly like the sorted-unique of the candidate rows
// (the reference's torch::_unique of the linear indices, spconv_ops.h:128).  A submanifold rulebook is memset + 2 launches
// instead of ~20 (pack, 3 radix passes of 5 launches, head flags, scan, finish, map, invert), a strided one memset +
// mark + scan + 2 launches instead of ~35.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int64_t sp_cell(int b, const int (&p)[3], const int* shape) {
  return (((int64_t)b * shape[0] + p[0]) * shape[1] + p[1]) * shape[2] + p[2];
}

// The cell grids are sized batch x in_shape (x out_shape): a row whose sample index or coordinates lie outside would index
// past them (the sort-based builder only produced wrong keys for such a row).  Such rows take no part: no cell, no pairs
// (-1 in both maps) - ADVICE round 3.  Preconditions that stay the caller's (spconv's own): unique voxels; with duplicates
// the highest row index owns the cell (atomicMax: deterministic), the other rows still read their neighbours.
__device__ __forceinline__ bool sp_row_in_grid(const int4& c, const SpGeom& g) {
  return (unsigned)c.x < (unsigned)g.batch && (unsigned)c.y < (unsigned)g.in_shape[0] &&
         (unsigned)c.z < (unsigned)g.in_shape[1] && (unsigned)c.w < (unsigned)g.in_shape[2];
}

// grid[cell of row j] = j
__global__ __launch_bounds__(256) void sp_grid_rows_k(const int32_t* __restrict__ coors, int64_t n, SpGeom g,
                                                      int32_t* __restrict__ grid) {
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) {
    const int4 c = ((const int4*)coors)[j];
    const int p[3] = {c.y, c.z, c.w};
    if (sp_row_in_grid(c, g)) atomicMax(&grid[sp_cell(c.x, p, g.in_shape)], (int32_t)j);
  }
}

// Submanifold: in2out[k][j] = row at pos(j) + pd - ko * dl (the output position offset k sends input j to), and
// out2in[k][i] = row at pos(i) - pd + ko * dl (the input position output i reads through offset k); -1 outside / empty.
__global__ __launch_bounds__(256) void sp_grid_subm_k(const int32_t* __restrict__ coors, int64_t n, SpGeom g,
                                                      const int32_t* __restrict__ grid, int32_t* __restrict__ in2out,
                                                      int32_t* __restrict__ out2in) {
  const int64_t total = (int64_t)g.kvol * n;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(e / n);
    const int64_t j = e - (int64_t)k * n;
    const int4 c = ((const int4*)coors)[j];
    const int pos[3] = {c.y, c.z, c.w};
    int ko[3];
    sp_koff(g, k, ko);
    int fwd[3], bwd[3];
    bool okf = sp_row_in_grid(c, g), okb = okf;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      fwd[d] = pos[d] + g.pd[d] - ko[d] * g.dl[d];
      bwd[d] = pos[d] - g.pd[d] + ko[d] * g.dl[d];
      okf = okf && fwd[d] >= 0 && fwd[d] < g.out_shape[d];
      okb = okb && bwd[d] >= 0 && bwd[d] < g.in_shape[d];
    }
    in2out[e] = okf ? grid[sp_cell(c.x, fwd, g.in_shape)] : -1;
    out2in[e] = okb ? grid[sp_cell(c.x, bwd, g.in_shape)] : -1;
  }
}

// Regular / transposed convolution, pass 1: flag the output cells touched by any (offset, input row)
__global__ __launch_bounds__(256) void sp_grid_mark_k(const int32_t* __restrict__ coors, int64_t n, SpGeom g,
                                                      int transpose, int32_t* __restrict__ flags) {
  const int64_t total = (int64_t)g.kvol * n;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(e / n);
    const int64_t j = e - (int64_t)k * n;
    const int4 c = ((const int4*)coors)[j];
    const int in[3] = {c.y, c.z, c.w};
    int ko[3], out[3];
    sp_koff(g, k, ko);
    if (sp_row_in_grid(c, g) && sp_out_pos(g, in, ko, transpose != 0, out)) flags[sp_cell(c.x, out, g.out_shape)] = 1;
  }
}

// pass 2 (after the exclusive scan of the flags -> pos): coordinates of the output voxels, in ascending cell order
__global__ __launch_bounds__(256) void sp_grid_outids_k(const int32_t* __restrict__ flags, const int32_t* __restrict__ pos,
                                                        int64_t cells, SpGeom g, int32_t* __restrict__ outids) {
  const int64_t per_b = (int64_t)g.out_shape[0] * g.out_shape[1] * g.out_shape[2];
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < cells; e += (int64_t)gridDim.x * blockDim.x) {
    if (!flags[e]) continue;
    const int64_t b = e / per_b;
    int64_t r = e - b * per_b;
    const int z = (int)(r / ((int64_t)g.out_shape[1] * g.out_shape[2]));
    r -= (int64_t)z * g.out_shape[1] * g.out_shape[2];
    const int y = (int)(r / g.out_shape[2]);
    ((int4*)outids)[pos[e]] = make_int4((int)b, z, y, (int)(r - (int64_t)y * g.out_shape[2]));
  }
}

// pass 3: both maps (out2in pre-filled with -1; (k, i) has at most one input row: no conflicts)
__global__ __launch_bounds__(256) void sp_grid_conv_maps_k(const int32_t* __restrict__ coors, int64_t n, SpGeom g,
                                                           int transpose, const int32_t* __restrict__ pos, int64_t m,
                                                           int32_t* __restrict__ in2out, int32_t* __restrict__ out2in) {
  const int64_t total = (int64_t)g.kvol * n;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(e / n);
    const int64_t j = e - (int64_t)k * n;
    const int4 c = ((const int4*)coors)[j];
    const int in[3] = {c.y, c.z, c.w};
    int ko[3], out[3];
    sp_koff(g, k, ko);
    int res = -1;
    if (sp_row_in_grid(c, g) && sp_out_pos(g, in, ko, transpose != 0, out)) {
      res = pos[sp_cell(c.x, out, g.out_shape)];
      out2in[(int64_t)k * m + res] = (int32_t)j;
    }
    in2out[e] = res;
  }
}

__global__ __launch_bounds__(256) void sp_fill_i32_k(int32_t* __restrict__ p, int64_t n, int32_t v) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) p[e] = v;
}

// out2in[k][in2out[k][j]] = j
__global__ __launch_bounds__(256) void sp_invert_map_k(const int32_t* __restrict__ in2out, int kvol, int64_t n,
                                                       int64_t m, int32_t* __restrict__ out2in) {
  const int64_t total = (int64_t)kvol * n;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int i = in2out[e];
    if (i >= 0) {
      const int64_t k = e / n;
      out2in[k * m + i] = (int32_t)(e - k * n);
    }
  }
}

__global__ __launch_bounds__(256) void sp_pair_flags_k(const int32_t* __restrict__ in2out, int64_t total,
                                                       int32_t* __restrict__ flags) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x)
    flags[e] = in2out[e] >= 0 ? 1 : 0;
}

// pairs[k][0][p] = j, pairs[k][1][p] = in2out[k][j] for the p-th valid j of offset k (ascending j); num[k]
__global__ __launch_bounds__(256) void sp_pair_write_k(const int32_t* __restrict__ in2out,
                                                       const int32_t* __restrict__ pos, const int32_t* __restrict__ tot,
                                                       int kvol, int64_t n, int32_t* __restrict__ pairs,
                                                       int32_t* __restrict__ num) {
  const int64_t total = (int64_t)kvol * n;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t k = e / n;
    const int64_t j = e - k * n;
    const int i = in2out[e];
    const int base = pos[k * n];
    if (i >= 0) {
      const int p = pos[e] - base;
      pairs[(k * 2 + 0) * n + p] = (int32_t)j;
      pairs[(k * 2 + 1) * n + p] = i;
    }
    if (j == 0) num[k] = ((k + 1 < kvol) ? pos[(k + 1) * n] : tot[0]) - base;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Y[r, :] = sum_k X[map[k][r], :] @ W[k]  (+ bias).  Workgroup = 64 rows x up to 128 columns, 4 waves:
// wave w -> rows 32 * (w & 1) .. +31, column tiles {2 * (w >> 1), 2 * (w >> 1) + 1} of 32.
// Per (offset, 32-channel chunk): gathered A tile [64][32] and the weight slab B [32][128] staged in LDS, then
// 16 k-steps of v_mfma_f32_32x32x2_f32 per column tile.  trans_w: W[k] is stored [cout_of_this_op][cin_of_this_op]
// (the data gradient reads the forward weights [K, Cin, Cout] with the roles swapped).
// ---------------------------------------------------------------------------------------------------------------
constexpr int kSpRows = 64;
constexpr int kSpCols = 128;

__global__ __launch_bounds__(256) void sp_gather_gemm_k(const float* __restrict__ x, int64_t ldx,
                                                        const int32_t* __restrict__ map, int64_t m, int kvol,
                                                        const float* __restrict__ w, int cin, int cout, int trans_w,
                                                        const float* __restrict__ bias, float* __restrict__ y,
                                                        int64_t ldy) {
  __shared__ float As[kSpRows][33];
  __shared__ float Bs[32][kSpCols + 4];
  __shared__ int idx[kSpRows];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t r0 = (int64_t)blockIdx.x * kSpRows;
  const int col0 = blockIdx.y * kSpCols;  // first output column of this workgroup
  const int rhalf = wave & 1, chalf = wave >> 1;
  const int l31 = lane & 31, kk = lane >> 5;
  f32x16 acc[2];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  const bool wave_cols = col0 + chalf * 64 < cout;  // this wave has at least one live column tile
  for (int k = 0; k < kvol; ++k) {
    int v = -1;
    if (tid < kSpRows && r0 + tid < m) v = map[(int64_t)k * m + r0 + tid];
    const int any = __syncthreads_or(v >= 0);  // also orders the previous offset's readers of idx
    if (!any) continue;
    if (tid < kSpRows) idx[tid] = v;
    __syncthreads();
    // does this wave's row half have a neighbour for this offset?  (uniform per wave)
    const bool half_live = __any(idx[rhalf * 32 + l31] >= 0);
    const float* wk = w + (int64_t)k * cin * cout;
    for (int c0 = 0; c0 < cin; c0 += 32) {
      {  // A: thread -> row tid >> 2, 8 channels at (tid & 3) * 8
        const int row = tid >> 2, seg = (tid & 3) * 8;
        const int src = idx[row];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int c = c0 + seg + u;
          As[row][seg + u] = (src >= 0 && c < cin) ? x[(int64_t)src * ldx + c] : 0.f;
        }
      }
      if (!trans_w) {  // B[c][n] = W[k][c0 + c][col0 + n]: thread -> channel tid >> 3, 16 columns at (tid & 7) * 16
        const int c = tid >> 3, seg = (tid & 7) * 16;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int n = col0 + seg + u;
          Bs[c][seg + u] = (c0 + c < cin && n < cout) ? wk[(int64_t)(c0 + c) * cout + n] : 0.f;
        }
      } else {  // W[k] stored [cout][cin]: thread -> column tid >> 1, 16 channels at (tid & 1) * 16
        const int nloc = tid >> 1, seg = (tid & 1) * 16;
        const int n = col0 + nloc;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int c = c0 + seg + u;
          Bs[seg + u][nloc] = (c < cin && n < cout) ? wk[(int64_t)n * cin + c] : 0.f;
        }
      }
      __syncthreads();
      if (half_live && wave_cols) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
          const float a = As[rhalf * 32 + l31][2 * s + kk];
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const float b = Bs[2 * s + kk][chalf * 64 + q * 32 + l31];
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q], 0, 0, 0);
          }
        }
      }
      __syncthreads();
    }
  }
  // D layout of 32x32: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int n = col0 + chalf * 64 + q * 32 + l31;
    if (n >= cout) continue;
    const float bv = bias ? bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t row = r0 + rhalf * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
      if (row < m) y[row * ldy + n] = acc[q][r] + bv;
    }
  }
}

// The same 64 x 128 tile with the gathers prefetched (K <= 32): the partner rows of all offsets are read up front,
// offsets without any partner in the tile drop out of the stage list, and while stage i (offset, 32-channel chunk) is
// multiplied out of one LDS buffer the A rows and the W slab of stage i + 1 are already on their way into registers
// (double-buffered LDS, one barrier per stage).
__global__ __launch_bounds__(256) void sp_gather_gemm_pf_k(const float* __restrict__ x, int64_t ldx,
                                                           const int32_t* __restrict__ map, int64_t m, int kvol,
                                                           const float* __restrict__ w, int cin, int cout, int trans_w,
                                                           const float* __restrict__ bias, float* __restrict__ y,
                                                           int64_t ldy) {
  __shared__ float As[2][kSpRows][33];
  __shared__ float Bs[2][32][kSpCols + 4];
  __shared__ int idx[32][kSpRows];
  __shared__ unsigned live_s[3];  // offsets with a partner in: the tile, rows 0..31, rows 32..63
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t r0 = (int64_t)blockIdx.x * kSpRows;
  const int col0 = blockIdx.y * kSpCols;
  const int rhalf = wave & 1, chalf = wave >> 1;
  const int l31 = lane & 31, kk = lane >> 5;
  if (wave == 0) {
    int v[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) v[k] = (k < kvol && r0 + lane < m) ? map[(int64_t)k * m + r0 + lane] : -1;
    unsigned any = 0, lo = 0, hi = 0;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      idx[k][lane] = v[k];
      const unsigned long long bal = __ballot(v[k] >= 0);
      any |= (bal != 0 ? 1u : 0u) << k;
      lo |= ((bal & 0xffffffffull) != 0 ? 1u : 0u) << k;
      hi |= ((bal >> 32) != 0 ? 1u : 0u) << k;
    }
    if (lane == 0) {
      live_s[0] = any;
      live_s[1] = lo;
      live_s[2] = hi;
    }
  }
  __syncthreads();
  const unsigned live = live_s[0], half_live = live_s[1 + rhalf];
  const bool wave_cols = col0 + chalf * 64 < cout;
  f32x16 acc[2];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  const int n_chunk = (cin + 31) / 32;
  const int arow = tid >> 2, aseg = (tid & 3) * 8;  // A staging: row, 8 channels
  float ra[8], rbv[16];
  auto fetch = [&](int k, int c0) {
    const int src = idx[k][arow];
    const float* px = x + (int64_t)(src >= 0 ? src : 0) * ldx + c0 + aseg;
#pragma unroll
    for (int u = 0; u < 8; ++u) ra[u] = (src >= 0 && c0 + aseg + u < cin) ? px[u] : 0.f;
    const float* wk = w + (int64_t)k * cin * cout;
    if (!trans_w) {  // B[c][n] = W[k][c0 + c][col0 + n]: thread -> channel tid >> 3, 16 columns at (tid & 7) * 16
      const int c = c0 + (tid >> 3), seg = col0 + (tid & 7) * 16;
#pragma unroll
      for (int u = 0; u < 16; ++u) rbv[u] = (c < cin && seg + u < cout) ? wk[(int64_t)c * cout + seg + u] : 0.f;
    } else {  // W[k] stored [cout][cin]: thread -> column tid >> 1, 16 channels at (tid & 1) * 16
      const int n = col0 + (tid >> 1), seg = c0 + (tid & 1) * 16;
#pragma unroll
      for (int u = 0; u < 16; ++u) rbv[u] = (seg + u < cin && n < cout) ? wk[(int64_t)n * cin + seg + u] : 0.f;
    }
  };
  auto next_live = [&](int k) {  // first populated offset >= k (32 if none)
    const unsigned rest = k < 32 ? (live >> k) : 0u;
    return rest ? k + __builtin_ctz(rest) : 32;
  };
  int k = next_live(0), ci = 0, buf = 0;
  if (k < 32) fetch(k, 0);
  while (k < 32) {
    int nk = k, nci = ci + 1;
    if (nci == n_chunk) {
      nci = 0;
      nk = next_live(k + 1);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) As[buf][arow][aseg + u] = ra[u];
    if (!trans_w) {
#pragma unroll
      for (int u = 0; u < 16; ++u) Bs[buf][tid >> 3][(tid & 7) * 16 + u] = rbv[u];
    } else {
#pragma unroll
      for (int u = 0; u < 16; ++u) Bs[buf][(tid & 1) * 16 + u][tid >> 1] = rbv[u];
    }
    if (nk < 32) fetch(nk, nci * 32);  // in flight during the MFMAs below
    __syncthreads();
    if (((half_live >> k) & 1) && wave_cols) {
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const float a = As[buf][rhalf * 32 + l31][2 * s + kk];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const float b = Bs[buf][2 * s + kk][chalf * 64 + q * 32 + l31];
          acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q], 0, 0, 0);
        }
      }
    }
    k = nk;
    ci = nci;
    buf ^= 1;
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int n = col0 + chalf * 64 + q * 32 + l31;
    if (n >= cout) continue;
    const float bv = bias ? bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t row = r0 + rhalf * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
      if (row < m) y[row * ldy + n] = acc[q][r] + bv;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Second form of the same contraction, used when W is stored [cin, cout] and K <= 32: the MFMAs run over COMPACTED
// rows and the gathers are prefetched.  A workgroup owns a segment of 128 output rows x 64 output columns whose sums
// live in LDS (ytile).  First the rows of the segment that have a partner are listed per kernel offset (ballots).
// Then the workgroup walks the stages (offset, 32 listed rows, 64 input channels) in a fixed order: while the
// gathered rows of stage i (in LDS) are multiplied with W[k] by v_mfma_f32_16x16x4_f32 -- wave w owns output columns
// 16 w .. 16 w + 15, fragments of W straight from global memory / L1 -- the rows of stage i + 1 are already on their
// way into registers (double-buffered LDS, one barrier per stage).  Products are added to the rows' sums with
// ds_add_f32: a wave is the only writer of its columns and the order is fixed, so the result is deterministic.
// With LiDAR-like sparsity (4 of 27 offsets populated) this issues ~1/4 of the MFMA work of sp_gather_gemm_k,
// which multiplies every row of a tile with every offset that has at least one partner in the tile.
// ---------------------------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kSegRows = 64, kSegCols = 64, kSegStage = 32, kSegChunk = 64, kSegMaxK = 32;
static_assert(kSegRows == 64 || kSegRows == 128, "row listing below handles one or two row waves");

template <bool VEC>
__global__ __launch_bounds__(256) void sp_conv_seg_k(const float* __restrict__ x, int64_t ldx,
                                                     const int32_t* __restrict__ map, int64_t m, int kvol,
                                                     const float* __restrict__ w, int cin, int cout,
                                                     const float* __restrict__ bias, float* __restrict__ y,
                                                     int64_t ldy) {
  __shared__ float ytile[kSegRows][kSegCols];
  __shared__ __attribute__((aligned(16))) float abuf[2][kSegStage][kSegChunk + 4];
  __shared__ int lsrc[kSegMaxK][kSegRows];
  __shared__ unsigned char lrow[kSegMaxK][kSegRows];
  __shared__ int cnt_s[kSegMaxK];
  __shared__ int wave_cnt[kSegMaxK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t r0 = (int64_t)blockIdx.x * kSegRows;
  const int col0 = blockIdx.y * kSegCols;
  const int l15 = lane & 15, kq = lane >> 4;
  const int ncol = col0 + wave * 16 + l15;  // this lane's output column
  const bool col_ok = ncol < cout;
  for (int e = tid; e < kSegRows * kSegCols; e += 256) (&ytile[0][0])[e] = 0.f;
  // ---- per-offset lists of the rows that have a partner (waves 0 / 1 hold rows 0..63 / 64..127) ----
  int v[kSegMaxK];  // all offsets' partners of this thread's row: independent loads, issued together
  constexpr int kRowWaves = kSegRows / 64;  // waves that hold rows (wave w: rows 64 w .. 64 w + 63)
#pragma unroll
  for (int k = 0; k < kSegMaxK; ++k)
    v[k] = (wave < kRowWaves && k < kvol && r0 + tid < m) ? map[(int64_t)k * m + r0 + tid] : -1;
  if (kRowWaves > 1) {
    if (wave == 0) {
#pragma unroll
      for (int k = 0; k < kSegMaxK; ++k) {
        const unsigned long long bal = __ballot(v[k] >= 0);
        if (lane == 0) wave_cnt[k] = __popcll(bal);
      }
    }
    __syncthreads();
  }
  if (wave < kRowWaves) {  // wave 1 appends behind wave 0
#pragma unroll
    for (int k = 0; k < kSegMaxK; ++k) {
      if (k < kvol) {  // uniform
        const unsigned long long bal = __ballot(v[k] >= 0);
        const int base = wave == 1 ? wave_cnt[k] : 0;
        if (wave == kRowWaves - 1 && lane == 0) cnt_s[k] = base + __popcll(bal);
        if (v[k] >= 0) {
          const int p = base + __popcll(bal & ((1ull << lane) - 1ull));
          lrow[k][p] = (unsigned char)tid;
          lsrc[k][p] = v[k];
        }
      }
    }
  }
  __syncthreads();
  // ---- stages ----
  const int n_chunk = (cin + kSegChunk - 1) / kSegChunk;
  const int srow = tid >> 3, sseg = (tid & 7) * 8;  // staging: listed row vb + srow, 8 channels at c0 + sseg
  float reg[8], bnext[kSegChunk / 4];
  auto fetch = [&](int k, int vb, int c0) {
    const float* wk = w + (int64_t)k * cin * cout;
#pragma unroll
    for (int s = 0; s < kSegChunk / 4; ++s) {  // this lane's W fragments of the stage: 16 independent loads
      const int ch = c0 + 4 * s + kq;
      bnext[s] = (col_ok && ch < cin) ? wk[(int64_t)ch * cout + ncol] : 0.f;
    }
    const int e = vb + srow;
    const int src = e < cnt_s[k] ? lsrc[k][e] : -1;
    const float* px = x + (int64_t)(src >= 0 ? src : 0) * ldx + c0 + sseg;
    if (VEC) {
      float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = lo;
      if (src >= 0 && c0 + sseg < cin) lo = *(const float4*)px;
      if (src >= 0 && c0 + sseg + 4 < cin) hi = *(const float4*)(px + 4);
      reg[0] = lo.x, reg[1] = lo.y, reg[2] = lo.z, reg[3] = lo.w;
      reg[4] = hi.x, reg[5] = hi.y, reg[6] = hi.z, reg[7] = hi.w;
    } else {
#pragma unroll
      for (int u = 0; u < 8; ++u) reg[u] = (src >= 0 && c0 + sseg + u < cin) ? px[u] : 0.f;
    }
  };
  // first stage
  int k = 0;
  while (k < kvol && cnt_s[k] == 0) ++k;
  int vb = 0, ci = 0, buf = 0;
  if (k < kvol) fetch(k, 0, 0);
  f32x4 acc[2];
  acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
  acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
  while (k < kvol) {
    // next stage in (k, vb, chunk) order
    int nk = k, nvb = vb, nci = ci + 1;
    if (nci == n_chunk) {
      nci = 0;
      nvb += kSegStage;
      if (nvb >= cnt_s[k]) {
        nvb = 0;
        ++nk;
        while (nk < kvol && cnt_s[nk] == 0) ++nk;
      }
    }
    float4* dst = (float4*)&abuf[buf][srow][sseg];
    dst[0] = make_float4(reg[0], reg[1], reg[2], reg[3]);
    dst[1] = make_float4(reg[4], reg[5], reg[6], reg[7]);
    float bcur[kSegChunk / 4];
#pragma unroll
    for (int s = 0; s < kSegChunk / 4; ++s) bcur[s] = bnext[s];
    if (nk < kvol) fetch(nk, nvb, nci * kSegChunk);  // in flight during the MFMAs below
    __syncthreads();
    const int cnt = cnt_s[k];
    const int c0 = ci * kSegChunk;
    // the two row blocks are independent accumulation chains: interleaved, so that a dependent MFMA never waits for
    // its predecessor (a second block without listed rows multiplies zeros: cheaper than a branch per step)
#pragma unroll
    for (int s = 0; s < kSegChunk / 4; ++s) {
      if (c0 + 4 * s < cin) {  // uniform
        const float a0 = abuf[buf][l15][4 * s + kq];
        const float a1 = abuf[buf][16 + l15][4 * s + kq];
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bcur[s], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bcur[s], acc[1], 0, 0, 0);
      }
    }
    if (ci == n_chunk - 1) {  // all channels of this row block done: D of 16x16: column = lane & 15, rows 4 * (lane >> 4) + r
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int e = vb + rb * 16 + 4 * kq + r;
          // plain read-modify-write (a float atomicAdd into LDS until round 5): a wave owns its 16 columns of the tile and the
          // rows lrow[k][.] of one offset are distinct, so no two lanes of the workgroup ever meet on an element; successive
          // offsets accumulate in program order - deterministic
          if (e < cnt) ytile[lrow[k][e]][wave * 16 + l15] += acc[rb][r];
        }
        acc[rb] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    }
    k = nk;
    vb = nvb;
    ci = nci;
    buf ^= 1;
  }
  __syncthreads();
  for (int e = tid; e < kSegRows * kSegCols; e += 256) {
    const int row = e / kSegCols, c = e - row * kSegCols;
    if (r0 + row < m && col0 + c < cout) y[(r0 + row) * ldy + col0 + c] = ytile[row][c] + (bias ? bias[col0 + c] : 0.f);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// dW[k] (cin x cout) = sum over the pairs p of offset k of  X[pa[p], :]^T  dY[pb[p], :].
// The pairs of all offsets are cut into chunks of kSpChunk pairs (an offset with num[k] pairs owns
// ceil(num[k] / kSpChunk) consecutive chunks; every workgroup finds its (offset, chunk) by walking num[], K <= 4096),
// so the load is balanced whatever the spread of the pair counts and there are enough workgroups to overlap their gather latencies (the centre offset of a submanifold convolution
// holds every voxel).  grid = (chunks upper bound, strips / 4): a wave owns a strip = 32 input channels x up to 128
// output columns (4 MFMA tiles); two pairs per v_mfma_f32_32x32x2_f32 step, four steps in flight (indices, then
// fragments, then MFMAs); fragments straight from global memory (a half-wave reads 128 contiguous bytes of one row).
// Partials [chunk][cin][cout] are summed per offset, in chunk order, by sp_wgrad_reduce_k.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kSpChunk = 512;  // small chunks: many workgroups in flight hide the gather latency of each one

__device__ __forceinline__ bool sp_find_chunk(const int32_t* __restrict__ num, int kvol, int chunk, int& k, int& first) {
  int c0 = 0;
  for (int i = 0; i < kvol; ++i) {
    const int nc = (num[i] + kSpChunk - 1) / kSpChunk;
    if (chunk < c0 + nc) {
      k = i;
      first = c0;
      return true;
    }
    c0 += nc;
  }
  return false;
}

// NQ = column tiles of 32 per strip (2: layers with up to 64 output columns, where a 128-column strip would multiply
// zeros half of the time; 4 otherwise).  Software pipeline: the pair indices of step i + 2 and the rows of step i + 1
// are in flight while step i is multiplied; every load is unconditional (clamped pair / channel index, zeroed by a
// select on the A side), so the s_waitcnt counters the compiler derives stay exact.
template <int NQ>
__global__ __launch_bounds__(256) void sp_wgrad_k(const float* __restrict__ x, int64_t ldx,
                                                  const float* __restrict__ dy, int64_t lddy,
                                                  const int32_t* __restrict__ pairs, int64_t pair_ld, int x_side,
                                                  const int32_t* __restrict__ num, int kvol, int cin, int cout,
                                                  float* __restrict__ part) {
  int k, first;
  if (!sp_find_chunk(num, kvol, blockIdx.x, k, first)) return;  // uniform
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, kk = lane >> 5;
  const int n_cog = (cout + 32 * NQ - 1) / (32 * NQ);  // column groups of 32 NQ
  const int strip = blockIdx.y * 4 + wave;
  const int n_ci = (cin + 31) / 32;
  const int ci = strip / n_cog, cog = strip - ci * n_cog;
  if (ci >= n_ci) return;
  const int np = num[k];
  const int p0 = (blockIdx.x - first) * kSpChunk;
  const int p1 = p0 + kSpChunk < np ? p0 + kSpChunk : np;
  if (p0 >= p1) return;
  const int32_t* pa = pairs + ((int64_t)k * 2 + x_side) * pair_ld;
  const int32_t* pb = pairs + ((int64_t)k * 2 + (1 - x_side)) * pair_ld;
  f32x16 acc[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  const int ca = ci * 32 + l31;
  const bool a_ok = ca < cin;
  const int ca_ld = a_ok ? ca : cin - 1;
  bool b_ok[NQ];
  int cb[NQ], cb_ld[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    cb[q] = cog * 32 * NQ + q * 32 + l31;
    b_ok[q] = cb[q] < cout;
    cb_ld[q] = b_ok[q] ? cb[q] : cout - 1;
  }
  int ia[4], ib[4];
  float a[4], b[4][NQ];
  auto load_idx = [&](int p) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int pp = p + 2 * u + kk;
      pp = pp < p1 ? pp : p1 - 1;
      ia[u] = pa[pp];
      ib[u] = pb[pp];
    }
  };
  auto load_rows = [&]() {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a[u] = x[(int64_t)ia[u] * ldx + ca_ld];
#pragma unroll
      for (int q = 0; q < NQ; ++q) b[u][q] = dy[(int64_t)ib[u] * lddy + cb_ld[q]];
    }
  };
  load_idx(p0);
  load_rows();
  load_idx(p0 + 8);
  for (int p = p0; p < p1; p += 8) {
    float ac[4], bc[4][NQ];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      ac[u] = (a_ok && p + 2 * u + kk < p1) ? a[u] : 0.f;
#pragma unroll
      for (int q = 0; q < NQ; ++q) bc[u][q] = b[u][q];
    }
    load_rows();       // rows of step p + 8 (their indices were requested a step ago)
    load_idx(p + 16);  // indices of step p + 16
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int q = 0; q < NQ; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[u], bc[u][q], acc[q], 0, 0, 0);
  }
  float* dst = part + (int64_t)blockIdx.x * cin * cout;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    if (!b_ok[q]) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = ci * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
      if (c < cin) dst[(int64_t)c * cout + cb[q]] = acc[q][r];
    }
  }
}

// dw[k][e] = sum over the chunks of offset k, in chunk order
__global__ __launch_bounds__(256) void sp_wgrad_reduce_k(const float* __restrict__ part, const int32_t* __restrict__ num,
                                                         int kvol, int64_t per_k, float* __restrict__ dw) {
  const int k = blockIdx.y;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= per_k) return;
  int first = 0;
  for (int i = 0; i < k; ++i) first += (num[i] + kSpChunk - 1) / kSpChunk;
  const int nc = (num[k] + kSpChunk - 1) / kSpChunk;
  float s = 0.f;
  for (int c = 0; c < nc; ++c) s += part[(int64_t)(first + c) * per_k + e];
  dw[(int64_t)k * per_k + e] = s;
}

// upper bound on the number of chunks: sum_k ceil(num[k] / chunk) <= total / chunk + K; without a known total every
// slot of the pair lists counts
int64_t sp_wgrad_chunks(int kvol, int64_t pair_ld, int64_t total_pairs) {
  const int64_t total = (total_pairs >= 0 && total_pairs <= (int64_t)kvol * pair_ld) ? total_pairs : (int64_t)kvol * pair_ld;
  return total / kSpChunk + kvol;
}

// ---------------------------------------------------------------------------------------------------------------
// Sparse max pooling (indiceMaxPool / indiceMaxPoolBackward, include/spconv/pool_ops.h:24-97, src/maxpool.cc:22-63) on
// the dense maps: out[i][c] = max(0, max_k in[out2in[k][i]][c]) -- the reference starts from a zero-filled output, so
// negative maxima clip to 0 -- and din[j][c] = sum over the outputs i = in2out[k][j] with out[i][c] == in[j][c] of
// dout[i][c] (every input that equals the maximum receives the gradient, as in the reference).  Gathers, no atomics.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sp_maxpool_fwd_k(const float* __restrict__ x, int64_t ldx,
                                                        const int32_t* __restrict__ out2in, int64_t m, int kvol, int c,
                                                        float* __restrict__ y, int64_t ldy) {
  const int64_t total = m * c;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e / c;
    const int ch = (int)(e - i * c);
    float best = 0.f;
    for (int k = 0; k < kvol; ++k) {
      const int j = out2in[(int64_t)k * m + i];
      if (j >= 0) {
        const float v = x[(int64_t)j * ldx + ch];
        if (best < v) best = v;
      }
    }
    y[i * ldy + ch] = best;
  }
}

__global__ __launch_bounds__(256) void sp_maxpool_bwd_k(const float* __restrict__ x, int64_t ldx,
                                                        const float* __restrict__ y, int64_t ldy,
                                                        const float* __restrict__ dy, int64_t lddy,
                                                        const int32_t* __restrict__ in2out, int64_t n, int kvol, int c,
                                                        float* __restrict__ dx, int64_t lddx) {
  const int64_t total = n * c;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t j = e / c;
    const int ch = (int)(e - j * c);
    const float v = x[j * ldx + ch];
    float g = 0.f;
    for (int k = 0; k < kvol; ++k) {
      const int i = in2out[(int64_t)k * n + j];
      if (i >= 0 && y[(int64_t)i * ldy + ch] == v) g += dy[(int64_t)i * lddy + ch];
    }
    dx[j * lddx + ch] = g;
  }
}

bool sp_geom(const int32_t* in_shape, const int32_t* out_shape, const int32_t* ks, const int32_t* st,
             const int32_t* pd, const int32_t* dl, SpGeom* g) {
  if (!in_shape || !out_shape || !ks || !st || !pd || !dl) return false;
  g->kvol = 1;
  g->batch = 1 << 30;   // set by the dense-grid entry points
  for (int d = 0; d < 3; ++d) {
    g->in_shape[d] = in_shape[d];
    g->out_shape[d] = out_shape[d];
    g->ks[d] = ks[d];
    g->st[d] = st[d];
    g->pd[d] = pd[d];
    g->dl[d] = dl[d];
    if (ks[d] < 1 || st[d] < 1 || dl[d] < 1 || pd[d] < 0 || in_shape[d] < 1 || out_shape[d] < 1) return false;
    g->kvol *= ks[d];
  }
  return g->kvol <= 4096;  // spconv_ops.h:49
}

}  // namespace

extern "C" {

int sst_spconv_candidates_i32(const int32_t* d_coors, int64_t n, const int32_t* in_shape, const int32_t* out_shape,
                              const int32_t* ksize, const int32_t* stride, const int32_t* padding,
                              const int32_t* dilation, int transpose, int32_t* d_rows, void* stream) {
  SpGeom g;
  if (n < 0 || !sp_geom(in_shape, out_shape, ksize, stride, padding, dilation, &g)) return SST_ERR_ARG;
  if ((int64_t)g.kvol * n > (1ll << 31) - 2) return SST_ERR_UNSUPPORTED;
  if (!d_rows || (n > 0 && !d_coors) || (((uintptr_t)d_coors | (uintptr_t)d_rows) & 15)) return SST_ERR_ARG;
  const int64_t total = (int64_t)g.kvol * n + 1;
  hipLaunchKernelGGL(sp_candidates_k, dim3(sst_grid_1d(total, 256)), dim3(256), 0, (hipStream_t)stream, d_coors, n, g,
                     transpose, (int4*)d_rows);
  SST_LAUNCH_CHECK();
  return SST_OK;
}

/* Dense-grid rulebooks (see the kernels).  Cells = batch x shape[0] x shape[1] x shape[2] of the INPUT shape (submanifold)
 * or of the OUTPUT shape (regular / transposed); the caller provides the cell arrays. */
int sst_spconv_grid_subm_i32(const int32_t* d_coors, int64_t n, int batch, const int32_t* shape, const int32_t* ksize,
                             const int32_t* dilation, int32_t* d_grid, int32_t* d_in2out, int32_t* d_out2in,
                             void* stream) {
  int32_t st[3] = {1, 1, 1}, pd[3];
  if (!ksize || batch < 1) return SST_ERR_ARG;
  for (int d = 0; d < 3; ++d) pd[d] = ksize[d] / 2;   // spconv_ops.h:74-77
  SpGeom g;
  if (n < 0 || !sp_geom(shape, shape, ksize, st, pd, dilation, &g)) return SST_ERR_ARG;
  if ((int64_t)g.kvol * n > (1ll << 31) - 2) return SST_ERR_UNSUPPORTED;
  g.batch = batch;
  if (n == 0) return SST_OK;
  if (!d_coors || !d_grid || !d_in2out || !d_out2in || (((uintptr_t)d_coors) & 15)) return SST_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int64_t cells = (int64_t)batch * shape[0] * shape[1] * shape[2];
  SST_HIP(hipMemsetAsync(d_grid, 0xFF, sizeof(int32_t) * cells, s));
  hipLaunchKernelGGL(sp_grid_rows_k, dim3(sst_grid_1d(n, 256)), dim3(256), 0, s, d_coors, n, g, d_grid);
  hipLaunchKernelGGL(sp_grid_subm_k, dim3(sst_grid_1d((int64_t)g.kvol * n, 256)), dim3(256), 0, s, d_coors, n, g, d_grid,
                     d_in2out, d_out2in);
  SST_LAUNCH_CHECK();
  return SST_OK;
}

int64_t sst_spconv_grid_conv_workspace_bytes(int64_t cells) {
  return 2 * sst_align_up((cells > 0 ? cells : 1) * (int64_t)sizeof(int32_t), 256) +
         sst_align_up(sst_scan_workspace_bytes(cells > 0 ? cells : 1), 256) + 256;
}

/* pass 1: marks + scan; d_num_out [1] = number of output voxels (device; the caller reads it to size outids / out2in) */
int sst_spconv_grid_conv_count_i32(const int32_t* d_coors, int64_t n, int batch, const int32_t* in_shape,
                                   const int32_t* out_shape, const int32_t* ksize, const int32_t* stride,
                                   const int32_t* padding, const int32_t* dilation, int transpose, void* d_workspace,
                                   int32_t* d_num_out, void* stream) {
  SpGeom g;
  if (n < 0 || batch < 1 || !sp_geom(in_shape, out_shape, ksize, stride, padding, dilation, &g)) return SST_ERR_ARG;
  if ((int64_t)g.kvol * n > (1ll << 31) - 2) return SST_ERR_UNSUPPORTED;
  if (!d_workspace || !d_num_out || (n > 0 && (!d_coors || (((uintptr_t)d_coors) & 15)))) return SST_ERR_ARG;
  g.batch = batch;
  hipStream_t s = (hipStream_t)stream;
  const int64_t cells = (int64_t)batch * out_shape[0] * out_shape[1] * out_shape[2];
  if (cells > (1ll << 31) - 2) return SST_ERR_UNSUPPORTED;
  const int64_t seg = sst_align_up(cells * (int64_t)sizeof(int32_t), 256);
  int32_t* flags = (int32_t*)d_workspace;
  int32_t* pos = (int32_t*)((char*)d_workspace + seg);
  void* scan_ws = (char*)d_workspace + 2 * seg;
  SST_HIP(hipMemsetAsync(flags, 0, sizeof(int32_t) * cells, s));
  if (n > 0)
    hipLaunchKernelGGL(sp_grid_mark_k, dim3(sst_grid_1d((int64_t)g.kvol * n, 256)), dim3(256), 0, s, d_coors, n, g, transpose,
                       flags);
  SST_LAUNCH_CHECK();
  return sst_exclusive_scan_i32(flags, pos, cells, d_num_out, scan_ws, stream);
}

/* pass 2: output coordinates [m, 4] and both maps; d_workspace as left by pass 1 */
int sst_spconv_grid_conv_maps_i32(const int32_t* d_coors, int64_t n, int batch, const int32_t* in_shape,
                                  const int32_t* out_shape, const int32_t* ksize, const int32_t* stride,
                                  const int32_t* padding, const int32_t* dilation, int transpose, void* d_workspace,
                                  int64_t m, int32_t* d_outids, int32_t* d_in2out, int32_t* d_out2in, void* stream) {
  SpGeom g;
  if (n < 0 || m < 0 || batch < 1 || !sp_geom(in_shape, out_shape, ksize, stride, padding, dilation, &g)) return SST_ERR_ARG;
  if (n == 0 || m == 0) return SST_OK;
  g.batch = batch;
  if (!d_workspace || !d_coors || !d_outids || !d_in2out || !d_out2in || (((uintptr_t)d_outids) & 15)) return SST_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int64_t cells = (int64_t)batch * out_shape[0] * out_shape[1] * out_shape[2];
  const int64_t seg = sst_align_up(cells * (int64_t)sizeof(int32_t), 256);
  const int32_t* flags = (const int32_t*)d_workspace;
  const int32_t* pos = (const int32_t*)((const char*)d_workspace + seg);
  SST_HIP(hipMemsetAsync(d_out2in, 0xFF, sizeof(int32_t) * g.kvol * m, s));
  hipLaunchKernelGGL(sp_grid_outids_k, dim3(sst_grid_1d(cells, 256)), dim3(256), 0, s, flags, pos, cells, g, d_outids);
  hipLaunchKernelGGL(sp_grid_conv_maps_k, dim3(sst_grid_1d((int64_t)g.kvol * n, 256)), dim3(256), 0, s, d_coors, n, g,
                     transpose, pos, m, d_in2out, d_out2in);
  SST_LAUNCH_CHECK();
  return SST_OK;
}

int sst_spconv_inverse_to_map_i32(const int32_t* d_inverse, int64_t total, int32_t* d_in2out, void* stream) {
  if (total < 0) return SST_ERR_ARG;
  if (total == 0) return SST_OK;
  if (!d_inverse || !d_in2out) return SST_ERR_ARG;
  hipLaunchKernelGGL(sp_inverse_to_map_k, dim3(sst_grid_1d(total, 256)), dim3(256), 0, (hipStream_t)stream, d_inverse,
                     total, d_in2out);
  SST_LAUNCH_CHECK();
  return SST_OK;
}

int sst_spconv_subm_map_i32(const int32_t* d_coors, int64_t n, const int32_t* shape, const int32_t* ksize,
                            const int32_t* dilation, const uint64_t* d_sorted_keys, const uint32_t* d_perm,
                            int32_t* d_in2out, void* stream) {
  // spconv_ops.h:74-77: a submanifold convolution always uses stride 1 and padding = ksize / 2
  int32_t st[3] = {1, 1, 1}, pd[3];
  if (!ksize) return SST_ERR_ARG;
  for (int d = 0; d < 3; ++d) pd[d] = ksize[d] / 2;
  SpGeom g;
  if (n < 0 || !sp_geom(shape, shape, ksize, st, pd, dilation, &g)) return SST_ERR_ARG;
  if ((int64_t)g.kvol * n > (1ll << 31) - 2) return SST_ERR_UNSUPPORTED;
  if (n == 0) return SST_OK;
  if (!d_coors || !d_sorted_keys || !d_perm || !d_in2out || (((uintptr_t)d_coors) & 15)) return SST_ERR_ARG;
  hipLaunchKernelGGL(sp_subm_map_k, dim3(sst_grid_1d((int64_t)g.kvol * n, 256)), dim3(256), 0, (hipStream_t)stream,
                     d_coors, n, g, d_sorted_keys, d_perm, d_in2out);
  SST_LAUNCH_CHECK();
  return SST_OK;
}

int sst_spconv_invert_map_i32(const int32_t* d_in2out, int kvol, int64_t n, int64_t m, int32_t* d_out2in,
                              void* stream) {
  if (kvol < 1 || n < 0 || m < 0) return SST_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (m > 0) {
    if (!d_out2in) return SST_ERR_ARG;
    hipLaunchKernelGGL(sp_fill_i32_k, dim3(sst_grid_1d((int64_t)kvol * m, 256)), dim3(256), 0, st, d_out2in,
                       (int64_t)kvol * m, -1);
  }
  if (n > 0 && m > 0) {
    if (!d_in2out) return SST_ERR_ARG;
    hipLaunchKernelGGL(sp_invert_map_k, dim3(sst_grid_1d((int64_t)kvol * n, 256)), dim3(256), 0, st, d_in2out, kvol, n,
                       m, d_out2in);
  }
  SST_LAUNCH_CHECK();
  return SST_OK;
}

int64_t sst_spconv_pair_lists_workspace_bytes(int kvol, int64_t n) {
  const int64_t total = (int64_t)(kvol > 0 ? kvol : 1) * (n > 0 ? n : 1);
  return 2 * sst_align_up(total * (int64_t)sizeof(int32_t), 256) + sst_align_up(sst_scan_workspace_bytes(total), 256) +
         512;
}

int sst_spconv_pair_lists_i32(const int32_t* d_in2out, int kvol, int64_t n, int32_t* d_pairs, int32_t* d_num,
                              void* d_workspace, void* stream) {
  if (kvol < 1 || n < 0 || !d_num) return SST_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) {
    SST_HIP(hipMemsetAsync(d_num, 0, sizeof(int32_t) * kvol, st));
    return SST_OK;
  }
  if (!d_in2out || !d_pairs || !d_workspace) return SST_ERR_ARG;
  const int64_t total = (int64_t)kvol * n;
  if (total > (1ll << 31) - 2) return SST_ERR_UNSUPPORTED;
  const int64_t seg = sst_align_up(total * (int64_t)sizeof(int32_t), 256);
  char* ws = (char*)d_workspace;
  int32_t* flags = (int32_t*)ws;
  int32_t* pos = (int32_t*)(ws + seg);
  void* scan_ws = ws + 2 * seg;
  int32_t* tot = (int32_t*)(ws + 2 * seg + sst_align_up(sst_scan_workspace_bytes(total), 256));
  const int grid = sst_grid_1d(total, 256);
  hipLaunchKernelGGL(sp_fill_i32_k, dim3(sst_grid_1d(2 * total, 256)), dim3(256), 0, st, d_pairs, 2 * total, -1);
  hipLaunchKernelGGL(sp_pair_flags_k, dim3(grid), dim3(256), 0, st, d_in2out, total, flags);
  const int rc = sst_exclusive_scan_i32(flags, pos, total, tot, scan_ws, stream);
  if (rc != SST_OK) return rc;
  hipLaunchKernelGGL(sp_pair_write_k, dim3(grid), dim3(256), 0, st, d_in2out, pos, tot, kvol, n, d_pairs, d_num);
  SST_LAUNCH_CHECK();
  return SST_OK;
}

int sst_spconv_gather_gemm_f32(const float* d_x, int64_t ldx, const int32_t* d_map, int64_t m, int kvol,
                               const float* d_w, int cin, int cout, int trans_w, const float* d_bias, float* d_y,
                               int64_t ldy, int form, void* stream) {
  if (m < 0 || kvol < 1 || cin < 1 || cout < 1 || ldx < cin || ldy < cout) return SST_ERR_ARG;
  if (m == 0) return SST_OK;
  if (!d_x || !d_map || !d_w || !d_y) return SST_ERR_ARG;
  if (sst_div_up(m, kSpRows) > 0x7fffffff || sst_div_up(cout, kSpCols) > 65535) return SST_ERR_UNSUPPORTED;
  // form: 0 = automatic (compacted rows for narrow outputs, where one 64-column group covers the layer), 1 =
  // uncompacted 64 x 128 tiles, 2 = compacted rows whenever the kernel applies.  SST_SPCONV_FORM overrides the argument.
  static int env_form = -1;
  if (env_form < 0) {
    const char* e = getenv("SST_SPCONV_FORM");
    env_form = e ? atoi(e) : 0;
  }
  if (env_form != 0) form = env_form;
  if (form == 0) form = cout <= kSegCols ? 2 : 1;
  if (!trans_w && form == 2 && kvol <= kSegMaxK) {
    const dim3 grid((unsigned)sst_div_up(m, kSegRows), (unsigned)sst_div_up(cout, kSegCols));
    const bool vec = (ldx % 4 == 0) && (cin % 4 == 0) && (((uintptr_t)d_x & 15) == 0);
    if (vec)
      hipLaunchKernelGGL(sp_conv_seg_k<true>, grid, dim3(256), 0, (hipStream_t)stream, d_x, ldx, d_map, m, kvol, d_w,
                         cin, cout, d_bias, d_y, ldy);
    else
      hipLaunchKernelGGL(sp_conv_seg_k<false>, grid, dim3(256), 0, (hipStream_t)stream, d_x, ldx, d_map, m, kvol, d_w,
                         cin, cout, d_bias, d_y, ldy);
  } else {
    const dim3 grid((unsigned)sst_div_up(m, kSpRows), (unsigned)sst_div_up(cout, kSpCols));
    if (kvol <= 32 && form != 3)
      hipLaunchKernelGGL(sp_gather_gemm_pf_k, grid, dim3(256), 0, (hipStream_t)stream, d_x, ldx, d_map, m, kvol, d_w, cin,
                         cout, trans_w, d_bias, d_y, ldy);
    else  // form 3: the version without prefetch (kept for comparison; the only one for K > 32)
      hipLaunchKernelGGL(sp_gather_gemm_k, grid, dim3(256), 0, (hipStream_t)stream, d_x, ldx, d_map, m, kvol, d_w, cin,
                         cout, trans_w, d_bias, d_y, ldy);
  }
  SST_LAUNCH_CHECK();
  return SST_OK;
}

int64_t sst_spconv_wgrad_workspace_bytes(int kvol, int64_t pair_ld, int64_t total_pairs, int cin, int cout) {
  return sp_wgrad_chunks(kvol > 0 ? kvol : 1, pair_ld > 0 ? pair_ld : 1, total_pairs) * cin * cout *
             (int64_t)sizeof(float) + 256;
}

int sst_spconv_wgrad_f32(const float* d_x, int64_t ldx, const float* d_dy, int64_t lddy, const int32_t* d_pairs,
                         int64_t pair_ld, int64_t total_pairs, int x_side, const int32_t* d_num, int kvol, int cin,
                         int cout, float* d_dw, void* d_workspace, void* stream) {
  if (kvol < 1 || cin < 1 || cout < 1 || ldx < cin || lddy < cout || pair_ld < 0 || (x_side != 0 && x_side != 1))
    return SST_ERR_ARG;
  if (!d_pairs || !d_num || !d_dw || !d_workspace) return SST_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int64_t per_k = (int64_t)cin * cout;
  if (pair_ld == 0 || !d_x || !d_dy) {
    SST_HIP(hipMemsetAsync(d_dw, 0, sizeof(float) * kvol * per_k, st));
    return SST_OK;
  }
  const int64_t chunks = sp_wgrad_chunks(kvol, pair_ld, total_pairs);
  // strips of 32 input channels x 64 or 128 output columns: the narrow strip whenever the columns beyond the last
  // multiple of 128 fit into 64 (64-channel layers: a 128-column strip would multiply zeros half of the time)
  const int rem = cout % 128;
  const int nq = (rem > 0 && rem <= 64) ? 2 : 4;
  const int strips = (int)(sst_div_up(cin, 32) * sst_div_up(cout, 32 * nq));
  if (kvol > 65535 || chunks > 0x7fffffff || sst_div_up(strips, 4) > 65535) return SST_ERR_UNSUPPORTED;
  float* part = (float*)d_workspace;
  const dim3 wg_grid((unsigned)chunks, (unsigned)sst_div_up(strips, 4));
  if (nq == 2)
    hipLaunchKernelGGL(sp_wgrad_k<2>, wg_grid, dim3(256), 0, st, d_x, ldx, d_dy, lddy, d_pairs, pair_ld, x_side, d_num,
                       kvol, cin, cout, part);
  else
    hipLaunchKernelGGL(sp_wgrad_k<4>, wg_grid, dim3(256), 0, st, d_x, ldx, d_dy, lddy, d_pairs, pair_ld, x_side, d_num,
                       kvol, cin, cout, part);
  hipLaunchKernelGGL(sp_wgrad_reduce_k, dim3((unsigned)sst_div_up(per_k, 256), (unsigned)kvol), dim3(256), 0, st, part,
                     d_num, kvol, per_k, d_dw);
  SST_LAUNCH_CHECK();
  return SST_OK;
}

int sst_spconv_maxpool_fwd_f32(const float* d_x, int64_t ldx, const int32_t* d_out2in, int64_t m, int kvol, int c,
                               float* d_y, int64_t ldy, void* stream) {
  if (m < 0 || kvol < 1 || c < 1 || ldx < c || ldy < c) return SST_ERR_ARG;
  if (m == 0) return SST_OK;
  if (!d_x || !d_out2in || !d_y) return SST_ERR_ARG;
  hipLaunchKernelGGL(sp_maxpool_fwd_k, dim3(sst_grid_1d(m * c, 256)), dim3(256), 0, (hipStream_t)stream, d_x, ldx,
                     d_out2in, m, kvol, c, d_y, ldy);
  SST_LAUNCH_CHECK();
  return SST_OK;
}

int sst_spconv_maxpool_bwd_f32(const float* d_x, int64_t ldx, const float* d_y, int64_t ldy, const float* d_dy,
                               int64_t lddy, const int32_t* d_in2out, int64_t n, int kvol, int c, float* d_dx,
                               int64_t lddx, void* stream) {
  if (n < 0 || kvol < 1 || c < 1 || ldx < c || ldy < c || lddy < c || lddx < c) return SST_ERR_ARG;
  if (n == 0) return SST_OK;
  if (!d_x || !d_in2out || !d_dx || !d_y || !d_dy) return SST_ERR_ARG;
  hipLaunchKernelGGL(sp_maxpool_bwd_k, dim3(sst_grid_1d(n * c, 256)), dim3(256), 0, (hipStream_t)stream, d_x, ldx, d_y,
                     ldy, d_dy, lddy, d_in2out, n, kvol, c, d_dx, lddx);
  SST_LAUNCH_CHECK();
  return SST_OK;
}

}  // extern "C"
