// Connected components of the "closer than dist in the xy plane" graph over cluster centres: the step between
// FSD's segmentor and its SIR point-group backbone.
//
// Reference: find_connected_componets, mmdet3d/models/detectors/single_stage_fsd.py:45-68 — per sample a dense
// N x N distance matrix on the GPU, `.cpu()`, scipy.sparse.csgraph.connected_components, labels shifted by a
// running base, `.to(device)`; the documented source of FSD's training-time instability
// (docs/overall_instructions.md:51).  scipy numbers the components in order of first appearance, i.e. by the
// smallest node index they contain.
//
// Here: lock-free union-find over all samples at once.  cc_pairs_k enumerates the pairs (j < i) tile by tile
// (256 x 256, the j tile staged in LDS), tests `same sample && sqrt(dx*dx + dy*dy) < dist` with the reference's
// own fp32 operations (separate multiplies, add, correctly rounded sqrt: no FMA contraction, so the edge set is
// bit-identical), and hooks the LARGER root under the smaller one with atomicCAS.  The root of a component is
// therefore its smallest node index whatever the execution order: flatten, flag the roots, exclusive-scan the
// flags and label[i] = rank of root(i) — scipy's numbering, deterministic.  With samples stored one after the
// other (they are: the centres come out of a sorted-unique) this also equals the reference's per-sample
// numbering with a running base.  N is a few 1e3..1e4 centres: the N^2/2 pair tests are ~1e8 simple operations.
#include "common.h"

namespace {

constexpr int kCcTile = 256;

__device__ __forceinline__ int cc_load(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ int cc_find(int* __restrict__ parent, int x) {
  int p = cc_load(parent + x);
  while (p != x) {
    const int gp = cc_load(parent + p);
    if (gp != p) __hip_atomic_store(parent + x, gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // path halving
    x = p;
    p = gp;
  }
  return x;
}

__device__ __forceinline__ void cc_union(int* __restrict__ parent, int a, int b) {
  while (true) {
    a = cc_find(parent, a);
    b = cc_find(parent, b);
    if (a == b) return;
    if (a < b) {
      const int t = a;
      a = b;
      b = t;
    }
    const int old = atomicCAS(parent + a, a, b);  // hook the larger root under the smaller one
    if (old == a) return;
    a = old;
  }
}

__global__ __launch_bounds__(256) void cc_init_k(int* __restrict__ parent, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) parent[i] = (int)i;
}

// blockIdx.x enumerates the tile pairs (ti >= tj) of the lower triangle
__global__ __launch_bounds__(kCcTile) void cc_pairs_k(const float* __restrict__ pts, int64_t ld,
                                                      const int32_t* __restrict__ batch, int n, float dist,
                                                      int n_tiles, int* __restrict__ parent) {
  __shared__ float xs[kCcTile], ys[kCcTile];
  __shared__ int bs[kCcTile];
  // row ti holds ti + 1 pairs: ti = floor((sqrt(8 b + 1) - 1) / 2), fixed up for rounding
  const int b = blockIdx.x;
  int ti = (int)((sqrtf(8.f * (float)b + 1.f) - 1.f) * 0.5f);
  while ((int64_t)(ti + 1) * (ti + 2) / 2 <= b) ++ti;
  while ((int64_t)ti * (ti + 1) / 2 > b) --ti;
  const int tj = b - (int)((int64_t)ti * (ti + 1) / 2);
  if (ti >= n_tiles) return;
  const int i = ti * kCcTile + threadIdx.x;
  const int j0 = tj * kCcTile;
  {
    const int j = j0 + threadIdx.x;
    const bool ok = j < n;
    xs[threadIdx.x] = ok ? pts[(int64_t)j * ld] : 0.f;
    ys[threadIdx.x] = ok ? pts[(int64_t)j * ld + 1] : 0.f;
    bs[threadIdx.x] = ok ? batch[j] : -1;
  }
  __syncthreads();
  if (i >= n) return;
  const float xi = pts[(int64_t)i * ld], yi = pts[(int64_t)i * ld + 1];
  const int bi = batch[i];
  const int jn = (n - j0) < kCcTile ? (n - j0) : kCcTile;
  for (int q = 0; q < jn; ++q) {
    const int j = j0 + q;
    if (j >= i) break;  // lower triangle only (the graph is undirected)
    if (bs[q] != bi) continue;
    const float dx = __fsub_rn(xi, xs[q]), dy = __fsub_rn(yi, ys[q]);
    const float d = __fsqrt_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)));
    if (d < dist) cc_union(parent, i, j);
  }
}

__global__ __launch_bounds__(256) void cc_flatten_k(int* __restrict__ parent, int64_t n, int32_t* __restrict__ is_root) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int x = (int)i;
  while (true) {  // no concurrent unions any more: plain walk (reads through the L2: other threads only write roots)
    const int p = cc_load(parent + x);
    if (p == x) break;
    x = p;
  }
  is_root[i] = (x == (int)i) ? 1 : 0;
  __hip_atomic_store(parent + i, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(256) void cc_label_k(const int* __restrict__ parent, const int32_t* __restrict__ rank,
                                                  int64_t n, int32_t* __restrict__ labels) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) labels[i] = rank[parent[i]];
}

}  // namespace

extern "C" {

int64_t sst_connected_components_workspace_bytes(int64_t n) {
  if (n < 1) n = 1;
  return 3 * sst_align_up(n * (int64_t)sizeof(int32_t), 256) + sst_scan_workspace_bytes(n) + 256;
}

int sst_connected_components_xy_f32(const float* d_points, int64_t ld, const int32_t* d_batch, int64_t n, float dist,
                                    int32_t* d_labels, int32_t* d_num_components, void* d_workspace, void* stream) {
  if (n < 0 || ld < 2 || n > 16000000) return n > 16000000 ? SST_ERR_UNSUPPORTED : SST_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) {
    if (d_num_components) SST_HIP(hipMemsetAsync(d_num_components, 0, sizeof(int32_t), st));
    return SST_OK;
  }
  if (!d_points || !d_batch || !d_labels || !d_workspace) return SST_ERR_ARG;
  char* ws = (char*)d_workspace;
  const int64_t seg = sst_align_up(n * (int64_t)sizeof(int32_t), 256);
  int* parent = (int*)ws;
  int32_t* is_root = (int32_t*)(ws + seg);
  int32_t* rank = (int32_t*)(ws + 2 * seg);
  void* scan_ws = ws + 3 * seg;
  const int g1 = sst_grid_1d(n, 256);
  hipLaunchKernelGGL(cc_init_k, dim3(g1), dim3(256), 0, st, parent, n);
  const int n_tiles = (int)sst_div_up(n, kCcTile);
  const int64_t n_pairs = (int64_t)n_tiles * (n_tiles + 1) / 2;
  hipLaunchKernelGGL(cc_pairs_k, dim3((unsigned)n_pairs), dim3(kCcTile), 0, st, d_points, ld, d_batch, (int)n, dist,
                     n_tiles, parent);
  hipLaunchKernelGGL(cc_flatten_k, dim3(g1), dim3(256), 0, st, parent, n, is_root);
  const int rc = sst_exclusive_scan_i32(is_root, rank, n, d_num_components, scan_ws, stream);
  if (rc != SST_OK) return rc;
  hipLaunchKernelGGL(cc_label_k, dim3(g1), dim3(256), 0, st, parent, rank, n, d_labels);
  SST_LAUNCH_CHECK();
  return SST_OK;
}

}  // extern "C"
