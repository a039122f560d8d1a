// Segmented point->voxel / point->cluster reduction and row gather/scatter for gfx950.
//
// Reference semantics:
//   forward  feats_reduce_kernel          mmdet3d/ops/voxel/src/scatter_points_cuda.cu:80-103 (+ :222-229)
//   backward add_reduce_traceback_grad    scatter_points_cuda.cu:105-133
//            max_reduce_traceback / scatter_grad   scatter_points_cuda.cu:135-179 (tie -> smallest point index)
//   torch_scatter.scatter_max / scatter(mean|sum)  call sites mmdet3d/ops/sst/sst_ops.py:172-177
//
// The reference reduces with float CAS atomics, one thread per point serial over channels.  Here the
// points are already grouped by a stable sort (sort_scan.hip), so each output element is produced by
// exactly one thread walking its group's CSR range: no atomics, deterministic, and a wave reads 64
// consecutive channels of one row (coalesced 256 B) — HBM-bound: (4C+4) B/point + 4C B/group.
#include <math.h>
#include <stdlib.h>
#include <hip/hip_ext.h>
#include "common.h"

namespace {

// Optional transform of every loaded value: relu(x * scale[ch] + shift[ch]) - the BatchNorm + ReLU of a DynamicVFE layer
// applied while its output is pooled (voxel_encoder.py:286-296: vfe -> scatter max), so that the activated [N, C] matrix of
// the LAST layer is never written (its backward pass recomputes the mask from the same expression: csrc/bn.hip).
struct seg_xf {
  const float* scale;
  const float* shift;
};
__device__ __forceinline__ float4 seg_xf_apply(float4 x, const float4 sc, const float4 sh) {
  x.x = x.x * sc.x + sh.x, x.y = x.y * sc.y + sh.y, x.z = x.z * sc.z + sh.z, x.w = x.w * sc.w + sh.w;
  x.x = x.x > 0.f ? x.x : 0.f, x.y = x.y > 0.f ? x.y : 0.f, x.z = x.z > 0.f ? x.z : 0.f, x.w = x.w > 0.f ? x.w : 0.f;
  return x;
}

// Work list of the long groups of a call (sst_segment_reduce_fwd_work_f32), int32 words:
//   header [8]: [0] entries, [1] next entry to take, [2] workgroups of the work kernel done, [3] partial slots handed out,
//               [4] groups cut into several chunks, [5] workgroups of the merge kernel done
//   entries [cap_e][3]: (output row g, chunk index, partial slot or -1) - a group of more than kSegChunk rows is cut into
//               chunks of kSegChunk rows, one entry = one workgroup's worth (a real sweep has voxels with thousands of points
//               and one group of all the clamped out-of-range points: one workgroup per group left 0.1 ms launches);
//   multi [cap_m][3]: (g, first partial slot, chunks) of every group with more than one chunk, merged in chunk order by
//               seg_reduce_merge_k; partial values [cap_p][cpad] floats, partial arg-max rows [cap_p][cpad] ints.
// The order of the lists has no influence on the result: a chunk is reduced by one workgroup in a fixed order, the chunks of a
// group are merged in chunk order.  Every kernel that is the last user of a counter zeroes it: the list is reusable as it is.
constexpr int kSegWorkHdr = 8;
constexpr int kSegChunk = 512;
struct seg_work {
  int32_t* w;
  int cap_e, cap_m, cap_p, cpad;
};
__device__ __forceinline__ int32_t* seg_work_entries(const seg_work& W) { return W.w + kSegWorkHdr; }
__device__ __forceinline__ int32_t* seg_work_multi(const seg_work& W) { return W.w + kSegWorkHdr + 3 * (int64_t)W.cap_e; }
__device__ __forceinline__ float* seg_work_pvals(const seg_work& W) {
  return (float*)(W.w + kSegWorkHdr + 3 * (int64_t)W.cap_e + 3 * (int64_t)W.cap_m);
}
__device__ __forceinline__ int32_t* seg_work_pargs(const seg_work& W) {
  return W.w + kSegWorkHdr + 3 * (int64_t)W.cap_e + 3 * (int64_t)W.cap_m + (int64_t)W.cap_p * W.cpad;
}
// one thread of a long group lists it
__device__ __forceinline__ void seg_work_push(const seg_work& W, int64_t g, int len) {
  const int nch = (len + kSegChunk - 1) / kSegChunk;
  const int s0 = atomicAdd(&W.w[0], nch);
  int p0 = -1;
  if (nch > 1) {
    p0 = atomicAdd(&W.w[3], nch);
    int32_t* mrec = seg_work_multi(W) + 3 * (int64_t)atomicAdd(&W.w[4], 1);
    mrec[0] = (int32_t)g, mrec[1] = p0, mrec[2] = nch;
  }
  int32_t* e = seg_work_entries(W) + 3 * (int64_t)s0;
  for (int i = 0; i < nch; ++i) e[3 * i] = (int32_t)g, e[3 * i + 1] = i, e[3 * i + 2] = nch > 1 ? p0 + i : -1;
}

// one thread per (group, channel) element; consecutive threads -> consecutive channels of one group
__global__ __launch_bounds__(256) void seg_reduce_fwd_k(const float* __restrict__ feats, int c,
                                                        const uint32_t* __restrict__ perm,
                                                        const int32_t* __restrict__ offsets,
                                                        const int32_t* __restrict__ gidx, int64_t m, int mode,
                                                        float* __restrict__ out, int32_t* __restrict__ argmax,
                                                        int32_t n_rows, const int32_t* __restrict__ d_mlim, int skip_len,
                                                        const seg_work work) {
  if (d_mlim != nullptr && (int64_t)*d_mlim < m) m = *d_mlim;  // device-side row count (m is then an upper bound)
  const int64_t total = m * c;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t g = e / c;
    const int ch = (int)(e - g * c);
    const int64_t gs = gidx != nullptr ? gidx[g] : g;
    const int beg = gs < 0 ? 0 : offsets[gs], end = gs < 0 ? 0 : offsets[gs + 1];   // negative entry: an empty group
    if (skip_len > 0 && end - beg > skip_len) {  // long group: seg_reduce_fwd_work_k takes it from the work list
      if (ch == 0) seg_work_push(work, g, end - beg);
      continue;
    }
    if (mode == SST_REDUCE_MAX) {
      float acc = -INFINITY;
      int32_t arg = n_rows;
      for (int p = beg; p < end; ++p) {
        const uint32_t row = perm[p];
        const float x = feats[(int64_t)row * c + ch];
        if (p == beg || x > acc) {  // strict '>' keeps the smallest row index on ties
          acc = x;
          arg = (int32_t)row;
        }
      }
      out[e] = end > beg ? acc : 0.f;  // a group without points reads 0, as torch_scatter.scatter_max fills it
      if (argmax != nullptr) argmax[e] = arg;
    } else {
      float acc = 0.f;
      for (int p = beg; p < end; ++p) acc += feats[(int64_t)perm[p] * c + ch];
      if (mode == SST_REDUCE_MEAN && end > beg) acc = acc / (float)(end - beg);
      out[e] = acc;
    }
  }
}

// float4 variant (c % 4 == 0, 16-byte aligned rows): one thread per (group, 4 channels), points walked four at a
// time so that four row ids and then four 16-byte row segments are in flight per thread instead of a dependent
// id -> value pair per point (the scalar kernel ran at 25-29 % of the HBM roof on 1e5..3e5 points).
__global__ __launch_bounds__(256) void seg_reduce_fwd_v4_k(const float* __restrict__ feats, int c,
                                                           const uint32_t* __restrict__ perm,
                                                           const int32_t* __restrict__ offsets,
                                                           const int32_t* __restrict__ gidx, int64_t m, int mode,
                                                           float* __restrict__ out, int32_t* __restrict__ argmax,
                                                           int32_t n_rows, const int32_t* __restrict__ d_mlim, int skip_len,
                                                           const seg_work work, const seg_xf xf) {
  if (d_mlim != nullptr && (int64_t)*d_mlim < m) m = *d_mlim;  // device-side row count (m is then an upper bound)
  const int c4 = c >> 2;
  const int64_t total = m * c4;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t g = e / c4;
    const int ch = (int)(e - g * c4) * 4;
    const int64_t gs = gidx != nullptr ? gidx[g] : g;
    const int beg = gs < 0 ? 0 : offsets[gs], end = gs < 0 ? 0 : offsets[gs + 1];   // negative entry: an empty group
    if (skip_len > 0 && end - beg > skip_len) {  // long group: seg_reduce_fwd_block_k / seg_reduce_fwd_work_k takes it
      if (work.w != nullptr && ch == 0) seg_work_push(work, g, end - beg);
      continue;
    }
    float4 acc = mode == SST_REDUCE_MAX ? make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY)
                                        : make_float4(0.f, 0.f, 0.f, 0.f);
    int32_t a0 = n_rows, a1 = n_rows, a2 = n_rows, a3 = n_rows;
    bool first = true;
    float4 xsc = make_float4(1.f, 1.f, 1.f, 1.f), xsh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (xf.scale != nullptr) xsc = *(const float4*)(xf.scale + ch), xsh = *(const float4*)(xf.shift + ch);
    for (int p = beg; p < end; p += 4) {
      uint32_t row[4];
      float4 x[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) row[u] = perm[p + u < end ? p + u : end - 1];
#pragma unroll
      for (int u = 0; u < 4; ++u) x[u] = *(const float4*)(feats + (int64_t)row[u] * c + ch);
      if (xf.scale != nullptr) {
#pragma unroll
        for (int u = 0; u < 4; ++u) x[u] = seg_xf_apply(x[u], xsc, xsh);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (p + u < end) {
          if (mode == SST_REDUCE_MAX) {
            // strict '>' keeps the smallest row index on ties (rows of a group are visited in ascending order)
            if (first || x[u].x > acc.x) acc.x = x[u].x, a0 = (int32_t)row[u];
            if (first || x[u].y > acc.y) acc.y = x[u].y, a1 = (int32_t)row[u];
            if (first || x[u].z > acc.z) acc.z = x[u].z, a2 = (int32_t)row[u];
            if (first || x[u].w > acc.w) acc.w = x[u].w, a3 = (int32_t)row[u];
            first = false;
          } else {
            acc.x += x[u].x, acc.y += x[u].y, acc.z += x[u].z, acc.w += x[u].w;
          }
        }
      }
    }
    if (mode == SST_REDUCE_MEAN && end > beg) {
      const float cnt = (float)(end - beg);
      acc.x = acc.x / cnt, acc.y = acc.y / cnt, acc.z = acc.z / cnt, acc.w = acc.w / cnt;
    }
    if (end <= beg) acc = make_float4(0.f, 0.f, 0.f, 0.f);  // empty group: 0 (torch_scatter.scatter_max's fill)
    *(float4*)(out + g * c + ch) = acc;
    if (mode == SST_REDUCE_MAX && argmax != nullptr) *(int4*)(argmax + g * c + ch) = make_int4(a0, a1, a2, a3);
  }
}

// Long groups (FSD clusters: hundreds to thousands of points each): ONE WORKGROUP per group.  The eight 32-lane halves of
// the block stride over the group's points (each half reads whole 16-byte-per-lane row segments, two points in flight),
// the eight partial results meet in LDS in a fixed order.  The thread-per-(group, 4 channels) kernel above walks such a
// group serially behind one load latency per 4 points (measured on FSD's clusters: 0.34 TB/s).  Launched beside it when
// the average group has at least 8 points: groups of more than kLongGroup (16) points are skipped there and taken here.  Ties of the maximum: smallest row index, as above (every half sees its rows
// in ascending order, the halves are combined with value first, then the smaller index).
// one group by the whole workgroup: float4 channel quads (c % 4 == 0).  A row is read by L = 8 / 16 / 32 lanes (the power of
// two that covers its quads), the 256 / L row parts stride over the group's rows with four rows in flight each, and meet in
// LDS in a fixed order.
__device__ __forceinline__ void seg_block_group_v4(const float* __restrict__ feats, int c, const uint32_t* __restrict__ perm,
                                                   int beg, int end, int64_t g, int mode, float* __restrict__ out,
                                                   int32_t* __restrict__ argmax, int32_t n_rows, float4 (*lds_v2)[32],
                                                   int4 (*lds_a2)[32], const seg_xf xf = seg_xf{nullptr, nullptr},
                                                   float* __restrict__ pval = nullptr, int32_t* __restrict__ parg = nullptr) {
  float4* lds_v = &lds_v2[0][0];
  int4* lds_a = &lds_a2[0][0];
  const int c4 = c >> 2;
  const int L = c4 <= 8 ? 8 : (c4 <= 16 ? 16 : 32);
  const int parts = 256 / L, part = threadIdx.x / L, l = threadIdx.x - part * L;
  for (int q0 = 0; q0 < c4; q0 += L) {  // L channel quads per pass (c <= 128: one pass)
    const int q = q0 + l;
    const bool live = q < c4;
    float4 acc = mode == SST_REDUCE_MAX ? make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY)
                                        : make_float4(0.f, 0.f, 0.f, 0.f);
    int4 arg = make_int4(n_rows, n_rows, n_rows, n_rows);
    float4 xsc = make_float4(1.f, 1.f, 1.f, 1.f), xsh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (xf.scale != nullptr && live) xsc = *(const float4*)(xf.scale + 4 * q), xsh = *(const float4*)(xf.shift + 4 * q);
    for (int p = beg + part; p < end; p += 4 * parts) {
      uint32_t r[4];
      float4 x[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) r[u] = perm[p + u * parts < end ? p + u * parts : p];
#pragma unroll
      for (int u = 0; u < 4; ++u) x[u] = live ? *(const float4*)(feats + (int64_t)r[u] * c + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (xf.scale != nullptr) {
#pragma unroll
        for (int u = 0; u < 4; ++u) x[u] = seg_xf_apply(x[u], xsc, xsh);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (p + u * parts < end) {   // rows of a part are visited in ascending order: strict '>' keeps the smallest row on ties
          if (mode == SST_REDUCE_MAX) {
            if (x[u].x > acc.x) acc.x = x[u].x, arg.x = (int)r[u];
            if (x[u].y > acc.y) acc.y = x[u].y, arg.y = (int)r[u];
            if (x[u].z > acc.z) acc.z = x[u].z, arg.z = (int)r[u];
            if (x[u].w > acc.w) acc.w = x[u].w, arg.w = (int)r[u];
          } else {
            acc.x += x[u].x, acc.y += x[u].y, acc.z += x[u].z, acc.w += x[u].w;
          }
        }
      }
    }
    lds_v[threadIdx.x] = acc;
    lds_a[threadIdx.x] = arg;
    __syncthreads();
    if (part == 0 && live) {
      float4 r = acc;
      int4 a = arg;
      for (int h = 1; h < parts; ++h) {
        const float4 v = lds_v[h * L + l];
        const int4 b = lds_a[h * L + l];
        if (mode == SST_REDUCE_MAX) {
          // a part that saw no point holds (-inf, n_rows): never wins; -inf values of real rows keep the smaller index
          if (v.x > r.x || (v.x == r.x && b.x < a.x)) r.x = v.x, a.x = b.x;
          if (v.y > r.y || (v.y == r.y && b.y < a.y)) r.y = v.y, a.y = b.y;
          if (v.z > r.z || (v.z == r.z && b.z < a.z)) r.z = v.z, a.z = b.z;
          if (v.w > r.w || (v.w == r.w && b.w < a.w)) r.w = v.w, a.w = b.w;
        } else {
          r.x += v.x, r.y += v.y, r.z += v.z, r.w += v.w;
        }
      }
      if (pval != nullptr) {   // a chunk of a group: the raw partial result, finished by seg_reduce_merge_k
        *(float4*)(pval + 4 * q) = r;
        *(int4*)(parg + 4 * q) = a;
      } else {
        if (mode == SST_REDUCE_MEAN && end > beg) {
          const float cnt = (float)(end - beg);
          r.x = r.x / cnt, r.y = r.y / cnt, r.z = r.z / cnt, r.w = r.w / cnt;
        }
        if (end <= beg) r = make_float4(0.f, 0.f, 0.f, 0.f);
        *(float4*)(out + g * c + 4 * q) = r;
        if (mode == SST_REDUCE_MAX && argmax != nullptr) *(int4*)(argmax + g * c + 4 * q) = a;
      }
    }
    __syncthreads();
  }
}

// one group by the whole workgroup, any width: thread = (row part, channel); cp = channels rounded up to a power of two
// (<= 256), 256 / cp row parts stride over the group's rows, the parts meet in LDS in a fixed order
__device__ __forceinline__ void seg_block_group_v1(const float* __restrict__ feats, int c, const uint32_t* __restrict__ perm,
                                                   int beg, int end, int64_t g, int mode, float* __restrict__ out,
                                                   int32_t* __restrict__ argmax, int32_t n_rows, float* lds_v, int* lds_a,
                                                   float* __restrict__ pval = nullptr, int32_t* __restrict__ parg = nullptr) {
  for (int c0 = 0; c0 < c; c0 += 256) {
    const int cw = c - c0 < 256 ? c - c0 : 256;
    int cp = 1;
    while (cp < cw) cp <<= 1;
    const int parts = 256 / cp, part = threadIdx.x / cp, ch = threadIdx.x - part * cp;
    const bool live = ch < cw;
    float acc = mode == SST_REDUCE_MAX ? -INFINITY : 0.f;
    int arg = n_rows;
    if (live) {
      for (int p = beg + part; p < end; p += parts) {
        const uint32_t r = perm[p];
        const float x = feats[(int64_t)r * c + c0 + ch];
        if (mode == SST_REDUCE_MAX) {
          if (x > acc) acc = x, arg = (int)r;
        } else {
          acc += x;
        }
      }
    }
    lds_v[threadIdx.x] = acc;
    lds_a[threadIdx.x] = arg;
    __syncthreads();
    if (part == 0 && live) {
      float r = acc;
      int a = arg;
      for (int h = 1; h < parts; ++h) {
        const float v = lds_v[h * cp + ch];
        const int b = lds_a[h * cp + ch];
        if (mode == SST_REDUCE_MAX) {
          if (v > r || (v == r && b < a)) r = v, a = b;
        } else {
          r += v;
        }
      }
      if (pval != nullptr) {
        pval[c0 + ch] = r;
        parg[c0 + ch] = a;
      } else {
        if (mode == SST_REDUCE_MEAN && end > beg) r = r / (float)(end - beg);
        if (end <= beg) r = 0.f;
        out[g * c + c0 + ch] = r;
        if (mode == SST_REDUCE_MAX && argmax != nullptr) argmax[g * c + c0 + ch] = a;
      }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void seg_reduce_fwd_block_k(const float* __restrict__ feats, int c,
                                                              const uint32_t* __restrict__ perm,
                                                              const int32_t* __restrict__ offsets,
                                                              const int32_t* __restrict__ gidx, int64_t m, int mode,
                                                              float* __restrict__ out, int32_t* __restrict__ argmax,
                                                              int32_t n_rows, const int32_t* __restrict__ d_mlim, int min_len) {
  __shared__ float4 lds_v[8][32];
  __shared__ int4 lds_a[8][32];
  if (d_mlim != nullptr && (int64_t)*d_mlim < m) m = *d_mlim;
  for (int64_t g = blockIdx.x; g < m; g += gridDim.x) {
    const int64_t gs = gidx != nullptr ? gidx[g] : g;
    const int beg = gs < 0 ? 0 : offsets[gs], end = gs < 0 ? 0 : offsets[gs + 1];   // negative entry: an empty group
    if (end - beg <= min_len) continue;     // short group: the thread-per-(group, 4 channels) kernel took it (uniform)
    seg_block_group_v4(feats, c, perm, beg, end, g, mode, out, argmax, n_rows, lds_v, lds_a);
  }
}

// The long groups of a call whose AVERAGE group is short (voxel grouping of a real LiDAR sweep: 1-10 points per voxel on
// average, thousands in the voxels next to the sensor): the thread-per-element kernels above put them on a work list instead
// of walking them serially (a 3 000-point voxel took 0.4-0.5 ms there, longer than the rest of the launch by 20 x); a fixed
// grid of workgroups takes the entries by ticket, one workgroup per group.  The last workgroup to finish zeroes the three
// counters: the list is reusable without a fill.  Nothing spins, no float atomics; the order of the list does not matter.
template <int V>
__global__ __launch_bounds__(256) void seg_reduce_fwd_work_k(const float* __restrict__ feats, int c,
                                                             const uint32_t* __restrict__ perm,
                                                             const int32_t* __restrict__ offsets,
                                                             const int32_t* __restrict__ gidx, int mode,
                                                             float* __restrict__ out, int32_t* __restrict__ argmax,
                                                             int32_t n_rows, const seg_work work, const seg_xf xf) {
  __shared__ float4 lds_v[8][32];
  __shared__ int4 lds_a[8][32];
  __shared__ int s_take;
  int32_t* hdr = work.w;
  const int count = __atomic_load_n(&hdr[0], __ATOMIC_RELAXED);
  if (count == 0) return;   // no long group (the usual voxel grouping): the counters are zero as they stand, nothing to take or reset
  const int32_t* entries = seg_work_entries(work);
  for (;;) {
    if (threadIdx.x == 0) s_take = atomicAdd(&hdr[1], 1);
    __syncthreads();
    const int take = s_take;
    __syncthreads();
    if (take >= count) break;
    const int64_t g = entries[3 * (int64_t)take];
    const int chunk = entries[3 * (int64_t)take + 1], pslot = entries[3 * (int64_t)take + 2];
    const int64_t gs = gidx != nullptr ? gidx[g] : g;
    int beg = gs < 0 ? 0 : offsets[gs], end = gs < 0 ? 0 : offsets[gs + 1];   // negative entry: an empty group
    float* pv = nullptr;
    int32_t* pa = nullptr;
    if (pslot >= 0) {        // one chunk of a group that was cut: rows [beg + chunk * kSegChunk, + kSegChunk)
      beg += chunk * kSegChunk;
      end = beg + kSegChunk < end ? beg + kSegChunk : end;
      pv = seg_work_pvals(work) + (int64_t)pslot * work.cpad;
      pa = seg_work_pargs(work) + (int64_t)pslot * work.cpad;
    }
    if (V == 4)
      seg_block_group_v4(feats, c, perm, beg, end, g, mode, out, argmax, n_rows, lds_v, lds_a, xf, pv, pa);
    else
      seg_block_group_v1(feats, c, perm, beg, end, g, mode, out, argmax, n_rows, (float*)&lds_v[0][0], (int*)&lds_a[0][0], pv, pa);
  }
  if (threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(&hdr[2], 1) == (int)gridDim.x - 1) {   // every workgroup has left its loop: nobody reads these counters again
      hdr[0] = 0;
      hdr[1] = 0;
      __threadfence();
      hdr[2] = 0;
    }
  }
}

// The groups that were cut into chunks: chunk partials merged in chunk order (MAX: value first, then the smaller row - the
// chunks hold ascending rows, so this is the tie rule of the serial walk; SUM / MEAN: added in chunk order).  Launched behind
// seg_reduce_fwd_work_k on the same stream; returns at once when no group was cut.
__global__ __launch_bounds__(256) void seg_reduce_merge_k(int c, const int32_t* __restrict__ offsets,
                                                          const int32_t* __restrict__ gidx, int mode, float* __restrict__ out,
                                                          int32_t* __restrict__ argmax, const seg_work work) {
  int32_t* hdr = work.w;
  const int n_multi = __atomic_load_n(&hdr[4], __ATOMIC_RELAXED);
  if (n_multi == 0) return;
  const int32_t* multi = seg_work_multi(work);
  const float* pv = seg_work_pvals(work);
  const int32_t* pa = seg_work_pargs(work);
  for (int mi = blockIdx.x; mi < n_multi; mi += gridDim.x) {
    const int64_t g = multi[3 * (int64_t)mi];
    const int p0 = multi[3 * (int64_t)mi + 1], nch = multi[3 * (int64_t)mi + 2];
    const int64_t gs = gidx != nullptr ? gidx[g] : g;
    const int len = offsets[gs + 1] - offsets[gs];
    for (int ch = threadIdx.x; ch < c; ch += blockDim.x) {
      float r = pv[(int64_t)p0 * work.cpad + ch];
      int a = pa[(int64_t)p0 * work.cpad + ch];
      for (int i = 1; i < nch; ++i) {
        const float v = pv[(int64_t)(p0 + i) * work.cpad + ch];
        const int b = pa[(int64_t)(p0 + i) * work.cpad + ch];
        if (mode == SST_REDUCE_MAX) {
          if (v > r || (v == r && b < a)) r = v, a = b;
        } else {
          r += v;
        }
      }
      if (mode == SST_REDUCE_MEAN) r = r / (float)len;
      out[g * c + ch] = r;
      if (mode == SST_REDUCE_MAX && argmax != nullptr) argmax[g * c + ch] = a;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(&hdr[5], 1) == (int)gridDim.x - 1) {
      hdr[3] = 0;
      hdr[4] = 0;
      __threadfence();
      hdr[5] = 0;
    }
  }
}

// Reduction over LONG, UNEVEN groups (FSD's clusters / RoI point sets: tens to thousands of points per group, Zipf-like;
// torch_scatter.scatter_max / scatter(mean) over cluster ids, voxel_encoder.py:696-764, single_stage_fsd.py:469), balanced
// and in ONE launch.  The per-group kernels above walk a 3 000-point cluster with one workgroup (16 us for 9 MB) or, for
// the 3-channel centroid means, with three threads (25 us for 0.2 MB).  Here the unit of work is a TILE of sorted positions,
// whatever groups they belong to:
//   * a workgroup = `lanes` row lanes x `cv` channel vectors (float4, or scalars when c % 4 != 0); a row lane walks 8
//     consecutive sorted rows (all loads in flight), reducing runs of equal group id;
//   * a run that begins and ends strictly inside a lane's span is a whole group: written straight to the output;
//   * the first and the last run of every lane meet in LDS, where lane 0's threads merge them in order: groups that are whole
//     inside the tile are written out; a run that began before the tile (record 0) or continues behind it (record 1) goes
//     to a global record;
//   * a group that crosses tiles is finished by the LAST of its tiles to arrive (one atomic ticket per crossing group and
//     tile, in a per-grouping counter array that is zeroed once and cleans itself): that workgroup merges the group's records
//     with all its threads, in a fixed order.
// Deterministic: every merge order is fixed, MAX carries (value, row) and prefers the smaller row on ties, like
// max_reduce_traceback_scatter_idx_kernel (scatter_points_cuda.cu:135-160).  No float atomics, nothing spins.  (A first
// version with one 64-bit atomic maximum per partial result was 2 x SLOWER than the unbalanced kernels: device-scope
// atomics cross the XCDs.)
constexpr int kSegSpan = 8;       // rows per row lane
constexpr int kSegNone = -3;      // "no record" (group ids are >= -1: -1 = rows of a discarded group)

struct seg_rec {                  // partial result of a run, one channel vector
  float4 v;
  int4 a;
};

// Records travel between workgroups that may run on different XCDs, whose L2s are not coherent with each other: they are
// written and read with AGENT-scope accesses (`sc1`: the store writes through to, the load reads from, the point that is
// coherent for the whole device), 16 bytes at a time, and the writer waits for its stores (s_waitcnt vmcnt(0)) BEFORE the
// workgroup takes the group's ticket.  History: the relaxed atomic load / store builtins compiled to one scoped access
// followed by seven plain ones; word-wise atomic exchanges worked only once their results were consumed (a fire-and-forget
// exchange is not ordered before the ticket by a workgroup-scope fence, which is `s_waitcnt lgkmcnt(0)` on this target:
// 1-8 % of the launches at 50 000 x 256 gave one wrong group) and cost 8 atomics per lane and record (a 160 000-row
// reduction spent most of its 93 us in them); a device-scope FENCE (__threadfence) writes back the whole L2 of the XCD and
// serialised the launch (50 us for 10 MB).
__device__ __forceinline__ void seg_rec_store(seg_rec* p, const seg_rec& r) {
  typedef float seg_f4 __attribute__((ext_vector_type(4)));
  typedef int seg_i4 __attribute__((ext_vector_type(4)));
  const seg_f4 v = {r.v.x, r.v.y, r.v.z, r.v.w};
  const seg_i4 a = {r.a.x, r.a.y, r.a.z, r.a.w};
  asm volatile(
      "global_store_dwordx4 %0, %1, off sc1\n\t"
      "global_store_dwordx4 %0, %2, off offset:16 sc1\n\t"
      "s_waitcnt vmcnt(0)"
      :
      : "v"(p), "v"(v), "v"(a)
      : "memory");
}
__device__ __forceinline__ seg_rec seg_rec_load(const seg_rec* p) {
  typedef float seg_f4 __attribute__((ext_vector_type(4)));
  typedef int seg_i4 __attribute__((ext_vector_type(4)));
  seg_f4 v;
  seg_i4 a;
  asm volatile(
      "global_load_dwordx4 %0, %2, off sc1\n\t"
      "global_load_dwordx4 %1, %2, off offset:16 sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(v), "=&v"(a)
      : "v"(p)
      : "memory");
  seg_rec r;
  r.v = make_float4(v.x, v.y, v.z, v.w);
  r.a = make_int4(a.x, a.y, a.z, a.w);
  return r;
}

// (v, a) <- merge((v, a), (w, b)); MAX: larger value, on ties the smaller row index (order-independent)
__device__ __forceinline__ void seg_merge(int mode, float4& v, int4& a, const float4& w, const int4& b) {
  if (mode == SST_REDUCE_MAX) {
    if (w.x > v.x || (w.x == v.x && b.x < a.x)) v.x = w.x, a.x = b.x;
    if (w.y > v.y || (w.y == v.y && b.y < a.y)) v.y = w.y, a.y = b.y;
    if (w.z > v.z || (w.z == v.z && b.z < a.z)) v.z = w.z, a.z = b.z;
    if (w.w > v.w || (w.w == v.w && b.w < a.w)) v.w = w.w, a.w = b.w;
  } else {
    v.x += w.x, v.y += w.y, v.z += w.z, v.w += w.w;
    a.x += b.x;                   // SUM / MEAN: a.x counts the rows merged so far
  }
}

template <int V>                  // channel vector width: 4 (16-byte accesses) or 1
__device__ __forceinline__ void seg_write(int mode, float* __restrict__ out, int32_t* __restrict__ argmax, int c, int64_t g,
                                          int ch, float4 v, int4 a) {
  if (mode == SST_REDUCE_MEAN) {
    const float cnt = (float)a.x;   // rows of the group (every row was merged exactly once)
    v.x = v.x / cnt, v.y = v.y / cnt, v.z = v.z / cnt, v.w = v.w / cnt;
  }
  if (V == 4) {
    *(float4*)(out + g * c + ch) = v;
    if (mode == SST_REDUCE_MAX && argmax != nullptr) *(int4*)(argmax + g * c + ch) = a;
  } else {
    out[g * c + ch] = v.x;
    if (mode == SST_REDUCE_MAX && argmax != nullptr) argmax[g * c + ch] = a.x;
  }
}

template <int V>
__global__ __launch_bounds__(256) void seg_tiles_k(const float* __restrict__ feats, int c, int cv, int lanes,
                                                   const uint32_t* __restrict__ perm,
                                                   const int32_t* __restrict__ inverse, int inverse_shift,
                                                   const int32_t* __restrict__ offsets, int32_t n, int mode,
                                                   float* __restrict__ out, int32_t* __restrict__ argmax,
                                                   seg_rec* __restrict__ recs, int32_t* __restrict__ counters, int dbg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char seg_smem[];
  seg_rec* lrec = (seg_rec*)seg_smem;                               // [lanes][2][cv]
  int* lgrp = (int*)(seg_smem + sizeof(seg_rec) * lanes * 2 * cv);  // [lanes][2]
  __shared__ int tile_grp[2];   // groups of the tile's two boundary records (kSegNone: none)
  __shared__ int finish[3];     // this workgroup finishes the group: flag, first tile, last tile
  const int tid = threadIdx.x;
  const int q = tid % cv, lane = tid / cv;
  const int ch = V * q;
  const int tile_rows = lanes * kSegSpan;
  const int64_t t0 = (int64_t)blockIdx.x * tile_rows;
  const int64_t t1 = t0 + tile_rows < n ? t0 + tile_rows : n;
  const float4 ident = mode == SST_REDUCE_MAX ? make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY)
                                              : make_float4(0.f, 0.f, 0.f, 0.f);
  const int4 noarg = mode == SST_REDUCE_MAX ? make_int4(n, n, n, n) : make_int4(0, 0, 0, 0);
  __shared__ int edge_grp[2];   // groups of the sorted rows right before / right behind the tile (kSegNone: none)
  if (tid < 2) {
    tile_grp[tid] = kSegNone;
    const int64_t pos = tid == 0 ? t0 - 1 : t1;
    edge_grp[tid] = (pos >= 0 && pos < n) ? inverse[perm[pos]] + inverse_shift : kSegNone;
  }
  if (lane < lanes) {
    const int64_t s0 = t0 + (int64_t)lane * kSegSpan;
    const int cnt = s0 >= n ? 0 : (int)(n - s0 < kSegSpan ? n - s0 : kSegSpan);
    uint32_t rows[kSegSpan];
    int grp[kSegSpan];
    float4 x[kSegSpan];
#pragma unroll
    for (int i = 0; i < kSegSpan; ++i) rows[i] = cnt > 0 ? perm[s0 + (i < cnt ? i : cnt - 1)] : 0u;
#pragma unroll
    for (int i = 0; i < kSegSpan; ++i) {
      grp[i] = cnt > 0 ? inverse[rows[i]] + inverse_shift : kSegNone;
      x[i] = ident;
      if (cnt > 0) {
        if (V == 4)
          x[i] = *(const float4*)(feats + (int64_t)rows[i] * c + ch);
        else
          x[i].x = feats[(int64_t)rows[i] * c + ch];
      }
    }
    // runs of equal group id inside the span: run 0 -> record 0, the last run (if there is a second one) -> record 1,
    // the runs between them are whole groups
    int cur = grp[0], n_closed = 0, first_grp = kSegNone;
    float4 acc = ident;
    int4 arg = noarg;
    seg_rec first;
    first.v = ident, first.a = noarg;
#pragma unroll
    for (int i = 0; i < kSegSpan; ++i) {
      if (i < cnt) {
        if (grp[i] != cur) {   // the run `cur` just ended
          if (n_closed == 0) {
            first.v = acc, first.a = arg, first_grp = cur;
          } else if (cur >= 0) {
            seg_write<V>(mode, out, argmax, c, cur, ch, acc, arg);
          }
          ++n_closed;
          cur = grp[i];
          acc = ident;
          arg = noarg;
        }
        const int r = mode == SST_REDUCE_MAX ? (int)rows[i] : 1;
        seg_merge(mode, acc, arg, x[i], make_int4(r, r, r, r));
      }
    }
    seg_rec last;
    last.v = acc, last.a = arg;
    int last_grp = cur;
    if (cnt == 0) {
      first_grp = last_grp = kSegNone;
    } else if (n_closed == 0) {   // the span is one run: its only record is record 0
      first = last;
      first_grp = cur;
      last_grp = kSegNone;
    }
    lrec[(lane * 2 + 0) * cv + q] = first;
    lrec[(lane * 2 + 1) * cv + q] = last;
    if (q == 0) {
      lgrp[lane * 2 + 0] = first_grp;
      lgrp[lane * 2 + 1] = last_grp;
    }
  }
  __syncthreads();
  if (dbg == 1) return;
  // ---- lane 0's threads merge the lane records of their channel vector in row order ----
  if (lane == 0) {
    int cur = kSegNone;
    float4 acc = ident;
    int4 arg = noarg;
    bool first_run = true;
    const int before = edge_grp[0], behind = edge_grp[1];
    auto close_run = [&](bool last_run) {
      const bool was_first = first_run;
      first_run = false;
      if (cur < 0) return;                 // nothing yet, or rows of a discarded group
      // the groups are contiguous in sorted order: a run goes on outside the tile iff the neighbouring row has its id
      const bool starts_here = !(was_first && cur == before), ends_here = !(last_run && cur == behind);
      if (starts_here && ends_here) {
        seg_write<V>(mode, out, argmax, c, cur, ch, acc, arg);
      } else {
        const int slot = starts_here ? 1 : 0;    // began before the tile: record 0; begins here and goes on: record 1
        seg_rec r;
        r.v = acc, r.a = arg;
        seg_rec_store(recs + ((int64_t)blockIdx.x * 2 + slot) * cv + q, r);
        if (q == 0) tile_grp[slot] = cur;
      }
    };
    for (int l = 0; l < lanes; ++l) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int g = lgrp[l * 2 + h];
        if (g == kSegNone) continue;
        const seg_rec r = lrec[(l * 2 + h) * cv + q];
        if (cur == kSegNone) {
          cur = g;
          acc = r.v;
          arg = r.a;
        } else if (g != cur) {
          close_run(false);
          cur = g;
          acc = r.v;
          arg = r.a;
        } else {
          seg_merge(mode, acc, arg, r.v, r.a);
        }
      }
    }
    if (cur != kSegNone) close_run(true);
  }
  // ---- groups that cross tiles: the last of their tiles to arrive merges their records ----
  if (dbg == 2) return;
  // every record store above waited for its own completion (seg_rec_store); the barrier then orders all of them, in every
  // thread, before the tickets are taken
  __syncthreads();
#pragma unroll 1
  for (int slot = 0; slot < 2; ++slot) {
    const int g = tile_grp[slot];
    if (g < 0) continue;   // uniform
    if (tid == 0) {
      const int ta = offsets[g] / tile_rows, tb = (offsets[g + 1] - 1) / tile_rows;
      const int before = atomicAdd(counters + g, 1);
      const int fin = before == tb - ta;       // tb - ta + 1 tiles hold a record of g
      if (fin) atomicExch(counters + g, 0);    // every other tile of g has already taken its ticket
      finish[0] = fin, finish[1] = ta, finish[2] = tb;
    }
    __syncthreads();
    if (finish[0]) {
      const int ta = finish[1], tb = finish[2];
      float4 acc = ident;
      int4 arg = noarg;
      if (lane < lanes) {
        for (int t = ta + lane; t <= tb; t += lanes) {
          const seg_rec r = seg_rec_load(recs + ((int64_t)t * 2 + (t == ta ? 1 : 0)) * cv + q);
          seg_merge(mode, acc, arg, r.v, r.a);
        }
        seg_rec r;
        r.v = acc, r.a = arg;
        lrec[lane * cv + q] = r;
      }
      __syncthreads();
      if (lane == 0) {
        for (int l = 1; l < lanes; ++l) {
          const seg_rec r = lrec[l * cv + q];
          seg_merge(mode, acc, arg, r.v, r.a);
        }
        seg_write<V>(mode, out, argmax, c, g, ch, acc, arg);
      }
    }
    __syncthreads();
  }
}

// SUM / MEAN backward: one thread per (point, channel)
__global__ __launch_bounds__(256) void seg_reduce_bwd_add_k(const float* __restrict__ gout, int c,
                                                            const int32_t* __restrict__ inverse, int shift,
                                                            const int32_t* __restrict__ offsets,
                                                            const int32_t* __restrict__ gidx, int64_t m,
                                                            int64_t n, int mode, float* __restrict__ gfeats,
                                                            const int32_t* __restrict__ d_mlim) {
  if (d_mlim != nullptr && (int64_t)*d_mlim < m) m = *d_mlim;
  const int64_t total = n * c;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e / c;
    const int ch = (int)(e - i * c);
    const int g = inverse[i] + shift;
    float v = 0.f;
    if (g >= 0 && g < m) {
      v = gout[(int64_t)g * c + ch];
      if (mode == SST_REDUCE_MEAN) {
        const int gs = gidx != nullptr ? gidx[g] : g;
        v = v / (float)(offsets[gs + 1] - offsets[gs]);
      }
    }
    gfeats[e] = v;
  }
}

// MAX backward: one thread per (group, channel) routes its gradient to the recorded argmax row.
__global__ __launch_bounds__(256) void seg_reduce_bwd_max_k(const float* __restrict__ gout, int c,
                                                            const int32_t* __restrict__ argmax, int64_t m,
                                                            int64_t n, float* __restrict__ gfeats,
                                                            const int32_t* __restrict__ d_mlim) {
  if (d_mlim != nullptr && (int64_t)*d_mlim < m) m = *d_mlim;
  const int64_t total = m * c;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int ch = (int)(e % c);
    const int32_t row = argmax[e];
    if (row >= 0 && row < n) gfeats[(int64_t)row * c + ch] = gout[e];
  }
}

__global__ __launch_bounds__(256) void gather_rows_k(const float* __restrict__ src, int64_t ld_src,
                                                     const int32_t* __restrict__ idx, int64_t n_out, int c,
                                                     float fill, float* __restrict__ out, int64_t ld_out) {
  const int64_t total = n_out * c;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e / c;
    const int ch = (int)(e - i * c);
    const int32_t r = idx[i];
    out[i * ld_out + ch] = (r >= 0) ? src[(int64_t)r * ld_src + ch] : fill;
  }
}

// out[i, :] = x[i, :] + table[idx[i], :] (c % 4 == 0): the first encoder layer's q / k input, feat + pos_embed
// (sst_basic_block_v2.py:62-66), as one pass instead of index cast + index_select + add
__global__ __launch_bounds__(256) void add_table_rows_k(const float* __restrict__ x, int64_t ldx, const float* __restrict__ table,
                                                        int64_t ldt, const int32_t* __restrict__ idx, int64_t m, int c4,
                                                        float* __restrict__ out, int64_t ldo) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int64_t total = m * c4;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e / c4;
    const int q = (int)(e - i * c4);
    const f4 a = *(const f4*)(x + i * ldx + 4 * q);
    const f4 b = *(const f4*)(table + (int64_t)idx[i] * ldt + 4 * q);
    *(f4*)(out + i * ldo + 4 * q) = a + b;
  }
}

__global__ __launch_bounds__(256) void scatter_rows_k(const float* __restrict__ src, int64_t ld_src,
                                                      const int32_t* __restrict__ idx, int64_t n_src, int c,
                                                      float* __restrict__ out, int64_t ld_out) {
  const int64_t total = n_src * c;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e / c;
    const int ch = (int)(e - i * c);
    const int32_t r = idx[i];
    if (r >= 0) out[(int64_t)r * ld_out + ch] = src[i * ld_src + ch];
  }
}

// (f1) decorate step of DynamicVFE / DynamicScatterVFE (voxel_encoders/voxel_encoder.py:252-271, 569-589) in one pass:
// out[i] = [ point features (c) | xyz - mean of its voxel, optionally / cluster_scale (3) | xyz - centre of its voxel (3) ]
// replacing ~15 element-wise launches, a gather and the `cat`.  Same fp32 operations in the same order
// (coordinate * voxel + offset with separate roundings), so the result is bit-identical to the composed ops.
template <typename CT>
__global__ __launch_bounds__(256) void vfe_decorate_k(const float* __restrict__ pts, int64_t ldp, int64_t n, int c,
                                                      const int32_t* __restrict__ inv, const float* __restrict__ mean,
                                                      int64_t ldm, float cluster_div, const CT* __restrict__ coors,
                                                      int64_t ldc, float vx, float vy, float vz, float ox, float oy,
                                                      float oz, int with_cluster, int with_center,
                                                      float* __restrict__ out, int64_t ldo) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float* p = pts + i * ldp;
    float* o = out + i * ldo;
    for (int k = 0; k < c; ++k) o[k] = p[k];
    int col = c;
    const float x = p[0], y = p[1], z = p[2];
    if (with_cluster) {
      int g = inv[i];
      if (g < 0) g = 0;  // a point without a voxel reads voxel 0 (the reference's zero-initialised canvas)
      const float* mu = mean + (int64_t)g * ldm;
      float fx = __fsub_rn(x, mu[0]), fy = __fsub_rn(y, mu[1]), fz = __fsub_rn(z, mu[2]);
      if (cluster_div != 1.f) {
        fx = __fdiv_rn(fx, cluster_div);
        fy = __fdiv_rn(fy, cluster_div);
        fz = __fdiv_rn(fz, cluster_div);
      }
      o[col] = fx;
      o[col + 1] = fy;
      o[col + 2] = fz;
      col += 3;
    }
    if (with_center) {
      const CT* cc = coors + i * ldc;  // (b, z, y, x)
      o[col] = __fsub_rn(x, __fadd_rn(__fmul_rn((float)cc[3], vx), ox));
      o[col + 1] = __fsub_rn(y, __fadd_rn(__fmul_rn((float)cc[2], vy), oy));
      o[col + 2] = __fsub_rn(z, __fadd_rn(__fmul_rn((float)cc[1], vz), oz));
    }
  }
}

// (a15 / f1) "features = cat([point_feats, voxel_feats[point -> voxel]], dim=1)" between the layers of DynamicVFE /
// DynamicScatterVFE / SIRLayer (voxel_encoders/voxel_encoder.py:288-291, :605-607, :745-750) in one pass: a gather by the
// inverse map and a concatenation become one coalesced kernel, the gathered [N, C] tensor is never materialised.
// idx < 0 reads group 0 (the zero-initialised canvas of DynamicVFE.map_voxel_center_to_point).
__global__ __launch_bounds__(256) void concat_gather_k(const float* __restrict__ x, int64_t ldx, int c1,
                                                       const float* __restrict__ g, int64_t ldg, int c2,
                                                       const int32_t* __restrict__ idx, int64_t n,
                                                       float* __restrict__ out) {
  const int q1 = c1 >> 2, q = (c1 + c2) >> 2;
  const int64_t total = n * q;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e / q;
    const int k = (int)(e - i * q);
    float4 v;
    if (k < q1) {
      v = *(const float4*)(x + i * ldx + 4 * k);
    } else {
      int r = idx[i];
      if (r < 0) r = 0;
      v = *(const float4*)(g + (int64_t)r * ldg + 4 * (k - q1));
    }
    *(float4*)(out + e * 4) = v;
  }
}

// (a14) recover_bev (mmdet3d/models/backbones/sst_v2.py:161-197): voxel features -> dense bird's-eye-view canvas.
// The reference loops over the samples, allocates a zero canvas per sample and assigns canvas[:, y * nx + x] = feat^T
// (a transposed, 4-byte-granular scatter), then stacks.  Here: (1) the voxel index of every occupied cell goes into an
// int32 cell map (0.9 MB per Waymo sample); (2) ONE pass writes every cell of the canvas exactly once - the voxel's row
// or zeros - in channels-last order (a cell's C channels are contiguous: 16-byte stores, every row read once), so
// there is no separate zero-fill of the 112 MB canvas.  The canvas is handed on as the channels-last view of the
// logical [B, C, ny, nx] tensor.
template <typename CT>
__global__ __launch_bounds__(256) void bev_map_k(const CT* __restrict__ coors, int64_t ldc, int64_t m, int ny, int nx,
                                                 int batch, int32_t* __restrict__ map) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x) {
    const CT* r = coors + i * ldc;  // (b, z, y, x)
    const int64_t b = (int64_t)r[0], y = (int64_t)r[2], x = (int64_t)r[3];
    if (b >= 0 && b < batch && y >= 0 && y < ny && x >= 0 && x < nx) map[(b * ny + y) * nx + x] = (int32_t)i;
  }
}

__global__ __launch_bounds__(256) void bev_fill_k(const float* __restrict__ feats, int64_t ldf,
                                                  const int32_t* __restrict__ map, int64_t cells, int c4,
                                                  float* __restrict__ out) {
  const int64_t total = cells * c4;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t cell = e / c4;
    const int q = (int)(e - cell * c4);
    const int32_t v = map[cell];
    float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
    if (v >= 0) val = *(const float4*)(feats + (int64_t)v * ldf + 4 * q);
    *(float4*)(out + (cell * c4 + q) * 4) = val;
  }
}

// gradient: rows of the canvas gradient (channels-last) at the voxels' cells
__global__ __launch_bounds__(256) void bev_gather_k(const float* __restrict__ gcanvas, const int32_t* __restrict__ map_of_voxel,
                                                    int64_t m, int c4, float* __restrict__ gfeats) {
  const int64_t total = m * c4;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e / c4;
    const int q = (int)(e - i * c4);
    const int32_t cell = map_of_voxel[i];
    float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
    if (cell >= 0) val = *(const float4*)(gcanvas + ((int64_t)cell * c4 + q) * 4);
    *(float4*)(gfeats + (i * c4 + q) * 4) = val;
  }
}

template <typename CT>
__global__ __launch_bounds__(256) void bev_cell_of_voxel_k(const CT* __restrict__ coors, int64_t ldc, int64_t m, int ny,
                                                           int nx, int batch, const int32_t* __restrict__ map,
                                                           int32_t* __restrict__ cell_of_voxel) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x) {
    const CT* r = coors + i * ldc;
    const int64_t b = (int64_t)r[0], y = (int64_t)r[2], x = (int64_t)r[3];
    int32_t cell = -1;
    if (b >= 0 && b < batch && y >= 0 && y < ny && x >= 0 && x < nx) {
      const int64_t cidx = (b * ny + y) * nx + x;
      if (map[cidx] == (int32_t)i) cell = (int32_t)cidx;  // a cell written by several voxels belongs to the one that won
    }
    cell_of_voxel[i] = cell;
  }
}

}  // namespace

extern "C" {

int sst_concat_gather_f32(const float* d_x, int64_t ldx, int c1, const float* d_g, int64_t ldg, int c2,
                          const int32_t* d_idx, int64_t n, float* d_out, void* stream) {
  if (n < 0 || c1 < 4 || c2 < 4 || (c1 & 3) || (c2 & 3) || ldx < c1 || ldg < c2 || (ldx & 3) || (ldg & 3)) return SST_ERR_ARG;
  if (n == 0) return SST_OK;
  if (!d_x || !d_g || !d_idx || !d_out) return SST_ERR_ARG;
  if (((uintptr_t)d_x | (uintptr_t)d_g | (uintptr_t)d_out) & 15) return SST_ERR_ARG;
  hipLaunchKernelGGL(concat_gather_k, dim3(sst_grid_1d(n * ((c1 + c2) >> 2), 256)), dim3(256), 0, (hipStream_t)stream,
                     d_x, ldx, c1, d_g, ldg, c2, d_idx, n, d_out);
  SST_LAUNCH_CHECK();
  return SST_OK;
}

int sst_recover_bev_f32(const float* d_feats, int64_t ldf, const void* d_coors, int coor_is_i64, int64_t ldc, int64_t m,
                        int batch, int ny, int nx, int c, int32_t* d_cell_map, int32_t* d_cell_of_voxel, float* d_canvas,
                        void* stream) {
  if (m < 0 || batch < 1 || ny < 1 || nx < 1 || c < 4 || (c & 3) || ldf < c) return SST_ERR_ARG;
  const int64_t cells = (int64_t)batch * ny * nx;
  if (cells * c >= ((int64_t)1 << 40) || cells >= ((int64_t)1 << 31) || m >= ((int64_t)1 << 31)) return SST_ERR_UNSUPPORTED;
  if (!d_cell_map || !d_canvas || (m > 0 && (!d_feats || !d_coors || !d_cell_of_voxel))) return SST_ERR_ARG;
  if ((((uintptr_t)d_feats | (uintptr_t)d_canvas) & 15) || (ldf & 3)) return SST_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  SST_HIP(hipMemsetAsync(d_cell_map, 0xff, sizeof(int32_t) * (size_t)cells, st));
  if (m > 0) {
    const int g = sst_grid_1d(m, 256);
    if (coor_is_i64) {
      hipLaunchKernelGGL(bev_map_k<int64_t>, dim3(g), dim3(256), 0, st, (const int64_t*)d_coors, ldc, m, ny, nx, batch,
                         d_cell_map);
      hipLaunchKernelGGL(bev_cell_of_voxel_k<int64_t>, dim3(g), dim3(256), 0, st, (const int64_t*)d_coors, ldc, m, ny, nx,
                         batch, d_cell_map, d_cell_of_voxel);
    } else {
      hipLaunchKernelGGL(bev_map_k<int32_t>, dim3(g), dim3(256), 0, st, (const int32_t*)d_coors, ldc, m, ny, nx, batch,
                         d_cell_map);
      hipLaunchKernelGGL(bev_cell_of_voxel_k<int32_t>, dim3(g), dim3(256), 0, st, (const int32_t*)d_coors, ldc, m, ny, nx,
                         batch, d_cell_map, d_cell_of_voxel);
    }
  }
  hipLaunchKernelGGL(bev_fill_k, dim3(sst_grid_1d(cells * (c >> 2), 256)), dim3(256), 0, st, d_feats, ldf, d_cell_map,
                     cells, c >> 2, d_canvas);
  SST_LAUNCH_CHECK();
  return SST_OK;
}

int sst_recover_bev_bwd_f32(const float* d_grad_canvas, const int32_t* d_cell_of_voxel, int64_t m, int c,
                            float* d_grad_feats, void* stream) {
  if (m < 0 || c < 4 || (c & 3)) return SST_ERR_ARG;
  if (m == 0) return SST_OK;
  if (!d_grad_canvas || !d_cell_of_voxel || !d_grad_feats) return SST_ERR_ARG;
  hipLaunchKernelGGL(bev_gather_k, dim3(sst_grid_1d(m * (c >> 2), 256)), dim3(256), 0, (hipStream_t)stream, d_grad_canvas,
                     d_cell_of_voxel, m, c >> 2, d_grad_feats);
  SST_LAUNCH_CHECK();
  return SST_OK;
}

int sst_vfe_decorate_f32(const float* d_points, int64_t ldp, int64_t n, int c, const int32_t* d_inverse,
                         const float* d_voxel_mean, int64_t ldm, float cluster_div, const void* d_coors,
                         int coor_is_i64, int64_t ldc, const float* voxel_size, const float* offsets,
                         int with_cluster, int with_center, float* d_out, int64_t ldo, void* stream) {
  if (n < 0 || c < 3 || ldp < c || ldo < c + 3 * (with_cluster != 0) + 3 * (with_center != 0)) return SST_ERR_ARG;
  if (n == 0) return SST_OK;
  if (!d_points || !d_out || (with_cluster && (!d_inverse || !d_voxel_mean || ldm < 3 || cluster_div == 0.f)) ||
      (with_center && (!d_coors || ldc < 4 || !voxel_size || !offsets)))
    return SST_ERR_ARG;
  const float vx = with_center ? voxel_size[0] : 0.f, vy = with_center ? voxel_size[1] : 0.f,
              vz = with_center ? voxel_size[2] : 0.f;
  const float ox = with_center ? offsets[0] : 0.f, oy = with_center ? offsets[1] : 0.f,
              oz = with_center ? offsets[2] : 0.f;
  const int grid = sst_grid_1d(n, 256);
  if (coor_is_i64)
    hipLaunchKernelGGL(vfe_decorate_k<int64_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, d_points, ldp, n, c,
                       d_inverse, d_voxel_mean, ldm, cluster_div, (const int64_t*)d_coors, ldc, vx, vy, vz, ox, oy, oz,
                       with_cluster, with_center, d_out, ldo);
  else
    hipLaunchKernelGGL(vfe_decorate_k<int32_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, d_points, ldp, n, c,
                       d_inverse, d_voxel_mean, ldm, cluster_div, (const int32_t*)d_coors, ldc, vx, vy, vz, ox, oy, oz,
                       with_cluster, with_center, d_out, ldo);
  SST_LAUNCH_CHECK();
  return SST_OK;
}

// one-shot kernel-bound events for the next forward launch on this thread (bench.py's roofline block of the FSD workloads;
// same scheme as sst_sra_attn_profile_next_fwd: hipExtLaunchKernelGGL start / stop events)
static thread_local hipEvent_t g_seg_ev[2] = {nullptr, nullptr};
int sst_segment_reduce_profile_next(void* start, void* stop) {
  g_seg_ev[0] = (hipEvent_t)start;
  g_seg_ev[1] = (hipEvent_t)stop;
  return SST_OK;
}

// capacities of a work list for n rows in m groups of c channels
static void seg_work_shape(int64_t n, int64_t m, int c, seg_work* W) {
  const int64_t chunks = n / kSegChunk + 1;
  W->cap_e = (int)sst_align_up(m + chunks, 4);   // multiples of 4: the partial records behind the lists stay 16-byte aligned
  W->cap_m = (int)sst_align_up(chunks, 4);
  W->cap_p = (int)(2 * chunks);
  W->cpad = (c + 3) & ~3;
}
int64_t sst_segment_reduce_work_words(int64_t n, int64_t m, int c) {
  if (n < 0 || m < 0 || c < 1 || n > 0x3fffffff || m > 0x3fffffff) return SST_ERR_ARG;
  seg_work W;
  seg_work_shape(n, m, c, &W);
  return kSegWorkHdr + 3 * (int64_t)W.cap_e + 3 * (int64_t)W.cap_m + 2 * (int64_t)W.cap_p * W.cpad;
}

int sst_segment_reduce_fwd_work_f32(const float* d_feats, int64_t n, int c, const uint32_t* d_perm,
                                    const int32_t* d_offsets, const int32_t* d_group_index, int64_t m, int mode,
                                    float* d_out, int32_t* d_argmax, const int32_t* d_m_limit, int32_t* d_work,
                                    int64_t work_capacity, const float* d_scale, const float* d_shift, void* stream) {
  if (n < 0 || m < 0 || c < 1 || mode < 0 || mode > 2) return SST_ERR_ARG;
  if (m == 0) return SST_OK;
  if (!d_offsets || !d_out || (n > 0 && (!d_feats || !d_perm))) return SST_ERR_ARG;
  if (d_work != nullptr && (work_capacity < sst_segment_reduce_work_words(n, m, c) || (((uintptr_t)d_work) & 15)))
    return SST_ERR_ARG;
  if ((d_scale != nullptr) != (d_shift != nullptr)) return SST_ERR_ARG;
  const seg_xf xf{d_scale, d_shift};
  const seg_xf no_xf{nullptr, nullptr};
  seg_work work{nullptr, 0, 0, 0, 0};
  const seg_work no_work{nullptr, 0, 0, 0, 0};
  if (d_work != nullptr) {
    seg_work_shape(n, m, c, &work);
    work.w = d_work;
  }
  hipEvent_t e0 = g_seg_ev[0], e1 = g_seg_ev[1];
  g_seg_ev[0] = g_seg_ev[1] = nullptr;
  const bool timed = e0 != nullptr && e1 != nullptr;
  const bool v4 = (c & 3) == 0 && (((uintptr_t)d_feats | (uintptr_t)d_out | (uintptr_t)d_argmax) & 15) == 0 && n > 0;
  // the transform exists in the float4 kernels with the work list only (what the voxel encoders call)
  if (d_scale != nullptr && (!v4 || d_work == nullptr || (((uintptr_t)d_scale | (uintptr_t)d_shift) & 15))) return SST_ERR_UNSUPPORTED;
  constexpr int kLongGroup = 16;       // without a work list: groups longer than this go to one workgroup each when n >= 8 m
  constexpr int kWorkGroupLen = 32;    // with a work list: groups longer than this, whatever the average
  constexpr int kWorkGrid = 512, kMergeGrid = 64;
  hipStream_t st = (hipStream_t)stream;
  if (d_work != nullptr && n > 0) {
    // robust form: the element kernel lists the long groups (cut into chunks of kSegChunk rows), a fixed grid of workgroups
    // reduces the chunks, a third launch merges the chunks of the groups that were cut
    if (v4) {
      const int grid = sst_grid_1d(m * (c >> 2), 256);
      if (timed)
        hipExtLaunchKernelGGL(seg_reduce_fwd_v4_k, dim3(grid), dim3(256), 0, st, e0, nullptr, 0, d_feats, c, d_perm, d_offsets,
                              d_group_index, m, mode, d_out, d_argmax, (int32_t)n, d_m_limit, kWorkGroupLen, work, xf);
      else
        hipLaunchKernelGGL(seg_reduce_fwd_v4_k, dim3(grid), dim3(256), 0, st, d_feats, c, d_perm, d_offsets, d_group_index, m,
                           mode, d_out, d_argmax, (int32_t)n, d_m_limit, kWorkGroupLen, work, xf);
      hipLaunchKernelGGL(seg_reduce_fwd_work_k<4>, dim3(kWorkGrid), dim3(256), 0, st, d_feats, c, d_perm, d_offsets,
                         d_group_index, mode, d_out, d_argmax, (int32_t)n, work, xf);
    } else {
      const int grid = sst_grid_1d(m * c, 256);
      if (timed)
        hipExtLaunchKernelGGL(seg_reduce_fwd_k, dim3(grid), dim3(256), 0, st, e0, nullptr, 0, d_feats, c, d_perm, d_offsets,
                              d_group_index, m, mode, d_out, d_argmax, (int32_t)n, d_m_limit, kWorkGroupLen, work);
      else
        hipLaunchKernelGGL(seg_reduce_fwd_k, dim3(grid), dim3(256), 0, st, d_feats, c, d_perm, d_offsets, d_group_index, m, mode,
                           d_out, d_argmax, (int32_t)n, d_m_limit, kWorkGroupLen, work);
      hipLaunchKernelGGL(seg_reduce_fwd_work_k<1>, dim3(kWorkGrid), dim3(256), 0, st, d_feats, c, d_perm, d_offsets,
                         d_group_index, mode, d_out, d_argmax, (int32_t)n, work, no_xf);
    }
    if (timed)
      hipExtLaunchKernelGGL(seg_reduce_merge_k, dim3(kMergeGrid), dim3(256), 0, st, nullptr, e1, 0, c, d_offsets, d_group_index,
                            mode, d_out, d_argmax, work);
    else
      hipLaunchKernelGGL(seg_reduce_merge_k, dim3(kMergeGrid), dim3(256), 0, st, c, d_offsets, d_group_index, mode, d_out,
                         d_argmax, work);
    SST_LAUNCH_CHECK();
    return SST_OK;
  }
  if (v4) {
    // groups of many points can only exist when the average is not tiny: the second kernel (one workgroup per LONG group,
    // every other block leaves at once) is launched when n >= 8 m - never for voxel grouping at 1-6 points per voxel
    const bool split = n >= 8 * m;
    const int grid = sst_grid_1d(m * (c >> 2), 256);
    if (timed)
      hipExtLaunchKernelGGL(seg_reduce_fwd_v4_k, dim3(grid), dim3(256), 0, st, e0, split ? nullptr : e1, 0, d_feats, c, d_perm,
                            d_offsets, d_group_index, m, mode, d_out, d_argmax, (int32_t)n, d_m_limit, split ? kLongGroup : 0,
                            no_work, no_xf);
    else
      hipLaunchKernelGGL(seg_reduce_fwd_v4_k, dim3(grid), dim3(256), 0, st, d_feats, c, d_perm, d_offsets, d_group_index, m,
                         mode, d_out, d_argmax, (int32_t)n, d_m_limit, split ? kLongGroup : 0, no_work, no_xf);
    if (split) {
      const int grid_b = (int)(m < 65536 ? m : 65536);
      if (timed)
        hipExtLaunchKernelGGL(seg_reduce_fwd_block_k, dim3(grid_b), dim3(256), 0, st, nullptr, e1, 0, d_feats, c, d_perm,
                              d_offsets, d_group_index, m, mode, d_out, d_argmax, (int32_t)n, d_m_limit, kLongGroup);
      else
        hipLaunchKernelGGL(seg_reduce_fwd_block_k, dim3(grid_b), dim3(256), 0, st, d_feats, c, d_perm, d_offsets, d_group_index,
                           m, mode, d_out, d_argmax, (int32_t)n, d_m_limit, kLongGroup);
    }
    SST_LAUNCH_CHECK();
    return SST_OK;
  }
  const int grid = sst_grid_1d(m * c, 256);
  if (timed)
    hipExtLaunchKernelGGL(seg_reduce_fwd_k, dim3(grid), dim3(256), 0, st, e0, e1, 0, d_feats, c, d_perm, d_offsets,
                          d_group_index, m, mode, d_out, d_argmax, (int32_t)n, d_m_limit, 0, no_work);
  else
    hipLaunchKernelGGL(seg_reduce_fwd_k, dim3(grid), dim3(256), 0, st, d_feats, c, d_perm, d_offsets, d_group_index, m, mode,
                       d_out, d_argmax, (int32_t)n, d_m_limit, 0, no_work);
  SST_LAUNCH_CHECK();
  return SST_OK;
}

int sst_segment_reduce_fwd_f32(const float* d_feats, int64_t n, int c, const uint32_t* d_perm,
                               const int32_t* d_offsets, const int32_t* d_group_index, int64_t m, int mode,
                               float* d_out, int32_t* d_argmax, const int32_t* d_m_limit, void* stream) {
  return sst_segment_reduce_fwd_work_f32(d_feats, n, c, d_perm, d_offsets, d_group_index, m, mode, d_out, d_argmax,
                                         d_m_limit, nullptr, 0, nullptr, nullptr, stream);
}

// geometry of seg_tiles_k for a width: channel vectors, row lanes per workgroup
static void seg_tiles_shape(int c, int64_t n, int* v, int* cv, int* lanes) {
  *v = (c % 4 == 0) ? 4 : 1;
  *cv = c / *v;
  *lanes = 256 / *cv;
  if (*lanes > 32) *lanes = 32;      // <= 256 rows per tile
  // narrow inputs (the 3-channel centroid means): fewer row lanes per tile while that is what it takes to give every CU a
  // tile - the merge of the lane records is serial in the lanes
  while (*lanes > 8 && n / ((int64_t)*lanes * kSegSpan) < 256) *lanes >>= 1;
}

int64_t sst_segment_long_scratch_bytes(int64_t n, int64_t m, int c) {
  if (n < 1 || m < 1 || c < 1 || c > 256) return 256;
  int v, cv, lanes;
  seg_tiles_shape(c, n, &v, &cv, &lanes);
  const int64_t tiles = sst_div_up(n, (int64_t)lanes * kSegSpan);
  return sst_align_up(m * 4, 256) + sst_align_up(tiles * 2 * cv * (int64_t)sizeof(seg_rec), 256);
}

int sst_segment_reduce_long_f32(const float* d_feats, int64_t n, int c, const uint32_t* d_perm, const int32_t* d_inverse,
                                int inverse_shift, const int32_t* d_offsets, int64_t m, int mode, void* d_scratch,
                                float* d_out, int32_t* d_argmax, void* stream) {
  if (n < 0 || m < 0 || c < 1 || mode < 0 || mode > 2) return SST_ERR_ARG;
  if (n == 0 || m == 0) return SST_OK;
  if (!d_feats || !d_perm || !d_inverse || !d_offsets || !d_scratch || !d_out) return SST_ERR_ARG;
  if (c > 256 || n > 0x7fffffff || (((uintptr_t)d_scratch) & 255)) return SST_ERR_UNSUPPORTED;
  int v, cv, lanes;
  seg_tiles_shape(c, n, &v, &cv, &lanes);
  if (256 % cv != 0 && cv * lanes > 256) return SST_ERR_UNSUPPORTED;
  if (v == 4 && (((uintptr_t)d_feats | (uintptr_t)d_out | (uintptr_t)d_argmax) & 15)) return SST_ERR_UNSUPPORTED;
  const int64_t tiles = sst_div_up(n, (int64_t)lanes * kSegSpan);
  if (tiles > 0x3fffffff) return SST_ERR_UNSUPPORTED;
  int32_t* counters = (int32_t*)d_scratch;                   // [m], zeroed once by the caller, left zeroed
  seg_rec* recs = (seg_rec*)((char*)d_scratch + sst_align_up(m * 4, 256));
  const size_t lds = sizeof(seg_rec) * lanes * 2 * cv + sizeof(int) * lanes * 2;
  hipEvent_t e0 = g_seg_ev[0], e1 = g_seg_ev[1];
  g_seg_ev[0] = g_seg_ev[1] = nullptr;
  const bool timed = e0 != nullptr && e1 != nullptr;
  const char* dbg_env = getenv("SST_SEG_TILES_DEBUG_PHASE");     // developer switch: stop after phase 1 / 2 (timing only)
  const int dbg = dbg_env ? atoi(dbg_env) : 0;
#define SST_SEG_LAUNCH(V)                                                                                                \
  do {                                                                                                                   \
    if (timed)                                                                                                           \
      hipExtLaunchKernelGGL(seg_tiles_k<V>, dim3((unsigned)tiles), dim3(256), lds, (hipStream_t)stream, e0, e1, 0, d_feats, \
                            c, cv, lanes, d_perm, d_inverse, inverse_shift, d_offsets, (int32_t)n, mode, d_out, d_argmax,  \
                            recs, counters, dbg);                                                                        \
    else                                                                                                                 \
      hipLaunchKernelGGL(seg_tiles_k<V>, dim3((unsigned)tiles), dim3(256), lds, (hipStream_t)stream, d_feats, c, cv, lanes, \
                         d_perm, d_inverse, inverse_shift, d_offsets, (int32_t)n, mode, d_out, d_argmax, recs, counters,  \
                         dbg);                                                                                           \
  } while (0)
  if (v == 4)
    SST_SEG_LAUNCH(4);
  else
    SST_SEG_LAUNCH(1);
#undef SST_SEG_LAUNCH
  SST_LAUNCH_CHECK();
  return SST_OK;
}

int sst_segment_reduce_bwd_f32(const float* d_grad_out, int64_t m, int c, const int32_t* d_inverse,
                               int inverse_shift, const int32_t* d_offsets, const int32_t* d_group_index,
                               const int32_t* d_argmax, int64_t n, int mode, float* d_grad_feats,
                               const int32_t* d_m_limit, void* stream) {
  if (n < 0 || m < 0 || c < 1 || mode < 0 || mode > 2) return SST_ERR_ARG;
  if (n == 0) return SST_OK;
  if (!d_grad_feats) return SST_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (mode == SST_REDUCE_MAX) {
    SST_HIP(hipMemsetAsync(d_grad_feats, 0, sizeof(float) * n * c, st));
    if (m == 0) return SST_OK;
    if (!d_grad_out || !d_argmax) return SST_ERR_ARG;
    const int grid = sst_grid_1d(m * c, 256);
    hipLaunchKernelGGL(seg_reduce_bwd_max_k, dim3(grid), dim3(256), 0, st, d_grad_out, c, d_argmax, m, n,
                       d_grad_feats, d_m_limit);
  } else {
    if (m == 0) {
      SST_HIP(hipMemsetAsync(d_grad_feats, 0, sizeof(float) * n * c, st));
      return SST_OK;
    }
    if (!d_grad_out || !d_inverse || !d_offsets) return SST_ERR_ARG;
    const int grid = sst_grid_1d(n * c, 256);
    hipLaunchKernelGGL(seg_reduce_bwd_add_k, dim3(grid), dim3(256), 0, st, d_grad_out, c, d_inverse, inverse_shift,
                       d_offsets, d_group_index, m, n, mode, d_grad_feats, d_m_limit);
  }
  SST_LAUNCH_CHECK();
  return SST_OK;
}

int sst_gather_rows_f32(const float* d_src, int64_t ld_src, const int32_t* d_idx, int64_t n_out, int c, float fill,
                        float* d_out, int64_t ld_out, void* stream) {
  if (n_out < 0 || c < 1) return SST_ERR_ARG;
  if (n_out == 0) return SST_OK;
  if (!d_src || !d_idx || !d_out) return SST_ERR_ARG;
  const int grid = sst_grid_1d(n_out * c, 256);
  hipLaunchKernelGGL(gather_rows_k, dim3(grid), dim3(256), 0, (hipStream_t)stream, d_src, ld_src, d_idx, n_out, c,
                     fill, d_out, ld_out);
  SST_LAUNCH_CHECK();
  return SST_OK;
}

int sst_add_table_rows_f32(const float* d_x, int64_t ldx, const float* d_table, int64_t ld_table, const int32_t* d_idx, int64_t m,
                           int c, float* d_out, int64_t ld_out, void* stream) {
  if (m < 0 || c < 4 || (c & 3)) return SST_ERR_ARG;
  if (m == 0) return SST_OK;
  if (!d_x || !d_table || !d_idx || !d_out || (ldx & 3) || (ld_table & 3) || (ld_out & 3) ||
      (((uintptr_t)d_x | (uintptr_t)d_table | (uintptr_t)d_out) & 15))
    return SST_ERR_ARG;
  const int grid = sst_grid_1d(m * (c >> 2), 256);
  hipLaunchKernelGGL(add_table_rows_k, dim3(grid), dim3(256), 0, (hipStream_t)stream, d_x, ldx, d_table, ld_table, d_idx, m, c >> 2,
                     d_out, ld_out);
  SST_LAUNCH_CHECK();
  return SST_OK;
}

int sst_scatter_rows_f32(const float* d_src, int64_t ld_src, const int32_t* d_idx, int64_t n_src, int c,
                         float* d_out, int64_t ld_out, void* stream) {
  if (n_src < 0 || c < 1) return SST_ERR_ARG;
  if (n_src == 0) return SST_OK;
  if (!d_src || !d_idx || !d_out) return SST_ERR_ARG;
  const int grid = sst_grid_1d(n_src * c, 256);
  hipLaunchKernelGGL(scatter_rows_k, dim3(grid), dim3(256), 0, (hipStream_t)stream, d_src, ld_src, d_idx, n_src, c,
                     d_out, ld_out);
  SST_LAUNCH_CHECK();
  return SST_OK;
}

}  // extern "C"
