// Regional grouping (window / shifted-window bucketing of non-empty voxels) for gfx950.
//
// Reference semantics:
//   get_window_coors                      mmdet3d/ops/sst/sst_ops.py:266-314
//   SSTInputLayerV2.drop_single_shift     mmdet3d/models/middle_encoders/sst_input_layer_v2.py:128-148
//   SSTInputLayerV2.drop_voxel            sst_input_layer_v2.py:150-226
//   get_flat2win_inds / make_continuous_inds   sst_ops.py:26-64, 316-331
//   get_inner_win_inds (TorchEx ingroup_indices, order unspecified)   sst_ops.py:244-264
//
// Everything here is int32 work on M ~ 1e5 voxels and W ~ 1e3 windows: L2-resident, launch-bound.
// The in-window order is defined as "ascending voxel index" (stable radix sort by window id), which
// makes voxel drop deterministic; the reference leaves it unspecified.
#include "common.h"

// from sort_scan.hip
int sst_scan_i32_internal(const int32_t* d_in, int32_t* d_out, int64_t n, int32_t* d_total, void* ws, hipStream_t st);
int64_t sst_scan_ws_internal(int64_t n);
int64_t sst_unique_ws_internal(int64_t n);
int sst_unique_keys_internal(uint64_t* keys, uint64_t* keys_alt, int64_t n, int key_bits, uint32_t* d_perm_out,
                             int32_t* d_inverse, int32_t* d_offsets, uint64_t* d_ukeys, int32_t* d_num_unique,
                             void* ws, hipStream_t st);

namespace {

// Exclusive scan of one int per thread over a 256-thread block (all threads must call it).
__device__ __forceinline__ int block_excl_scan_256_w(int v, int& total, int* lds) {
  const int incl = sst_wave_incl_scan(v);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 63) lds[wave] = incl;
  __syncthreads();
  int woff = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const int s = lds[w];
    if (w < wave) woff += s;
    tot += s;
  }
  __syncthreads();
  total = tot;
  return woff + incl - v;
}

struct win_params {
  int wx, wy, wz;        // window shape
  int nwx, nwy, nwz;     // max windows per axis (ceil(s/w) + 1)
  int sh0[3];            // shift (x,y,z) for the non-shifted partition (= window shape; z 0 if sz == wz)
  int sh1[3];            // shift for the shifted partition (= window shape // 2)
};

template <typename T>
__global__ __launch_bounds__(256) void window_coors_k(const T* __restrict__ coors, int64_t m, win_params wp,
                                                      int32_t* __restrict__ win0, int32_t* __restrict__ ciw0,
                                                      int32_t* __restrict__ win1, int32_t* __restrict__ ciw1) {
  const int per_sample = wp.nwx * wp.nwy * wp.nwz;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x) {
    const T* r = coors + i * 4;
    const int b = (int)r[0], z = (int)r[1], y = (int)r[2], x = (int)r[3];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int* sh = s ? wp.sh1 : wp.sh0;
      const int xs = x + sh[0], ys = y + sh[1], zs = z + sh[2];
      const int wxi = xs / wp.wx, wyi = ys / wp.wy, wzi = zs / wp.wz;
      const int id = b * per_sample + wxi * wp.nwy * wp.nwz + wyi * wp.nwz + wzi;
      int32_t* w = s ? win1 : win0;
      int32_t* c = s ? ciw1 : ciw0;
      w[i] = id;
      c[i * 3 + 0] = zs - wzi * wp.wz;
      c[i * 3 + 1] = ys - wyi * wp.wy;
      c[i * 3 + 2] = xs - wxi * wp.wx;
    }
  }
}

struct level_table {
  int n;
  int cap[8];
  int lo[8];
  int hi[8];
};

__global__ void keys_from_i32_k(const int32_t* __restrict__ in, int64_t n, uint64_t* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = (uint64_t)(uint32_t)in[i];
}

// g[i] = flag[perm[i]] (or 1)
__global__ void gather_flag_k(const int32_t* __restrict__ flag, const uint32_t* __restrict__ perm, int64_t n,
                              int32_t* __restrict__ g) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    g[i] = (flag == nullptr) ? 1 : (flag[perm[i]] != 0 ? 1 : 0);
}

// c = exclusive scan of g in sorted order, c[n] = total.  For the voxel at sorted position i:
//   rank  = flagged voxels of the same segment that precede it, cnt = flagged voxels in its segment
__global__ void seg_rank_cnt_k(const int32_t* __restrict__ c, const uint32_t* __restrict__ perm,
                               const int32_t* __restrict__ inv, const int32_t* __restrict__ off, int64_t n,
                               int32_t* __restrict__ rank, int32_t* __restrict__ cnt) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t row = perm[i];
    const int seg = inv[row];
    const int cb = c[off[seg]], ce = c[off[seg + 1]];
    rank[row] = c[i] - cb;
    if (cnt != nullptr) cnt[row] = ce - cb;
  }
}

// drop_single_shift (sst_input_layer_v2.py:128-148): level by window population, keep iff rank < cap.
__global__ void drop_level_k(const int32_t* __restrict__ cnt, const int32_t* __restrict__ rank,
                             const int32_t* __restrict__ flag_in, const int32_t* __restrict__ inv, int64_t n,
                             level_table lt, int32_t* __restrict__ level, int32_t* __restrict__ keep,
                             int32_t* __restrict__ seglevel) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int lv = -1, kp = 0;
    if (flag_in == nullptr || flag_in[i] != 0) {
      const int c = cnt[i];
      int cap = 0;
      for (int l = 0; l < lt.n; ++l)
        if (c >= lt.lo[l] && c < lt.hi[l]) {  // later levels override earlier ones, as the reference loop does
          lv = l;
          cap = lt.cap[l];
        }
      kp = rank[i] < cap ? 1 : 0;
      seglevel[inv[i]] = lv;  // same value from every survivor of the window
    }
    level[i] = lv;
    keep[i] = kp;
  }
}

// Single block: for the non-empty windows of one shift, in ascending window-id (= segment) order,
//   tokbase[seg]   exclusive sum of survivors             -> CSR offsets of the token list
//   cwin[seg]      rank among non-empty windows           -> plan window id
//   cwl[seg]       rank among non-empty windows of the same drop level  (make_continuous_inds per level)
__global__ __launch_bounds__(256) void seg_plan_k(const int32_t* __restrict__ c, const int32_t* __restrict__ off,
                                                  const int32_t* __restrict__ nseg_p,
                                                  const int32_t* __restrict__ seglevel, int n_levels,
                                                  int32_t* __restrict__ tokbase, int32_t* __restrict__ cwl,
                                                  int32_t* __restrict__ winoff, int32_t* __restrict__ winlevel,
                                                  int32_t* __restrict__ n_windows_out,
                                                  int32_t* __restrict__ max_tokens_out) {
  __shared__ int lds[4];
  __shared__ int lds_max;
  if (threadIdx.x == 0) lds_max = 0;
  const int nseg = *nseg_p;
  int carry_tok = 0, carry_win = 0, my_max = 0;
  int carry_lvl[8];
#pragma unroll
  for (int l = 0; l < 8; ++l) carry_lvl[l] = 0;
  for (int base = 0; base < nseg; base += 256) {
    const int seg = base + threadIdx.x;
    int cnt = 0, lv = -1;
    if (seg < nseg) {
      cnt = c[off[seg + 1]] - c[off[seg]];
      lv = seglevel[seg];
    }
    const int nonempty = cnt > 0 ? 1 : 0;
    my_max = cnt > my_max ? cnt : my_max;
    int total;
    const int ex_tok = block_excl_scan_256_w(cnt, total, lds);
    const int tb = carry_tok + ex_tok;
    carry_tok += total;
    const int ex_win = block_excl_scan_256_w(nonempty, total, lds);
    const int cw = carry_win + ex_win;
    carry_win += total;
    int my_cwl = 0;
#pragma unroll
    for (int l = 0; l < 8; ++l) {
      if (l < n_levels) {
        const int f = (nonempty && lv == l) ? 1 : 0;
        const int ex = block_excl_scan_256_w(f, total, lds);
        if (f) my_cwl = carry_lvl[l] + ex;
        carry_lvl[l] += total;
      }
    }
    if (seg < nseg) {
      tokbase[seg] = tb;
      cwl[seg] = my_cwl;
      if (nonempty) {
        winoff[cw] = tb;
        winlevel[cw] = lv;
      }
    }
  }
  atomicMax(&lds_max, my_max);  // LDS, integer: the largest surviving window population of this shift
  __syncthreads();
  if (threadIdx.x == 0) {
    winoff[carry_win] = carry_tok;
    *n_windows_out = carry_win;
    *max_tokens_out = lds_max;
  }
}

__global__ void plan_fill_k(const int32_t* __restrict__ keep, const int32_t* __restrict__ inv,
                            const int32_t* __restrict__ inner, const int32_t* __restrict__ level,
                            const int32_t* __restrict__ newidx, const int32_t* __restrict__ tokbase,
                            const int32_t* __restrict__ cwl, level_table lt, int64_t n, int32_t* __restrict__ tok,
                            int32_t* __restrict__ flat2win) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int f2w = -1;
    if (keep[i]) {
      const int seg = inv[i];
      const int in = inner[i];
      tok[tokbase[seg] + in] = newidx[i];
      const int lv = level[i];
      const int cap = (lv >= 0 && lv < lt.n) ? lt.cap[lv] : 0;
      f2w = cwl[seg] * cap + in;
    }
    flat2win[i] = f2w;
  }
}

}  // namespace

// Launch order of the windows for the register-resident attention kernels (sst_amd/kernels.py WindowPlan.order): window ids by
// ascending token count, ties in id order = torch.sort(sizes, stable=True)[1].  One workgroup: wave q owns the q-th contiguous
// slice of the ids; per (wave, size) counts -> exclusive positions in (size, wave) order -> every wave places its ids chunk by
// chunk, the lanes of one size taking consecutive places by their rank in the ballot.  Replaces four small launches per
// partition and step (difference, sort, index cast ...) by one.
constexpr int kOrdWaves = 16, kOrdBins = 512;

__global__ __launch_bounds__(1024) void window_order_k(const int32_t* __restrict__ winoff, int n, int nbins,
                                                       int32_t* __restrict__ order) {
  __shared__ int hist[kOrdWaves][kOrdBins];   // counts, then next free position, per (wave, size)
  __shared__ int tot[kOrdBins];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < kOrdWaves * kOrdBins; i += 1024) (&hist[0][0])[i] = 0;
  __syncthreads();
  const int per = (n + kOrdWaves - 1) / kOrdWaves;
  const int w0 = wave * per, w1 = (w0 + per < n) ? w0 + per : n;
  for (int w = w0 + lane; w < w1; w += 64) {
    int s = winoff[w + 1] - winoff[w];
    s = s < 0 ? 0 : (s < nbins ? s : nbins - 1);
    atomicAdd(&hist[wave][s], 1);           // integer counts: the totals do not depend on the order of the adds
  }
  __syncthreads();
  for (int s = tid; s < nbins; s += 1024) {
    int t = 0;
    for (int q = 0; q < kOrdWaves; ++q) t += hist[q][s];
    tot[s] = t;
  }
  __syncthreads();
  if (wave == 0) {                          // exclusive scan of the per-size totals: 64 sizes at a time
    int carry = 0;
    for (int b = 0; b < nbins; b += 64) {
      const int s = b + lane;
      const int v = s < nbins ? tot[s] : 0;
      const int inc = sst_wave_incl_scan(v);
      if (s < nbins) tot[s] = carry + inc - v;
      carry += __shfl(inc, 63, 64);
    }
  }
  __syncthreads();
  for (int s = tid; s < nbins; s += 1024) {
    int run = tot[s];
    for (int q = 0; q < kOrdWaves; ++q) {
      const int c = hist[q][s];
      hist[q][s] = run;
      run += c;
    }
  }
  __syncthreads();
  volatile int* next = hist[wave];
  for (int c0 = w0; c0 < w1; c0 += 64) {
    const int w = c0 + lane;
    const bool valid = w < w1;
    int s = -1;
    if (valid) {
      s = winoff[w + 1] - winoff[w];
      s = s < 0 ? 0 : (s < nbins ? s : nbins - 1);
    }
    unsigned long long todo = __ballot(valid);
    while (todo) {
      const int leader = __ffsll((long long)todo) - 1;
      const int sl = __shfl(s, leader, 64);
      const unsigned long long same = __ballot(valid && s == sl);
      const int first = next[sl];
      if (valid && s == sl) order[first + __popcll(same & ((1ull << lane) - 1ull))] = w;
      if (lane == leader) next[sl] = first + __popcll(same);
      todo &= ~same;
    }
  }
}

extern "C" {

int sst_window_coors(const void* d_coors, int coor_is_i64, int64_t m, const int32_t sparse_shape[3],
                     const int32_t window_shape[3], int32_t* d_win0, int32_t* d_ciw0, int32_t* d_win1,
                     int32_t* d_ciw1, void* stream) {
  if (m < 0 || !sparse_shape || !window_shape) return SST_ERR_ARG;
  for (int a = 0; a < 3; ++a)
    if (sparse_shape[a] < 1 || window_shape[a] < 1) return SST_ERR_ARG;
  if (m == 0) return SST_OK;
  if (!d_coors || !d_win0 || !d_ciw0 || !d_win1 || !d_ciw1) return SST_ERR_ARG;
  win_params wp;
  wp.wx = window_shape[0];
  wp.wy = window_shape[1];
  wp.wz = window_shape[2];
  // max_num_win = ceil(sparse / win) + 1 (sst_ops.py:280-283)
  wp.nwx = (sparse_shape[0] + wp.wx - 1) / wp.wx + 1;
  wp.nwy = (sparse_shape[1] + wp.wy - 1) / wp.wy + 1;
  wp.nwz = (sparse_shape[2] + wp.wz - 1) / wp.wz + 1;
  wp.sh0[0] = wp.wx;
  wp.sh0[1] = wp.wy;
  wp.sh0[2] = wp.wz;
  wp.sh1[0] = wp.wx / 2;
  wp.sh1[1] = wp.wy / 2;
  wp.sh1[2] = wp.wz / 2;
  if (sparse_shape[2] == wp.wz) {  // "compatibility between 2D window and 3D window" (sst_ops.py:291-293)
    wp.sh0[2] = 0;
    wp.sh1[2] = 0;
  }
  const int grid = sst_grid_1d(m, 256);
  hipStream_t st = (hipStream_t)stream;
  if (coor_is_i64)
    hipLaunchKernelGGL(window_coors_k<int64_t>, dim3(grid), dim3(256), 0, st, (const int64_t*)d_coors, m, wp, d_win0,
                       d_ciw0, d_win1, d_ciw1);
  else
    hipLaunchKernelGGL(window_coors_k<int32_t>, dim3(grid), dim3(256), 0, st, (const int32_t*)d_coors, m, wp, d_win0,
                       d_ciw0, d_win1, d_ciw1);
  SST_LAUNCH_CHECK();
  return SST_OK;
}

int64_t sst_region_batching_workspace_bytes(int64_t m) {
  const int64_t n = m > 0 ? m : 1;
  return 2 * sst_align_up(8 * n, 256)                // key ping-pong
         + 2 * (3 * sst_align_up(4 * (n + 1), 256))  // per shift: perm, inv, off
         + 9 * sst_align_up(4 * (n + 1), 256)        // g, c, rank, cnt, keep0, seglevel x2, tokbase, cwl
         + 1024 + sst_unique_ws_internal(n) + sst_scan_ws_internal(n + 1);
}

int sst_region_batching(const int32_t* d_win0, const int32_t* d_win1, int64_t m, int win_bits,
                        const int32_t* h_levels, int n_levels, int32_t* d_keep, int32_t* d_newidx,
                        int32_t* d_level0, int32_t* d_level1, int32_t* d_inner0, int32_t* d_inner1,
                        int32_t* d_flat2win0, int32_t* d_flat2win1, int32_t* d_tok0, int32_t* d_tok1,
                        int32_t* d_winoff0, int32_t* d_winoff1, int32_t* d_winlevel0, int32_t* d_winlevel1,
                        int32_t* d_counts, void* d_workspace, void* stream) {
  if (m < 0 || n_levels < 1 || n_levels > 8 || !h_levels || win_bits < 1 || win_bits > 31 || !d_counts)
    return SST_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  SST_HIP(hipMemsetAsync(d_counts, 0, 8 * sizeof(int32_t), st));
  if (m == 0) {
    if (d_winoff0) SST_HIP(hipMemsetAsync(d_winoff0, 0, sizeof(int32_t), st));
    if (d_winoff1) SST_HIP(hipMemsetAsync(d_winoff1, 0, sizeof(int32_t), st));
    return SST_OK;
  }
  if (!d_win0 || !d_win1 || !d_keep || !d_newidx || !d_level0 || !d_level1 || !d_inner0 || !d_inner1 ||
      !d_flat2win0 || !d_flat2win1 || !d_tok0 || !d_tok1 || !d_winoff0 || !d_winoff1 || !d_winlevel0 ||
      !d_winlevel1 || !d_workspace)
    return SST_ERR_ARG;
  if (m >= (int64_t)1 << 31) return SST_ERR_UNSUPPORTED;

  level_table lt;
  lt.n = n_levels;
  for (int l = 0; l < 8; ++l) {
    lt.cap[l] = l < n_levels ? h_levels[l * 3 + 0] : 0;
    lt.lo[l] = l < n_levels ? h_levels[l * 3 + 1] : 0;
    lt.hi[l] = l < n_levels ? h_levels[l * 3 + 2] : 0;
  }

  sst_carver cv(d_workspace);
  uint64_t* ka = cv.take<uint64_t>(m);
  uint64_t* kb = cv.take<uint64_t>(m);
  uint32_t* perm[2];
  int32_t* inv[2];
  int32_t* off[2];
  for (int s = 0; s < 2; ++s) {
    perm[s] = cv.take<uint32_t>(m + 1);
    inv[s] = cv.take<int32_t>(m + 1);
    off[s] = cv.take<int32_t>(m + 1);
  }
  int32_t* g = cv.take<int32_t>(m + 1);
  int32_t* c = cv.take<int32_t>(m + 1);
  int32_t* rank = cv.take<int32_t>(m + 1);
  int32_t* cnt = cv.take<int32_t>(m + 1);
  int32_t* keep0 = cv.take<int32_t>(m + 1);
  int32_t* seglevel[2] = {cv.take<int32_t>(m + 1), cv.take<int32_t>(m + 1)};
  int32_t* tokbase = cv.take<int32_t>(m + 1);
  int32_t* cwl = cv.take<int32_t>(m + 1);
  int32_t* nseg = cv.take<int32_t>(2);  // nseg[0], nseg[1]
  void* uniq_ws = (void*)cv.take<char>(sst_unique_ws_internal(m));
  void* scan_ws = (void*)cv.take<char>(sst_scan_ws_internal(m + 1));

  const int grid = sst_grid_1d(m, 256);
  const int32_t* win[2] = {d_win0, d_win1};
  int32_t* level[2] = {d_level0, d_level1};
  int32_t* inner[2] = {d_inner0, d_inner1};
  int32_t* flat2win[2] = {d_flat2win0, d_flat2win1};
  int32_t* tok[2] = {d_tok0, d_tok1};
  int32_t* winoff[2] = {d_winoff0, d_winoff1};
  int32_t* winlevel[2] = {d_winlevel0, d_winlevel1};
  int rc;

  // 1. group voxels by window id for both shifts (stable)
  for (int s = 0; s < 2; ++s) {
    hipLaunchKernelGGL(keys_from_i32_k, dim3(grid), dim3(256), 0, st, win[s], m, ka);
    rc = sst_unique_keys_internal(ka, kb, m, win_bits, perm[s], inv[s], off[s], nullptr, nseg + s, uniq_ws, st);
    if (rc != SST_OK) return rc;
  }

  // 2. shift 0 on all voxels, shift 1 on the survivors of shift 0
  const int32_t* flag_in = nullptr;
  int32_t* keep_out[2] = {keep0, d_keep};
  for (int s = 0; s < 2; ++s) {
    hipLaunchKernelGGL(gather_flag_k, dim3(grid), dim3(256), 0, st, flag_in, perm[s], m, g);
    rc = sst_scan_i32_internal(g, c, m, c + m, scan_ws, st);
    if (rc != SST_OK) return rc;
    hipLaunchKernelGGL(seg_rank_cnt_k, dim3(grid), dim3(256), 0, st, c, perm[s], inv[s], off[s], m, rank, cnt);
    hipLaunchKernelGGL(drop_level_k, dim3(grid), dim3(256), 0, st, cnt, rank, flag_in, inv[s], m, lt, level[s],
                       keep_out[s], seglevel[s]);
    flag_in = keep0;
  }
  // NOTE (reference quirk, sst_input_layer_v2.py:186-194): level0 is NOT recomputed after the shift-1 filter.

  // 3. survivors -> new numbering
  rc = sst_scan_i32_internal(d_keep, d_newidx, m, d_counts + 0, scan_ws, st);
  if (rc != SST_OK) return rc;

  // 4. per shift: final in-window order among survivors, window CSR, per-level contiguous window ids
  for (int s = 0; s < 2; ++s) {
    hipLaunchKernelGGL(gather_flag_k, dim3(grid), dim3(256), 0, st, d_keep, perm[s], m, g);
    rc = sst_scan_i32_internal(g, c, m, c + m, scan_ws, st);
    if (rc != SST_OK) return rc;
    hipLaunchKernelGGL(seg_rank_cnt_k, dim3(grid), dim3(256), 0, st, c, perm[s], inv[s], off[s], m, inner[s],
                       (int32_t*)nullptr);
    hipLaunchKernelGGL(seg_plan_k, dim3(1), dim3(256), 0, st, c, off[s], nseg + s, seglevel[s], n_levels, tokbase, cwl,
                       winoff[s], winlevel[s], d_counts + 1 + s, d_counts + 3 + s);
    hipLaunchKernelGGL(plan_fill_k, dim3(grid), dim3(256), 0, st, d_keep, inv[s], inner[s], level[s], d_newidx,
                       tokbase, cwl, lt, m, tok[s], flat2win[s]);
  }
  SST_LAUNCH_CHECK();
  return SST_OK;
}


int sst_window_order_i32(const int32_t* d_winoff, int n_windows, int max_tokens, int32_t* d_order, void* stream) {
  if (n_windows < 0 || max_tokens < 0) return SST_ERR_ARG;
  if (n_windows == 0) return SST_OK;
  if (!d_winoff || !d_order) return SST_ERR_ARG;
  if (max_tokens >= kOrdBins || n_windows > (1 << 20)) return SST_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(window_order_k, dim3(1), dim3(1024), 0, (hipStream_t)stream, d_winoff, n_windows, max_tokens + 1, d_order);
  SST_LAUNCH_CHECK();
  return SST_OK;
}

}  // extern "C"
