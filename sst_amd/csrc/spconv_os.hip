// (§8 f4) sparse 3-D convolution, second generation of the contraction kernel: output-stationary implicit GEMM with the
// weights of one kernel offset staged through LDS and the partner rows gathered straight into MFMA operands.
//
// Replaces the same reference code as csrc/spconv.hip's sp_gather_gemm_k / sp_conv_seg_k: indiceConv and the data
// gradient of indiceConvBackward (mmdet3d/ops/spconv/include/spconv/spconv_ops.h:256-446; per kernel offset gather ->
// mm -> scatter-add there), on the dense maps out2in / in2out of the rulebook:
//     Y[r, :] = sum_k X[map[k][r], :] W[k]        (map[k][r] = -1: no partner)
//
// Why a second kernel.  On LiDAR-like voxel sets every level below the first has 17-18 of 27 partners per voxel and,
// with voxels numbered by ascending (b, z, y, x), 85-97 % of the (16..128-row block, offset) slots that have ANY partner
// are fully populated (tools/rulebook_stats.py).  Compacting rows per offset (sp_conv_seg_k: LDS row lists, LDS atomics,
// 32-row stages behind a barrier) buys nothing there; what the contraction needs is what a dense tall GEMM needs:
//   * a workgroup owns 64 or 128 output rows x 64 or 128 output columns; every wave owns 16 or 32 of the rows and ALL
//     the columns of the group, so the accumulators never leave registers (no LDS adds, no atomics, deterministic);
//   * the TRANSPOSED product Y^T = W[k]^T X_g^T on v_mfma_f32_16x16x4_f32: a lane ends up with 4 consecutive output
//     columns of one row (16-byte stores), and the gathered X fragment is the MFMA B operand as it comes from memory - a
//     16-byte load gives a lane 4 consecutive channels of its partner row = 4 consecutive k-steps (idiom of
//     csrc/dense_f32.hip); no LDS round trip for X;
//   * W[k] (one 64-channel chunk: 16 / 32 KB) is staged through LDS in MFMA-fragment order (packed once per call by
//     sp_os_pack_w_k, which also absorbs the transposition the data gradient needs), double-buffered, ONE barrier per
//     stage; the partner rows and the W slab of stage s + 1 are requested before the MFMAs of stage s are issued
//     (128-256 MFMAs = 1.7-3.4 us per wave and stage: more than an L2 / MALL round trip);
//   * loads are unconditional (absent partners read row 0 and are zeroed by a select, channel tails read a clamped
//     address against zero weights), so the compiler's s_waitcnt bookkeeping stays static (DESIGN.md, round-2 finding);
//   * offsets without a partner in the workgroup's rows are never staged; a wave whose own rows have none skips the MFMAs;
//   * XCD-aware numbering: workgroup b runs on XCD b % 8.  Runs of kOsXcdChunk consecutive units go to one XCD (their
//     halos overlap in its L2), the runs round-robin over the XCDs.  NOT one contiguous eighth of the rows per XCD: with
//     voxels numbered by (b, z, y, x) the work per tile follows z - on the top level of FSD's U-Net the ground slab is
//     3/4 of the rows with 4-5 partners each, the objects above it have 15+, and the last two XCDs were handed 2.3 x the
//     MFMA work of the others (the launch took as long as they did);
//   * tiles are LAUNCHED in the order of decreasing work when the caller passes the permutation computed from
//     sp_os_tile_work_k (once per rulebook and map): workgroups are dispatched in blockIdx order, the work of a tile
//     varies 6 .. 27 staged offsets on that level, and in row order the heavy tiles - the objects, at the highest z -
//     came last, so the launch ended with a few long workgroups on an otherwise empty chip (CUs busy 55 % of the launch:
//     SQ_BUSY_CU_CYCLES; 292 -> 182 us together with the point above).
#include <stdlib.h>

#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Debug instrumentation (build with -DSST_OS_TIMING, see tools/conv_os_phases.py): per-wave phase times of sp_conv_os_k
// in ticks of the constant 100 MHz clock.  Compiled out of the product library.
#ifdef SST_OS_TIMING
constexpr int kOsTsUnits = 4096, kOsTsSlots = 16;
__device__ unsigned long long g_os_ts[kOsTsUnits * 4 * kOsTsSlots];
#define OS_TS_DECL unsigned long long ts_prev_ = 0, ts_acc_[4] = {0, 0, 0, 0}; unsigned ts_n_[2] = {0, 0}
#define OS_NOW() wall_clock64()
#define OS_MARK(i)                                                                                                   \
  do {                                                                                                               \
    if (lane == 0 && unit < kOsTsUnits) g_os_ts[(unit * 4 + wave) * kOsTsSlots + (i)] = OS_NOW();                    \
  } while (0)
#define OS_LAP_START() ts_prev_ = OS_NOW()
#define OS_LAP(j)                                                                                                    \
  do {                                                                                                               \
    const unsigned long long now_ = OS_NOW();                                                                        \
    ts_acc_[j] += now_ - ts_prev_;                                                                                   \
    ts_prev_ = now_;                                                                                                 \
  } while (0)
#define OS_COUNT(j) ++ts_n_[j]
#define OS_FLUSH()                                                                                                   \
  do {                                                                                                               \
    if (lane == 0 && unit < kOsTsUnits) {                                                                            \
      unsigned long long* t_ = g_os_ts + (unit * 4 + wave) * kOsTsSlots;                                             \
      for (int j_ = 0; j_ < 4; ++j_) t_[8 + j_] = ts_acc_[j_];                                                       \
      t_[12] = ts_n_[0];                                                                                             \
      t_[13] = ts_n_[1];                                                                                             \
    }                                                                                                                \
  } while (0)
#else
#define OS_TS_DECL
#define OS_MARK(i)
#define OS_LAP_START()
#define OS_LAP(j)
#define OS_COUNT(j)
#define OS_FLUSH()
#endif

constexpr int kOsMaxK = 32;   // kernel offsets the index image holds (3 x 3 x 3 = 27)
constexpr int kOsChunk = 64;  // input channels per stage
constexpr int kOsXcdChunk = 4;  // consecutive units handed to one XCD

// Packed weights: slab (k, column group cg, channel chunk cc) = [4 j][NCT ct][64 lanes][4 t] floats with
//   value = W[k][c = 64 cc + 16 j + 4 (lane >> 4) + t][n = 16 NCT cg + 16 ct + (lane & 15)]   (0 outside cin x cout),
// i.e. the A fragments of the four k-steps a 16-byte X load feeds, one ds_read_b128 per lane.
// trans_w: W[k] is stored [n][c] (the data gradient reads the forward weights with the roles swapped).
__global__ __launch_bounds__(256) void sp_os_pack_w_k(const float* __restrict__ w, int kvol, int cin, int cout,
                                                      int trans_w, int nct, int n_cg, int n_cc,
                                                      float* __restrict__ wp) {
  const int64_t total = (int64_t)kvol * n_cg * n_cc * 64 * 16 * nct;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int t = (int)(e & 3), lane = (int)((e >> 2) & 63);
    int64_t rest = e >> 8;
    const int ct = (int)(rest % nct);
    rest /= nct;
    const int j = (int)(rest & 3);
    int64_t slab = rest >> 2;
    const int cc = (int)(slab % n_cc);
    slab /= n_cc;
    const int cg = (int)(slab % n_cg);
    const int k = (int)(slab / n_cg);
    const int c = cc * kOsChunk + 16 * j + 4 * (lane >> 4) + t;
    const int n = cg * 16 * nct + 16 * ct + (lane & 15);
    float v = 0.f;
    if (c < cin && n < cout)
      v = trans_w ? w[((int64_t)k * cout + n) * cin + c] : w[((int64_t)k * cin + c) * cout + n];
    wp[e] = v;
  }
}

template <int NCT, int RB>
__global__ __launch_bounds__(256, (NCT == 4 && RB == 1) ? 4 : 1) void sp_conv_os_k(const float* __restrict__ x, int64_t ldx,
                                                    const int32_t* __restrict__ map, int64_t m, int kvol,
                                                    const float* __restrict__ wp, int cin, int cout,
                                                    const float* __restrict__ bias, float* __restrict__ y, int64_t ldy,
                                                    int n_units, int n_cg, int n_cc, int xcd_chunk, int vec_store,
                                                    const int32_t* __restrict__ tile_order) {
  constexpr int ROWS = 64 * RB;          // rows of the workgroup: 4 waves x RB blocks of 16
  constexpr int SLAB = 64 * 16 * NCT;    // floats of one packed W slab
  constexpr int WREG = SLAB / 4 / 256;   // 16-byte pieces of a slab per thread
  extern __shared__ __attribute__((aligned(16))) float os_smem[];
  float (*wbuf)[SLAB] = (float (*)[SLAB])os_smem;                      // [2][SLAB]
  int (*idx)[ROWS] = (int (*)[ROWS])(os_smem + 2 * SLAB);              // [kvol][ROWS]
  unsigned* live_w = (unsigned*)(os_smem + 2 * SLAB + kvol * ROWS);     // [4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, kq = lane >> 4;
  // ---- which (row tile, column group) ----
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int unit = ((slot / xcd_chunk) * 8 + xcd) * xcd_chunk + slot % xcd_chunk;
  if (unit >= n_units) return;  // uniform
  const int pos = unit / n_cg, cg = unit - pos * n_cg;
  const int tile = tile_order ? tile_order[pos] : pos;   // heaviest tiles first (sst_spconv_os_tile_work_i32)
  const int64_t r0 = (int64_t)tile * ROWS;
  OS_TS_DECL;
  OS_MARK(0);
  // ---- partner rows of the tile for every offset -> LDS; which offsets are populated, per wave (ONE pass: a wave loads
  // 64 consecutive rows of one offset per step, so a ballot tells for every 16-row block whether that offset has a partner)
  if (tid < 4) live_w[tid] = 0u;
  __syncthreads();
  {
    // all the index loads of the thread are issued before the first one is used (one memory round trip for the image,
    // not one per step)
    constexpr int NIT = (kOsMaxK * ROWS + 255) / 256;
    int vals[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int e = it * 256 + tid;
      const int k = e / ROWS, row = e - k * ROWS;
      vals[it] = (k < kvol && r0 + row < m) ? map[(int64_t)k * m + r0 + row] : -1;
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int e = it * 256 + tid;
      const int k = e / ROWS, row = e - k * ROWS;   // uniform k per wave (ROWS is a multiple of 64)
      const int v = vals[it];                        // (-1 behind the last offset: no bit set below)
      if (k < kvol) idx[k][row] = v;
      // a wave holds 64 consecutive rows of ONE offset here: the ballot tells for each of their four 16-row blocks whether
      // the offset has a partner; lane q < 4 reports block q to the wave that owns it
      const unsigned long long b = __ballot(v >= 0);
      const int blk = (row & ~63) / 16 + (lane & 3);
      if (lane < 4 && k < kvol && ((b >> (16 * (lane & 3))) & 0xffffull)) atomicOr(&live_w[blk / RB], 1u << k);
    }
  }
  __syncthreads();
  OS_MARK(1);
  const unsigned wave_live = live_w[wave];
  const unsigned live = live_w[0] | live_w[1] | live_w[2] | live_w[3];
  const int wave_row = wave * 16 * RB;

  f32x4 acc[RB][NCT];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) acc[rb][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (live != 0u) {
    const int c_last = cin - 4;  // cin % 4 == 0 (checked by the host): the last whole 16-byte piece of a row
    f32x4 xc[RB][4], xn[RB][4], wr[WREG];
    int sn[RB];
    auto fetch_w = [&](int k, int cc) {
      const f32x4* src = (const f32x4*)(wp + ((int64_t)(k * n_cg + cg) * n_cc + cc) * SLAB);
#pragma unroll
      for (int u = 0; u < WREG; ++u) wr[u] = src[tid + 256 * u];
    };
    auto gather_x = [&](int k, int cc, f32x4 (&dst)[RB][4], int (&s)[RB]) {
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        s[rb] = idx[k][wave_row + 16 * rb + l15];
        const float* p = x + (int64_t)(s[rb] >= 0 ? s[rb] : 0) * ldx;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          int c = cc * kOsChunk + 16 * j + 4 * kq;
          c = c < c_last ? c : c_last;  // beyond cin the packed weights are zero: any finite value will do
          dst[rb][j] = *(const f32x4*)(p + c);
        }
      }
    };
    auto next_live = [&](int k) {  // first populated offset >= k (32 if none)
      const unsigned rest = k < 32 ? (live >> k) : 0u;
      return rest ? k + __builtin_ctz(rest) : 32;
    };
    int k = next_live(0), cc = 0, buf = 0;
    fetch_w(k, 0);
    gather_x(k, 0, xn, sn);
#pragma unroll
    for (int u = 0; u < WREG; ++u) *(f32x4*)(&wbuf[0][4 * (tid + 256 * u)]) = wr[u];
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int j = 0; j < 4; ++j) xc[rb][j] = sn[rb] >= 0 ? xn[rb][j] : zero4;  // rows without a partner contribute 0
    __syncthreads();
    OS_MARK(2);
    while (true) {
      OS_LAP_START();
      OS_COUNT(0);
      int nk = k, ncc = cc + 1;
      if (ncc == n_cc) {
        ncc = 0;
        nk = next_live(k + 1);
      }
      const bool has_next = nk < 32;
      if (!has_next) {  // keep the loads unconditional: the last stage requests itself once more
        nk = k;
        ncc = cc;
      }
      fetch_w(nk, ncc);
      gather_x(nk, ncc, xn, sn);
      OS_LAP(0);   // loads of the next stage issued
      if ((wave_live >> k) & 1u) {  // uniform per wave
        OS_COUNT(1);
        const float* wb = &wbuf[buf][4 * lane];
        // column tiles in groups of 4; the fragments of group q + 1 are read while the MFMAs of group q issue
        constexpr int CG4 = NCT / 4, G = 4 * CG4;
        f32x4 wf[4], wnx[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) wf[u] = *(const f32x4*)(wb + u * 256);
#pragma clang loop unroll(full)
        for (int q = 0; q < G; ++q) {
          const int j = q / CG4, g4 = q - j * CG4;
          const int qn = q + 1 < G ? q + 1 : q;
          const int jn = qn / CG4, gn = qn - jn * CG4;
#pragma unroll
          for (int u = 0; u < 4; ++u) wnx[u] = *(const f32x4*)(wb + ((jn * NCT + 4 * gn + u) * 256));
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
              for (int rb = 0; rb < RB; ++rb)
                acc[rb][4 * g4 + u] =
                    __builtin_amdgcn_mfma_f32_16x16x4f32(wf[u][t], xc[rb][j][t], acc[rb][4 * g4 + u], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int u = 0; u < 4; ++u) wf[u] = wnx[u];
        }
      }
      OS_LAP(1);   // MFMAs issued (the clock read waits for the LDS fragments, not for the matrix pipe)
#pragma unroll
      for (int u = 0; u < WREG; ++u) *(f32x4*)(&wbuf[buf ^ 1][4 * (tid + 256 * u)]) = wr[u];
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int j = 0; j < 4; ++j) xc[rb][j] = sn[rb] >= 0 ? xn[rb][j] : zero4;
#ifdef SST_OS_TIMING
      __builtin_amdgcn_s_waitcnt(0);
#endif
      OS_LAP(2);   // loads arrived, W in LDS
      __syncthreads();
      OS_LAP(3);   // barrier
      if (!has_next) break;
      k = nk;
      cc = ncc;
      buf ^= 1;
    }
  }
  OS_MARK(3);
  // ---- epilogue: lane = (row l15 of its block, 4 consecutive columns at 4 kq of every column tile) ----
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const int64_t row = r0 + wave_row + 16 * rb + l15;
    if (row >= m) continue;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      const int n = cg * 16 * NCT + 16 * ct + 4 * kq;
      if (n >= cout) continue;
      f32x4 v = acc[rb][ct];
      if (vec_store && n + 3 < cout) {
        if (bias) v += *(const f32x4*)(bias + n);
        *(f32x4*)(y + row * ldy + n) = v;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n + e < cout) y[row * ldy + n + e] = v[e] + (bias ? bias[n + e] : 0.f);
      }
    }
  }
  OS_MARK(4);
  OS_FLUSH();
}

// work[tile] = number of (16-row block, offset) slots of the tile with at least one partner: the MFMA work of the tile in
// units of 16 rows x one offset.  One wave per tile.
__global__ __launch_bounds__(256) void sp_os_tile_work_k(const int32_t* __restrict__ map, int64_t m, int kvol,
                                                         int tile_rows, int64_t n_tiles, int32_t* __restrict__ work) {
  const int lane = threadIdx.x & 63;
  const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tile >= n_tiles) return;  // uniform per wave
  const int64_t r0 = tile * tile_rows;
  int slots = 0;
  for (int k = 0; k < kvol; ++k)
    for (int r = 0; r < tile_rows; r += 64) {
      const int64_t row = r0 + r + lane;
      const bool has = row < m && map[(int64_t)k * m + row] >= 0;
      const unsigned long long b = __ballot(has);
#pragma unroll
      for (int q = 0; q < 4; ++q) slots += ((b >> (16 * q)) & 0xffffull) != 0ull ? 1 : 0;
    }
  if (lane == 0) work[tile] = slots;
}

// ---------------------------------------------------------------------------------------------------------------
// Filter gradient of indiceConvBackward (spconv_ops.h:359-446): dW[k] (cin x cout) = sum over the pairs p of offset k of
// X[pa[p], :]^T dY[pb[p], :], second generation.  The first-generation kernel (sp_wgrad_k) feeds the MFMAs with one 4-byte
// load per lane and operand straight from global memory (two gather latencies per 8 pairs, 512-pair chunks so that enough
// workgroups overlap them, hence 6 k partial tiles of 16 KB for a 3 M-pair level and a 37 us reduction per call).  Here:
//   * a workgroup owns a 64 x 64 block of dW[k] and a chunk of 2048 pairs of that offset; wave w a 32 x 32 quadrant;
//   * a stage = 64 pairs: the 64 gathered X rows and the 64 gathered dY rows (64 channels of each: 16-byte loads, whole
//     256-byte row pieces per 16 lanes) go global -> registers -> LDS TRANSPOSED ([channel][pair], XOR-swizzled by 4-pair
//     groups so that both the transposing writes and the 16-byte fragment reads are conflict-free); the contraction runs
//     over the pairs, so a ds_read_b128 hands a lane 4 consecutive pairs of its channel = 4 MFMA k-steps
//     (v_mfma_f32_16x16x4_f32) for both operands;
//   * software pipeline: pair indices of stage s + 2 and rows of stage s + 1 are in flight while stage s is multiplied,
//     double-buffered LDS, one barrier per stage, every load unconditional (clamped pair index / channel, zeroed on the X
//     side by a select);
//   * partial blocks per chunk are summed in chunk order by sp_wgrad_os_reduce_k (deterministic): 4 x fewer, larger chunks.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kWgChunk = 2048;  // pairs per workgroup (upper bound; os_wgrad_chunk_size picks 512 .. 2048)
constexpr int kWgStage = 64;    // pairs per stage
constexpr int kWgLd = 68;       // LDS row stride (floats) of a [64 channels][64 pairs] image

__device__ __forceinline__ bool os_find_chunk(const int32_t* __restrict__ num, int kvol, int chunk, int csize, int& k,
                                              int& first) {
  int c0 = 0;
  for (int i = 0; i < kvol; ++i) {
    const int nc = (num[i] + csize - 1) / csize;
    if (chunk < c0 + nc) {
      k = i;
      first = c0;
      return true;
    }
    c0 += nc;
  }
  return false;
}

__global__ __launch_bounds__(256) void sp_wgrad_os_k(const float* __restrict__ x, int64_t ldx,
                                                     const float* __restrict__ dy, int64_t lddy,
                                                     const int32_t* __restrict__ pairs, int64_t pair_ld, int x_side,
                                                     const int32_t* __restrict__ num, int kvol, int cin, int cout,
                                                     int n_bj, int csize, float* __restrict__ part) {
  __shared__ __attribute__((aligned(16))) float At[2][64 * kWgLd];
  __shared__ __attribute__((aligned(16))) float Bt[2][64 * kWgLd];
  int k, first;
  if (!os_find_chunk(num, kvol, blockIdx.x, csize, k, first)) return;  // uniform
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, kq = lane >> 4;
  const int bi = blockIdx.y / n_bj, bj = blockIdx.y - bi * n_bj;
  const int ci0 = bi * 64, co0 = bj * 64;
  const int np = num[k];
  const int p0 = (blockIdx.x - first) * csize;
  const int p1 = p0 + csize < np ? p0 + csize : np;
  const int32_t* pa = pairs + ((int64_t)k * 2 + x_side) * pair_ld;
  const int32_t* pb = pairs + ((int64_t)k * 2 + (1 - x_side)) * pair_ld;
  // staging role of this thread: pairs prow + 16 u (u < 4) of the stage, channels 4 ch4 .. + 3 of the block
  const int ch4 = tid & 15, prow = tid >> 4;
  int ca = ci0 + 4 * ch4, cb = co0 + 4 * ch4;
  ca = ca < cin - 4 ? ca : cin - 4;      // channel tails: a clamped (finite) piece whose products are never stored
  cb = cb < cout - 4 ? cb : cout - 4;
  int ia[4], ib[4];
  f32x4 ra[4], rb[4];
  bool live[4];
  auto load_idx = [&](int p) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int pp = p + prow + 16 * u;
      pp = pp < p1 ? pp : p1 - 1;
      ia[u] = pa[pp];
      ib[u] = pb[pp];
    }
  };
  auto load_rows = [&](int p) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      live[u] = p + prow + 16 * u < p1;
      ra[u] = *(const f32x4*)(x + (int64_t)ia[u] * ldx + ca);
      rb[u] = *(const f32x4*)(dy + (int64_t)ib[u] * lddy + cb);
    }
  };
  // [channel row][pair]: 4-pair group g of row r sits at group g ^ ((r >> 4) & 3) (conflict-free writes and reads)
  auto store_rows = [&](int buf) {
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int pair = prow + 16 * u;
      const int col = (((pair >> 2) ^ (ch4 >> 2)) << 2) + (pair & 3);   // rows 4 ch4 + e: (row >> 4) == ch4 >> 2
      const f32x4 va = live[u] ? ra[u] : zero4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        At[buf][(4 * ch4 + e) * kWgLd + col] = va[e];
        Bt[buf][(4 * ch4 + e) * kWgLd + col] = rb[u][e];
      }
    }
  };
  f32x4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int wi = (wave & 1) * 32, wj = (wave >> 1) * 32;   // this wave's quadrant of the 64 x 64 block
  load_idx(p0);
  load_rows(p0);
  load_idx(p0 + kWgStage);
  store_rows(0);
  __syncthreads();
  int buf = 0;
  for (int p = p0; p < p1; p += kWgStage) {
    load_rows(p + kWgStage);        // rows of the next stage (their indices were requested a stage ago)
    load_idx(p + 2 * kWgStage);     // indices of the stage after it
    __builtin_amdgcn_sched_barrier(0);   // the loads are issued HERE, in front of the MFMAs that hide their latency
    const float* a_img = At[buf];
    const float* b_img = Bt[buf];
#pragma unroll
    for (int s = 0; s < kWgStage / 16; ++s) {
      f32x4 af[2], bf[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int ra_ = wi + 16 * h + l15, rb_ = wj + 16 * h + l15;
        af[h] = *(const f32x4*)(a_img + ra_ * kWgLd + (((4 * s + kq) ^ ((ra_ >> 4) & 3)) << 2));
        bf[h] = *(const f32x4*)(b_img + rb_ * kWgLd + (((4 * s + kq) ^ ((rb_ >> 4) & 3)) << 2));
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[a][t], bf[b][t], acc[a][b], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    store_rows(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  // D[i][j]: lane = (j = l15, rows i = 4 kq + r): dW[k][ci0 + wi + 16 a + 4 kq + r][co0 + wj + 16 b + l15]
  float* dst = part + (int64_t)blockIdx.x * cin * cout;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int n = co0 + wj + 16 * b + l15;
      if (n >= cout) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = ci0 + wi + 16 * a + 4 * kq + r;
        if (c < cin) dst[(int64_t)c * cout + n] = acc[a][b][r];
      }
    }
}

// dw[k][e] = sum over the chunks of offset k, in chunk order
__global__ __launch_bounds__(256) void sp_wgrad_os_reduce_k(const float* __restrict__ part, const int32_t* __restrict__ num,
                                                            int kvol, int64_t per_k, int csize, float* __restrict__ dw) {
  const int k = blockIdx.y;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= per_k) return;
  int first = 0;
  for (int i = 0; i < k; ++i) first += (num[i] + csize - 1) / csize;
  const int nc = (num[k] + csize - 1) / csize;
  float s = 0.f;
  for (int c = 0; c < nc; ++c) s += part[(int64_t)(first + c) * per_k + e];
  dw[(int64_t)k * per_k + e] = s;
}

// ---------------------------------------------------------------------------------------------------------------
// The same filter gradient from the EXACT three-way bf16 split of both gathered operands (csrc/dense_f32x6.hip, csrc/wgrad_x6.hip,
// csrc/spconv_os_x6.hip: x = x0 + x1 + x2, dy = d0 + d1 + d2, the six products with i + j <= 2 on v_mfma_f32_16x16x32_bf16, fp32
// accumulation, two accumulator sets): 48 bf16 instructions = 768 matrix-pipe cycles per wave and 64-pair stage against
// 64 x v_mfma_f32_16x16x4_f32 = 2 048 for sp_wgrad_os_k, which that pipe bounds (4.8 of 25.5 ms of an FSD step, 10.1 of 49 ms
// of an FSDv2 step).  Same decomposition (64 x 64 block of dW[k], chunk of pairs, wave = 32 x 32 quadrant, stage = 64 pairs,
// rows global -> registers -> LDS transposed, indices two stages ahead), different LDS image: per operand three bf16 images
// [64 channels][64 pairs] of 128-byte rows, written as 8-byte groups of 4 consecutive pairs (a thread stages 4 consecutive
// pairs x 4 channels: its split yields the three parts of 4 pairs per channel), read as the 16-byte fragments the
// instruction wants (lane (l15, g): 8 consecutive pairs 32 ks + 8 g .. of channel row l15).  16-byte unit G of row r sits at
// G ^ f(r), f(r) = ((r >> 1) & 7) ^ ((r >> 3) & 7): the sixteen rows of a fragment read cover all 64 banks once, the sixteen
// channel blocks of a write land two per 8-byte slot.  One buffer (48 KB, three workgroups per CU): the stage in registers
// waits for a barrier behind the products of the stage in LDS.
__device__ __forceinline__ unsigned wx6_pack2(float lo, float hi) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  const bf2 v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ void wx6_split2(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) {
  p0 = wx6_pack2(a, b);
  const float ra = a - __uint_as_float(p0 << 16), rb = b - __uint_as_float(p0 & 0xffff0000u);
  p1 = wx6_pack2(ra, rb);
  p2 = wx6_pack2(ra - __uint_as_float(p1 << 16), rb - __uint_as_float(p1 & 0xffff0000u));
}
typedef unsigned wx6_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned wx6_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 wx6_mma(wx6_u32x4 a, wx6_u32x4 b, f32x4 c) {
  typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), c, 0, 0, 0);
}
__device__ __forceinline__ int wx6_unit(int row, int G) { return (G ^ ((row >> 1) & 7) ^ ((row >> 3) & 7)) & 7; }

__global__ __launch_bounds__(256) void sp_wgrad_os_x6_k(const float* __restrict__ x, int64_t ldx,
                                                        const float* __restrict__ dy, int64_t lddy,
                                                        const int32_t* __restrict__ pairs, int64_t pair_ld, int x_side,
                                                        const int32_t* __restrict__ num, int kvol, int cin, int cout,
                                                        int n_bj, int csize, float* __restrict__ part) {
  __shared__ __attribute__((aligned(16))) unsigned char img[6][64 * 128];   // X parts 0..2, dY parts 0..2
  int k, first;
  if (!os_find_chunk(num, kvol, blockIdx.x, csize, k, first)) return;  // uniform
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int bi = blockIdx.y / n_bj, bj = blockIdx.y - bi * n_bj;
  const int ci0 = bi * 64, co0 = bj * 64;
  const int np = num[k];
  const int p0 = (blockIdx.x - first) * csize;
  const int p1 = p0 + csize < np ? p0 + csize : np;
  const int32_t* pa = pairs + ((int64_t)k * 2 + x_side) * pair_ld;
  const int32_t* pb = pairs + ((int64_t)k * 2 + (1 - x_side)) * pair_ld;
  // staging role of this thread: pairs 4 pq + u (u < 4) of the stage, channels 4 ch4 .. + 3 of the block
  const int ch4 = tid & 15, pq = tid >> 4;
  int ca = ci0 + 4 * ch4, cb = co0 + 4 * ch4;
  ca = ca < cin - 4 ? ca : cin - 4;      // channel tails: a clamped (finite) piece whose products are never stored
  cb = cb < cout - 4 ? cb : cout - 4;
  int ia[4], ib[4];
  f32x4 ra[4], rb[4];
  bool live[4];
  auto load_idx = [&](int p) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int pp = p + 4 * pq + u;
      pp = pp < p1 ? pp : p1 - 1;
      ia[u] = pa[pp];
      ib[u] = pb[pp];
    }
  };
  auto load_rows = [&](int p) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      live[u] = p + 4 * pq + u < p1;
      ra[u] = *(const f32x4*)(x + (int64_t)ia[u] * ldx + ca);
      rb[u] = *(const f32x4*)(dy + (int64_t)ib[u] * lddy + cb);
    }
  };
  auto store_rows = [&]() {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = 4 * ch4 + e;
      const int off = row * 128 + wx6_unit(row, pq >> 1) * 16 + (pq & 1) * 8;
      unsigned a0[2], a1[2], a2[2], b0[2], b1[2], b2[2];
      wx6_split2(live[0] ? ra[0][e] : 0.f, live[1] ? ra[1][e] : 0.f, a0[0], a1[0], a2[0]);
      wx6_split2(live[2] ? ra[2][e] : 0.f, live[3] ? ra[3][e] : 0.f, a0[1], a1[1], a2[1]);
      wx6_split2(rb[0][e], rb[1][e], b0[0], b1[0], b2[0]);
      wx6_split2(rb[2][e], rb[3][e], b0[1], b1[1], b2[1]);
      *(wx6_u32x2*)(img[0] + off) = (wx6_u32x2){a0[0], a0[1]};
      *(wx6_u32x2*)(img[1] + off) = (wx6_u32x2){a1[0], a1[1]};
      *(wx6_u32x2*)(img[2] + off) = (wx6_u32x2){a2[0], a2[1]};
      *(wx6_u32x2*)(img[3] + off) = (wx6_u32x2){b0[0], b0[1]};
      *(wx6_u32x2*)(img[4] + off) = (wx6_u32x2){b1[0], b1[1]};
      *(wx6_u32x2*)(img[5] + off) = (wx6_u32x2){b2[0], b2[1]};
    }
  };
  f32x4 acc[2][2], cor[2][2];   // leading product | the five corrections
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = cor[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int wi = (wave & 1) * 32, wj = (wave >> 1) * 32;   // this wave's quadrant of the 64 x 64 block
  load_idx(p0);
  load_rows(p0);
  load_idx(p0 + kWgStage);
  store_rows();
  __syncthreads();
  for (int p = p0; p < p1; p += kWgStage) {
    load_rows(p + kWgStage);        // rows of the next stage (their indices were requested a stage ago)
    load_idx(p + 2 * kWgStage);     // indices of the stage after it
    __builtin_amdgcn_sched_barrier(0);   // the loads are issued HERE, in front of the products that hide their latency
#pragma unroll
    for (int ks = 0; ks < kWgStage / 32; ++ks) {
      wx6_u32x4 af[2][3], bf[2][3];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int ra_ = wi + 16 * h + l15, rb_ = wj + 16 * h + l15;
        const int oa = ra_ * 128 + wx6_unit(ra_, 4 * ks + g) * 16, ob = rb_ * 128 + wx6_unit(rb_, 4 * ks + g) * 16;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          af[h][q] = *(const wx6_u32x4*)(img[q] + oa);
          bf[h][q] = *(const wx6_u32x4*)(img[3 + q] + ob);
        }
      }
      // smallest products first (x2 d0, x0 d2, x1 d1 ~ 2^-16; x1 d0, x0 d1 ~ 2^-8), product by product over the four tiles
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) cor[a][b] = wx6_mma(af[a][2], bf[b][0], cor[a][b]);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) cor[a][b] = wx6_mma(af[a][0], bf[b][2], cor[a][b]);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) cor[a][b] = wx6_mma(af[a][1], bf[b][1], cor[a][b]);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) cor[a][b] = wx6_mma(af[a][1], bf[b][0], cor[a][b]);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) cor[a][b] = wx6_mma(af[a][0], bf[b][1], cor[a][b]);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = wx6_mma(af[a][0], bf[b][0], acc[a][b]);
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();                // every wave has read this stage's images
    store_rows();
    __syncthreads();
  }
  // D[i][j]: lane = (j = l15, rows i = 4 g + r): dW[k][ci0 + wi + 16 a + 4 g + r][co0 + wj + 16 b + l15]
  float* dst = part + (int64_t)blockIdx.x * cin * cout;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int n = co0 + wj + 16 * b + l15;
      if (n >= cout) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = ci0 + wi + 16 * a + 4 * g + r;
        if (c < cin) dst[(int64_t)c * cout + n] = acc[a][b][r] + cor[a][b][r];
      }
    }
}

// pairs per workgroup: 2048 when that still gives every CU a few workgroups, down to 512 for the thin levels (a level with
// 5 partners per voxel has 330 chunks of 2048 pairs for 256 CUs)
int os_wgrad_chunk_size(int kvol, int64_t pair_ld, int64_t total_pairs, int n_blocks) {
  static int env = -1;   // SST_SPCONV_WGRAD_CHUNK overrides (A/B measurements)
  if (env < 0) {
    const char* e = getenv("SST_SPCONV_WGRAD_CHUNK");
    env = e ? atoi(e) : 0;
  }
  if (env >= 64 && env <= kWgChunk && env % 64 == 0) return env;
  const int64_t total = (total_pairs >= 0 && total_pairs <= (int64_t)kvol * pair_ld) ? total_pairs : (int64_t)kvol * pair_ld;
  // 2 workgroups fit a CU (70 KB of LDS each): below one full round of 512, size the chunks so that the launch IS about
  // one round (a level with 5 partners per voxel: 330 chunks of 2048 pairs would leave a third of the CUs with twice the work)
  if ((total / kWgChunk + kvol / 2) * n_blocks >= 480) return kWgChunk;
  int64_t c = total * n_blocks / 480;
  c = (c + kWgStage - 1) / kWgStage * kWgStage;
  return (int)(c < 512 ? 512 : (c > kWgChunk ? kWgChunk : c));
}

int64_t os_wgrad_chunks(int kvol, int64_t pair_ld, int64_t total_pairs, int csize) {
  const int64_t total = (total_pairs >= 0 && total_pairs <= (int64_t)kvol * pair_ld) ? total_pairs : (int64_t)kvol * pair_ld;
  return total / csize + kvol;
}

struct os_cfg {
  int nct, rb, n_cg, n_cc;
  int64_t n_tiles;
};

// tile shape of a call.  64-row workgroups (one 16-row block per wave) everywhere: they run 4 per CU (109 VGPRs, 40 KB of
// LDS) where the 128-row form runs 3, have twice as many workgroups to balance, and measured 10-30 % faster on every level
// of FSD's U-Net although they stage W twice as often; 128 columns per workgroup only while that still leaves a few
// workgroups per CU
os_cfg os_pick(int64_t m, int cout, int tile_cfg) {
  static int env_cfg = -1;  // SST_SPCONV_OS_TILE = 10 * NCT + RB overrides the automatic choice (A/B measurements)
  if (env_cfg < 0) {
    const char* e = getenv("SST_SPCONV_OS_TILE");
    env_cfg = e ? atoi(e) : 0;
  }
  if (tile_cfg == 0) tile_cfg = env_cfg;
  const int env_nct = tile_cfg / 10, env_rb = tile_cfg % 10;
  os_cfg c;
  c.nct = cout <= 64 ? 4 : 8;
  c.rb = 1;
  auto units = [&](int nct, int rb) { return sst_div_up(m, 64 * rb) * sst_div_up(cout, 16 * nct); };
  if (c.nct == 8 && units(8, c.rb) < 2048) c.nct = 4;
  if (env_nct == 4 || env_nct == 8) c.nct = env_nct;
  if (env_rb == 1 || env_rb == 2) c.rb = env_rb;
  c.n_cg = (int)sst_div_up(cout, 16 * c.nct);
  c.n_tiles = sst_div_up(m, 64 * c.rb);
  return c;
}

}  // namespace

extern "C" {

int64_t sst_spconv_conv_os_workspace_bytes(int kvol, int cin, int cout) {
  if (kvol < 1 || cin < 1 || cout < 1) return 256;
  // the packed size does not depend on the tile shape: column groups x 16 NCT columns = cout rounded up to 64 or 128
  const int64_t cols = sst_div_up(cout, 128) * 128;
  return (int64_t)kvol * sst_div_up(cin, kOsChunk) * kOsChunk * cols * (int64_t)sizeof(float) + 256;
}

int sst_spconv_conv_os_f32(const float* d_x, int64_t ldx, const int32_t* d_map, int64_t m, int kvol, const float* d_w,
                           int cin, int cout, int trans_w, const float* d_bias, float* d_y, int64_t ldy,
                           int tile_cfg, const int32_t* d_tile_order, void* d_workspace, void* stream) {
  if (m < 0 || kvol < 1 || cin < 1 || cout < 1 || ldx < cin || ldy < cout) return SST_ERR_ARG;
  if (tile_cfg != 0 && tile_cfg != 41 && tile_cfg != 42 && tile_cfg != 81 && tile_cfg != 82) return SST_ERR_ARG;
  if (m == 0) return SST_OK;
  if (!d_x || !d_map || !d_w || !d_y || !d_workspace) return SST_ERR_ARG;
  if (kvol > kOsMaxK || (cin & 3) || (ldx & 3) || (((uintptr_t)d_x) & 15) || (((uintptr_t)d_workspace) & 15))
    return SST_ERR_UNSUPPORTED;
  const os_cfg c = os_pick(m, cout, tile_cfg);
  const int n_cc = (int)sst_div_up(cin, kOsChunk);
  const int64_t n_units = c.n_tiles * c.n_cg;
  if (n_units > 0x3fffffff) return SST_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  float* wp = (float*)d_workspace;
  const int64_t packed = (int64_t)kvol * c.n_cg * n_cc * 64 * 16 * c.nct;
  hipLaunchKernelGGL(sp_os_pack_w_k, dim3(sst_grid_1d(packed, 256)), dim3(256), 0, st, d_w, kvol, cin, cout, trans_w,
                     c.nct, c.n_cg, n_cc, wp);
  static int xcd_chunk = -1;  // SST_SPCONV_OS_XCD_CHUNK: units per run (A/B measurements)
  if (xcd_chunk < 0) {
    const char* e = getenv("SST_SPCONV_OS_XCD_CHUNK");
    xcd_chunk = e && atoi(e) > 0 ? atoi(e) : kOsXcdChunk;
  }
  const int chunk = n_units >= 64 * (int64_t)xcd_chunk ? xcd_chunk : 1;
  const dim3 grid((unsigned)(sst_div_up(n_units, 8 * chunk) * 8 * chunk));
  const int vec_store = ((ldy & 3) == 0 && (((uintptr_t)d_y) & 15) == 0 && (!d_bias || (((uintptr_t)d_bias) & 15) == 0)) ? 1 : 0;
#define SST_OS_LAUNCH(NCT, RB)                                                                                         \
  do {                                                                                                                 \
    constexpr int lds_max = (2 * 64 * 16 * NCT + kOsMaxK * 64 * RB + 4) * (int)sizeof(float);                          \
    const int lds = (2 * 64 * 16 * NCT + kvol * 64 * RB + 4) * (int)sizeof(float);                                     \
    static unsigned long long attr_set = 0;                                                                                      \
    if (sst_first_use_on_device(&attr_set)) {                                                                                                   \
      SST_HIP(hipFuncSetAttribute((const void*)sp_conv_os_k<NCT, RB>, hipFuncAttributeMaxDynamicSharedMemorySize,      \
                                  lds_max));                                                                           \
      sst_mark_device(&attr_set);                                                                                                 \
    }                                                                                                                  \
    hipLaunchKernelGGL((sp_conv_os_k<NCT, RB>), grid, dim3(256), lds, st, d_x, ldx, d_map, m, kvol, wp, cin, cout,     \
                       d_bias, d_y, ldy, (int)n_units, c.n_cg, n_cc, chunk, vec_store, d_tile_order);                        \
  } while (0)
  if (c.nct == 4 && c.rb == 2)
    SST_OS_LAUNCH(4, 2);
  else if (c.nct == 4)
    SST_OS_LAUNCH(4, 1);
  else if (c.rb == 2)
    SST_OS_LAUNCH(8, 2);
  else
    SST_OS_LAUNCH(8, 1);
#undef SST_OS_LAUNCH
  SST_LAUNCH_CHECK();
  return SST_OK;
}

int sst_spconv_conv_os_tile_rows(int64_t m, int cout, int tile_cfg) {
  if (m < 0 || cout < 1) return SST_ERR_ARG;
  return 64 * os_pick(m, cout, tile_cfg).rb;
}

int sst_spconv_os_tile_work_i32(const int32_t* d_map, int64_t m, int kvol, int tile_rows, int32_t* d_work, void* stream) {
  if (m < 0 || kvol < 1 || (tile_rows != 64 && tile_rows != 128)) return SST_ERR_ARG;
  if (m == 0) return SST_OK;
  if (!d_map || !d_work) return SST_ERR_ARG;
  const int64_t n_tiles = sst_div_up(m, tile_rows);
  hipLaunchKernelGGL(sp_os_tile_work_k, dim3((unsigned)sst_div_up(n_tiles, 4)), dim3(256), 0, (hipStream_t)stream, d_map, m,
                     kvol, tile_rows, n_tiles, d_work);
  SST_LAUNCH_CHECK();
  return SST_OK;
}

#ifdef SST_OS_TIMING
int sst_debug_conv_os_timestamps(void* host_dst, int64_t bytes) {
  if (bytes > (int64_t)sizeof(g_os_ts)) bytes = sizeof(g_os_ts);
  SST_HIP(hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_os_ts), (size_t)bytes, 0, hipMemcpyDeviceToHost));
  return SST_OK;
}
#endif

int64_t sst_spconv_wgrad_os_workspace_bytes(int kvol, int64_t pair_ld, int64_t total_pairs, int cin, int cout) {
  kvol = kvol > 0 ? kvol : 1;
  pair_ld = pair_ld > 0 ? pair_ld : 1;
  const int n_blocks = (int)(sst_div_up(cin > 0 ? cin : 1, 64) * sst_div_up(cout > 0 ? cout : 1, 64));
  const int csize = os_wgrad_chunk_size(kvol, pair_ld, total_pairs, n_blocks);
  return os_wgrad_chunks(kvol, pair_ld, total_pairs, csize) * cin * cout * (int64_t)sizeof(float) + 256;
}

static int wgrad_os_any(int split, const float* d_x, int64_t ldx, const float* d_dy, int64_t lddy, const int32_t* d_pairs,
                            int64_t pair_ld, int64_t total_pairs, int x_side, const int32_t* d_num, int kvol, int cin,
                            int cout, float* d_dw, void* d_workspace, void* stream) {
  if (kvol < 1 || cin < 1 || cout < 1 || ldx < cin || lddy < cout || pair_ld < 0 || (x_side != 0 && x_side != 1))
    return SST_ERR_ARG;
  if (!d_pairs || !d_num || !d_dw || !d_workspace) return SST_ERR_ARG;
  if ((cin & 3) || (cout & 3) || (ldx & 3) || (lddy & 3) || (((uintptr_t)d_x | (uintptr_t)d_dy) & 15))
    return SST_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int64_t per_k = (int64_t)cin * cout;
  if (pair_ld == 0 || !d_x || !d_dy) {
    SST_HIP(hipMemsetAsync(d_dw, 0, sizeof(float) * kvol * per_k, st));
    return SST_OK;
  }
  const int n_bi = (int)sst_div_up(cin, 64), n_bj = (int)sst_div_up(cout, 64);
  const int csize = os_wgrad_chunk_size(kvol, pair_ld, total_pairs, n_bi * n_bj);
  const int64_t chunks = os_wgrad_chunks(kvol, pair_ld, total_pairs, csize);
  if (kvol > 65535 || chunks > 0x7fffffff || n_bi * n_bj > 65535) return SST_ERR_UNSUPPORTED;
  float* part = (float*)d_workspace;
  if (split)
    hipLaunchKernelGGL(sp_wgrad_os_x6_k, dim3((unsigned)chunks, (unsigned)(n_bi * n_bj)), dim3(256), 0, st, d_x, ldx, d_dy, lddy,
                       d_pairs, pair_ld, x_side, d_num, kvol, cin, cout, n_bj, csize, part);
  else
    hipLaunchKernelGGL(sp_wgrad_os_k, dim3((unsigned)chunks, (unsigned)(n_bi * n_bj)), dim3(256), 0, st, d_x, ldx, d_dy, lddy,
                       d_pairs, pair_ld, x_side, d_num, kvol, cin, cout, n_bj, csize, part);
  hipLaunchKernelGGL(sp_wgrad_os_reduce_k, dim3((unsigned)sst_div_up(per_k, 256), (unsigned)kvol), dim3(256), 0, st, part,
                     d_num, kvol, per_k, csize, d_dw);
  SST_LAUNCH_CHECK();
  return SST_OK;
}

int sst_spconv_wgrad_os_f32(const float* d_x, int64_t ldx, const float* d_dy, int64_t lddy, const int32_t* d_pairs,
                            int64_t pair_ld, int64_t total_pairs, int x_side, const int32_t* d_num, int kvol, int cin,
                            int cout, float* d_dw, void* d_workspace, void* stream) {
  return wgrad_os_any(0, d_x, ldx, d_dy, lddy, d_pairs, pair_ld, total_pairs, x_side, d_num, kvol, cin, cout, d_dw, d_workspace,
                      stream);
}

int sst_spconv_wgrad_os_f32x6(const float* d_x, int64_t ldx, const float* d_dy, int64_t lddy, const int32_t* d_pairs,
                              int64_t pair_ld, int64_t total_pairs, int x_side, const int32_t* d_num, int kvol, int cin,
                              int cout, float* d_dw, void* d_workspace, void* stream) {
  return wgrad_os_any(1, d_x, ldx, d_dy, lddy, d_pairs, pair_ld, total_pairs, x_side, d_num, kvol, cin, cout, d_dw, d_workspace,
                      stream);
}

}  // extern "C"
