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

// Debug instrumentation (build with -DSST_WGRAD_TIMING, see tools/wgrad_phases.py): per-wave timestamps of the wide
// kernel's phases, {constant 100 MHz clock, shader clock} pairs.  Compiled out of the product library.
#ifdef SST_WGRAD_TIMING
__device__ unsigned long long g_wg_ts[1024 * 4 * 8 * 2];
#define SST_TS(i)                                                                      \
  do {                                                                                 \
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 1024) {                                \
      if ((i) == 5) __builtin_amdgcn_s_waitcnt(0);                                     \
      unsigned long long* t_ = g_wg_ts + ((blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + (i)) * 2; \
      t_[0] = wall_clock64();                                                          \
      t_[1] = clock64();                                                               \
    }                                                                                  \
  } while (0)
#else
#define SST_TS(i)
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int64_t sst_dev_align_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }

constexpr int kWgTileO = 128, kWgTileI = 64;

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

template <int U, int KW>  // U: k-steps (of 2 rows) per unrolled group; KW: K-slices inside one workgroup
__global__ __launch_bounds__(256 * KW) void wgrad_k(const float* __restrict__ dy, const float* __restrict__ x, int64_t m,
                                               int out, int in, int64_t ld_dy, int64_t ld_x, int64_t rows_per_split,
                                               float* __restrict__ part_w, float* __restrict__ part_b) {
  const int o_tiles = (out + kWgTileO - 1) / kWgTileO;
  const int i_tiles = (in + kWgTileI - 1) / kWgTileI;
  const int tiles = o_tiles * i_tiles;
  // XCD-aware mapping: workgroup b runs on XCD b % 8 (8 XCDs, private L2s).  All tiles of one K-slice read the
  // same rows of dY / X, so they are given block ids that are congruent mod 8: the slice is fetched from HBM
  // once and the other tiles hit in that XCD's L2.  (gridDim.x = nsplit8 * tiles, nsplit8 a multiple of 8.)
  const int xcd = blockIdx.x & 7;
  const int rest = blockIdx.x >> 3;
  const int tt = rest % tiles;
  const int s = (rest / tiles) * 8 + xcd;
  const int ot = tt / i_tiles, it = tt - ot * i_tiles;
  // 4*KW waves: wave = kw*4 + tile-wave.  The KW wave groups split this workgroup's K range (more waves per
  // SIMD to hide the HBM latency of the fragment loads without multiplying the split-K partials); their
  // accumulators are combined through LDS at the end.
  extern __shared__ __attribute__((aligned(16))) float red[];  // [(KW-1)][4 waves][34 values][64 lanes]
  const int wave_all = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int kw = wave_all >> 2, wave = wave_all & 3;
  const int wo = wave >> 1, wi = wave & 1;
  const int o0 = ot * kWgTileO + wo * 64;
  const int i0 = it * kWgTileI + wi * 32;
  const bool active = (o0 < out) && (i0 < in);  // inactive waves still reach the barrier; empty K slices write zeros
  const int col = lane & 31, kk = lane >> 5;
  const int64_t ks0 = (int64_t)s * rows_per_split < m ? (int64_t)s * rows_per_split : m;
  const int64_t ks1 = ks0 + rows_per_split < m ? ks0 + rows_per_split : m;
  // this wave group's slice of [ks0, ks1), boundaries on multiples of 2*U rows
  const int64_t per = sst_dev_align_up((ks1 - ks0 + KW - 1) / KW, 4 * U);
  int64_t k0 = ks0 + (int64_t)kw * per;
  int64_t k1 = k0 + per < ks1 ? k0 + per : ks1;
  if (k0 > ks1) k0 = ks1;
  if (!active) k1 = k0;

  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
  float bs0 = 0.f, bs1 = 0.f;
  // ragged dims (out, in not multiples of 32: SIR / VFE layers with 84, 133, 11 ... channels): columns past the
  // matrix are read from its last column (in bounds) and multiplied by zero; their products are never stored
  const bool a0ok = (o0 + col) < out, a1ok = (o0 + 32 + col) < out, bok = (i0 + col) < in;
  const float a0m = a0ok ? 1.f : 0.f, a1m = a1ok ? 1.f : 0.f, bm = bok ? 1.f : 0.f;
  const int ca0 = a0ok ? o0 + col : out - 1, ca1 = a1ok ? o0 + 32 + col : out - 1, cb = bok ? i0 + col : in - 1;
  const int o1off = ca1 - ca0;
  const float* pa = dy + (k0 + kk) * ld_dy + ca0;
  const float* pb = x + (k0 + kk) * ld_x + cb;
  // software pipeline: the loads of group g+1 are in flight while the MFMAs of group g issue.  The prefetch is
  // UNCONDITIONAL (past the last group the pointers simply stop advancing and the group is re-read): a
  // conditional prefetch puts a branch between the loads and their first use, and the waitcnt insertion then
  // waits at the join for (almost) every outstanding load, the fresh prefetch included (vmcnt(1) instead of
  // vmcnt(#prefetch) — measured: no overlap at all).
  auto load = [&](float (&a0)[U], float (&a1)[U], float (&b)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      a0[u] = pa[(int64_t)u * 2 * ld_dy] * a0m;
      a1[u] = pa[(int64_t)u * 2 * ld_dy + o1off] * a1m;
      b[u] = pb[(int64_t)u * 2 * ld_x] * bm;
    }
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
  // An EVEN number of groups goes through the pipeline (no exit test between a prefetch and its use — LLVM would
  // sink the loads below the exit branch); rows left over are handled by the tail loop.
  const int64_t ng = ((k1 - k0) / (4 * U)) * 2;
  float A0[U], A1[U], B0[U], C0[U], C1[U], D0[U];
  if (ng > 0) {
    load(A0, A1, B0);
    for (int64_t gi = 0; gi < ng; gi += 2) {
      pa += (int64_t)2 * U * ld_dy;
      pb += (int64_t)2 * U * ld_x;
      load(C0, C1, D0);
      comp(A0, A1, B0);
      const int64_t adv = gi + 2 < ng ? 2 * U : 0;
      pa += adv * ld_dy;
      pb += adv * ld_x;
      load(A0, A1, B0);
      comp(C0, C1, D0);
      // scheduling pattern for this block: [prefetch][MFMA group][prefetch][MFMA group]
      __builtin_amdgcn_sched_group_barrier(0x020, 3 * U, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 2 * U, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 3 * U, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 2 * U, 0);
    }
    pa += (int64_t)2 * U * ld_dy;
    pb += (int64_t)2 * U * ld_x;
  }
  int64_t k = k0 + ng * 2 * U;
  for (; k < k1; k += 2) {  // ragged tail, row-guarded
    const bool ok = (k + kk) < k1;
    const float a0 = ok ? pa[0] * a0m : 0.f;
    const float a1 = ok ? pa[o1off] * a1m : 0.f;
    const float b = ok ? pb[0] * bm : 0.f;
    acc0 = mfma32(a0, b, acc0);
    acc1 = mfma32(a1, b, acc1);
    bs0 += a0;
    bs1 += a1;
    pa += 2 * ld_dy;
    pb += 2 * ld_x;
  }
  if (KW > 1) {  // combine the K-slices of this workgroup: groups 1.. hand their accumulators to group 0
    if (kw > 0) {
      float* dst = red + ((size_t)((kw - 1) * 4 + wave) * 34) * 64 + lane;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        dst[r * 64] = acc0[r];
        dst[(16 + r) * 64] = acc1[r];
      }
      dst[32 * 64] = bs0;
      dst[33 * 64] = bs1;
    }
    __syncthreads();
    if (kw > 0 || !active) return;
#pragma unroll
    for (int g2 = 1; g2 < KW; ++g2) {
      const float* src = red + ((size_t)((g2 - 1) * 4 + wave) * 34) * 64 + lane;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc0[r] += src[r * 64];
        acc1[r] += src[(16 + r) * 64];
      }
      bs0 += src[32 * 64];
      bs1 += src[33 * 64];
    }
  } else if (!active) {
    return;
  }
  // C/D layout of 32x32: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  const int64_t pstride = (int64_t)out * in + (part_b != nullptr ? out : 0);  // per-split record: [dW | db]
  float* pw = part_w + (int64_t)s * pstride;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * kk;
    if (bok && o0 + row < out) pw[(int64_t)(o0 + row) * in + i0 + col] = acc0[r];
    if (bok && o0 + 32 + row < out) pw[(int64_t)(o0 + 32 + row) * in + i0 + col] = acc1[r];
  }
  if (part_b != nullptr && it == 0 && wi == 0) {
    bs0 += __shfl_xor(bs0, 32, 64);
    bs1 += __shfl_xor(bs1, 32, 64);
    if (lane < 32) {
      if (o0 + lane < out) part_b[(int64_t)s * pstride + o0 + lane] = bs0;
      if (o0 + 32 + lane < out) part_b[(int64_t)s * pstride + o0 + 32 + lane] = bs1;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Wide variant (out % 128 == 0, in % 64 == 0, out/128 * in/64 <= 4): ONE workgroup covers the whole dW for its
// K-slice, so every row of dY and X is read exactly once from L2 (the narrow kernel above re-reads X per
// out-tile and dY per in-tile and is bound by L2->L1 traffic at ~10 flop/B).  A wave owns a 128(out) x 64(in)
// tile = 4 x 2 MFMA tiles with INTERLEAVED columns: lane l loads dY[row][4*(l&31) .. +3] as one float4 and
// X[row][2*(l&31) .. +1] as one float2; component q of the float4 feeds MFMA tile q, whose row i = l&31
// therefore means output row 4*i + q (and column 2*j + p for the X tiles).  2 wide loads feed 8 MFMAs.
// Waves not needed to cover the output split the K range (KW groups), combined through LDS.
// ------------------------------------------------------------------------------------------------
template <int U>
__global__ __launch_bounds__(256) void wgrad_wide_k(const float* __restrict__ dy, const float* __restrict__ x,
                                                    int64_t m, int out, int in, int64_t ld_dy, int64_t ld_x,
                                                    int64_t rows_per_split, float* __restrict__ part_w,
                                                    float* __restrict__ part_b, int tiles) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [(KW-1) * ntile][130 values][64 lanes]
  SST_TS(0);
  const int o_w = out / 128, i_w = in / 64;  // wave tiles along out / in
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int col = lane & 31, kk = lane >> 5;
  int ntile, kwn, kw, tw, ow, iw, s;
  if (tiles > 0) {
    // Tiled mode: the workgroup owns ONE 128 x 64 tile of dW; its four waves split the K slice and meet in LDS.
    // tiles < 0 would be meaningless; tiles = o_w * i_w.  A K slice is shared by `tiles` workgroups that read the
    // same rows at the same time: consecutive j below differ in the tile only and blockIdx % 8 (the XCD, hence the
    // L2) is the same for all of them, so the second reader of a line hits in L2.
    ntile = 1;
    kwn = 4;
    kw = wave;
    tw = 0;
    const int b = blockIdx.x;
    const int nsplit = gridDim.x / tiles;
    int tile;
    if ((nsplit & 7) == 0) {
      const int j = b >> 3;
      tile = j % tiles;
      s = (j / tiles) * 8 + (b & 7);
    } else {
      tile = b % tiles;
      s = b / tiles;
    }
    ow = tile / i_w;
    iw = tile - ow * i_w;
  } else {
    ntile = o_w * i_w;  // 1, 2 or 4: one workgroup covers all of dW
    kwn = 4 / ntile;    // K groups inside the workgroup
    kw = wave / ntile;
    tw = wave - kw * ntile;
    ow = tw / i_w;
    iw = tw - ow * i_w;
    s = blockIdx.x;
  }
  const int o0 = ow * 128, i0 = iw * 64;
  const int64_t ks0 = (int64_t)s * rows_per_split < m ? (int64_t)s * rows_per_split : m;
  const int64_t ks1 = ks0 + rows_per_split < m ? ks0 + rows_per_split : m;
  const int64_t per = sst_dev_align_up((ks1 - ks0 + kwn - 1) / kwn, 4 * U);
  int64_t k0 = ks0 + (int64_t)kw * per;
  if (k0 > ks1) k0 = ks1;
  const int64_t k1 = k0 + per < ks1 ? k0 + per : ks1;

  f32x16 acc[4][2];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[q][p][r] = 0.f;
  float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
  const float* pa = dy + (k0 + kk) * ld_dy + o0 + 4 * col;
  const float* pb = x + (k0 + kk) * ld_x + i0 + 2 * col;

  // Software pipeline with a prefetch distance of one group (2*U rows): the loads of group g+1 are in flight while
  // the 8*U MFMAs of group g issue.  The compiler cannot express this: its scheduler sinks the prefetch next to
  // the first use (register pressure) and its waitcnt insertion then waits for everything, so with one wave per
  // SIMD every group paid a full memory latency (measured 54 % of the fp32 MFMA rate).  The loads are therefore
  // issued from inline asm (invisible to the waitcnt pass) and the wait is explicit: `s_waitcnt vmcnt(2*U)`
  // leaves exactly the fresh prefetch outstanding.  The wait statement takes the group's registers as in/out
  // operands, so every MFMA that consumes them is ordered after it by data dependence.
  float bs0 = 0.f, bs1 = 0.f, bs2 = 0.f, bs3 = 0.f;
  auto load = [&](f32x4 (&a)[U], f32x2 (&b)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(a[u]) : "v"(pa + (int64_t)u * 2 * ld_dy));
      asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(b[u]) : "v"(pb + (int64_t)u * 2 * ld_x));
    }
  };
  auto wait_prev = [&](f32x4 (&a)[U], f32x2 (&b)[U]) {  // all but the newest 2*U loads have landed
    static_assert(U == 2 || U == 4, "operand list below");
    if constexpr (U == 4)
      asm volatile("s_waitcnt vmcnt(8)"
                   : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
    else
      asm volatile("s_waitcnt vmcnt(4)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]));
  };
  // The MFMAs of the pipelined loop are asm statements as well: volatile asm keeps program order, so the previous
  // contents of a register group are dead before the next prefetch into it is issued (with builtin MFMAs, which
  // may float below the load statements, the allocator would have to copy the still-in-flight registers).
  // No MFMA -> MFMA hazard arises inside the loop: an accumulator is revisited after 7 other 16-pass MFMAs.
  auto comp = [&](const f32x4 (&a)[U], const f32x2 (&b)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
#define SST_MFMA(Q, P, AV, BV) \
  asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[Q][P]) : "v"(AV), "v"(BV))
      SST_MFMA(0, 0, a[u].x, b[u].x);
      SST_MFMA(1, 0, a[u].y, b[u].x);
      SST_MFMA(2, 0, a[u].z, b[u].x);
      SST_MFMA(3, 0, a[u].w, b[u].x);
      SST_MFMA(0, 1, a[u].x, b[u].y);
      SST_MFMA(1, 1, a[u].y, b[u].y);
      SST_MFMA(2, 1, a[u].z, b[u].y);
      SST_MFMA(3, 1, a[u].w, b[u].y);
#undef SST_MFMA
      // (asm for the same reason: a floating v_add would keep the old registers alive across the next prefetch)
      asm volatile("v_add_f32 %0, %0, %1" : "+v"(bs0) : "v"(a[u].x));
      asm volatile("v_add_f32 %0, %0, %1" : "+v"(bs1) : "v"(a[u].y));
      asm volatile("v_add_f32 %0, %0, %1" : "+v"(bs2) : "v"(a[u].z));
      asm volatile("v_add_f32 %0, %0, %1" : "+v"(bs3) : "v"(a[u].w));
    }
  };
  // An even number of groups goes through the pipeline; rows left over are handled by the tail loop.
  const int64_t ng = ((k1 - k0) / (4 * U)) * 2;
  f32x4 A0[U], C0[U];
  f32x2 B0[U], D0[U];
  SST_TS(1);
  if (ng > 0) {
    load(A0, B0);
    for (int64_t gi = 0; gi < ng; gi += 2) {
      pa += (int64_t)2 * U * ld_dy;
      pb += (int64_t)2 * U * ld_x;
      load(C0, D0);
      wait_prev(A0, B0);
      comp(A0, B0);
      const int64_t adv = gi + 2 < ng ? 2 * U : 0;  // past the end the last group is simply read again
      pa += adv * ld_dy;
      pb += adv * ld_x;
      load(A0, B0);
      wait_prev(C0, D0);
      comp(C0, D0);
    }
    // The redundant last prefetch must land before its registers are reused: the statement names them as in/out
    // operands, which keeps them allocated until here (otherwise the allocator hands them out right after the
    // loop and the late load overwrites e.g. a pointer).  The last MFMAs (16 passes, opaque to the hazard
    // recognizer) must have written their accumulators before anything reads them.
    if constexpr (U == 4)
      asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7"
                   : "+v"(A0[0]), "+v"(A0[1]), "+v"(A0[2]), "+v"(A0[3]), "+v"(B0[0]), "+v"(B0[1]), "+v"(B0[2]),
                     "+v"(B0[3])
                   :
                   : "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7"
                   : "+v"(A0[0]), "+v"(A0[1]), "+v"(B0[0]), "+v"(B0[1])
                   :
                   : "memory");
    pa += (int64_t)2 * U * ld_dy;
    pb += (int64_t)2 * U * ld_x;
    bsum = make_float4(bs0, bs1, bs2, bs3);
    SST_TS(3);
  }
  for (int64_t k = k0 + ng * 2 * U; k < k1; k += 2) {  // ragged tail, row-guarded
    const bool ok = (k + kk) < k1;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    float2 b = make_float2(0.f, 0.f);
    if (ok) {
      a = *(const float4*)pa;
      b = *(const float2*)pb;
    }
    acc[0][0] = mfma32(a.x, b.x, acc[0][0]);
    acc[1][0] = mfma32(a.y, b.x, acc[1][0]);
    acc[2][0] = mfma32(a.z, b.x, acc[2][0]);
    acc[3][0] = mfma32(a.w, b.x, acc[3][0]);
    acc[0][1] = mfma32(a.x, b.y, acc[0][1]);
    acc[1][1] = mfma32(a.y, b.y, acc[1][1]);
    acc[2][1] = mfma32(a.z, b.y, acc[2][1]);
    acc[3][1] = mfma32(a.w, b.y, acc[3][1]);
    bsum.x += a.x;
    bsum.y += a.y;
    bsum.z += a.z;
    bsum.w += a.w;
    pa += 2 * ld_dy;
    pb += 2 * ld_x;
  }
  if (tiles > 0) {
    // Tiled mode: all four waves (the four K quarters of one 128 x 64 tile) take part in the reduction.  Every wave
    // writes its 8 MFMA tiles to LDS; wave w then sums, in the fixed order of the K quarters, the tile pair
    // (q = w, p = 0 / 1) = output rows o0 + 4*row + w, and stores that quarter of the partial record.
    float* dst = red + (size_t)wave * 132 * 64 + lane;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[((q * 2 + p) * 16 + r) * 64] = acc[q][p][r];
    dst[128 * 64] = bsum.x;
    dst[129 * 64] = bsum.y;
    dst[130 * 64] = bsum.z;
    dst[131 * 64] = bsum.w;
    __syncthreads();
    SST_TS(4);
    const int q = wave;
    const int64_t pstride = (int64_t)out * in + (part_b != nullptr ? out : 0);
    float* pw = part_w + (int64_t)s * pstride;
    const float* src = red + ((size_t)q * 2 * 16) * 64 + lane;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v0 = 0.f, v1 = 0.f;
#pragma unroll
      for (int g2 = 0; g2 < 4; ++g2) {
        v0 += src[((size_t)g2 * 132 + r) * 64];
        v1 += src[((size_t)g2 * 132 + 16 + r) * 64];
      }
      const int row = (r & 3) + 8 * (r >> 2) + 4 * kk;
      *(float2*)(pw + (int64_t)(o0 + 4 * row + q) * in + i0 + 2 * col) = make_float2(v0, v1);
    }
    if (part_b != nullptr && iw == 0) {
      float b = 0.f;
#pragma unroll
      for (int g2 = 0; g2 < 4; ++g2) b += red[((size_t)g2 * 132 + 128 + q) * 64 + lane];
      b += __shfl_xor(b, 32, 64);
      if (lane < 32) part_b[(int64_t)s * pstride + o0 + 4 * lane + q] = b;
    }
    SST_TS(5);
    return;
  }
  if (kwn > 1) {  // K groups 1.. hand their accumulators to group 0 of the same wave tile
    if (kw > 0) {
      float* dst = red + ((size_t)((kw - 1) * ntile + tw) * 132) * 64 + lane;
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
          for (int r = 0; r < 16; ++r) dst[((q * 2 + p) * 16 + r) * 64] = acc[q][p][r];
      dst[128 * 64] = bsum.x;
      dst[129 * 64] = bsum.y;
      dst[130 * 64] = bsum.z;
      dst[131 * 64] = bsum.w;
    }
    __syncthreads();
    if (kw > 0) return;
    for (int g2 = 1; g2 < kwn; ++g2) {
      const float* src = red + ((size_t)((g2 - 1) * ntile + tw) * 132) * 64 + lane;
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[q][p][r] += src[((q * 2 + p) * 16 + r) * 64];
      bsum.x += src[128 * 64];
      bsum.y += src[129 * 64];
      bsum.z += src[130 * 64];
      bsum.w += src[131 * 64];
    }
  }
  // MFMA tile (q,p): D value r of lane (col, kk) = tile[row = (r&3) + 8*(r>>2) + 4*kk][col]
  //   -> dW[o0 + 4*row + q][i0 + 2*col + p]
  SST_TS(4);
  const int64_t pstride = (int64_t)out * in + (part_b != nullptr ? out : 0);
  float* pw = part_w + (int64_t)s * pstride;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * kk;
      *(float2*)(pw + (int64_t)(o0 + 4 * row + q) * in + i0 + 2 * col) = make_float2(acc[q][0][r], acc[q][1][r]);
    }
  }
  if (part_b != nullptr && iw == 0) {
    bsum.x += __shfl_xor(bsum.x, 32, 64);
    bsum.y += __shfl_xor(bsum.y, 32, 64);
    bsum.z += __shfl_xor(bsum.z, 32, 64);
    bsum.w += __shfl_xor(bsum.w, 32, 64);
    if (lane < 32) *(float4*)(part_b + (int64_t)s * pstride + o0 + 4 * lane) = bsum;
  }
  SST_TS(5);
}

// record e of every split summed: e < n_w -> dw[e], else db[e - n_w].
// Block = 32 consecutive elements x 8 interleaved slices of the splits, combined through LDS (deterministic).
__global__ __launch_bounds__(256) void wgrad_reduce_k(const float* __restrict__ part, int nsplit, int64_t n_w,
                                                      int64_t n_b, float* __restrict__ dw, float* __restrict__ db) {
  __shared__ float red[8][33];
  const int cx = threadIdx.x & 31, gy = threadIdx.x >> 5;
  const int64_t e = (int64_t)blockIdx.x * 32 + cx;
  const int64_t n = n_w + n_b;
  float a0 = 0.f, a1 = 0.f;
  if (e < n) {
    int s = gy;
    for (; s + 8 < nsplit; s += 16) {
      a0 += part[(int64_t)s * n + e];
      a1 += part[(int64_t)(s + 8) * n + e];
    }
    if (s < nsplit) a0 += part[(int64_t)s * n + e];
  }
  red[gy][cx] = a0 + a1;
  __syncthreads();
  if (gy == 0 && e < n) {
    float r = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) r += red[k][cx];
    if (e < n_w)
      dw[e] = r;
    else
      db[e - n_w] = r;
  }
}

constexpr int kGroupMax = 8;
struct wg_reduce_problem {
  const float* part;
  int nsplit;
  int64_t n_w, n_b;
  float* dw;
  float* db;
};
struct wg_reduce_group {
  wg_reduce_problem pr[kGroupMax];
  int n;
};

// wgrad_reduce_k for several problems in one launch (blockIdx.y = problem)
__global__ __launch_bounds__(256) void wgrad_reduce_group_k(const wg_reduce_group args) {
  __shared__ float red[8][33];
  const wg_reduce_problem& pr = args.pr[blockIdx.y];
  const int cx = threadIdx.x & 31, gy = threadIdx.x >> 5;
  const int64_t n = pr.n_w + pr.n_b;
  if ((int64_t)blockIdx.x * 32 >= n) return;
  const int64_t e = (int64_t)blockIdx.x * 32 + cx;
  float a0 = 0.f, a1 = 0.f;
  if (e < n) {
    int s = gy;
    for (; s + 8 < pr.nsplit; s += 16) {
      a0 += pr.part[(int64_t)s * n + e];
      a1 += pr.part[(int64_t)(s + 8) * n + e];
    }
    if (s < pr.nsplit) a0 += pr.part[(int64_t)s * n + e];
  }
  red[gy][cx] = a0 + a1;
  __syncthreads();
  if (gy == 0 && e < n) {
    float r = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) r += red[k][cx];
    if (e < pr.n_w)
      pr.dw[e] = r;
    else
      pr.db[e - pr.n_w] = r;
  }
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
  s = (s + 7) / 8 * 8;  // multiple of the XCD count
  if (s < 8) s = 8;
  if (s > 512) s = 512;
  int64_t rps = sst_div_up(m > 0 ? m : 1, s);
  rps = sst_align_up(rps, 16);
  if (rps < 64) rps = 64;
  *rows_per_split = rps;
  return s;  // some trailing slices may be empty (they write zero partials)
}

int wide_splits(int64_t m, int64_t* rows_per_split) {
  static int target = 0;
  if (target == 0) {
    const char* e = getenv("SST_WGRAD_WIDE_WGS");
    target = e ? atoi(e) : 256;
    if (target < 1) target = 256;
  }
  int64_t rps = sst_align_up(sst_div_up(m > 0 ? m : 1, target), 16);
  if (rps < 64) rps = 64;
  *rows_per_split = rps;
  return (int)sst_div_up(m > 0 ? m : 1, rps);
}

}  // namespace

extern "C" {

#ifdef SST_WGRAD_TIMING
int sst_debug_wgrad_timestamps(void* host_dst, int64_t bytes) {
  SST_HIP(hipDeviceSynchronize());
  SST_HIP(hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_wg_ts), (size_t)bytes));
  return SST_OK;
}
#endif

int64_t sst_weight_grad_workspace_bytes(int64_t m, int out, int in) {
  int64_t rps;
  int s = pick_splits(m, out, in, &rps);
  const int sw = wide_splits(m, &rps);
  if (sw > s) s = sw;
  return sst_align_up((int64_t)s * ((int64_t)out * in + out) * sizeof(float), 256) + 256;
}

// split-K partial records of one problem into d_workspace (the kernel choice of sst_weight_grad_f32); *splits = records written
static int launch_partials(const float* d_dy, const float* d_x, int64_t m, int out, int in, int64_t ld_dy, int64_t ld_x,
                           bool want_bias, void* d_workspace, hipStream_t st, int* splits) {
  const int64_t nw = (int64_t)out * in;
  float* part_w = (float*)d_workspace;
  float* part_b = want_bias ? part_w + nw : nullptr;  // bias sums sit behind the dW block of each split record
  int64_t rps;
  int s;
  // Tiled mode (default): any out % 128 == 0, in % 64 == 0 with up to 64 tiles; SST_WGRAD_TILED=0 selects the older
  // form where one workgroup covers all of dW (at most 4 wave tiles).
  static int tiled_env = -1;
  if (tiled_env < 0) {
    const char* e = getenv("SST_WGRAD_TILED");
    tiled_env = e ? atoi(e) : 1;
  }
  const int ntile = (out % 128 == 0 && in % 64 == 0) ? (out / 128) * (in / 64) : 0;
  const bool aligned = (ld_dy % 4 == 0) && (ld_x % 2 == 0) && (((uintptr_t)d_dy & 15) == 0) && (((uintptr_t)d_x & 7) == 0);
  const bool tiled = tiled_env != 0 && aligned && ntile >= 1 && ntile <= 64;
  const bool wide = tiled || (aligned && ntile >= 1 && ntile <= 4 && ntile != 3);
  if (wide) {
    size_t lds;
    int grid;
    if (tiled) {
      // K slices: as many as fill the 256 CUs with `ntile` workgroups each, a multiple of 8 so that the workgroups
      // sharing a slice sit on one XCD; every wave's chunk (a quarter of the slice) is a multiple of 16 rows.
      int s0 = (256 / ntile) & ~7;
      if (s0 < 8) s0 = 8;
      rps = sst_align_up(sst_div_up(m, s0), 64);
      s = (int)sst_div_up(m, rps);
      grid = s * ntile;
      lds = (size_t)4 * 132 * 64 * sizeof(float);
    } else {
      s = wide_splits(m, &rps);
      grid = s;
      lds = (size_t)(4 / ntile - 1) * ntile * 132 * 64 * sizeof(float);
    }
    static int wide_u = 0;
    if (wide_u == 0) {
      const char* e = getenv("SST_WGRAD_U");
      wide_u = e ? atoi(e) : 4;
    }
    static unsigned long long configured = 0;  // once: the attribute call costs tens of microseconds on the host
    if (sst_first_use_on_device(&configured)) {
      const int lds_max = 4 * 132 * 64 * (int)sizeof(float);
      SST_HIP(hipFuncSetAttribute((const void*)wgrad_wide_k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
      SST_HIP(hipFuncSetAttribute((const void*)wgrad_wide_k<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
      sst_mark_device(&configured);
    }
    const int tiles_arg = tiled ? ntile : 0;
    if (wide_u == 2) {
      hipLaunchKernelGGL(wgrad_wide_k<2>, dim3((unsigned)grid), dim3(256), lds, st, d_dy, d_x, m, out, in, ld_dy, ld_x,
                         rps, part_w, part_b, tiles_arg);
    } else {
      hipLaunchKernelGGL(wgrad_wide_k<4>, dim3((unsigned)grid), dim3(256), lds, st, d_dy, d_x, m, out, in, ld_dy, ld_x,
                         rps, part_w, part_b, tiles_arg);
    }
  } else {
    s = pick_splits(m, out, in, &rps);
    const int tiles = ((out + kWgTileO - 1) / kWgTileO) * ((in + kWgTileI - 1) / kWgTileI);
    constexpr int KW = 2;
    const size_t lds = (size_t)(KW - 1) * 4 * 34 * 64 * sizeof(float);
    static unsigned long long configured_narrow = 0;
    if (sst_first_use_on_device(&configured_narrow)) {
      SST_HIP(hipFuncSetAttribute((const void*)wgrad_k<8, KW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      sst_mark_device(&configured_narrow);
    }
    hipLaunchKernelGGL((wgrad_k<8, KW>), dim3((unsigned)(s * tiles)), dim3(256 * KW), lds, st, d_dy, d_x, m, out, in,
                       ld_dy, ld_x, rps, part_w, part_b);
  }
  *splits = s;
  return SST_OK;
}

int sst_weight_grad_f32(const float* d_dy, const float* d_x, int64_t m, int out, int in, int64_t ld_dy, int64_t ld_x,
                        float* d_dw, float* d_db, void* d_workspace, void* stream) {
  if (m < 0 || out < 1 || in < 1 || out > 4096 || in > 4096) return SST_ERR_UNSUPPORTED;
  if (!d_dw || !d_workspace || ld_dy < out || ld_x < in) return SST_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (m == 0) {
    SST_HIP(hipMemsetAsync(d_dw, 0, sizeof(float) * out * in, st));
    if (d_db) SST_HIP(hipMemsetAsync(d_db, 0, sizeof(float) * out, st));
    return SST_OK;
  }
  if (!d_dy || !d_x) return SST_ERR_ARG;
  int s = 0;
  const int rc = launch_partials(d_dy, d_x, m, out, in, ld_dy, ld_x, d_db != nullptr, d_workspace, st, &s);
  if (rc) return rc;
  const int64_t nw = (int64_t)out * in, nb = d_db ? out : 0;
  hipLaunchKernelGGL(wgrad_reduce_k, dim3((unsigned)sst_div_up(nw + nb, 32)), dim3(256), 0, st, (const float*)d_workspace, s,
                     nw, nb, d_dw, d_db);
  SST_LAUNCH_CHECK();
  return SST_OK;
}

// Several problems (the five parameter gradients of an encoder layer): the same split-K kernels, launched one after the
// other into their own slices of the workspace, and ONE reduction launch for all of them (5 us saved per problem).
int64_t sst_weight_grad_group_workspace_bytes(const sst_wgrad_problem_f32* problems, int n) {
  int64_t total = 0;
  for (int i = 0; i < n; ++i) total += sst_weight_grad_workspace_bytes(problems[i].m, problems[i].out, problems[i].in);
  return total;
}

int sst_weight_grad_group_f32(const sst_wgrad_problem_f32* problems, int n, void* d_workspace, void* stream) {
  if (n < 1 || n > kGroupMax || !problems || !d_workspace) return SST_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  wg_reduce_group args;
  args.n = n;
  char* ws = (char*)d_workspace;
  unsigned max_blocks = 1;
  for (int i = 0; i < n; ++i) {
    const sst_wgrad_problem_f32& q = problems[i];
    if (q.m < 1 || q.out < 1 || q.in < 1 || q.out > 4096 || q.in > 4096) return SST_ERR_UNSUPPORTED;
    if (q.x_add_rows || q.x_add_index) return SST_ERR_UNSUPPORTED;   // rows added to X on load: the exact-split group only
    if (!q.dy || !q.x || !q.dw || q.ld_dy < q.out || q.ld_x < q.in) return SST_ERR_ARG;
    int s = 0;
    const int rc = launch_partials(q.dy, q.x, q.m, q.out, q.in, q.ld_dy, q.ld_x, q.db != nullptr, ws, st, &s);
    if (rc) return rc;
    args.pr[i].part = (const float*)ws;
    args.pr[i].nsplit = s;
    args.pr[i].n_w = (int64_t)q.out * q.in;
    args.pr[i].n_b = q.db ? q.out : 0;
    args.pr[i].dw = q.dw;
    args.pr[i].db = q.db;
    const unsigned blocks = (unsigned)sst_div_up(args.pr[i].n_w + args.pr[i].n_b, 32);
    if (blocks > max_blocks) max_blocks = blocks;
    ws += sst_weight_grad_workspace_bytes(q.m, q.out, q.in);
  }
  hipLaunchKernelGGL(wgrad_reduce_group_k, dim3(max_blocks, (unsigned)n), dim3(256), 0, st, args);
  SST_LAUNCH_CHECK();
  return SST_OK;
}

}  // extern "C"
