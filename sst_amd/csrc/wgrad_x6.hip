// Weight / bias gradients of the tall linear layers on the bf16 matrix pipe with the EXACT three-way operand split of
// csrc/dense_f32x6.hip ("f32x6"):   dW[out, in] = dY[M, out]^T X[M, in],  db[out] = column sums of dY,   fp32 in, fp32 out;
// dy = d0 + d1 + d2, x = x0 + x1 + x2 (bf16 parts, exact), the six products d_i x_j with i + j <= 2 accumulated in fp32 by
// v_mfma_f32_16x16x32_bf16 - what is dropped is below 2^-24 of a product, i.e. below the rounding of an fp32 FMA chain.
// Replaces, in that mode, wgrad_wide_k of csrc/wgrad.hip (exact fp32 matrix pipe: 2.65 + 0.3 ms of the 12.6 ms step, bound
// by that pipe at 78 % of its sustained rate) for the five parameter gradients of an SRA encoder layer
// (autograd of sst_basic_block_v2.py:41-126): the contraction runs over the TOKENS, 6 x 16 instead of 8 x 32 matrix-pipe
// cycles per 16 x 16 x 32 block, so the kernel is bound by reading dY and X once (HBM).
//
// The contraction index is the row index of both operands in memory (token-major rows), while an MFMA operand wants 8
// consecutive k (= tokens) of ONE column per lane: the transposition happens on the way into LDS.  Per 32-token step a
// thread loads a 4-token x 4-column block of one operand (four 16-byte loads of consecutive rows: full lines across the
// wave), splits it in registers into the three bf16 parts and stores, per part, the block TRANSPOSED as 4 columns x
// (4 tokens = 8 bytes) = 32 contiguous bytes (two ds_write_b128).  LDS image of a 16-column tile:
//     [token-quad parity][token-quad >> 1][16 columns][4 tokens] bf16      (1 KB; tiles 1040 B apart)
// so that the fragment of lane (column c, token group g) - tokens 8 g .. 8 g + 7 - is two ds_read_b64 512 B apart whose 32
// lanes (16 columns x 2 token groups) cover all 64 banks exactly once, and the eight lanes of a ds_write_b128 group hit
// eight different 4-bank slots (the 16-byte tile skew separates odd from even tiles).
//
// One launch for ALL problems of a group (the two or three gradients that share an operand in the backward of a layer):
// workgroup = (128 x 128 tile of one dW, token slice), 8 waves = 4 (out) x 2 (in) sub-tiles of 32 x 64; id = tile * S + slice
// with S % 8 == 0: the tiles of one slice read the same dY / X rows and sit on the same XCD (ids S apart).  Partial tiles
// (fp32) go to the workspace, ONE reduction launch sums them in slice order (deterministic).
#include <math.h>
#include <stdlib.h>
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
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
__device__ __forceinline__ void split2(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) {
  p0 = pack2(a, b);
  const float ra = a - lo_f(p0), rb = b - hi_f(p0);
  p1 = pack2(ra, rb);
  p2 = pack2(ra - lo_f(p1), rb - hi_f(p1));
}

constexpr int kMaxProblems = 8;
constexpr int kTile = 128;                    // dW tile: 128 (out) x 128 (in)
constexpr int kStep = 32;                     // tokens per step
constexpr int kTileLds = 1040;                // bytes of one 16-column tile image (1 KB + 16-byte skew)
constexpr int kImage = 8 * kTileLds;          // 128 columns of one part of one operand
constexpr int kStage = 2 * 3 * kImage;        // both operands, three parts
constexpr int kLdsBytes = 2 * kStage;         // double buffered: 99 840 B

struct x6_problem {
  const float* dy;
  const float* x;
  int64_t m, ld_dy, ld_x;
  int out, in;
  int tile0;        // first tile (global numbering) of this problem; tiles run over (out / 128) x (in / 128), in fastest
  int db_off;       // offset of this problem's bias-gradient partials in the db workspace, -1: no bias gradient
  float* dw;        // [out, in] contiguous
  float* db;        // [out] or NULL
  const float* x_add_rows;      // not NULL: the X operand is x + x_add_rows[x_add_index[token]] (rows of width `in`, the positional
  const int32_t* x_add_index;   // table of an encoder layer): "x + pos" is formed on the way into LDS, never in memory
};
struct x6_group {
  x6_problem p[kMaxProblems];
  int n, tiles, slices;
  int64_t tokens_per_slice;
};

// partial dW tiles: part[(tile * slices + slice) * 128 * 128]; partial db: dbp[(db_off + out column) * slices + slice]
__global__ __launch_bounds__(512, 2) void wgrad_x6_k(const x6_group G, float* __restrict__ part, float* __restrict__ dbp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int tile = blockIdx.x / G.slices, slice = blockIdx.x - tile * G.slices;
  int pi = 0;
#pragma unroll
  for (int q = 1; q < kMaxProblems; ++q)
    if (q < G.n && tile >= G.p[q].tile0) pi = q;
  const x6_problem P = G.p[pi];
  const int tiles_k = P.in / kTile;
  const int tl = tile - P.tile0, tn = tl / tiles_k, tk = tl - tn * tiles_k;
  const int n0 = tn * kTile, k0 = tk * kTile;
  const int64_t t_begin = (int64_t)slice * G.tokens_per_slice;
  int64_t t_end = t_begin + G.tokens_per_slice;
  t_end = t_end < P.m ? t_end : P.m;

  // staging role: threads 0..255 the dY block (columns n0 ..), 256..511 the X block (columns k0 ..): token quad tq, columns 4 cg ..
  const int op = tid >> 8, tq = (tid >> 5) & 7, cg = tid & 31;
  const float* src = op == 0 ? P.dy + n0 + 4 * cg : P.x + k0 + 4 * cg;
  const int64_t ld = op == 0 ? P.ld_dy : P.ld_x;
  // LDS address of this thread's 32 bytes (columns 4 cg .. 4 cg + 3 of tile cg / 4) inside a part image
  const int st_off = (cg >> 2) * kTileLds + (tq & 1) * 512 + (tq >> 1) * 128 + (cg & 3) * 32;
  // ring of kPf register sets: the rows of step s + kPf are requested right after the set of step s has been consumed, so that
  // kPf - 1 steps of matrix work (0.4 us each) stand between a request and its use - one step does not cover an HBM round trip
  // (measured: the kernel is NOT bound there - ablations: skeleton of LDS fragment reads + barriers alone 63 of 107 us per group,
  // each wave re-reads 18 fragments (18 KB) per step from LDS for its 48 products - a ring of 4 sets changed nothing).  Loads are UNCONDITIONAL (row clamped, zeroed at the split): no branch
  // around them, so the compiler's vmcnt counts stay exact.
  constexpr int kPf = 2;
  f32x4 rows[kPf][4];
  const int64_t t_last = t_end > 0 ? t_end - 1 : 0;
  // positional rows of the X operand (x_add_rows): requested with the rows they are added to; their token indices are loaded
  // one request ahead so that a table address never waits for its index
  const bool xadd = op == 1 && P.x_add_rows != nullptr;
  const float* addsrc = xadd ? P.x_add_rows + k0 + 4 * cg : nullptr;
  f32x4 prow[kPf][4];
  int32_t pidx[4] = {0, 0, 0, 0};
  auto load_idx = [&](int64_t t0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int64_t t = t0 + 4 * tq + r;
      t = t < t_last ? t : t_last;
      pidx[r] = P.x_add_index[t];
    }
  };
  auto load_rows = [&](int64_t t0, f32x4 (&dst)[4], f32x4 (&pdst)[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int64_t t = t0 + 4 * tq + r;
      t = t < t_last ? t : t_last;
      dst[r] = *(const f32x4*)(src + t * ld);
    }
    if (xadd) {
#pragma unroll
      for (int r = 0; r < 4; ++r) pdst[r] = *(const f32x4*)(addsrc + (int64_t)pidx[r] * P.in);
      load_idx(t0 + kStep);      // the indices of the NEXT request (requests are one step apart)
    }
  };
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};
  // the split of one column (4 tokens) of the staged block: called between the MFMA groups of the running step, so that its ~30
  // VALU instructions issue in the shadow of the matrix pipe instead of after it (all 8 waves of the workgroup move in lock
  // step from barrier to barrier: work that is not interleaved is serial)
  u32x4 img[3][2];
  auto split_col = [&](int e, int64_t t0, const f32x4 (&rws)[4], const f32x4 (&prw)[4]) {
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = (t0 + 4 * tq + r < t_end) ? (xadd ? rws[r][e] + prw[r][e] : rws[r][e]) : 0.f;
    if (op == 0) bsum[e] += (v[0] + v[1]) + (v[2] + v[3]);
    unsigned a0, a1, a2, b0, b1, b2;
    split2(v[0], v[1], a0, a1, a2);
    split2(v[2], v[3], b0, b1, b2);
    img[0][e >> 1][2 * (e & 1)] = a0;
    img[0][e >> 1][2 * (e & 1) + 1] = b0;
    img[1][e >> 1][2 * (e & 1)] = a1;
    img[1][e >> 1][2 * (e & 1) + 1] = b1;
    img[2][e >> 1][2 * (e & 1)] = a2;
    img[2][e >> 1][2 * (e & 1) + 1] = b2;
  };
  auto store_img = [&](int buf) {
    unsigned char* base = lds + buf * kStage + op * 3 * kImage + st_off;
#pragma unroll
    for (int im = 0; im < 3; ++im) {
      *(u32x4*)(base + im * kImage) = img[im][0];
      *(u32x4*)(base + im * kImage + 16) = img[im][1];
    }
  };
  // compute role: wave = (wn, wk): out rows 32 wn .. + 31 (2 tiles), in columns 64 wk .. + 63 (4 tiles) of the 128 x 128 tile.
  // Transposed product D'[k][n] = sum_t X[t][k] dY[t][n]: A = X columns, B = dY columns, so that a lane ends with 4
  // consecutive k of one n = 16 contiguous bytes of dW[n][k ..].
  const int wn = wave & 3, wk = wave >> 2;
  const int frag_off = g * 128 + l15 * 8;
  // two accumulator sets: the leading product d0 x0 alone (one accumulation per step into the large running sum) and the five
  // correction products (2^-8 .. 2^-16 of it: their roundings are that much smaller) - summed once at the end.  With a single
  // set the 6 x 44 accumulations per slice put the result 2.1 x further from float64 than the fp32-pipe kernel at 90 k tokens.
  f32x4 acc[4][2], acl[4][2];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = acl[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  auto frag = [&](const unsigned char* image, int tile16) -> u32x4 {
    const unsigned char* p = image + tile16 * kTileLds + frag_off;
    const u32x2 lo = *(const u32x2*)p, hi = *(const u32x2*)(p + 512);
    return (u32x4){lo[0], lo[1], hi[0], hi[1]};
  };

  if (xadd) load_idx(t_begin);
#pragma unroll
  for (int u = 0; u < kPf; ++u) load_rows(t_begin + (int64_t)u * kStep, rows[u], prow[u]);
#pragma unroll
  for (int e = 0; e < 4; ++e) split_col(e, t_begin, rows[0], prow[0]);
  store_img(0);
  load_rows(t_begin + (int64_t)kPf * kStep, rows[0], prow[0]);
  __syncthreads();
  // step s (tokens t_begin + 32 s ..): products on buffer s & 1, and between its four MFMA groups the four columns of step
  // s + 1 (ring set (s + 1) % kPf) are split; then stored into buffer (s + 1) & 1 and the set refilled with step s + 1 + kPf
  const int64_t n_steps = (t_end - t_begin + kStep - 1) / kStep;
  for (int64_t s0 = 0; s0 < n_steps; s0 += kPf) {
#pragma unroll
    for (int u = 0; u < kPf; ++u) {
      const int64_t s = s0 + u;
      if (s >= n_steps) break;
      const int64_t t1 = t_begin + (s + 1) * kStep;
      const unsigned char* dyi = lds + (u & 1) * kStage;       // dY parts (s0 is a multiple of kPf, kPf even: s & 1 == u & 1)
      const unsigned char* xi = dyi + 3 * kImage;               // X parts
      u32x4 b0[2], b1[2], b2[2];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        b0[b] = frag(dyi, 2 * wn + b);
        b1[b] = frag(dyi + kImage, 2 * wn + b);
        b2[b] = frag(dyi + 2 * kImage, 2 * wn + b);
      }
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const u32x4 a0 = frag(xi, 4 * wk + a), a1 = frag(xi + kImage, 4 * wk + a), a2 = frag(xi + 2 * kImage, 4 * wk + a);
        // corrections, smallest first: (2,0) (0,2) (1,1) ~ 2^-16, (1,0) (0,1) ~ 2^-8; then the leading product
#pragma unroll
        for (int b = 0; b < 2; ++b) acl[a][b] = mma32(a2, b0[b], acl[a][b]);
#pragma unroll
        for (int b = 0; b < 2; ++b) acl[a][b] = mma32(a0, b2[b], acl[a][b]);
#pragma unroll
        for (int b = 0; b < 2; ++b) acl[a][b] = mma32(a1, b1[b], acl[a][b]);
#pragma unroll
        for (int b = 0; b < 2; ++b) acl[a][b] = mma32(a1, b0[b], acl[a][b]);
#pragma unroll
        for (int b = 0; b < 2; ++b) acl[a][b] = mma32(a0, b1[b], acl[a][b]);
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = mma32(a0, b0[b], acc[a][b]);
        split_col(a, t1, rows[(u + 1) % kPf], prow[(u + 1) % kPf]);
      }
      store_img((u + 1) & 1);
      load_rows(t1 + (int64_t)kPf * kStep, rows[(u + 1) % kPf], prow[(u + 1) % kPf]);
      __syncthreads();
    }
  }
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] += acl[a][b];
  // D'[k = 4 g + r][n = l15] of sub-tile (a, b): dW[n0 + 32 wn + 16 b + l15][k0 + 64 wk + 16 a + 4 g .. + 3]
  float* dst = part + ((int64_t)tile * G.slices + slice) * (kTile * kTile);
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
      *(f32x4*)(dst + (32 * wn + 16 * b + l15) * kTile + 64 * wk + 16 * a + 4 * g) = acc[a][b];
  // bias gradient: column sums of the dY block, from the k = 0 tiles only (every tile of a row of tiles sees the same dY)
  if (P.db_off >= 0 && tk == 0) {
    float* red = (float*)lds;               // [8 token quads][128 columns]; the stage buffers are dead (barrier above)
    if (op == 0) *(f32x4*)(red + tq * 128 + 4 * cg) = (f32x4){bsum[0], bsum[1], bsum[2], bsum[3]};
    __syncthreads();
    if (tid < 128) {
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) s += red[q * 128 + tid];
      dbp[((int64_t)P.db_off + n0 + tid) * G.slices + slice] = s;
    }
  }
}

// dW[problem][n][k] = sum over slices of the partial tiles, in float64 (one rounding at the end; slice order fixed:
// deterministic).  Workgroup = 32 output quads (16 bytes each) x 8 slice groups: thread (quad, group) sums the slices group,
// group + 8, ... with every load independent, the groups meet in LDS.  (First version: one thread walking all slices of its
// quad, 64 workgroups - 26 us for 17 MB; second: 64 quads x 4 groups, 256 workgroups - 14 us; 16 quads x 16 groups with
// 4-byte stores: 256-byte runs per wave and load, 30 us.)
//
// Optional riders (sst_colsum_rider, csrc/common.h; the FIRST workgroups of the launch): the column sums of another kernel's
// block partials - the LayerNorm backward's d(gamma) | d(beta) partials [nb][2 c] (csrc/dense.hip add_ln_bwd_*), in the
// arithmetic of colsum_partials_k (32 strided partial sums per column, added in order), so that the one-call layer executor
// needs no finishing launch of its own for them.
struct x6_riders {
  sst_colsum_rider r[2];
  int n, blocks0;          // blocks0: workgroups of rider 0 (the rest of the rider blocks belong to rider 1)
};

constexpr int kRedQuads = 32, kRedGroups = 8;

__global__ __launch_bounds__(256) void wgrad_x6_reduce_k(const x6_group G, const float* __restrict__ part,
                                                         const float* __restrict__ dbp, int rider_blocks,
                                                         const x6_riders R) {
  __shared__ double red[kRedGroups][kRedQuads][4];
  if ((int)blockIdx.x < rider_blocks) {
    const int second = (int)blockIdx.x >= R.blocks0 ? 1 : 0;
    const sst_colsum_rider& J = R.r[second];
    const int rblk = (int)blockIdx.x - (second ? R.blocks0 : 0);
    // column sums of the rider (the FIRST workgroups: they start with the launch and end inside it): this workgroup = 32
    // columns; thread (cx, gq) forms the four strided sums gy = gq + 8 j side by side, each in colsum_partials_k's order
    float* fr = (float*)&red[0][0][0];                       // [32][33] floats
    const int cx = threadIdx.x & 31, gq = threadIdx.x >> 5;
    const int i = rblk * 32 + cx;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (i < J.width) {
      // the partials were written several kernels ago (600 MB of operands have passed through the caches since): every load
      // is an HBM round trip.  All 64 of them are requested before the first is added (row clamped, value masked): a loop that
      // waited per step took 30 us.  nb <= 512 (the LayerNorm backward's grid cap; checked by the host).
      const float* src = J.partials + i;
      // 512 partial rows at a time (the LayerNorm backward's grid cap; the one-kernel layer tail, csrc/layer_tail_x6.hip, has one
      // row per 64 tokens): same order of additions per strided sum as colsum_partials_k whatever nb is
      for (int base = 0; base < J.nb; base += 512) {
        float v[16][4];
#pragma unroll
        for (int it = 0; it < 16; ++it)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int b = base + 32 * it + gq + 8 * j;
            v[it][j] = src[(int64_t)(b < J.nb ? b : J.nb - 1) * J.width];
          }
#pragma unroll
        for (int it = 0; it < 16; ++it)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (base + 32 * it + gq + 8 * j < J.nb) acc[j] += v[it][j];
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) fr[(gq + 8 * j) * 33 + cx] = acc[j];
    __syncthreads();
    if (gq == 0 && i < J.width) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 32; ++k) t += fr[k * 33 + cx];
      if (i < J.split)
        J.out0[i] = t;
      else
        J.out1[i - J.split] = t;
    }
    return;
  }
  const int blk = (int)blockIdx.x - rider_blocks;
  const int ql = threadIdx.x & (kRedQuads - 1), sg = threadIdx.x / kRedQuads;
  const int64_t quad = (int64_t)blk * kRedQuads + ql;          // over tiles x 128 x 32
  const int64_t tile = quad / (kTile * kTile / 4);
  const int e4 = (int)(quad - tile * (kTile * kTile / 4));
  double d0 = 0.0, d1 = 0.0, d2 = 0.0, d3 = 0.0;
  if (tile < G.tiles) {
    const float* src = part + tile * G.slices * (int64_t)(kTile * kTile) + 4 * e4;
#pragma unroll 8
    for (int q = sg; q < G.slices; q += kRedGroups) {
      const f32x4 v = *(const f32x4*)(src + (int64_t)q * (kTile * kTile));
      d0 += v[0], d1 += v[1], d2 += v[2], d3 += v[3];
    }
  }
  red[sg][ql][0] = d0, red[sg][ql][1] = d1, red[sg][ql][2] = d2, red[sg][ql][3] = d3;
  __syncthreads();
  if (sg == 0 && tile < G.tiles) {
    f32x4 s;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double t = 0.0;
#pragma unroll
      for (int k = 0; k < kRedGroups; ++k) t += red[k][ql][r];
      s[r] = (float)t;
    }
    int pi = 0;
#pragma unroll
    for (int q = 1; q < kMaxProblems; ++q)
      if (q < G.n && tile >= G.p[q].tile0) pi = q;
    const int tiles_k = G.p[pi].in / kTile;
    const int tl = (int)tile - G.p[pi].tile0, tn = tl / tiles_k, tk = tl - tn * tiles_k;
    const int n = tn * kTile + e4 / 32, k = tk * kTile + (e4 % 32) * 4;
    *(f32x4*)(G.p[pi].dw + (int64_t)n * G.p[pi].in + k) = s;
  }
  // bias gradients ride on the first blocks: one thread per output column
  const int64_t col = (int64_t)blk * 256 + threadIdx.x;
  int base = 0;
  for (int q = 0; q < G.n; ++q) {
    if (G.p[q].db_off < 0) continue;
    if (col >= base && col < base + G.p[q].out) {
      const float* src = dbp + ((int64_t)G.p[q].db_off + (col - base)) * G.slices;
      double s = 0.0;
      for (int i = 0; i < G.slices; ++i) s += src[i];
      G.p[q].db[col - base] = (float)s;
    }
    base += G.p[q].out;
  }
}

bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

struct x6_plan {
  x6_group g;
  int64_t part_bytes, db_bytes;
};

// problems -> launch geometry; false when a problem is outside what the kernel is built for
bool make_plan(const sst_wgrad_problem_f32* pr, int n, x6_plan* plan) {
  if (!pr || n < 1 || n > kMaxProblems) return false;
  x6_group& G = plan->g;
  G.n = n;
  int tiles = 0, db_cols = 0;
  int64_t m = pr[0].m;
  for (int i = 0; i < n; ++i) {
    const sst_wgrad_problem_f32& p = pr[i];
    if (p.m != m || p.m < 1 || p.out < kTile || p.in < kTile || (p.out % kTile) || (p.in % kTile) || (p.ld_dy & 3) ||
        (p.ld_x & 3) || !p.dy || !p.x || !p.dw || !aligned16(p.dy) || !aligned16(p.x) || !aligned16(p.dw))
      return false;
    G.p[i].dy = p.dy;
    G.p[i].x = p.x;
    G.p[i].m = p.m;
    G.p[i].ld_dy = p.ld_dy;
    G.p[i].ld_x = p.ld_x;
    G.p[i].out = p.out;
    G.p[i].in = p.in;
    G.p[i].tile0 = tiles;
    G.p[i].db_off = p.db ? db_cols : -1;
    G.p[i].dw = p.dw;
    G.p[i].db = p.db;
    if ((p.x_add_rows != nullptr) != (p.x_add_index != nullptr) || (p.x_add_rows && !aligned16(p.x_add_rows))) return false;
    G.p[i].x_add_rows = p.x_add_rows;
    G.p[i].x_add_index = p.x_add_index;
    tiles += (p.out / kTile) * (p.in / kTile);
    if (p.db) db_cols += p.out;
  }
  G.tiles = tiles;
  // one workgroup per CU (100 KB of LDS each): tiles x slices <= 256, slices a multiple of 8 (same-XCD placement of a slice)
  int slices = (256 / tiles) & ~7;
  if (slices < 8) slices = 8;
  const int64_t max_slices = sst_div_up(m, (int64_t)kStep);
  while (slices > 8 && slices > max_slices) slices -= 8;
  G.slices = slices;
  G.tokens_per_slice = sst_align_up(sst_div_up(m, (int64_t)slices), (int64_t)kStep);
  plan->part_bytes = sst_align_up((int64_t)tiles * slices * kTile * kTile * 4, 256);
  plan->db_bytes = sst_align_up((int64_t)(db_cols > 0 ? db_cols : 1) * slices * 4, 256);
  return true;
}

}  // namespace

extern "C" {

int64_t sst_weight_grad_group_f32x6_workspace_bytes(const sst_wgrad_problem_f32* problems, int n) {
  x6_plan plan;
  if (!make_plan(problems, n, &plan)) return SST_ERR_UNSUPPORTED;
  return plan.part_bytes + plan.db_bytes;
}

}  // extern "C"

// the group launch with a rider for the reduction kernel (csrc/layer_exec.hip: the LayerNorm backward's parameter-gradient
// partials): partials [nb][width] -> out0 (columns < split) | out1
int sst_internal_weight_grad_group_f32x6(const sst_wgrad_problem_f32* problems, int n, void* d_workspace,
                                         const sst_colsum_rider* riders, int n_riders, void* stream) {
  x6_plan plan;
  if (!make_plan(problems, n, &plan)) return SST_ERR_UNSUPPORTED;
  if (!d_workspace || !aligned16(d_workspace)) return SST_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  float* part = (float*)d_workspace;
  float* dbp = (float*)((char*)d_workspace + plan.part_bytes);
  static unsigned long long configured = 0;
  if (sst_first_use_on_device(&configured)) {
    SST_HIP(hipFuncSetAttribute((const void*)wgrad_x6_k, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
    sst_mark_device(&configured);
  }
  const int64_t quads = (int64_t)plan.g.tiles * (kTile * kTile / 4);
  const int tile_blocks = (int)sst_div_up(quads, (int64_t)kRedQuads);
  if (n_riders < 0 || n_riders > 2 || (n_riders > 0 && !riders)) return SST_ERR_ARG;
  x6_riders R;
  R.n = n_riders, R.blocks0 = 0;
  int extra = 0;
  for (int i = 0; i < n_riders; ++i) {
    const sst_colsum_rider& q = riders[i];
    if (!q.partials || q.nb < 1 || q.width < 1 || !q.out0 || !q.out1) return SST_ERR_ARG;
    R.r[i] = q;
    if (i == 0) R.blocks0 = (q.width + 31) / 32;
    extra += (q.width + 31) / 32;
  }
  hipLaunchKernelGGL(wgrad_x6_k, dim3((unsigned)(plan.g.tiles * plan.g.slices)), dim3(512), kLdsBytes, st, plan.g, part, dbp);
  hipLaunchKernelGGL(wgrad_x6_reduce_k, dim3((unsigned)(tile_blocks + extra)), dim3(256), 0, st, plan.g, part, dbp, extra, R);
  SST_LAUNCH_CHECK();
  return SST_OK;
}

extern "C" {

int sst_weight_grad_group_f32x6(const sst_wgrad_problem_f32* problems, int n, void* d_workspace, void* stream) {
  return sst_internal_weight_grad_group_f32x6(problems, n, d_workspace, nullptr, 0, stream);
}

}  // extern "C"
