// (section 8 f4, exact-split mode) sparse 3-D convolution on the bf16 matrix pipe with the EXACT three-way operand split of
// csrc/dense_f32x6.hip: the contraction of csrc/spconv_os.hip (indiceConv and the data gradient of indiceConvBackward,
// mmdet3d/ops/spconv/include/spconv/spconv_ops.h:256-446),
//     Y[r, :] = sum_k X[map[k][r], :] W[k],
// x = x0 + x1 + x2 and w = w0 + w1 + w2 (bf16 parts, 8 + 8 + 8 significand bits: exact), the six products x_i w_j with
// i + j <= 2 accumulated in fp32 by v_mfma_f32_16x16x32_bf16: what is dropped is below 2^-24 of a product.  Two accumulator
// sets per column tile - the leading product alone, the five corrections together - summed once at the end (the long chains
// of this contraction, 27 offsets x C_in / 32 steps, are where a single set drifts: csrc/wgrad_x6.hip).  Everything stays fp32
// in memory.  Why: with all workgroups resident the fp32 kernel is bound by the fp32 matrix pipe (64 x v_mfma_f32_16x16x4_f32 =
// 2 048 pipe cycles per wave and 64-channel stage); the same stage is 48 bf16 instructions = 768 cycles here.
// Admissibility (same arithmetic class as the fp32 kernel): tests/test_gpu_spconv.py::test_exact_split_convolution_kernel -
// error against the float64 oracle <= 2 x the fp32 kernel's.  `sst_amd.spconv.set_conv_precision('f32x6')`.
//
// Structure = csrc/spconv_os_x3.hip (64-row workgroups, a wave owns 16 rows x all columns of the group, W[k] packed once per
// call into fragment order - here THREE bf16 images, 6 bytes per element - and staged through LDS double-buffered, partner rows
// gathered straight into MFMA operands and split in registers, heaviest-first launch order, XCD round-robin).
#include <stdlib.h>

#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int kX6MaxK = 32;    // kernel offsets the index image holds
constexpr int kX6Chunk = 64;   // input channels per stage
constexpr int kX6XcdChunk = 4;

__device__ __forceinline__ unsigned x6_pack2(float lo, float hi) {  // one v_cvt_pk_bf16_f32 (round to nearest even)
  const bf16x2 v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float x6_lo_f(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float x6_hi_f(unsigned p) { return __uint_as_float(p & 0xffff0000u); }
__device__ __forceinline__ f32x4 x6_mma(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// 8 consecutive fp32 values -> their three bf16 parts (p0 = bf16(x), p1 = bf16(x - p0), p2 = x - p0 - p1: exact)
__device__ __forceinline__ void x6_split2(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) {
  p0 = x6_pack2(a, b);
  const float ra = a - x6_lo_f(p0), rb = b - x6_hi_f(p0);
  p1 = x6_pack2(ra, rb);
  p2 = x6_pack2(ra - x6_lo_f(p1), rb - x6_hi_f(p1));
}
__device__ __forceinline__ void x6_split8(const f32x4& a, const f32x4& b, u32x4& p0, u32x4& p1, u32x4& p2) {
  unsigned q0[4], q1[4], q2[4];
  x6_split2(a[0], a[1], q0[0], q1[0], q2[0]);
  x6_split2(a[2], a[3], q0[1], q1[1], q2[1]);
  x6_split2(b[0], b[1], q0[2], q1[2], q2[2]);
  x6_split2(b[2], b[3], q0[3], q1[3], q2[3]);
  p0 = (u32x4){q0[0], q0[1], q0[2], q0[3]};
  p1 = (u32x4){q1[0], q1[1], q1[2], q1[3]};
  p2 = (u32x4){q2[0], q2[1], q2[2], q2[3]};
}

// Packed weights: slab (k, column group cg, channel chunk cc) = [2 ks][NCT ct][3 parts][64 lanes][4 words of 2 bf16], with
//   the 8 values of lane (l15, g) = W[k][c = 64 cc + 32 ks + 8 g + 0..7][n = 16 NCT cg + 16 ct + l15]   (0 outside cin x cout):
// the A operand (16 columns x 32 channels) of v_mfma_f32_16x16x32_bf16, one ds_read_b128 per lane and image.
// trans_w: W[k] is stored [n][c] (the data gradient reads the forward weights with the roles swapped).
__global__ __launch_bounds__(256) void sp_x6_pack_w_k(const float* __restrict__ w, int kvol, int cin, int cout, int trans_w,
                                                      int nct, int n_cg, int n_cc, unsigned* __restrict__ wp) {
  const int64_t total = (int64_t)kvol * n_cg * n_cc * 2 * nct * 64;   // one thread = one lane's 8 values (three parts)
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int lane = (int)(e & 63);
    int64_t rest = e >> 6;
    const int ct = (int)(rest % nct);
    rest /= nct;
    const int ks = (int)(rest & 1);
    int64_t slab = rest >> 1;
    const int cc = (int)(slab % n_cc);
    slab /= n_cc;
    const int cg = (int)(slab % n_cg);
    const int k = (int)(slab / n_cg);
    const int n = cg * 16 * nct + 16 * ct + (lane & 15);
    const int c0 = cc * kX6Chunk + 32 * ks + 8 * (lane >> 4);
    float v[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int c = c0 + t;
      v[t] = 0.f;
      if (c < cin && n < cout) v[t] = trans_w ? w[((int64_t)k * cout + n) * cin + c] : w[((int64_t)k * cin + c) * cout + n];
    }
    u32x4 p0, p1, p2;
    x6_split8((f32x4){v[0], v[1], v[2], v[3]}, (f32x4){v[4], v[5], v[6], v[7]}, p0, p1, p2);
    // slab base in words: ((k, cg, cc) * 2 * nct * 3 * 64 * 4); inside: [ks][ct][part][lane][4]
    unsigned* dst = wp + ((((int64_t)(k * n_cg + cg) * n_cc + cc) * 2 + ks) * nct + ct) * 3 * 256;
    *(u32x4*)(dst + lane * 4) = p0;
    *(u32x4*)(dst + 256 + lane * 4) = p1;
    *(u32x4*)(dst + 512 + lane * 4) = p2;
  }
}

template <int NCT>
__global__ __launch_bounds__(256, NCT == 4 ? 2 : 1) void sp_conv_os_x6_k(
    const float* __restrict__ x, int64_t ldx, const int32_t* __restrict__ map, int64_t m, int kvol,
    const unsigned* __restrict__ wp, int cin, int cout, const float* __restrict__ bias, float* __restrict__ y, int64_t ldy,
    int n_units, int n_cg, int n_cc, int xcd_chunk, int vec_store, const int32_t* __restrict__ tile_order, int n_split,
    float* __restrict__ partial) {
  constexpr int ROWS = 64;
  constexpr int SLAB = 64 * 4 * 2 * NCT * 3;   // 32-bit words of one packed W slab (three part images)
  constexpr int WREG = SLAB / 4 / 256;   // 16-byte pieces of a slab per thread
  extern __shared__ __attribute__((aligned(16))) unsigned x6_smem[];
  unsigned (*wbuf)[SLAB] = (unsigned (*)[SLAB])x6_smem;                    // [2][SLAB]
  int (*idx)[ROWS] = (int (*)[ROWS])(x6_smem + 2 * SLAB);                  // [kvol][ROWS]
  unsigned* live_w = (unsigned*)(x6_smem + 2 * SLAB + kvol * ROWS);         // [4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, kq = lane >> 4;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int unit_s = ((slot / xcd_chunk) * 8 + xcd) * xcd_chunk + slot % xcd_chunk;
  if (unit_s >= n_units * n_split) return;  // uniform
  // THIN levels (a few thousand rows: 30-200 (tile, column group) units for 256 CUs, each walking 27 offsets x the channel
  // chunks alone): the offsets are dealt out over n_split workgroups per unit (offset k belongs to workgroup k % n_split), which
  // write partial tiles that sp_x6_split_reduce_k adds in a fixed order.  n_split = 1: the plain kernel.
  const int unit = unit_s / n_split, split = unit_s - unit * n_split;
  const int pos = unit / n_cg, cg = unit - pos * n_cg;
  const int tile = tile_order ? tile_order[pos] : pos;
  const int64_t r0 = (int64_t)tile * ROWS;
  // ---- partner rows of the tile for every offset -> LDS; which offsets are populated, per wave ----
  if (tid < 4) live_w[tid] = 0u;
  __syncthreads();
  {
    constexpr int NIT = (kX6MaxK * ROWS + 255) / 256;
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
      const int k = e / ROWS, row = e - k * ROWS;   // a wave holds the 64 rows of ONE offset
      const int v = vals[it];
      if (k < kvol) idx[k][row] = v;
      const unsigned long long b = __ballot(v >= 0);
      if (lane < 4 && k < kvol && ((b >> (16 * (lane & 3))) & 0xffffull)) atomicOr(&live_w[lane & 3], 1u << k);
    }
  }
  __syncthreads();
  unsigned mine = 0xffffffffu;
  if (n_split > 1) {
    mine = 0u;
    for (int k = split; k < 32; k += n_split) mine |= 1u << k;
  }
  const unsigned wave_live = live_w[wave] & mine;
  const unsigned live = (live_w[0] | live_w[1] | live_w[2] | live_w[3]) & mine;
  const int wave_row = wave * 16;

  f32x4 acc[NCT], acl[NCT];      // leading product | the five corrections
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) acc[ct] = acl[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (live != 0u) {
    const int c_last = cin - 4;  // cin % 4 == 0 (host check)
    f32x4 xn[4];                 // requested stage: channels 32 ks + 8 kq + 4 h + 0..3 at xn[2 ks + h]
    u32x4 x0[2], x1[2], x2[2], wr[WREG];
    int sn;
    auto fetch_w = [&](int k, int cc) {
      const u32x4* src = (const u32x4*)(wp + ((int64_t)(k * n_cg + cg) * n_cc + cc) * SLAB);
#pragma unroll
      for (int u = 0; u < WREG; ++u) wr[u] = src[tid + 256 * u];
    };
    auto gather_x = [&](int k, int cc) {
      sn = idx[k][wave_row + l15];
      const float* p = x + (int64_t)(sn >= 0 ? sn : 0) * ldx;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int c = cc * kX6Chunk + 32 * (j >> 1) + 8 * kq + 4 * (j & 1);
        c = c < c_last ? c : c_last;  // beyond cin the packed weights are zero: any finite value will do
        xn[j] = *(const f32x4*)(p + c);
      }
    };
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto commit = [&](int buf) {   // requested W -> LDS buffer, requested X -> split operands of the current stage
#pragma unroll
      for (int u = 0; u < WREG; ++u) *(u32x4*)(&wbuf[buf][4 * (tid + 256 * u)]) = wr[u];
      const bool has = sn >= 0;     // rows without a partner contribute 0
      x6_split8(has ? xn[0] : zero4, has ? xn[1] : zero4, x0[0], x1[0], x2[0]);
      x6_split8(has ? xn[2] : zero4, has ? xn[3] : zero4, x0[1], x1[1], x2[1]);
    };
    auto next_live = [&](int k) {
      const unsigned rest = k < 32 ? (live >> k) : 0u;
      return rest ? k + __builtin_ctz(rest) : 32;
    };
    int k = next_live(0), cc = 0, buf = 0;
    fetch_w(k, 0);
    gather_x(k, 0);
    commit(0);
    __syncthreads();
    while (true) {
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
      gather_x(nk, ncc);
      if ((wave_live >> k) & 1u) {  // uniform per wave
        const unsigned* wb = &wbuf[buf][4 * lane];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
          for (int ct = 0; ct < NCT; ++ct) {
            const u32x4 w0 = *(const u32x4*)(wb + ((ks * NCT + ct) * 3 + 0) * 256);
            const u32x4 w1 = *(const u32x4*)(wb + ((ks * NCT + ct) * 3 + 1) * 256);
            const u32x4 w2 = *(const u32x4*)(wb + ((ks * NCT + ct) * 3 + 2) * 256);
            // corrections, smallest first: (0,2) (2,0) (1,1) ~ 2^-16, (0,1) (1,0) ~ 2^-8; then the leading product
            acl[ct] = x6_mma(w0, x2[ks], acl[ct]);
            acl[ct] = x6_mma(w2, x0[ks], acl[ct]);
            acl[ct] = x6_mma(w1, x1[ks], acl[ct]);
            acl[ct] = x6_mma(w0, x1[ks], acl[ct]);
            acl[ct] = x6_mma(w1, x0[ks], acl[ct]);
            acc[ct] = x6_mma(w0, x0[ks], acc[ct]);
          }
        }
      }
      commit(buf ^ 1);
      __syncthreads();
      if (!has_next) break;
      k = nk;
      cc = ncc;
      buf ^= 1;
    }
  }
  // ---- epilogue: lane = (row l15 of the wave's block, 4 consecutive columns at 4 kq of every column tile) ----
  const int64_t row = r0 + wave_row + l15;
  if (row >= m) return;
  if (n_split > 1) {   // partial tile of this workgroup's offsets: [split][row][cout padded to 4], no bias
    const int64_t cpad = (int64_t)((cout + 3) & ~3);
    float* dst = partial + ((int64_t)split * m + row) * cpad;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      const int n = cg * 16 * NCT + 16 * ct + 4 * kq;
      if (n < cout) *(f32x4*)(dst + n) = acc[ct] + acl[ct];
    }
    return;
  }
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) {
    const int n = cg * 16 * NCT + 16 * ct + 4 * kq;
    if (n >= cout) continue;
    f32x4 v = acc[ct] + acl[ct];
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

// y[row][n] = bias[n] + sum over the splits, in split order (deterministic)
__global__ __launch_bounds__(256) void sp_x6_split_reduce_k(const float* __restrict__ partial, int64_t m, int cout, int n_split,
                                                            const float* __restrict__ bias, float* __restrict__ y, int64_t ldy) {
  const int64_t cpad = (int64_t)((cout + 3) & ~3), c4 = cpad >> 2;
  const int64_t total = m * c4;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = e / c4;
    const int n = (int)(e - row * c4) * 4;
    f32x4 v = *(const f32x4*)(partial + row * cpad + n);
    for (int sp = 1; sp < n_split; ++sp) v += *(const f32x4*)(partial + ((int64_t)sp * m + row) * cpad + n);
#pragma unroll
    for (int t = 0; t < 4; ++t)
      if (n + t < cout) y[row * ldy + n + t] = v[t] + (bias ? bias[n + t] : 0.f);
  }
}

// tile shape and offset split of a call
struct x6_cfg {
  int nct, n_cg, n_cc, n_split;
  int64_t n_tiles, pack_words;
};
x6_cfg x6_pick(int64_t m, int kvol, int cin, int cout) {
  x6_cfg c;
  c.n_tiles = sst_div_up(m, 64);
  c.nct = cout <= 64 ? 4 : 8;
  if (c.nct == 8 && c.n_tiles * sst_div_up(cout, 128) < 2048) c.nct = 4;   // as sp_conv_os_k's os_pick
  c.n_cg = (int)sst_div_up(cout, 16 * c.nct);
  c.n_cc = (int)sst_div_up(cin, kX6Chunk);
  c.pack_words = (int64_t)kvol * c.n_cg * c.n_cc * (64 * 4 * 2 * c.nct * 3);
  // offsets dealt out over workgroups while the launch has fewer than ~3 workgroups per CU and an offset's worth of work is
  // left to each (SST_SPCONV_X6_SPLIT overrides: 1 = never)
  static int env = -1;
  if (env < 0) {
    const char* e = getenv("SST_SPCONV_X6_SPLIT");
    env = e ? atoi(e) : 0;
  }
  const int64_t units = c.n_tiles * c.n_cg;
  int s = 1;
  while (s < 8 && units * s * 2 <= 1024 && kvol >= 2 * s * 2) s *= 2;
  if (env >= 1 && env <= 16) s = env < kvol ? env : kvol;
  c.n_split = s;
  return c;
}

}  // namespace

extern "C" {

// packed weights of one call: kvol x column groups x channel chunks slabs of 2 x nct x 3 x 64 x 4 words (worst case nct = 8)
int64_t sst_spconv_conv_os_f32x6_workspace_bytes(int kvol, int cin, int cout) {
  if (kvol < 1 || cin < 1 || cout < 1) return SST_ERR_ARG;
  const int64_t slabs4 = (int64_t)kvol * sst_div_up(cout, 64) * sst_div_up(cin, kX6Chunk) * (64 * 4 * 2 * 4 * 3);
  const int64_t slabs8 = (int64_t)kvol * sst_div_up(cout, 128) * sst_div_up(cin, kX6Chunk) * (64 * 4 * 2 * 8 * 3);
  return (slabs4 > slabs8 ? slabs4 : slabs8) * 4 + 256;
}
// ... plus the partial tiles of a call on m rows when its offsets are dealt out over several workgroups (thin levels):
// what sst_spconv_conv_os_rows_f32x6 wants for its own choice of the split
int64_t sst_spconv_conv_os_f32x6_workspace_bytes_rows(int kvol, int cin, int cout, int64_t m) {
  const int64_t pack = sst_spconv_conv_os_f32x6_workspace_bytes(kvol, cin, cout);
  if (pack < 0 || m < 0) return SST_ERR_ARG;
  const x6_cfg c = x6_pick(m > 0 ? m : 1, kvol, cin, cout);
  const int64_t part = c.n_split > 1 ? (int64_t)c.n_split * m * ((cout + 3) & ~3) * 4 : 0;
  return sst_align_up(pack, 256) + part + 256;
}

// same arguments as sst_spconv_conv_os_f32 + the size of the workspace: sst_spconv_conv_os_f32x6_workspace_bytes_rows bytes let
// the call deal the offsets of a thin level out over several workgroups; with less (sst_spconv_conv_os_f32x6_workspace_bytes:
// the packed weights alone) it runs unsplit.  tile_cfg must be 0.
int sst_spconv_conv_os_rows_f32x6(const float* d_x, int64_t ldx, const int32_t* d_map, int64_t m, int kvol, const float* d_w,
                                  int cin, int cout, int trans_w, const float* d_bias, float* d_y, int64_t ldy, int tile_cfg,
                                  const int32_t* d_tile_order, void* d_workspace, int64_t workspace_bytes, void* stream) {
  if (m < 0 || kvol < 1 || cin < 1 || cout < 1 || ldx < cin || ldy < cout || tile_cfg != 0) return SST_ERR_ARG;
  if (m == 0) return SST_OK;
  if (!d_x || !d_map || !d_w || !d_y || !d_workspace) return SST_ERR_ARG;
  if (kvol > kX6MaxK || (cin & 3) || (ldx & 3) || (((uintptr_t)d_x) & 15) || (((uintptr_t)d_workspace) & 255) ||
      m > 0x3fffffff)
    return SST_ERR_UNSUPPORTED;
  x6_cfg c = x6_pick(m, kvol, cin, cout);
  const int64_t pack_bytes = sst_align_up(sst_spconv_conv_os_f32x6_workspace_bytes(kvol, cin, cout), 256);
  const int64_t cpad = (cout + 3) & ~3;
  while (c.n_split > 1 && pack_bytes + (int64_t)c.n_split * m * cpad * 4 > workspace_bytes) c.n_split >>= 1;
  if (workspace_bytes < c.pack_words * 4) return SST_ERR_ARG;
  const int nct = c.nct, n_cg = c.n_cg, n_cc = c.n_cc;
  const int64_t n_units = c.n_tiles * n_cg;
  if (n_units * c.n_split > 0x3fffffff) return SST_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  unsigned* wp = (unsigned*)d_workspace;
  float* partial = c.n_split > 1 ? (float*)((char*)d_workspace + pack_bytes) : nullptr;
  const int64_t lanes = (int64_t)kvol * n_cg * n_cc * 2 * nct * 64;
  hipLaunchKernelGGL(sp_x6_pack_w_k, dim3(sst_grid_1d(lanes, 256)), dim3(256), 0, st, d_w, kvol, cin, cout, trans_w, nct, n_cg,
                     n_cc, wp);
  const int64_t wgs = n_units * c.n_split;
  const int chunk = wgs >= 64 * (int64_t)kX6XcdChunk ? kX6XcdChunk : 1;
  const dim3 grid((unsigned)(sst_div_up(wgs, 8 * chunk) * 8 * chunk));
  const int vec_store = ((ldy & 3) == 0 && (((uintptr_t)d_y) & 15) == 0 && (!d_bias || (((uintptr_t)d_bias) & 15) == 0)) ? 1 : 0;
  const int lds = (2 * 64 * 4 * 2 * nct * 3 + kvol * 64 + 4) * (int)sizeof(unsigned);
  if (nct == 4) {
    static unsigned long long attr4 = 0;
    if (sst_first_use_on_device(&attr4)) {
      SST_HIP(hipFuncSetAttribute((const void*)sp_conv_os_x6_k<4>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (2 * 64 * 4 * 2 * 4 * 3 + kX6MaxK * 64 + 4) * (int)sizeof(unsigned)));
      sst_mark_device(&attr4);
    }
    hipLaunchKernelGGL(sp_conv_os_x6_k<4>, grid, dim3(256), lds, st, d_x, ldx, d_map, m, kvol, wp, cin, cout, d_bias, d_y, ldy,
                       (int)n_units, n_cg, n_cc, chunk, vec_store, d_tile_order, c.n_split, partial);
  } else {
    static unsigned long long attr8 = 0;
    if (sst_first_use_on_device(&attr8)) {
      SST_HIP(hipFuncSetAttribute((const void*)sp_conv_os_x6_k<8>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (2 * 64 * 4 * 2 * 8 * 3 + kX6MaxK * 64 + 4) * (int)sizeof(unsigned)));
      sst_mark_device(&attr8);
    }
    hipLaunchKernelGGL(sp_conv_os_x6_k<8>, grid, dim3(256), lds, st, d_x, ldx, d_map, m, kvol, wp, cin, cout, d_bias, d_y, ldy,
                       (int)n_units, n_cg, n_cc, chunk, vec_store, d_tile_order, c.n_split, partial);
  }
  if (c.n_split > 1)
    hipLaunchKernelGGL(sp_x6_split_reduce_k, dim3(sst_grid_1d(m * (cpad >> 2), 256)), dim3(256), 0, st, partial, m, cout, c.n_split,
                       d_bias, d_y, ldy);
  SST_LAUNCH_CHECK();
  return SST_OK;
}

// the unsplit call (workspace: sst_spconv_conv_os_f32x6_workspace_bytes)
int sst_spconv_conv_os_f32x6(const float* d_x, int64_t ldx, const int32_t* d_map, int64_t m, int kvol, const float* d_w,
                             int cin, int cout, int trans_w, const float* d_bias, float* d_y, int64_t ldy, int tile_cfg,
                             const int32_t* d_tile_order, void* d_workspace, void* stream) {
  return sst_spconv_conv_os_rows_f32x6(d_x, ldx, d_map, m, kvol, d_w, cin, cout, trans_w, d_bias, d_y, ldy, tile_cfg, d_tile_order,
                                       d_workspace, kvol > 0 && cin > 0 && cout > 0
                                                        ? sst_align_up(sst_spconv_conv_os_f32x6_workspace_bytes(kvol, cin, cout), 256)
                                                        : 0,
                                       stream);
}

}  // extern "C"
