// (§8 f4, precision mode) sparse 3-D convolution in SPLIT precision: the contraction of csrc/spconv_os.hip
// (indiceConv and the data gradient of indiceConvBackward, mmdet3d/ops/spconv/include/spconv/spconv_ops.h:256-446),
//     Y[r, :] = sum_k X[map[k][r], :] W[k],
// with every fp32 product formed as three bf16 products of split operands,
//     x w ~= x_hi w_hi + x_lo w_hi + x_hi w_lo      (x = x_hi + x_lo, x_hi = bf16(x), x_lo = bf16(x - x_hi); fp32 accumulation)
// on v_mfma_f32_16x16x32_bf16 - the scheme of csrc/dense_f32x3.hip for the dense layers.  Everything stays fp32 in memory.
// Why: with all workgroups resident the fp32 kernel is bound by the fp32 matrix pipe (64 x v_mfma_f32_16x16x4_f32 = 2 048
// pipe cycles per wave and 64-channel stage); the same stage is 24 bf16 instructions = 384 cycles here.  What remains is
// the skeleton of the stage (gathers, LDS staging, barrier).  Error ~1e-5 of the output scale per layer (tighter than the
// TF32 products the reference's torch 1.8 used on Ampere, `allow_tf32` default); the exact-fp32 kernel stays the default and
// the parity mode, this one is `sst_amd.spconv.set_conv_precision('f32x3')`.
//
// Same structure as sp_conv_os_k<NCT, 1>: 64-row workgroups (a wave owns 16 rows x all columns of the group), W[k] packed
// once per call - here as bf16 hi / lo fragment images, same 4 bytes per element - and staged through LDS double-buffered,
// partner rows gathered straight into MFMA operands (a lane's two 16-byte loads = the 8 consecutive channels the
// 16x16x32 B operand wants), heaviest-first launch order, runs of 4 units round-robin over the XCDs.
#include <stdlib.h>

#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int kX3MaxK = 32;    // kernel offsets the index image holds
constexpr int kX3Chunk = 64;   // input channels per stage
constexpr int kX3XcdChunk = 4;

__device__ __forceinline__ unsigned x3_pack2(float lo, float hi) {  // one v_cvt_pk_bf16_f32 (round to nearest even)
  const bf16x2 v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float x3_lo_f(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float x3_hi_f(unsigned p) { return __uint_as_float(p & 0xffff0000u); }
__device__ __forceinline__ f32x4 x3_mma(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// 8 consecutive fp32 values -> their bf16 heads and the bf16 of what the heads leave
__device__ __forceinline__ void x3_split8(const f32x4& a, const f32x4& b, u32x4& hi, u32x4& lo) {
  hi[0] = x3_pack2(a[0], a[1]);
  hi[1] = x3_pack2(a[2], a[3]);
  hi[2] = x3_pack2(b[0], b[1]);
  hi[3] = x3_pack2(b[2], b[3]);
  lo[0] = x3_pack2(a[0] - x3_lo_f(hi[0]), a[1] - x3_hi_f(hi[0]));
  lo[1] = x3_pack2(a[2] - x3_lo_f(hi[1]), a[3] - x3_hi_f(hi[1]));
  lo[2] = x3_pack2(b[0] - x3_lo_f(hi[2]), b[1] - x3_hi_f(hi[2]));
  lo[3] = x3_pack2(b[2] - x3_lo_f(hi[3]), b[3] - x3_hi_f(hi[3]));
}

// Packed weights: slab (k, column group cg, channel chunk cc) = [2 ks][NCT ct][2 hi|lo][64 lanes][4 words of 2 bf16], with
//   the 8 values of lane (l15, g) = W[k][c = 64 cc + 32 ks + 8 g + 0..7][n = 16 NCT cg + 16 ct + l15]   (0 outside cin x cout):
// the A operand (16 columns x 32 channels) of v_mfma_f32_16x16x32_bf16, one ds_read_b128 per lane and image.
// trans_w: W[k] is stored [n][c] (the data gradient reads the forward weights with the roles swapped).
__global__ __launch_bounds__(256) void sp_x3_pack_w_k(const float* __restrict__ w, int kvol, int cin, int cout, int trans_w,
                                                      int nct, int n_cg, int n_cc, unsigned* __restrict__ wp) {
  const int64_t total = (int64_t)kvol * n_cg * n_cc * 2 * nct * 64;   // one thread = one lane's 8 values (hi and lo)
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
    const int c0 = cc * kX3Chunk + 32 * ks + 8 * (lane >> 4);
    float v[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int c = c0 + t;
      v[t] = 0.f;
      if (c < cin && n < cout) v[t] = trans_w ? w[((int64_t)k * cout + n) * cin + c] : w[((int64_t)k * cin + c) * cout + n];
    }
    u32x4 hi, lo;
    x3_split8((f32x4){v[0], v[1], v[2], v[3]}, (f32x4){v[4], v[5], v[6], v[7]}, hi, lo);
    // slab base in words: ((k, cg, cc) * 2 * nct * 2 * 64 * 4); inside: [ks][ct][hi|lo][lane][4]
    unsigned* dst = wp + ((((int64_t)(k * n_cg + cg) * n_cc + cc) * 2 + ks) * nct + ct) * 2 * 256;
    *(u32x4*)(dst + lane * 4) = hi;
    *(u32x4*)(dst + 256 + lane * 4) = lo;
  }
}

template <int NCT>
__global__ __launch_bounds__(256, NCT == 4 ? 4 : 2) void sp_conv_os_x3_k(
    const float* __restrict__ x, int64_t ldx, const int32_t* __restrict__ map, int64_t m, int kvol,
    const unsigned* __restrict__ wp, int cin, int cout, const float* __restrict__ bias, float* __restrict__ y, int64_t ldy,
    int n_units, int n_cg, int n_cc, int xcd_chunk, int vec_store, const int32_t* __restrict__ tile_order) {
  constexpr int ROWS = 64;
  constexpr int SLAB = 64 * 16 * NCT;    // 32-bit words of one packed W slab (hi + lo images)
  constexpr int WREG = SLAB / 4 / 256;   // 16-byte pieces of a slab per thread
  extern __shared__ __attribute__((aligned(16))) unsigned x3_smem[];
  unsigned (*wbuf)[SLAB] = (unsigned (*)[SLAB])x3_smem;                    // [2][SLAB]
  int (*idx)[ROWS] = (int (*)[ROWS])(x3_smem + 2 * SLAB);                  // [kvol][ROWS]
  unsigned* live_w = (unsigned*)(x3_smem + 2 * SLAB + kvol * ROWS);         // [4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, kq = lane >> 4;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int unit = ((slot / xcd_chunk) * 8 + xcd) * xcd_chunk + slot % xcd_chunk;
  if (unit >= n_units) return;  // uniform
  const int pos = unit / n_cg, cg = unit - pos * n_cg;
  const int tile = tile_order ? tile_order[pos] : pos;
  const int64_t r0 = (int64_t)tile * ROWS;
  // ---- partner rows of the tile for every offset -> LDS; which offsets are populated, per wave ----
  if (tid < 4) live_w[tid] = 0u;
  __syncthreads();
  {
    constexpr int NIT = (kX3MaxK * ROWS + 255) / 256;
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
  const unsigned wave_live = live_w[wave];
  const unsigned live = live_w[0] | live_w[1] | live_w[2] | live_w[3];
  const int wave_row = wave * 16;

  f32x4 acc[NCT];
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (live != 0u) {
    const int c_last = cin - 4;  // cin % 4 == 0 (host check)
    f32x4 xn[4];                 // requested stage: channels 32 ks + 8 kq + 4 h + 0..3 at xn[2 ks + h]
    u32x4 xh[2], xl[2], wr[WREG];
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
        int c = cc * kX3Chunk + 32 * (j >> 1) + 8 * kq + 4 * (j & 1);
        c = c < c_last ? c : c_last;  // beyond cin the packed weights are zero: any finite value will do
        xn[j] = *(const f32x4*)(p + c);
      }
    };
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto commit = [&](int buf) {   // requested W -> LDS buffer, requested X -> split operands of the current stage
#pragma unroll
      for (int u = 0; u < WREG; ++u) *(u32x4*)(&wbuf[buf][4 * (tid + 256 * u)]) = wr[u];
      const bool has = sn >= 0;     // rows without a partner contribute 0
      x3_split8(has ? xn[0] : zero4, has ? xn[1] : zero4, xh[0], xl[0]);
      x3_split8(has ? xn[2] : zero4, has ? xn[3] : zero4, xh[1], xl[1]);
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
            const u32x4 wh = *(const u32x4*)(wb + ((ks * NCT + ct) * 2 + 0) * 256);
            const u32x4 wl = *(const u32x4*)(wb + ((ks * NCT + ct) * 2 + 1) * 256);
            acc[ct] = x3_mma(wh, xh[ks], acc[ct]);
            acc[ct] = x3_mma(wh, xl[ks], acc[ct]);
            acc[ct] = x3_mma(wl, xh[ks], acc[ct]);
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
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) {
    const int n = cg * 16 * NCT + 16 * ct + 4 * kq;
    if (n >= cout) continue;
    f32x4 v = acc[ct];
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

}  // namespace

extern "C" {

// same arguments and workspace size (sst_spconv_conv_os_workspace_bytes) as sst_spconv_conv_os_f32; tile_cfg must be 0
int sst_spconv_conv_os_f32x3(const float* d_x, int64_t ldx, const int32_t* d_map, int64_t m, int kvol, const float* d_w,
                             int cin, int cout, int trans_w, const float* d_bias, float* d_y, int64_t ldy, int tile_cfg,
                             const int32_t* d_tile_order, void* d_workspace, void* stream) {
  if (m < 0 || kvol < 1 || cin < 1 || cout < 1 || ldx < cin || ldy < cout || tile_cfg != 0) return SST_ERR_ARG;
  if (m == 0) return SST_OK;
  if (!d_x || !d_map || !d_w || !d_y || !d_workspace) return SST_ERR_ARG;
  if (kvol > kX3MaxK || (cin & 3) || (ldx & 3) || (((uintptr_t)d_x) & 15) || (((uintptr_t)d_workspace) & 15) ||
      m > 0x3fffffff)
    return SST_ERR_UNSUPPORTED;
  const int64_t n_tiles = sst_div_up(m, 64);
  int nct = cout <= 64 ? 4 : 8;
  if (nct == 8 && n_tiles * sst_div_up(cout, 128) < 2048) nct = 4;   // as sp_conv_os_k's os_pick
  const int n_cg = (int)sst_div_up(cout, 16 * nct), n_cc = (int)sst_div_up(cin, kX3Chunk);
  const int64_t n_units = n_tiles * n_cg;
  if (n_units > 0x3fffffff) return SST_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  unsigned* wp = (unsigned*)d_workspace;
  const int64_t lanes = (int64_t)kvol * n_cg * n_cc * 2 * nct * 64;
  hipLaunchKernelGGL(sp_x3_pack_w_k, dim3(sst_grid_1d(lanes, 256)), dim3(256), 0, st, d_w, kvol, cin, cout, trans_w, nct, n_cg,
                     n_cc, wp);
  const int chunk = n_units >= 64 * (int64_t)kX3XcdChunk ? kX3XcdChunk : 1;
  const dim3 grid((unsigned)(sst_div_up(n_units, 8 * chunk) * 8 * chunk));
  const int vec_store = ((ldy & 3) == 0 && (((uintptr_t)d_y) & 15) == 0 && (!d_bias || (((uintptr_t)d_bias) & 15) == 0)) ? 1 : 0;
  const int lds = (2 * 64 * 16 * nct + kvol * 64 + 4) * (int)sizeof(unsigned);
  if (nct == 4) {
    static unsigned long long attr4 = 0;
    if (sst_first_use_on_device(&attr4)) {
      SST_HIP(hipFuncSetAttribute((const void*)sp_conv_os_x3_k<4>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (2 * 64 * 16 * 4 + kX3MaxK * 64 + 4) * (int)sizeof(unsigned)));
      sst_mark_device(&attr4);
    }
    hipLaunchKernelGGL(sp_conv_os_x3_k<4>, grid, dim3(256), lds, st, d_x, ldx, d_map, m, kvol, wp, cin, cout, d_bias, d_y, ldy,
                       (int)n_units, n_cg, n_cc, chunk, vec_store, d_tile_order);
  } else {
    static unsigned long long attr8 = 0;
    if (sst_first_use_on_device(&attr8)) {
      SST_HIP(hipFuncSetAttribute((const void*)sp_conv_os_x3_k<8>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (2 * 64 * 16 * 8 + kX3MaxK * 64 + 4) * (int)sizeof(unsigned)));
      sst_mark_device(&attr8);
    }
    hipLaunchKernelGGL(sp_conv_os_x3_k<8>, grid, dim3(256), lds, st, d_x, ldx, d_map, m, kvol, wp, cin, cout, d_bias, d_y, ldy,
                       (int)n_units, n_cg, n_cc, chunk, vec_store, d_tile_order);
  }
  SST_LAUNCH_CHECK();
  return SST_OK;
}

}  // extern "C"
