// The part of a post-norm SRA encoder layer behind the attention core (models/sst/sst_basic_block_v2.py:113-118) as ONE kernel
// per direction in the REDUCED-PRECISION mode (bf16 storage, fp32 accumulation / LayerNorm statistics; what the reference's fp16
// training of these layers corresponds to on this hardware) - the bf16 twin of csrc/layer_tail_x6.hip:
//   forward : o, x -> [out-proj + b + x] = s1 -> LN1 -> y1 -> [W1 + b1] = pre -> act -> h -> [W2 + b2 + y1] = s2 -> LN2 -> y2 (, y2 + pos)
//   backward: dy2 (, dy2p), s2 -> LN2' -> ds2 -> [W2^T] * act'(pre) = dpre -> [W1^T] + ds2 = d(y1) -> LN1'(s1) -> ds1 -> [W_o^T] = d_o
// A wave carries 16 tokens through the chain in registers: the accumulator tile of v_mfma_f32_16x16x32_bf16 with the weight rows
// permuted (w_lds_row) leaves lane (c, g) with 8 consecutive columns of token c - packed to bf16 that is the B operand of the
// next product's k-step AND the 16 bytes the tensor stores for that lane.  Every value is rounded to bf16 exactly where the
// launch-per-product sequence of csrc/layer_exec.hip rounded it (what is stored is what the next product reads); d(y1) stays
// fp32 between the second product and norm1's backward (the sequence stored it as bf16).
// Weights: bf16 images (fragment-contiguous: the 16 rows x 32 k of an A operand are one 1 KiB block in lane order) of five
// phases - the out-projection and four feed-forward chunks of 64 hidden columns (first-product part 16 KiB | second-product
// part 16 KiB) - packed once per layer call from the fp32 masters, fetched with LDS-DMA into two 32 KiB slots, the next phase's
// images landing while the running one is multiplied.  Workgroup = 8 waves = 128 tokens, five barriers.
// Global loads / stores go through a wave-private LDS block so that 4 lanes touch 64 contiguous bytes of one row.
#include <math.h>
#include <stdlib.h>
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short bf16_t;

__device__ __forceinline__ unsigned pack2(float lo, float hi) {  // one v_cvt_pk_bf16_f32 (RNE)
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float lo_f(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float hi_f(unsigned p) { return __uint_as_float(p & 0xffff0000u); }
__device__ __forceinline__ f32x4 mma32(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ u32x4 pack8(const f32x4& a, const f32x4& b) {
  return (u32x4){pack2(a[0], a[1]), pack2(a[2], a[3]), pack2(b[0], b[1]), pack2(b[2], b[3])};
}
__device__ __forceinline__ void unpack8(const u32x4& p, f32x4& a, f32x4& b) {
  a = (f32x4){lo_f(p[0]), hi_f(p[0]), lo_f(p[1]), hi_f(p[1])};
  b = (f32x4){lo_f(p[2]), hi_f(p[2]), lo_f(p[3]), hi_f(p[3])};
}
__device__ __forceinline__ float erf_as(float z, float& e) {  // Abramowitz & Stegun 7.1.26 (as csrc/dense_bf16.hip)
  const float az = fabsf(z);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, az, 1.f));
  e = __expf(-az * az);
  const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
  return copysignf(fmaf(-poly, e, 1.f), z);
}
__device__ __forceinline__ float gelu_f(float x) {
  float e;
  return 0.5f * x * (1.f + erf_as(x * 0.70710678118654752f, e));
}
__device__ __forceinline__ float gelu_grad_f(float x) {
  float e;
  const float phi = 0.5f * (1.f + erf_as(x * 0.70710678118654752f, e));
  return fmaf(x * 0.3989422804014327f, e, phi);
}
__device__ __forceinline__ int w_lds_row(int n) {  // see csrc/dense_bf16.hip: a lane ends with 8 consecutive columns of a row
  const int tp = n >> 5, within = n & 31;
  return 16 * (2 * tp + ((within >> 2) & 1)) + ((within >> 3) << 2) + (within & 3);
}
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row_sum16(float v) {   // sum over the 16 lanes of a DPP row, every lane gets the total
  v = dpp_add<0xB1>(v);
  v = dpp_add<0x4E>(v);
  v = dpp_add<0x141>(v);
  return dpp_add<0x140>(v);
}
__device__ __forceinline__ void barrier_drain() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int kC = 128, kFF = 256, kHC = 64, kNF = kFF / kHC;   // 4 feed-forward chunks of 64 hidden columns
constexpr int kFrag = 1024;
constexpr int kP1 = 4 * 4 * kFrag;      // first product of a chunk: 4 row tiles (64 hidden columns) x 4 k-steps (K = 128)
constexpr int kP2 = 8 * 2 * kFrag;      // second product: 8 row tiles (128 outputs) x 2 k-steps (the chunk's 64 hidden columns)
constexpr int kSlot = kP1 + kP2;        // 32 KiB = the out-projection's 8 row tiles x 4 k-steps as well
static_assert(kSlot == 8 * 4 * kFrag, "slot");
constexpr int kPhases = kNF + 1;
constexpr int kPackDir = kPhases * kSlot;
constexpr int kWaves = 8, kNTH = 64 * kWaves, kRowsPerWg = 16 * kWaves;
constexpr int kScr = 1024;              // per wave: a 16-token x 32-column bf16 block between the fragment and the row layout
constexpr int kParF = 6 * kC + kFF;
constexpr int kLdsFwd = 2 * kSlot + kParF * 4 + kWaves * kScr;
constexpr int kLdsBwd = 2 * kSlot + kWaves * 2 * kC * 4 + kWaves * kScr;   // 80 KiB: two workgroups per CU (gamma comes from L2)

struct tail_weights {
  const float* wo;   // [128][128]
  const float* w1;   // [256][128]
  const float* w2;   // [128][256]
};

// byte offset inside an image of the 16 bytes (row lr of the image, k8 .. k8 + 7), the image having `ks` k-steps per row tile
__device__ __forceinline__ int frag_off(int lr, int k8, int ks) {
  return ((lr >> 4) * ks + (k8 >> 5)) * kFrag + (16 * ((k8 >> 3) & 3) + (lr & 15)) * 16;
}

// packed[dir][phase][kSlot]: dir 0 forward (phase 0 = out-projection, 1 .. 4 = feed-forward chunks), dir 1 backward (0 .. 3 =
// feed-forward chunks of the TRANSPOSED products, 4 = out-projection^T).  One workgroup per phase image, an item = 8 k of one row.
constexpr int kPackMany = 16;
struct pack_batch {          // the weights of up to 16 layers (a whole encoder stack in one launch)
  tail_weights w[kPackMany];
  unsigned char* dst[kPackMany];
};
__global__ __launch_bounds__(512) void encoder_tail_pack_bf16_k(const pack_batch B) {
  const int layer = blockIdx.x / (2 * kPhases), blk = blockIdx.x % (2 * kPhases);
  const tail_weights W = B.w[layer];
  const int phase = blk % kPhases, dir = blk / kPhases;
  unsigned char* img = B.dst[layer] + (size_t)blk * kSlot;
  const bool outproj = dir == 0 ? phase == 0 : phase == kNF;
  const int j = dir == 0 ? phase - 1 : phase;
  for (int it = threadIdx.x; it < kSlot / 16; it += 512) {
    float v[8];
    int off;
    if (outproj) {   // 128 rows x 16 items
      const int r = it >> 4, k8 = (it & 15) * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = dir == 0 ? W.wo[r * kC + k8 + e] : W.wo[(k8 + e) * kC + r];
      off = frag_off(w_lds_row(r), k8, 4);
    } else if (it < kP1 / 16) {   // first product: 64 rows (hidden 64 j + r) x 16 items (K = 128)
      const int r = it >> 4, k8 = (it & 15) * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = dir == 0 ? W.w1[(kHC * j + r) * kC + k8 + e] : W.w2[(k8 + e) * kFF + kHC * j + r];
      off = frag_off(w_lds_row(r), k8, 4);
    } else {                      // second product: 128 rows x 8 items (K = the chunk's 64 hidden columns)
      const int i2 = it - kP1 / 16, r = i2 >> 3, k8 = (i2 & 7) * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = dir == 0 ? W.w2[r * kFF + kHC * j + k8 + e] : W.w1[(kHC * j + k8 + e) * kC + r];
      off = kP1 + frag_off(w_lds_row(r), k8, 2);
    }
    *(u32x4*)(img + off) = (u32x4){pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
  }
}

__device__ __forceinline__ void dma_pieces(const unsigned char* __restrict__ src, unsigned char* dst, int pieces, int wave, int lane) {
  for (int i = wave; i < pieces; i += kWaves)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i * 1024 + lane * 16),
                                     (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
}

// ---- global memory <-> fragment layout through a wave-private 1 KiB LDS block (see csrc/layer_tail_x6.hip) -----------------
// fragment side: lane (c, g) holds columns 8 g .. 8 g + 7 of token c (16 bytes); row side: lane l holds row l / 4, 16-byte slot
// l % 4 - four lanes = 64 contiguous bytes of a row.  Slots are XOR-swizzled with the row: both sides are bank-conflict-free.
struct tile_io {
  unsigned char* scr;
  int frag_off, row_off;
  int64_t rrow;
  bool rvalid;
  int rcol;
};
__device__ __forceinline__ tile_io make_tile_io(unsigned char* scr, int lane, int64_t r0, int64_t m) {
  tile_io t;
  const int c = lane & 15, g = lane >> 4, r = lane >> 2, p = lane & 3;
  t.scr = scr;
  t.frag_off = c * 64 + ((g ^ ((c >> 2) & 3)) << 4);
  t.row_off = r * 64 + ((p ^ ((r >> 2) & 3)) << 4);
  t.rvalid = r0 + r < m;
  t.rrow = t.rvalid ? r0 + r : m - 1;
  t.rcol = p * 8;
  return t;
}
template <bool STREAM>
__device__ __forceinline__ void store32(const tile_io& t, bf16_t* __restrict__ base, int64_t ld, int col0, const u32x4& v) {
  *(u32x4*)(t.scr + t.frag_off) = v;
  const u32x4 w = *(const u32x4*)(t.scr + t.row_off);
  if (t.rvalid) {
    if (STREAM)
      __builtin_nontemporal_store(w, (u32x4*)(base + t.rrow * ld + col0 + t.rcol));
    else
      *(u32x4*)(base + t.rrow * ld + col0 + t.rcol) = w;
  }
}
__device__ __forceinline__ u32x4 load32_issue(const tile_io& t, const bf16_t* __restrict__ base, int64_t ld, int col0) {
  return *(const u32x4*)(base + t.rrow * ld + col0 + t.rcol);
}
__device__ __forceinline__ u32x4 load32_finish(const tile_io& t, const u32x4& w) {
  *(u32x4*)(t.scr + t.row_off) = w;
  return *(const u32x4*)(t.scr + t.frag_off);
}

// ---- products -----------------------------------------------------------------------------------------------------------------
template <int TILES, int KS>
__device__ __forceinline__ void mma_tiles(const unsigned char* img, int lane_off, const u32x4 (&b)[KS], f32x4 (&acc)[TILES]) {
#pragma unroll
  for (int s = 0; s < KS; ++s)
#pragma unroll
    for (int T = 0; T < TILES; ++T) acc[T] = mma32(*(const u32x4*)(img + lane_off + (T * KS + s) * kFrag), b[s], acc[T]);
}

__device__ __forceinline__ void ln_stats(f32x4 (&v)[4][2], float eps, float& mean, float& rstd) {
  float sum = 0.f;
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int r = 0; r < 4; ++r) sum += v[s][0][r] + v[s][1][r];
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  mean = sum * (1.f / 128.f);
  float sq = 0.f;
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[s][h][r] -= mean;
        sq = fmaf(v[s][h][r], v[s][h][r], sq);
      }
  sq += __shfl_xor(sq, 16, 64);
  sq += __shfl_xor(sq, 32, 64);
  rstd = rsqrtf(sq * (1.f / 128.f) + eps);
}

struct tail_fwd_params {
  const bf16_t *o, *x;
  const unsigned char* packed;
  const float *b_out, *b1, *b2, *n1w, *n1b, *n2w, *n2b;
  float eps;
  int64_t m;
  bf16_t *s1, *y1, *pre, *h, *s2, *y2, *y2p;
  float *st1, *st2;
  const float* pos_table;
  const int32_t* pos_idx;
};

template <int ACT>   // 1 = GELU(erf), 2 = ReLU
__global__ __launch_bounds__(kNTH, 4) void encoder_tail_fwd_bf16_k(const tail_fwd_params P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  float* par = (float*)(lds + 2 * kSlot);   // b_o | g1 | be1 | b2 | g2 | be2 | b1[256]
  const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15, wave = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)blockIdx.x * kRowsPerWg + wave * 16;
  const bool valid = r0 + c < P.m;
  const int64_t row = valid ? r0 + c : P.m - 1;
  const tile_io io = make_tile_io(lds + 2 * kSlot + kParF * 4 + wave * kScr, lane, r0, P.m);
  const unsigned char* packed = P.packed;
  dma_pieces(packed, lds, kSlot / 1024, wave, lane);                    // out-projection -> slot 0
  dma_pieces(packed + kSlot, lds + kSlot, kSlot / 1024, wave, lane);    // feed-forward chunk 0 -> slot 1
  u32x4 ob[4], xb[4];
  {
    u32x4 wo[4], wx[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) wo[s] = load32_issue(io, P.o, kC, 32 * s);
#pragma unroll
    for (int s = 0; s < 4; ++s) wx[s] = load32_issue(io, P.x, kC, 32 * s);
#pragma unroll
    for (int s = 0; s < 4; ++s) ob[s] = load32_finish(io, wo[s]);
#pragma unroll
    for (int s = 0; s < 4; ++s) xb[s] = load32_finish(io, wx[s]);
  }
  for (int i = threadIdx.x; i < kC; i += kNTH) {
    par[i] = P.b_out ? P.b_out[i] : 0.f;
    par[kC + i] = P.n1w[i];
    par[2 * kC + i] = P.n1b[i];
    par[3 * kC + i] = P.b2 ? P.b2[i] : 0.f;
    par[4 * kC + i] = P.n2w[i];
    par[5 * kC + i] = P.n2b[i];
  }
  for (int i = threadIdx.x; i < kFF; i += kNTH) par[6 * kC + i] = P.b1 ? P.b1[i] : 0.f;
  const int lane_off = lane * 16;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc[8];
#pragma unroll
  for (int T = 0; T < 8; ++T) acc[T] = z4;
  barrier_drain();   // both slots have landed, the parameters are in LDS
  mma_tiles<8, 4>(lds, lane_off, ob, acc);
  u32x4 yb[4];       // y1 (bf16): B operand of every chunk's first product, and the residual of the second LayerNorm
  {
    f32x4 v[4][2];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int n0 = 32 * s + 8 * g;
      f32x4 xa, xc;
      unpack8(xb[s], xa, xc);
      v[s][0] = acc[2 * s] + *(const f32x4*)(par + n0) + xa;
      v[s][1] = acc[2 * s + 1] + *(const f32x4*)(par + n0 + 4) + xc;
      if (P.s1 != nullptr) store32<true>(io, P.s1, kC, 32 * s, pack8(v[s][0], v[s][1]));
    }
    float mean, rstd;
    ln_stats(v, P.eps, mean, rstd);
    if (valid && g == 0) ((float2*)P.st1)[row] = make_float2(mean, rstd);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int n0 = 32 * s + 8 * g;
      const f32x4 y0 = v[s][0] * rstd * *(const f32x4*)(par + kC + n0) + *(const f32x4*)(par + 2 * kC + n0);
      const f32x4 y1 = v[s][1] * rstd * *(const f32x4*)(par + kC + n0 + 4) + *(const f32x4*)(par + 2 * kC + n0 + 4);
      yb[s] = pack8(y0, y1);
      store32<true>(io, P.y1, kC, 32 * s, yb[s]);
    }
#pragma unroll
    for (int T = 0; T < 8; ++T) acc[T] = z4;
  }
  barrier_drain();   // slot 0 is free
  dma_pieces(packed + 2 * (size_t)kSlot, lds, kSlot / 1024, wave, lane);   // chunk 1 -> slot 0
#pragma unroll 1
  for (int j = 0; j < kNF; ++j) {
    const unsigned char* slot = lds + ((j + 1) & 1) * kSlot;
    f32x4 a1[4] = {z4, z4, z4, z4};
    mma_tiles<4, 4>(slot, lane_off, yb, a1);
    u32x4 hb[2];
#pragma unroll
    for (int tp = 0; tp < 2; ++tp) {
      const int nl = kHC * j + 32 * tp + 8 * g;
      const f32x4 p0 = a1[2 * tp] + *(const f32x4*)(par + 6 * kC + nl), p1 = a1[2 * tp + 1] + *(const f32x4*)(par + 6 * kC + nl + 4);
      f32x4 h0, h1;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        h0[r] = ACT == 1 ? gelu_f(p0[r]) : fmaxf(p0[r], 0.f);
        h1[r] = ACT == 1 ? gelu_f(p1[r]) : fmaxf(p1[r], 0.f);
      }
      store32<true>(io, P.pre, kFF, kHC * j + 32 * tp, pack8(p0, p1));
      hb[tp] = pack8(h0, h1);
      store32<true>(io, P.h, kFF, kHC * j + 32 * tp, hb[tp]);
    }
    mma_tiles<8, 2>(slot + kP1, lane_off, hb, acc);
    if (j + 1 < kNF) {
      barrier_drain();   // this chunk's slot is free; the next chunk's images (requested a phase ago) have landed
      if (j + 2 < kNF) dma_pieces(packed + (size_t)(j + 3) * kSlot, lds + ((j + 1) & 1) * kSlot, kSlot / 1024, wave, lane);
    }
  }
  // s2 = y1 + linear2 + b2; y2 = LN2(s2) (; y2p = y2 + positional rows of the next layer)
  f32x4 v[4][2];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int n0 = 32 * s + 8 * g;
    f32x4 ya, yc;
    unpack8(yb[s], ya, yc);
    v[s][0] = acc[2 * s] + *(const f32x4*)(par + 3 * kC + n0) + ya;
    v[s][1] = acc[2 * s + 1] + *(const f32x4*)(par + 3 * kC + n0 + 4) + yc;
    if (P.s2 != nullptr) store32<true>(io, P.s2, kC, 32 * s, pack8(v[s][0], v[s][1]));
  }
  float mean, rstd;
  ln_stats(v, P.eps, mean, rstd);
  if (valid && g == 0) ((float2*)P.st2)[row] = make_float2(mean, rstd);
  const float* prow = P.pos_table != nullptr ? P.pos_table + (size_t)P.pos_idx[row] * kC : nullptr;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int n0 = 32 * s + 8 * g;
    const f32x4 y0 = v[s][0] * rstd * *(const f32x4*)(par + 4 * kC + n0) + *(const f32x4*)(par + 5 * kC + n0);
    const f32x4 y1 = v[s][1] * rstd * *(const f32x4*)(par + 4 * kC + n0 + 4) + *(const f32x4*)(par + 5 * kC + n0 + 4);
    store32<false>(io, P.y2, kC, 32 * s, pack8(y0, y1));
    if (prow != nullptr)
      store32<false>(io, P.y2p, kC, 32 * s, pack8(y0 + *(const f32x4*)(prow + n0), y1 + *(const f32x4*)(prow + n0 + 4)));
  }
}

struct tail_bwd_params {
  const bf16_t *dy2, *dy2p, *s2, *pre, *s1;
  const float *st2, *st1, *n2w, *n1w;
  const unsigned char* packed;
  int64_t m;
  bf16_t *ds2, *dpre, *ds1, *d_o;
  float *part2, *part1;   // [gridDim.x][256] each
};

// LayerNorm backward of a token (d: upstream gradient, sv: the LayerNorm's input; both fp32 values of bf16 tensors): d <- d(input),
// sv <- xhat; the tile's column sums of d * xhat | d go to red[256] of this wave (invalid tokens masked)
__device__ __forceinline__ void ln_bwd(f32x4 (&d)[4][2], f32x4 (&sv)[4][2], const float2 st, const float* gamma, int g, int c,
                                       bool valid, float* red) {
  float sg = 0.f, sgx = 0.f;
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const f32x4 w = *(const f32x4*)(gamma + 32 * s + 8 * g + 4 * h);
      sv[s][h] = (sv[s][h] - st.x) * st.y;
      f32x4 a, b;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        a[r] = row_sum16(valid ? d[s][h][r] * sv[s][h][r] : 0.f);
        b[r] = row_sum16(valid ? d[s][h][r] : 0.f);
      }
      if (c == 0) {
        *(f32x4*)(red + 32 * s + 8 * g + 4 * h) = a;
        *(f32x4*)(red + kC + 32 * s + 8 * g + 4 * h) = b;
      }
      d[s][h] *= w;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        sg += d[s][h][r];
        sgx = fmaf(d[s][h][r], sv[s][h][r], sgx);
      }
    }
  sg += __shfl_xor(sg, 16, 64);
  sg += __shfl_xor(sg, 32, 64);
  sgx += __shfl_xor(sgx, 16, 64);
  sgx += __shfl_xor(sgx, 32, 64);
  const float mg = sg * (1.f / 128.f), mgx = sgx * (1.f / 128.f);
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r = 0; r < 4; ++r) d[s][h][r] = st.y * (d[s][h][r] - mg - sv[s][h][r] * mgx);
}

template <int ACT>
__global__ __launch_bounds__(kNTH, 4) void encoder_tail_bwd_bf16_k(const tail_bwd_params P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  float* red = (float*)(lds + 2 * kSlot);       // [waves][256]: norm2's column sums, later norm1's
  const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15, wave = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)blockIdx.x * kRowsPerWg + wave * 16;
  const bool valid = r0 + c < P.m;
  const int64_t row = valid ? r0 + c : P.m - 1;
  const tile_io io = make_tile_io(lds + 2 * kSlot + kWaves * 2 * kC * 4 + wave * kScr, lane, r0, P.m);
  const unsigned char* packed = P.packed;
  dma_pieces(packed, lds, kSlot / 1024, wave, lane);                    // chunk 0 -> slot 0
  dma_pieces(packed + kSlot, lds + kSlot, kSlot / 1024, wave, lane);    // chunk 1 -> slot 1
  f32x4 d[4][2], sv[4][2];
  {
    u32x4 wd[4], we[4], ws[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) wd[s] = load32_issue(io, P.dy2, kC, 32 * s);
    if (P.dy2p != nullptr) {
#pragma unroll
      for (int s = 0; s < 4; ++s) we[s] = load32_issue(io, P.dy2p, kC, 32 * s);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) ws[s] = load32_issue(io, P.s2, kC, 32 * s);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      unpack8(load32_finish(io, wd[s]), d[s][0], d[s][1]);
      if (P.dy2p != nullptr) {   // the second gradient arriving at the LayerNorm output (its "+ positional rows" copy)
        f32x4 ea, eb;
        unpack8(load32_finish(io, we[s]), ea, eb);
        d[s][0] += ea;
        d[s][1] += eb;
      }
      unpack8(load32_finish(io, ws[s]), sv[s][0], sv[s][1]);
    }
  }
  const float2 st2 = ((const float2*)P.st2)[row];
  barrier_drain();   // both slots have landed
  u32x4 db[4];       // ds2 (bf16): B operand of every chunk's first product, and the residual branch of d(y1)
  ln_bwd(d, sv, st2, P.n2w, g, c, valid, red + wave * 2 * kC);
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    db[s] = pack8(d[s][0], d[s][1]);
    store32<true>(io, P.ds2, kC, 32 * s, db[s]);
  }
  const int lane_off = lane * 16;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc[8];
#pragma unroll
  for (int T = 0; T < 8; ++T) acc[T] = z4;
  // norm2's parameter-gradient partials leave now: the reduction area is used again by norm1's
  barrier_drain();
  if (threadIdx.x < 2 * kC) {
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) a += red[w * 2 * kC + threadIdx.x];   // fixed order: deterministic
    P.part2[(int64_t)blockIdx.x * 2 * kC + threadIdx.x] = a;
  }
  u32x4 pq[2];
  pq[0] = load32_issue(io, P.pre, kFF, 0);
  pq[1] = load32_issue(io, P.pre, kFF, 32);
#pragma unroll 1
  for (int j = 0; j < kNF; ++j) {
    const unsigned char* slot = lds + (j & 1) * kSlot;
    u32x4 qf[2];
    qf[0] = load32_finish(io, pq[0]);
    qf[1] = load32_finish(io, pq[1]);
    if (j + 1 < kNF) {   // the next chunk's pre-activation
      pq[0] = load32_issue(io, P.pre, kFF, kHC * (j + 1));
      pq[1] = load32_issue(io, P.pre, kFF, kHC * (j + 1) + 32);
    }
    f32x4 a1[4] = {z4, z4, z4, z4};
    mma_tiles<4, 4>(slot, lane_off, db, a1);
    u32x4 hb[2];
#pragma unroll
    for (int tp = 0; tp < 2; ++tp) {
      f32x4 q0, q1;
      unpack8(qf[tp], q0, q1);
      f32x4 p0 = a1[2 * tp], p1 = a1[2 * tp + 1];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        p0[r] *= ACT == 1 ? gelu_grad_f(q0[r]) : (q0[r] > 0.f ? 1.f : 0.f);
        p1[r] *= ACT == 1 ? gelu_grad_f(q1[r]) : (q1[r] > 0.f ? 1.f : 0.f);
      }
      hb[tp] = pack8(p0, p1);
      store32<true>(io, P.dpre, kFF, kHC * j + 32 * tp, hb[tp]);
    }
    mma_tiles<8, 2>(slot + kP1, lane_off, hb, acc);
    barrier_drain();   // this chunk's slot is free; the next phase's images (requested a phase ago) have landed
    if (j + 2 < kPhases) dma_pieces(packed + (size_t)(j + 2) * kSlot, lds + (j & 1) * kSlot, kSlot / 1024, wave, lane);
  }
  // d(y1) = ds2 + dpre W1; norm1 backward -> ds1 (= d(x) of the residual, = d(out-projection output))
  f32x4 dd[4][2], xs[4][2];
  {
    u32x4 ws[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) ws[s] = load32_issue(io, P.s1, kC, 32 * s);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      f32x4 ya, yb;
      unpack8(db[s], ya, yb);
      dd[s][0] = acc[2 * s] + ya;
      dd[s][1] = acc[2 * s + 1] + yb;
      unpack8(load32_finish(io, ws[s]), xs[s][0], xs[s][1]);
    }
  }
  const float2 st1 = ((const float2*)P.st1)[row];
  ln_bwd(dd, xs, st1, P.n1w, g, c, valid, red + wave * 2 * kC);
  u32x4 sb[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    sb[s] = pack8(dd[s][0], dd[s][1]);
    store32<true>(io, P.ds1, kC, 32 * s, sb[s]);
  }
#pragma unroll
  for (int T = 0; T < 8; ++T) acc[T] = z4;
  // the out-projection^T images are in slot (kNF & 1) = slot 0 (phase 4, requested behind chunk 2, drained by chunk 3's barrier)
  mma_tiles<8, 4>(lds + (kNF & 1) * kSlot, lane_off, sb, acc);
#pragma unroll
  for (int s = 0; s < 4; ++s) store32<false>(io, P.d_o, kC, 32 * s, pack8(acc[2 * s], acc[2 * s + 1]));
  barrier_drain();   // all waves' column sums of norm1 are in LDS
  if (threadIdx.x < 2 * kC) {
    float b = 0.f;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) b += red[w * 2 * kC + threadIdx.x];
    P.part1[(int64_t)blockIdx.x * 2 * kC + threadIdx.x] = b;
  }
}

bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

template <typename K>
int configure(K kernel, int lds_bytes, unsigned long long* mask) {
  if (sst_first_use_on_device(mask)) {
    SST_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    sst_mark_device(mask);
  }
  return SST_OK;
}

__global__ __launch_bounds__(1024) void tail_colsum_bf16_k(const float* __restrict__ partials, int nb, int width,
                                                           float* __restrict__ out0, float* __restrict__ out1, int split) {
  __shared__ float red[32][33];
  const int cx = threadIdx.x & 31, gy = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + cx;
  float acc = 0.f;
  if (i < width)
    for (int b = gy; b < nb; b += 32) acc += partials[(int64_t)b * width + i];
  red[gy][cx] = acc;
  __syncthreads();
  if (gy == 0 && i < width) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) t += red[k][cx];
    if (i < split)
      out0[i] = t;
    else
      out1[i - split] = t;
  }
}

}  // namespace

// the backward kernel without finishing launches for the LayerNorm parameter gradients (csrc/layer_exec.hip: riders of the
// weight-gradient reduction): partials [rows][256] at *part2 (norm2) and *part1 (norm1) inside args->workspace
int sst_internal_encoder_tail_bwd_bf16(const sst_encoder_tail_bwd_bf16_args* a, float** part2, float** part1, int* partial_rows,
                                       void* stream) {
  if (!a || a->m <= 0 || (a->act != 1 && a->act != 2)) return SST_ERR_ARG;
  const void* need[] = {a->dy2, a->s2, a->st2, a->n2w, a->pre, a->s1, a->st1, a->n1w, a->packed,
                        a->ds2, a->dpre, a->ds1, a->d_o, a->workspace};
  for (const void* p : need)
    if (!p || !aligned16(p)) return SST_ERR_ARG;
  if (a->dy2p && !aligned16(a->dy2p)) return SST_ERR_ARG;
  const int64_t rows = sst_div_up(a->m, (int64_t)kRowsPerWg);
  if (rows > 0x7fffffff / 2) return SST_ERR_UNSUPPORTED;
  tail_bwd_params P;
  P.dy2 = (const bf16_t*)a->dy2, P.dy2p = (const bf16_t*)a->dy2p, P.s2 = (const bf16_t*)a->s2, P.pre = (const bf16_t*)a->pre;
  P.s1 = (const bf16_t*)a->s1, P.st2 = a->st2, P.st1 = a->st1, P.n2w = a->n2w, P.n1w = a->n1w;
  P.packed = (const unsigned char*)a->packed + kPackDir, P.m = a->m;
  P.ds2 = (bf16_t*)a->ds2, P.dpre = (bf16_t*)a->dpre, P.ds1 = (bf16_t*)a->ds1, P.d_o = (bf16_t*)a->d_o;
  P.part2 = (float*)a->workspace;
  P.part1 = P.part2 + rows * 2 * kC;
  hipStream_t st = (hipStream_t)stream;
  static unsigned long long cfg1 = 0, cfg2 = 0;
  if (a->act == 1) {
    const int rc = configure(encoder_tail_bwd_bf16_k<1>, kLdsBwd, &cfg1);
    if (rc) return rc;
    hipLaunchKernelGGL(encoder_tail_bwd_bf16_k<1>, dim3((unsigned)rows), dim3(kNTH), kLdsBwd, st, P);
  } else {
    const int rc = configure(encoder_tail_bwd_bf16_k<2>, kLdsBwd, &cfg2);
    if (rc) return rc;
    hipLaunchKernelGGL(encoder_tail_bwd_bf16_k<2>, dim3((unsigned)rows), dim3(kNTH), kLdsBwd, st, P);
  }
  SST_LAUNCH_CHECK();
  *part2 = P.part2, *part1 = P.part1, *partial_rows = (int)rows;
  return SST_OK;
}

extern "C" {

int64_t sst_encoder_tail_pack_bf16_bytes(void) { return 2 * (int64_t)kPackDir; }

int sst_encoder_tail_pack_bf16_many(const float* const* d_w_out, const float* const* d_w1, const float* const* d_w2,
                                    void* const* d_packed, int n, void* stream) {
  if (n < 0 || (n > 0 && (!d_w_out || !d_w1 || !d_w2 || !d_packed))) return SST_ERR_ARG;
  for (int base = 0; base < n; base += kPackMany) {
    pack_batch B;
    const int cnt = n - base < kPackMany ? n - base : kPackMany;
    for (int i = 0; i < cnt; ++i) {
      if (!d_w_out[base + i] || !d_w1[base + i] || !d_w2[base + i] || !d_packed[base + i] || !aligned16(d_packed[base + i]))
        return SST_ERR_ARG;
      B.w[i].wo = d_w_out[base + i], B.w[i].w1 = d_w1[base + i], B.w[i].w2 = d_w2[base + i];
      B.dst[i] = (unsigned char*)d_packed[base + i];
    }
    hipLaunchKernelGGL(encoder_tail_pack_bf16_k, dim3(cnt * 2 * kPhases), dim3(512), 0, (hipStream_t)stream, B);
  }
  SST_LAUNCH_CHECK();
  return SST_OK;
}

int sst_encoder_tail_pack_bf16(const float* d_w_out, const float* d_w1, const float* d_w2, void* d_packed, void* stream) {
  return sst_encoder_tail_pack_bf16_many(&d_w_out, &d_w1, &d_w2, &d_packed, 1, stream);
}

int64_t sst_encoder_tail_bwd_bf16_workspace_bytes(int64_t m) {
  if (m < 0) return SST_ERR_ARG;
  return sst_align_up(2 * sst_div_up(m > 0 ? m : 1, (int64_t)kRowsPerWg) * 2 * kC * (int64_t)sizeof(float), 256);
}

int sst_encoder_tail_fwd_bf16(const sst_encoder_tail_fwd_bf16_args* a, void* stream) {
  if (!a || a->m < 0 || (a->act != 1 && a->act != 2)) return SST_ERR_ARG;
  if (a->m == 0) return SST_OK;
  const void* need[] = {a->o, a->x, a->packed, a->n1w, a->n1b, a->n2w, a->n2b, a->st1, a->y1, a->pre, a->h, a->st2, a->y2};
  for (const void* p : need)
    if (!p || !aligned16(p)) return SST_ERR_ARG;
  if ((a->s1 && !aligned16(a->s1)) || (a->s2 && !aligned16(a->s2))) return SST_ERR_ARG;
  if ((a->pos_table != nullptr) != (a->pos_idx != nullptr) || (a->pos_table != nullptr) != (a->y2p != nullptr)) return SST_ERR_ARG;
  if (a->pos_table && (!aligned16(a->pos_table) || !aligned16(a->y2p))) return SST_ERR_ARG;
  tail_fwd_params P;
  P.o = (const bf16_t*)a->o, P.x = (const bf16_t*)a->x, P.packed = (const unsigned char*)a->packed;
  P.b_out = a->b_out, P.b1 = a->b1, P.b2 = a->b2, P.n1w = a->n1w, P.n1b = a->n1b, P.n2w = a->n2w, P.n2b = a->n2b;
  P.eps = a->eps, P.m = a->m;
  P.s1 = (bf16_t*)a->s1, P.y1 = (bf16_t*)a->y1, P.pre = (bf16_t*)a->pre, P.h = (bf16_t*)a->h, P.s2 = (bf16_t*)a->s2;
  P.y2 = (bf16_t*)a->y2, P.y2p = (bf16_t*)a->y2p, P.st1 = a->st1, P.st2 = a->st2;
  P.pos_table = a->pos_table, P.pos_idx = a->pos_idx;
  const int64_t rows = sst_div_up(a->m, (int64_t)kRowsPerWg);
  if (rows > 0x7fffffff) return SST_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  static unsigned long long cfg1 = 0, cfg2 = 0;
  if (a->act == 1) {
    const int rc = configure(encoder_tail_fwd_bf16_k<1>, kLdsFwd, &cfg1);
    if (rc) return rc;
    hipLaunchKernelGGL(encoder_tail_fwd_bf16_k<1>, dim3((unsigned)rows), dim3(kNTH), kLdsFwd, st, P);
  } else {
    const int rc = configure(encoder_tail_fwd_bf16_k<2>, kLdsFwd, &cfg2);
    if (rc) return rc;
    hipLaunchKernelGGL(encoder_tail_fwd_bf16_k<2>, dim3((unsigned)rows), dim3(kNTH), kLdsFwd, st, P);
  }
  SST_LAUNCH_CHECK();
  return SST_OK;
}

int sst_encoder_tail_bwd_bf16(const sst_encoder_tail_bwd_bf16_args* a, void* stream) {
  if (!a || a->m < 0) return SST_ERR_ARG;
  if (!a->dn2w || !a->dn2b || !a->dn1w || !a->dn1b) return SST_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (a->m == 0) {
    for (float* p : {a->dn2w, a->dn2b, a->dn1w, a->dn1b}) SST_HIP(hipMemsetAsync(p, 0, sizeof(float) * kC, st));
    return SST_OK;
  }
  float *part2 = nullptr, *part1 = nullptr;
  int rows = 0;
  const int rc = sst_internal_encoder_tail_bwd_bf16(a, &part2, &part1, &rows, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(tail_colsum_bf16_k, dim3(2 * kC / 32), dim3(1024), 0, st, part2, rows, 2 * kC, a->dn2w, a->dn2b, kC);
  hipLaunchKernelGGL(tail_colsum_bf16_k, dim3(2 * kC / 32), dim3(1024), 0, st, part1, rows, 2 * kC, a->dn1w, a->dn1b, kC);
  SST_LAUNCH_CHECK();
  return SST_OK;
}

}  // extern "C"
