// The part of a post-norm SRA encoder layer that follows the attention core (models/sst/sst_basic_block_v2.py:113-118:
//   src = norm1(src + out_proj(attn));  src = norm2(src + linear2(act(linear1(src)))) )
// as ONE kernel forward and ONE kernel backward in the exact-split arithmetic of csrc/dense_f32x6.hip ("f32x6": fp32 operands,
// three bf16 parts each, the six products with i + j <= 2 on v_mfma_f32_16x16x32_bf16, fp32 accumulation).
//
// Why: the launch-per-product sequence of csrc/layer_exec.hip moved 69 [M, 128] tensors per layer through HBM (3.2 GB at 90 k
// voxels, 38 of the 41 GB of a training step; profiles/r05/k_step_traffic.txt) and the dense kernels already ran at 0.7 of the
// copy rate: what was left to take were the BYTES.  Between the attention core and the next layer every intermediate of a
// token depends on that token alone, so a wave can carry its 16 tokens through the whole chain in registers:
//   forward : o, x -> [out-proj + b + x] = s1 -> LN1 -> y1 -> [W1 + b1] = pre -> act -> h -> [W2 + b2 + y1] = s2 -> LN2 -> y2 (, y2 + pos)
//             reads 2 tensors of width 128, writes s1, y1, pre (256), h (256), s2, y2, y2p: 11 instead of 16 widths of 128
//   backward: dy2 (, dy2p), s2 -> LN2' -> ds2 -> [W2^T] * act'(pre) = dpre -> [W1^T] + ds2 = dy1 -> LN1'(s1) -> ds1 -> [W_o^T] = d_o
//             reads 5, writes 5 widths of 128 instead of 18; d(gamma) | d(beta) of both LayerNorms leave as per-workgroup partials.
// What makes the chain free of shuffles: the accumulator tile of v_mfma_f32_16x16x32_bf16 (lane (c, g): token c, 4 output rows)
// with the weight rows permuted in LDS (w_lds_row: a lane ends with 8 consecutive output columns of its token per pair of tiles) is
// exactly the B-operand fragment (token c, k = 8 g .. 8 g + 7) of the next product's k-step - the output of one product is split
// into its three bf16 parts and multiplied again without leaving the lane.
//
// What makes it fit: three bf16 images of W1 and W2 are 393 KB, LDS is 160 KB.  The feed-forward width is cut into 8 chunks of
// 32 hidden columns: chunk j needs W1[32 j .., :] (32 x 128) and W2[:, 32 j ..] (128 x 32), 57 KB as images; the out-projection
// is cut into two k-halves (128 x 64, 55 KB each).  A workgroup (8 waves, 128 tokens) walks the 10 chunks with two LDS slots:
// the fp32 weights of chunk i + 1 are requested from L2 before chunk i is multiplied, split and stored after it, one barrier per
// chunk.  Weight traffic is L2 -> LDS only (0.5 MB per 128 tokens against 0.7 MB of HBM traffic for the same tokens).
#include <math.h>
#include <stdlib.h>
#include "common.h"

#ifndef SST_NT_OUT
#define SST_NT_OUT false
#endif
#ifndef SST_NT_BWD
#define SST_NT_BWD true
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack2(float lo, float hi) {  // one v_cvt_pk_bf16_f32 (RNE)
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float lo_f(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float hi_f(unsigned p) { return __uint_as_float(p & 0xffff0000u); }
__device__ __forceinline__ f32x4 mma32(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// two floats -> their three bf16 parts, packed pairwise (exact: see csrc/dense_f32x6.hip)
__device__ __forceinline__ void split2(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) {
  p0 = pack2(a, b);
  const float ra = a - lo_f(p0), rb = b - hi_f(p0);
  p1 = pack2(ra, rb);
  p2 = pack2(ra - lo_f(p1), rb - hi_f(p1));
}
struct img3 {   // the three parts of 8 consecutive k of one token / one weight row
  u32x4 p0, p1, p2;
};
__device__ __forceinline__ img3 split8(const f32x4& a, const f32x4& b) {
  unsigned q0[4], q1[4], q2[4];
  split2(a[0], a[1], q0[0], q1[0], q2[0]);
  split2(a[2], a[3], q0[1], q1[1], q2[1]);
  split2(b[0], b[1], q0[2], q1[2], q2[2]);
  split2(b[2], b[3], q0[3], q1[3], q2[3]);
  img3 r;
  r.p0 = (u32x4){q0[0], q0[1], q0[2], q0[3]};
  r.p1 = (u32x4){q1[0], q1[1], q1[2], q1[3]};
  r.p2 = (u32x4){q2[0], q2[1], q2[2], q2[3]};
  return r;
}
// the fp32 values back from their parts: (p2 + p1) + p0, both additions exact (p1 + p2 = x - p0 was formed exactly)
__device__ __forceinline__ void join8(const img3& v, f32x4& a, f32x4& b) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    a[2 * i] = (lo_f(v.p2[i]) + lo_f(v.p1[i])) + lo_f(v.p0[i]);
    a[2 * i + 1] = (hi_f(v.p2[i]) + hi_f(v.p1[i])) + hi_f(v.p0[i]);
    b[2 * i] = (lo_f(v.p2[2 + i]) + lo_f(v.p1[2 + i])) + lo_f(v.p0[2 + i]);
    b[2 * i + 1] = (hi_f(v.p2[2 + i]) + hi_f(v.p1[2 + i])) + hi_f(v.p0[2 + i]);
  }
}

__device__ __forceinline__ float erf_as(float z, float& e) {  // Abramowitz & Stegun 7.1.26, |error| <= 1.5e-7 (as dense_f32x6.hip)
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

// sum over the 16 lanes of a DPP row (the 16 tokens of a tile held by the lanes of one k group), every lane gets the total:
// quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror - four v_add_f32 with a DPP operand, no LDS crossbar
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row_sum16(float v) {
  v = dpp_add<0xB1>(v);
  v = dpp_add<0x4E>(v);
  v = dpp_add<0x141>(v);
  return dpp_add<0x140>(v);
}

constexpr int kC = 128, kFF = 256, kHC = 32, kNF = kFF / kHC;   // 8 feed-forward chunks of 32 hidden columns
// LDS images are FRAGMENT-contiguous: the 16 rows x 32 k (bf16) an MFMA A operand covers are one 1 KiB block in lane order
// (lane (c, g) = row c, k 8 g .. 8 g + 7 at byte 16 * (16 g + c)), a fragment read is base + 16 * lane.  ds_read_b128 is served in
// four groups of 16 NON-contiguous lanes ({0-3, 12-15, 20-27}, ... - MI355X_MICROARCH.md, LDS): a row-major image with a padded
// row stride (K * 2 + 16 bytes, csrc/dense_f32x6.hip) puts two lanes of every group on the same banks - SQ_LDS_BANK_CONFLICT was
// 42 % of SQ_LDS_IDX_ACTIVE; lane order has every group cover all 64 banks exactly once, and needs no padding.
constexpr int kFrag = 1024;
constexpr int kI1 = 2 * 4 * kFrag;      // a chunk's first product: 2 row tiles (32 hidden columns) x 4 k-steps (K = 128)
constexpr int kI2 = 8 * 1 * kFrag;      // ... second product: 8 row tiles (128 outputs) x 1 k-step (the chunk's 32 hidden columns)
constexpr int kIA = 8 * 2 * kFrag;      // an out-projection k-half: 8 row tiles x 2 k-steps
constexpr int kPart1 = 3 * kI1, kPart2 = 3 * kI2;
constexpr int kSlot = kPart1 + kPart2;                   // 48 KiB
static_assert(3 * kIA <= kSlot, "slot");
#ifndef SST_TAIL_WAVES
#define SST_TAIL_WAVES 4
#endif
constexpr int kWaves = SST_TAIL_WAVES, kNTH = 64 * kWaves, kRowsPerWg = 16 * kWaves;   // two workgroups per CU
static_assert(kNTH >= 2 * kC, "one thread per LayerNorm parameter-gradient column");
constexpr int kNChunks = kNF + 2;
constexpr int kPackDir = kNChunks * kSlot;               // packed weight images of one direction: 10 chunk images of kSlot bytes
constexpr int kParF = 6 * kC + kFF;                       // forward parameters in LDS: b_o, g1, be1, b2, g2, be2 | b1
constexpr int kArea = kSlot;                               // LDS weight area of a workgroup: first-product part | second-product part
constexpr int kScr = 2048;                               // per wave: one 16-token x 32-column fp32 block on its way between layouts
constexpr int kLdsFwd = kArea + kParF * 4 + kWaves * kScr;
constexpr int kLdsBwd = kArea + 2 * kC * 4 + kWaves * 2 * kC * 4 + kWaves * kScr;   // gamma2 | gamma1 | one [waves][256] reduction area (used twice)

// ---- weight images ------------------------------------------------------------------------------------------------------------
// The three bf16 images of every chunk are formed ONCE per layer call by encoder_tail_pack_k (global fp32 -> registers -> split
// -> the chunk's LDS image, copied out as it lies in LDS, padding included) and the layer kernels fetch a chunk's image with
// LDS-DMA (global_load_lds_dwordx4: no registers, no VALU - the split of the weights inside the layer kernels was 1 760 of their
// 4 540 vector instructions per 16 tokens).  chunk ids: forward 0, 1 = out-projection k-halves, 2 .. 9 = feed-forward chunks;
// backward 0 .. 7 = feed-forward chunks, 8, 9 = out-projection k-halves (of the TRANSPOSED products).
struct tail_weights {
  const float* wo;   // [128][128]
  const float* w1;   // [256][128]
  const float* w2;   // [128][256]
};

template <bool BWD>
__device__ __forceinline__ void stage_load(const tail_weights W, int chunk, int t, float (&r)[16]) {
  if (!BWD) {
    if (chunk < 2) {   // out-projection half q: rows n, k = 64 q + k8 ..: two items per thread
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int idx = t + 512 * u, n = idx >> 3, k8 = (idx & 7) * 8;
        const float* p = W.wo + n * kC + 64 * chunk + k8;
        const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) r[8 * u + e] = a[e], r[8 * u + 4 + e] = b[e];
      }
    } else {
      const int j = chunk - 2;
      {   // W1 rows 32 j + n (n = t >> 4), k8 = (t & 15) * 8
        const float* p = W.w1 + (size_t)(kHC * j + (t >> 4)) * kC + (t & 15) * 8;
        const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = a[e], r[4 + e] = b[e];
      }
      {   // W2 rows n = t >> 2, columns 32 j + (t & 3) * 8 ..
        const float* p = W.w2 + (size_t)(t >> 2) * kFF + kHC * j + (t & 3) * 8;
        const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) r[8 + e] = a[e], r[12 + e] = b[e];
      }
    }
  } else {
    if (chunk >= kNF) {   // d_o = ds1 W_o: image rows = in-features j, k = out-features n; half q: n = 64 q + k8 + e
      const int q = chunk - kNF;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int idx = t + 512 * u, jn = idx & 127, k8 = (idx >> 7) * 8;
        const float* p = W.wo + (size_t)(64 * q + k8) * kC + jn;
#pragma unroll
        for (int e = 0; e < 8; ++e) r[8 * u + e] = p[e * kC];
      }
    } else {
      const int j = chunk;
      {   // first product dpre_j = ds2 W2[:, 32 j ..]: image rows = hidden column h (t & 31), k = n: W2[k8 + e][32 j + h]
        const float* p = W.w2 + (size_t)((t >> 5) * 8) * kFF + kHC * j + (t & 31);
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = p[e * kFF];
      }
      {   // second product dy1 += dpre_j W1[32 j .., :]: image rows = n' (t & 127), k = hidden: W1[32 j + k8 + e][n']
        const float* p = W.w1 + (size_t)(kHC * j + (t >> 7) * 8) * kC + (t & 127);
#pragma unroll
        for (int e = 0; e < 8; ++e) r[8 + e] = p[e * kC];
      }
    }
  }
}

// byte offset inside an image of the 16 bytes (row lr of the image, k8 .. k8 + 7), the image having `ks` k-steps per row tile
__device__ __forceinline__ int frag_off(int lr, int k8, int ks) {
  return ((lr >> 4) * ks + (k8 >> 5)) * kFrag + (16 * ((k8 >> 3) & 3) + (lr & 15)) * 16;
}

__device__ __forceinline__ void store_images(unsigned char* base, int image_bytes, int off, const float* v) {
  const f32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
  const img3 s = split8(a, b);
  *(u32x4*)(base + off) = s.p0;
  *(u32x4*)(base + image_bytes + off) = s.p1;
  *(u32x4*)(base + 2 * image_bytes + off) = s.p2;
}

template <bool BWD>
__device__ __forceinline__ void stage_store(unsigned char* slot, int chunk, int t, const float (&r)[16]) {
  const bool outproj = BWD ? chunk >= kNF : chunk < 2;
  if (outproj) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int idx = t + 512 * u;
      const int n = BWD ? (idx & 127) : (idx >> 3), k8 = BWD ? (idx >> 7) * 8 : (idx & 7) * 8;
      store_images(slot, kIA, frag_off(w_lds_row(n), k8, 2), r + 8 * u);
    }
  } else {
    const int n1 = BWD ? (t & 31) : (t >> 4), k1 = BWD ? (t >> 5) * 8 : (t & 15) * 8;
    store_images(slot, kI1, frag_off(w_lds_row(n1), k1, 4), r);
    const int n2 = BWD ? (t & 127) : (t >> 2), k2 = BWD ? (t >> 7) * 8 : (t & 3) * 8;
    store_images(slot + kPart1, kI2, frag_off(w_lds_row(n2), k2, 1), r + 8);
  }
}

// the chunk images of both directions: packed[dir][chunk][kSlot bytes], dir 0 = forward, 1 = backward.  One workgroup per image;
// a launch takes the weights of up to 16 layers (a whole encoder stack: one launch per forward pass instead of one per layer -
// twelve 5 us launches of a 10.4 ms step)
constexpr int kPackMany = 16;
struct pack_batch {
  tail_weights w[kPackMany];
  unsigned char* dst[kPackMany];
};
__global__ __launch_bounds__(512) void encoder_tail_pack_k(const pack_batch B) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int layer = blockIdx.x / (2 * kNChunks), blk = blockIdx.x % (2 * kNChunks);
  const tail_weights W = B.w[layer];
  const int chunk = blk % kNChunks, dir = blk / kNChunks;
  float r[16];
  if (dir == 0) {
    stage_load<false>(W, chunk, threadIdx.x, r);
    stage_store<false>(lds, chunk, threadIdx.x, r);
  } else {
    stage_load<true>(W, chunk, threadIdx.x, r);
    stage_store<true>(lds, chunk, threadIdx.x, r);
  }
  __syncthreads();
  unsigned char* dst = B.dst[layer] + (size_t)blk * kSlot;
  for (int i = threadIdx.x; i < kSlot / 16; i += 512) *(u32x4*)(dst + 16 * i) = *(const u32x4*)(lds + 16 * i);   // pads: whatever
}

// LDS-DMA of `pieces` KiB from src to the LDS offset dst (both 1 KiB pieces apart), the workgroup's waves taking pieces in turn.
// Completion: the issuing wave's vmcnt, then a barrier (every caller below drains with __syncthreads()).
__device__ __forceinline__ void dma_pieces(const unsigned char* __restrict__ src, unsigned char* dst, int pieces, int wave, int lane) {
  for (int i = wave; i < pieces; i += kWaves)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i * 1024 + lane * 16),
                                     (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
}

// barrier that does NOT drain the vector-memory counter (stores / DMA stay in flight across it): the LDS reads of the phase it
// ends are retired (lgkmcnt(0)), nothing else is waited for.  Only where no DMA has to have landed by this point.
__device__ __forceinline__ void barrier_lds_only() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// THE barrier of the chunk choreography: every vector-memory operation of this wave - its LDS-DMA pieces among them - and every
// LDS operation has completed, then the workgroup meets.  Written out: __syncthreads() leaves the vmcnt wait to the compiler's
// view of which LDS-DMA may alias which ds_read, and a missing one shows as rare wrong tiles in workgroups that start late
// (found by running the kernel twice on the same input: 2 % of the s2 rows differed).
__device__ __forceinline__ void barrier_drain() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- global memory <-> the MFMA fragment layout, through a wave-private LDS block ---------------------------------------------
// In the fragment layout lane (c, g) holds 32 bytes of token c: consecutive lanes are consecutive ROWS, so a dwordx4 access of a
// wave touches 64 different 16-byte pieces (16 rows x 4 pieces with holes between them) - measured: the forward kernel's stores
// alone took 70 of its 188 us (2 TB/s against 6.7 TB/s of a plain fill), and every fragment-shaped load pays the same way.
// Here a 16-token x 32-column block (2 KiB) passes through LDS so that in global memory lane l touches row l / 8, piece l % 8:
// 8 consecutive lanes = one full 128-byte line.  The 16-byte slots of a row are XOR-swizzled with the row so that both the
// fragment-side and the row-side LDS accesses are bank-conflict-free (ds_*_b128 lane groups: MI355X_MICROARCH.md, LDS).  One
// wave, in-order LDS: no barrier.
struct tile_io {
  unsigned char* scr;   // this wave's 2 KiB
  int frag_off0;        // fragment side: slot 2 g of row c (the second half: ^ 16)
  int row_off[2];       // row side, instruction i: row 8 i + lane / 8, slot lane % 8
  int64_t rrow[2];      // the global rows of the row side (clamped to m - 1 for loads)
  bool rvalid[2];
  int rcol;             // (lane % 8) * 4
};
__device__ __forceinline__ tile_io make_tile_io(unsigned char* scr, int lane, int64_t r0, int64_t m) {
  tile_io t;
  const int c = lane & 15, g = lane >> 4;
  t.scr = scr;
  t.frag_off0 = c * 128 + (((2 * g) ^ (c & 7)) << 4);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = 8 * i + (lane >> 3), p = lane & 7;
    t.row_off[i] = r * 128 + ((p ^ (r & 7)) << 4);
    t.rvalid[i] = r0 + r < m;
    t.rrow[i] = t.rvalid[i] ? r0 + r : m - 1;
  }
  t.rcol = (lane & 7) * 4;
  return t;
}
// columns col0 .. col0 + 31 of the wave's 16 tokens: fragment registers (v0: 8 g .. + 3, v1: 8 g + 4 .. + 7 of token c) -> memory
// STREAM: the tensor is read again only much later (kept for the backward pass): nontemporal stores - they do not queue behind
// the cache's write-back of earlier lines (measured on the forward kernel: 190 -> 175 us with every store streamed)
template <bool STREAM>
__device__ __forceinline__ void store32(const tile_io& t, float* __restrict__ base, int64_t ld, int col0, const f32x4& v0,
                                        const f32x4& v1) {
  *(f32x4*)(t.scr + t.frag_off0) = v0;
  *(f32x4*)(t.scr + (t.frag_off0 ^ 16)) = v1;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const f32x4 w = *(const f32x4*)(t.scr + t.row_off[i]);
    if (t.rvalid[i]) {
      if (STREAM)
        __builtin_nontemporal_store(w, (f32x4*)(base + t.rrow[i] * ld + col0 + t.rcol));
      else
        *(f32x4*)(base + t.rrow[i] * ld + col0 + t.rcol) = w;
    }
  }
}
__device__ __forceinline__ void load32_issue(const tile_io& t, const float* __restrict__ base, int64_t ld, int col0, f32x4 (&w)[2]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) w[i] = *(const f32x4*)(base + t.rrow[i] * ld + col0 + t.rcol);
}
__device__ __forceinline__ void load32_finish(const tile_io& t, const f32x4 (&w)[2], f32x4& v0, f32x4& v1) {
#pragma unroll
  for (int i = 0; i < 2; ++i) *(f32x4*)(t.scr + t.row_off[i]) = w[i];
  v0 = *(const f32x4*)(t.scr + t.frag_off0);
  v1 = *(const f32x4*)(t.scr + (t.frag_off0 ^ 16));
}

// ---- the three products on a wave's 16 tokens -------------------------------------------------------------------------------
// out-projection k-half (two k-steps, 8 output tiles): acc[T] += W(slot)[T] x(ks), x split on the way (smallest products first)
template <int HALF>
__device__ __forceinline__ void mma_half(const unsigned char* slot, int lane_off, const f32x4 (&x)[4][2], f32x4 (&acc)[8]) {
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
    const img3 xs = split8(x[2 * HALF + s2][0], x[2 * HALF + s2][1]);
#pragma unroll
    for (int t0 = 0; t0 < 8; t0 += 4) {
      u32x4 a0[4], a1[4], a2[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int off = lane_off + ((t0 + u) * 2 + s2) * kFrag;
        a0[u] = *(const u32x4*)(slot + off);
        a1[u] = *(const u32x4*)(slot + kIA + off);
        a2[u] = *(const u32x4*)(slot + 2 * kIA + off);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[t0 + u] = mma32(a0[u], xs.p2, acc[t0 + u]);
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[t0 + u] = mma32(a2[u], xs.p0, acc[t0 + u]);
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[t0 + u] = mma32(a1[u], xs.p1, acc[t0 + u]);
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[t0 + u] = mma32(a0[u], xs.p1, acc[t0 + u]);
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[t0 + u] = mma32(a1[u], xs.p0, acc[t0 + u]);
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[t0 + u] = mma32(a0[u], xs.p0, acc[t0 + u]);
    }
  }
}

// first product of a feed-forward chunk: 32 output columns (two tiles) over K = 128 from the kept parts of the input tile.
// Per tile three accumulators - the leading product, the two 2^-8 corrections, the three 2^-16 corrections - so that a
// dependent MFMA is at least four instructions behind its predecessor with the fragments of ONE k-step in registers, and the
// leading sum does not absorb the corrections' roundings.
__device__ __forceinline__ void mma_first(const unsigned char* slot, int lane_off1, const img3 (&y)[4], f32x4& out0, f32x4& out1) {
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  f32x4 am[2] = {z, z}, ca[2] = {z, z}, cb[2] = {z, z};
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    u32x4 a0[2], a1[2], a2[2];
#pragma unroll
    for (int T = 0; T < 2; ++T) {
      const int off = lane_off1 + (T * 4 + s) * kFrag;
      a0[T] = *(const u32x4*)(slot + off);
      a1[T] = *(const u32x4*)(slot + kI1 + off);
      a2[T] = *(const u32x4*)(slot + 2 * kI1 + off);
    }
#pragma unroll
    for (int T = 0; T < 2; ++T) ca[T] = mma32(a0[T], y[s].p2, ca[T]);
#pragma unroll
    for (int T = 0; T < 2; ++T) cb[T] = mma32(a0[T], y[s].p1, cb[T]);
#pragma unroll
    for (int T = 0; T < 2; ++T) ca[T] = mma32(a2[T], y[s].p0, ca[T]);
#pragma unroll
    for (int T = 0; T < 2; ++T) am[T] = mma32(a0[T], y[s].p0, am[T]);
#pragma unroll
    for (int T = 0; T < 2; ++T) ca[T] = mma32(a1[T], y[s].p1, ca[T]);
#pragma unroll
    for (int T = 0; T < 2; ++T) cb[T] = mma32(a1[T], y[s].p0, cb[T]);
  }
  out0 = am[0] + (cb[0] + ca[0]);
  out1 = am[1] + (cb[1] + ca[1]);
}

// second product of a feed-forward chunk: one k-step (the chunk's 32 hidden columns), 8 output tiles
__device__ __forceinline__ void mma_second(const unsigned char* slot2, int lane_off2, const img3& h, f32x4 (&acc)[8]) {
#pragma unroll
  for (int t0 = 0; t0 < 8; t0 += 4) {
    u32x4 a0[4], a1[4], a2[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int off = lane_off2 + (t0 + u) * kFrag;
      a0[u] = *(const u32x4*)(slot2 + off);
      a1[u] = *(const u32x4*)(slot2 + kI2 + off);
      a2[u] = *(const u32x4*)(slot2 + 2 * kI2 + off);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[t0 + u] = mma32(a0[u], h.p2, acc[t0 + u]);
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[t0 + u] = mma32(a2[u], h.p0, acc[t0 + u]);
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[t0 + u] = mma32(a1[u], h.p1, acc[t0 + u]);
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[t0 + u] = mma32(a0[u], h.p1, acc[t0 + u]);
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[t0 + u] = mma32(a1[u], h.p0, acc[t0 + u]);
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[t0 + u] = mma32(a0[u], h.p0, acc[t0 + u]);
  }
}

// LayerNorm of a token's 128 values held as v[4][2] (lane (c, g): columns 32 s + 8 g + 0 .. 7 of token c): two-pass statistics
// across the four k groups of the token (lanes c, c + 16, c + 32, c + 48), v <- v - mean on return
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
  const float *o, *x;
  const unsigned char* packed;   // this direction's 10 chunk images
  const float *b_out, *b1, *b2, *n1w, *n1b, *n2w, *n2b;
  float eps;
  int64_t m;
  float *s1, *st1, *y1, *pre, *h, *s2, *st2, *y2;
  const float* pos_table;
  const int32_t* pos_idx;
  float* y2p;
};

template <int ACT>   // 1 = GELU(erf), 2 = ReLU
__global__ __launch_bounds__(kNTH, 2) void encoder_tail_fwd_x6_k(const tail_fwd_params P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  float* par = (float*)(lds + kArea);   // b_o | g1 | be1 | b2 | g2 | be2 | b1[256]
  const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15, wave = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)blockIdx.x * kRowsPerWg + wave * 16;
  const bool valid = r0 + c < P.m;
  const int64_t row = valid ? r0 + c : P.m - 1;
  const tile_io io = make_tile_io(lds + kArea + kParF * 4 + wave * kScr, lane, r0, P.m);
  const unsigned char* packed = P.packed;
  unsigned char* const part2 = lds + kPart1;
  dma_pieces(packed, lds, 3 * kIA / 1024, wave, lane);   // out-projection half 0
  // the wave's o tile (B operand of the out-projection) and x tile (the residual: k-step s <-> columns 32 s ..)
  f32x4 ob[4][2], xr[4][2];
  {
    f32x4 wo[4][2], wx[4][2];
#pragma unroll
    for (int s = 0; s < 4; ++s) load32_issue(io, P.o, kC, 32 * s, wo[s]);
#pragma unroll
    for (int s = 0; s < 4; ++s) load32_issue(io, P.x, kC, 32 * s, wx[s]);
#pragma unroll
    for (int s = 0; s < 4; ++s) load32_finish(io, wo[s], ob[s][0], ob[s][1]);
#pragma unroll
    for (int s = 0; s < 4; ++s) load32_finish(io, wx[s], xr[s][0], xr[s][1]);
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

  const int lane_offA = lane * 16, lane_off1 = lane * 16, lane_off2 = lane * 16;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc[8];
#pragma unroll
  for (int T = 0; T < 8; ++T) acc[T] = z4;
  img3 yi[4];   // y1's parts: B operand of every chunk's first product, and the residual of the second LayerNorm

  // chunk choreography, ONE slot per workgroup (the CU's second workgroup computes while this one waits).  Out-projection halves
  // take the whole slot: DMA, drain, barrier, multiply, barrier.  A feed-forward chunk uses the slot's two parts in turn - while
  // the first product reads part 1 the DMA of the chunk's second-product images lands in part 2, while the second product reads
  // part 2 the next chunk's first-product images land in part 1 - so no DMA latency is exposed after the first chunk.
  // Every barrier_drain() drains vmcnt and lgkmcnt before its s_barrier: "landed for all waves" and "nobody reads any more".
  barrier_drain();
  mma_half<0>(lds, lane_offA, ob, acc);
  barrier_drain();
  dma_pieces(packed + kSlot, lds, 3 * kIA / 1024, wave, lane);
  barrier_drain();
  mma_half<1>(lds, lane_offA, ob, acc);
  barrier_drain();
  dma_pieces(packed + 2 * kSlot, lds, kSlot / 1024, wave, lane);   // feed-forward chunk 0, both parts (landed at the next barrier)
  {
    // s1 = out-proj + b_o + x; y1 = LN1(s1)
    f32x4 v[4][2];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int n0 = 32 * s + 8 * g;
      v[s][0] = acc[2 * s] + *(const f32x4*)(par + n0) + xr[s][0];
      v[s][1] = acc[2 * s + 1] + *(const f32x4*)(par + n0 + 4) + xr[s][1];
      if (P.s1 != nullptr) store32<true>(io, P.s1, kC, 32 * s, v[s][0], v[s][1]);
    }
    float mean, rstd;
    ln_stats(v, P.eps, mean, rstd);
    if (valid && g == 0) ((float2*)P.st1)[row] = make_float2(mean, rstd);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int n0 = 32 * s + 8 * g;
      const f32x4 y0 = v[s][0] * rstd * *(const f32x4*)(par + kC + n0) + *(const f32x4*)(par + 2 * kC + n0);
      const f32x4 y1 = v[s][1] * rstd * *(const f32x4*)(par + kC + n0 + 4) + *(const f32x4*)(par + 2 * kC + n0 + 4);
      store32<true>(io, P.y1, kC, 32 * s, y0, y1);
      yi[s] = split8(y0, y1);
    }
#pragma unroll
    for (int T = 0; T < 8; ++T) acc[T] = z4;
  }
  barrier_drain();   // chunk 0's images have landed
  f32x4 sp0 = z4, sp1 = z4, sh0 = z4, sh1 = z4;   // pre | h of the previous chunk, stored one phase later
#pragma unroll 1
  for (int j = 0; j < kNF; ++j) {
    if (j > 0) {   // behind the barrier that needed no store to be finished; acknowledged by the next drain, a whole phase away
      store32<true>(io, P.pre, kFF, kHC * (j - 1), sp0, sp1);
      store32<true>(io, P.h, kFF, kHC * (j - 1), sh0, sh1);
    }
    f32x4 p0, p1;
    mma_first(lds, lane_off1, yi, p0, p1);
    const int nl = kHC * j + 8 * g;
    p0 += *(const f32x4*)(par + 6 * kC + nl);
    p1 += *(const f32x4*)(par + 6 * kC + nl + 4);
    f32x4 h0, h1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      h0[r] = ACT == 1 ? gelu_f(p0[r]) : fmaxf(p0[r], 0.f);
      h1[r] = ACT == 1 ? gelu_f(p1[r]) : fmaxf(p1[r], 0.f);
    }
    const img3 hs = split8(h0, h1);
    barrier_drain();   // part 1 is free; this chunk's part 2 has landed; the previous chunk's stores are acknowledged
    if (j + 1 < kNF) dma_pieces(packed + (size_t)(j + 3) * kSlot, lds, kPart1 / 1024, wave, lane);
    mma_second(part2, lane_off2, hs, acc);
    sp0 = p0, sp1 = p1, sh0 = h0, sh1 = h1;
    barrier_drain();   // part 2 is free, the next chunk's part 1 has landed (nothing else is in flight)
    if (j + 1 < kNF) dma_pieces(packed + (size_t)(j + 3) * kSlot + kPart1, part2, kPart2 / 1024, wave, lane);
  }
  store32<true>(io, P.pre, kFF, kHC * (kNF - 1), sp0, sp1);
  store32<true>(io, P.h, kFF, kHC * (kNF - 1), sh0, sh1);
  // s2 = y1 + linear2 + b2; y2 = LN2(s2) (; y2p = y2 + positional rows of the next layer)
  f32x4 v[4][2];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int n0 = 32 * s + 8 * g;
    f32x4 ya, yb;
    join8(yi[s], ya, yb);
    v[s][0] = acc[2 * s] + *(const f32x4*)(par + 3 * kC + n0) + ya;
    v[s][1] = acc[2 * s + 1] + *(const f32x4*)(par + 3 * kC + n0 + 4) + yb;
    if (P.s2 != nullptr) store32<true>(io, P.s2, kC, 32 * s, v[s][0], v[s][1]);
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
    store32<SST_NT_OUT>(io, P.y2, kC, 32 * s, y0, y1);
    if (prow != nullptr) store32<SST_NT_OUT>(io, P.y2p, kC, 32 * s, y0 + *(const f32x4*)(prow + n0), y1 + *(const f32x4*)(prow + n0 + 4));
  }
}

struct tail_bwd_params {
  const float *dy2, *dy2p, *s2, *st2, *n2w, *pre, *s1, *st1, *n1w;
  const unsigned char* packed;   // this direction's 10 chunk images
  int64_t m;
  float *ds2, *dpre, *ds1, *d_o;
  float *part2, *part1;   // [gridDim.x][256] each: d(gamma) | d(beta) partials of norm2 / norm1
};

// LayerNorm backward of a token held as d[4][2] (upstream gradient) and sv[4][2] (the LayerNorm's input), gamma from LDS:
// d <- d(input), sv <- xhat.  The parameter-gradient contributions d * xhat | d of the tile's 16 tokens (invalid tokens masked)
// are summed over the tokens on the way (DPP row sums) into red[256] of this wave: d(gamma) at [n], d(beta) at [128 + n].
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
      d[s][h] *= w;   // g = dy * gamma
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
__global__ __launch_bounds__(kNTH, 2) void encoder_tail_bwd_x6_k(const tail_bwd_params P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  float* par = (float*)(lds + kArea);        // gamma2 | gamma1
  float* red = par + 2 * kC;                  // [waves][256]: norm2's column sums, later norm1's
  const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15, wave = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)blockIdx.x * kRowsPerWg + wave * 16;
  const bool valid = r0 + c < P.m;
  const int64_t row = valid ? r0 + c : P.m - 1;
  const tile_io io = make_tile_io(lds + kArea + 2 * kC * 4 + kWaves * 2 * kC * 4 + wave * kScr, lane, r0, P.m);
  const unsigned char* packed = P.packed;
  unsigned char* const part2 = lds + kPart1;
  dma_pieces(packed, lds, kSlot / 1024, wave, lane);   // feed-forward chunk 0, both parts: lands behind the first LayerNorm backward
  f32x4 d[4][2], sv[4][2];
  {
    f32x4 wd[4][2], we[4][2];
#pragma unroll
    for (int s = 0; s < 4; ++s) load32_issue(io, P.dy2, kC, 32 * s, wd[s]);
    if (P.dy2p != nullptr) {   // the second gradient arriving at the LayerNorm output (its "+ positional rows" copy)
#pragma unroll
      for (int s = 0; s < 4; ++s) load32_issue(io, P.dy2p, kC, 32 * s, we[s]);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        wd[s][0] += we[s][0];
        wd[s][1] += we[s][1];
      }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) load32_issue(io, P.s2, kC, 32 * s, we[s]);
#pragma unroll
    for (int s = 0; s < 4; ++s) load32_finish(io, wd[s], d[s][0], d[s][1]);
#pragma unroll
    for (int s = 0; s < 4; ++s) load32_finish(io, we[s], sv[s][0], sv[s][1]);
  }
  const float2 st2 = ((const float2*)P.st2)[row];
  for (int i = threadIdx.x; i < kC; i += kNTH) {
    par[i] = P.n2w[i];
    par[kC + i] = P.n1w[i];
  }
  barrier_drain();    // gamma is in LDS

  img3 di[4];   // ds2's parts: B operand of every chunk's first product, and the residual branch of d(y1)
  {
    ln_bwd(d, sv, st2, par, g, c, valid, red + wave * 2 * kC);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      store32<SST_NT_BWD>(io, P.ds2, kC, 32 * s, d[s][0], d[s][1]);
      di[s] = split8(d[s][0], d[s][1]);
    }
  }
  const int lane_offA = lane * 16, lane_off1 = lane * 16, lane_off2 = lane * 16;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc[8];
#pragma unroll
  for (int T = 0; T < 8; ++T) acc[T] = z4;
  f32x4 xs1[4][2];   // ds1 (fp32): B operand of the out-projection's data gradient
  f32x4 pq[2];       // the pre-activation of the running chunk (row side), requested one chunk ahead
  load32_issue(io, P.pre, kFF, 0, pq);

  // chunk choreography as in the forward kernel: one slot, its two parts alternate between "being read" and "being filled"
  // feed-forward chunk j: dpre_j = (ds2 W2[:, 32 j ..]) act'(pre_j), d(y1) += dpre_j W1[32 j .., :]
  f32x4 sd0 = z4, sd1 = z4;   // dpre of the previous chunk, stored one phase later (see the forward kernel)
  auto ffn_chunk = [&](int j, const f32x4 q0, const f32x4 q1) __attribute__((always_inline)) {
    if (j > 0) store32<SST_NT_BWD>(io, P.dpre, kFF, kHC * (j - 1), sd0, sd1);
    f32x4 p0, p1;
    mma_first(lds, lane_off1, di, p0, p1);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      p0[r] *= ACT == 1 ? gelu_grad_f(q0[r]) : (q0[r] > 0.f ? 1.f : 0.f);
      p1[r] *= ACT == 1 ? gelu_grad_f(q1[r]) : (q1[r] > 0.f ? 1.f : 0.f);
    }
    const img3 hs = split8(p0, p1);
    barrier_drain();   // part 1 is free (and this chunk's part 2 has landed)
    if (j + 1 < kNF) dma_pieces(packed + (size_t)(j + 1) * kSlot, lds, kPart1 / 1024, wave, lane);
    mma_second(part2, lane_off2, hs, acc);
    sd0 = p0, sd1 = p1;
    barrier_drain();   // part 2 is free, the next chunk's part 1 has landed
    if (j + 1 < kNF)
      dma_pieces(packed + (size_t)(j + 1) * kSlot + kPart1, part2, kPart2 / 1024, wave, lane);
    else
      dma_pieces(packed + (size_t)kNF * kSlot, lds, 3 * kIA / 1024, wave, lane);   // out-projection half 0: lands behind norm1's backward
  };
  // norm2's parameter-gradient partials of this workgroup leave now: the reduction area is used again by norm1's
  barrier_drain();   // chunk 0's images have landed; all waves' column sums are in LDS
  if (threadIdx.x < 2 * kC) {
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) a += red[w * 2 * kC + threadIdx.x];   // fixed order: deterministic
    P.part2[(int64_t)blockIdx.x * 2 * kC + threadIdx.x] = a;
  }
#pragma unroll 1
  for (int j = 0; j < kNF - 1; ++j) {
    f32x4 q0, q1;
    load32_finish(io, pq, q0, q1);
    load32_issue(io, P.pre, kFF, kHC * (j + 1), pq);   // the next chunk's pre-activation
    ffn_chunk(j, q0, q1);
  }
  {
    f32x4 q0, q1;
    load32_finish(io, pq, q0, q1);
    ffn_chunk(kNF - 1, q0, q1);
    store32<SST_NT_BWD>(io, P.dpre, kFF, kHC * (kNF - 1), sd0, sd1);
    {   // norm1's input (requested here: a tile of it held across the chunk spills)
      f32x4 ws[4][2];
#pragma unroll
      for (int s = 0; s < 4; ++s) load32_issue(io, P.s1, kC, 32 * s, ws[s]);
#pragma unroll
      for (int s = 0; s < 4; ++s) load32_finish(io, ws[s], xs1[s][0], xs1[s][1]);
    }
    // d(y1) = ds2 + dpre W1; norm1 backward -> ds1 (= d(x) of the residual, = d(out-projection output))
    f32x4 dd[4][2];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      f32x4 ya, yb;
      join8(di[s], ya, yb);
      dd[s][0] = acc[2 * s] + ya;
      dd[s][1] = acc[2 * s + 1] + yb;
    }
    const float2 st1 = ((const float2*)P.st1)[row];
    barrier_drain();   // norm2's column sums have been read by every thread: the reduction area is free
    ln_bwd(dd, xs1, st1, par + kC, g, c, valid, red + wave * 2 * kC);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      store32<SST_NT_BWD>(io, P.ds1, kC, 32 * s, dd[s][0], dd[s][1]);
      xs1[s][0] = dd[s][0];
      xs1[s][1] = dd[s][1];
    }
#pragma unroll
    for (int T = 0; T < 8; ++T) acc[T] = z4;
    barrier_drain();   // out-projection half 0 has landed
    mma_half<0>(lds, lane_offA, xs1, acc);
    barrier_drain();
    dma_pieces(packed + (size_t)(kNF + 1) * kSlot, lds, 3 * kIA / 1024, wave, lane);
    barrier_drain();
    mma_half<1>(lds, lane_offA, xs1, acc);
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) store32<SST_NT_OUT>(io, P.d_o, kC, 32 * s, acc[2 * s], acc[2 * s + 1]);
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

}  // namespace

// the backward kernel without a finishing launch for the LayerNorm parameter gradients: their per-workgroup partials stay in
// the workspace as [rows][256] (norm2's first, norm1's `rows` rows later) for the caller's reduction launch (csrc/layer_exec.hip)
int sst_internal_encoder_tail_bwd_f32x6(const sst_encoder_tail_bwd_args* a, float** part2, float** part1, int* partial_rows,
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
  P.dy2 = a->dy2, P.dy2p = a->dy2p, P.s2 = a->s2, P.st2 = a->st2, P.n2w = a->n2w, P.pre = a->pre, P.s1 = a->s1, P.st1 = a->st1;
  P.n1w = a->n1w, P.packed = (const unsigned char*)a->packed + kPackDir, P.m = a->m;
  P.ds2 = a->ds2, P.dpre = a->dpre, P.ds1 = a->ds1, P.d_o = a->d_o;
  P.part2 = (float*)a->workspace;
  P.part1 = P.part2 + rows * 2 * kC;
  hipStream_t st = (hipStream_t)stream;
  static unsigned long long cfg1 = 0, cfg2 = 0;
  if (a->act == 1) {
    const int rc = configure(encoder_tail_bwd_x6_k<1>, kLdsBwd, &cfg1);
    if (rc) return rc;
    hipLaunchKernelGGL(encoder_tail_bwd_x6_k<1>, dim3((unsigned)rows), dim3(kNTH), kLdsBwd, st, P);
  } else {
    const int rc = configure(encoder_tail_bwd_x6_k<2>, kLdsBwd, &cfg2);
    if (rc) return rc;
    hipLaunchKernelGGL(encoder_tail_bwd_x6_k<2>, dim3((unsigned)rows), dim3(kNTH), kLdsBwd, st, P);
  }
  SST_LAUNCH_CHECK();
  *part2 = P.part2, *part1 = P.part1, *partial_rows = (int)rows;
  return SST_OK;
}

// out[i] = sum over the partial rows, in the arithmetic of csrc/dense.hip colsum_partials_k / the rider of csrc/wgrad_x6.hip:
// 32 strided sums per column (rows gy, gy + 32, ...), added in order
namespace {
__global__ __launch_bounds__(1024) void tail_colsum_k(const float* __restrict__ partials, int nb, int width, float* __restrict__ out0,
                                                      float* __restrict__ out1, int split) {
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

extern "C" {

int64_t sst_encoder_tail_pack_bytes(void) { return 2 * (int64_t)kPackDir; }

int sst_encoder_tail_pack_f32x6_many(const float* const* d_w_out, const float* const* d_w1, const float* const* d_w2,
                                     void* const* d_packed, int n, void* stream) {
  if (n < 0 || (n > 0 && (!d_w_out || !d_w1 || !d_w2 || !d_packed))) return SST_ERR_ARG;
  static unsigned long long cfg = 0;
  const int rc = configure(encoder_tail_pack_k, kSlot, &cfg);
  if (rc) return rc;
  for (int base = 0; base < n; base += kPackMany) {
    pack_batch B;
    const int cnt = n - base < kPackMany ? n - base : kPackMany;
    for (int i = 0; i < cnt; ++i) {
      const float *wo = d_w_out[base + i], *w1 = d_w1[base + i], *w2 = d_w2[base + i];
      void* dst = d_packed[base + i];
      if (!wo || !w1 || !w2 || !dst || !aligned16(wo) || !aligned16(w1) || !aligned16(w2) || !aligned16(dst)) return SST_ERR_ARG;
      B.w[i].wo = wo, B.w[i].w1 = w1, B.w[i].w2 = w2;
      B.dst[i] = (unsigned char*)dst;
    }
    hipLaunchKernelGGL(encoder_tail_pack_k, dim3(cnt * 2 * kNChunks), dim3(512), kSlot, (hipStream_t)stream, B);
  }
  SST_LAUNCH_CHECK();
  return SST_OK;
}

int sst_encoder_tail_pack_f32x6(const float* d_w_out, const float* d_w1, const float* d_w2, void* d_packed, void* stream) {
  return sst_encoder_tail_pack_f32x6_many(&d_w_out, &d_w1, &d_w2, &d_packed, 1, stream);
}

int64_t sst_encoder_tail_bwd_workspace_bytes(int64_t m) {
  if (m < 0) return SST_ERR_ARG;
  return sst_align_up(2 * sst_div_up(m > 0 ? m : 1, (int64_t)kRowsPerWg) * 2 * kC * (int64_t)sizeof(float), 256);
}

int sst_encoder_tail_fwd_f32x6(const sst_encoder_tail_fwd_args* a, void* stream) {
  if (!a || a->m < 0 || (a->act != 1 && a->act != 2)) return SST_ERR_ARG;
  if (a->m == 0) return SST_OK;
  const void* need[] = {a->o, a->x, a->packed, a->n1w, a->n1b, a->n2w, a->n2b, a->st1, a->y1, a->pre, a->h, a->st2, a->y2};
  for (const void* p : need)
    if (!p || !aligned16(p)) return SST_ERR_ARG;
  if ((a->s1 && !aligned16(a->s1)) || (a->s2 && !aligned16(a->s2))) return SST_ERR_ARG;
  if ((a->pos_table != nullptr) != (a->pos_idx != nullptr) || (a->pos_table != nullptr) != (a->y2p != nullptr)) return SST_ERR_ARG;
  if (a->pos_table && (!aligned16(a->pos_table) || !aligned16(a->y2p))) return SST_ERR_ARG;
  tail_fwd_params P;
  P.o = a->o, P.x = a->x, P.packed = (const unsigned char*)a->packed;
  P.b_out = a->b_out, P.b1 = a->b1, P.b2 = a->b2, P.n1w = a->n1w, P.n1b = a->n1b, P.n2w = a->n2w, P.n2b = a->n2b;
  P.eps = a->eps, P.m = a->m;
  P.s1 = a->s1, P.st1 = a->st1, P.y1 = a->y1, P.pre = a->pre, P.h = a->h, P.s2 = a->s2, P.st2 = a->st2, P.y2 = a->y2;
  P.pos_table = a->pos_table, P.pos_idx = a->pos_idx, P.y2p = a->y2p;
  const int64_t rows = sst_div_up(a->m, (int64_t)kRowsPerWg);
  if (rows > 0x7fffffff) return SST_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  static unsigned long long cfg1 = 0, cfg2 = 0;
  if (a->act == 1) {
    const int rc = configure(encoder_tail_fwd_x6_k<1>, kLdsFwd, &cfg1);
    if (rc) return rc;
    hipLaunchKernelGGL(encoder_tail_fwd_x6_k<1>, dim3((unsigned)rows), dim3(kNTH), kLdsFwd, st, P);
  } else {
    const int rc = configure(encoder_tail_fwd_x6_k<2>, kLdsFwd, &cfg2);
    if (rc) return rc;
    hipLaunchKernelGGL(encoder_tail_fwd_x6_k<2>, dim3((unsigned)rows), dim3(kNTH), kLdsFwd, st, P);
  }
  SST_LAUNCH_CHECK();
  return SST_OK;
}

int sst_encoder_tail_bwd_f32x6(const sst_encoder_tail_bwd_args* a, void* stream) {
  if (!a || a->m < 0) return SST_ERR_ARG;
  if (!a->dn2w || !a->dn2b || !a->dn1w || !a->dn1b) return SST_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (a->m == 0) {
    for (float* p : {a->dn2w, a->dn2b, a->dn1w, a->dn1b}) SST_HIP(hipMemsetAsync(p, 0, sizeof(float) * kC, st));
    return SST_OK;
  }
  float *part2 = nullptr, *part1 = nullptr;
  int rows = 0;
  const int rc = sst_internal_encoder_tail_bwd_f32x6(a, &part2, &part1, &rows, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(tail_colsum_k, dim3(2 * kC / 32), dim3(1024), 0, st, part2, rows, 2 * kC, a->dn2w, a->dn2b, kC);
  hipLaunchKernelGGL(tail_colsum_k, dim3(2 * kC / 32), dim3(1024), 0, st, part1, rows, 2 * kC, a->dn1w, a->dn1b, kC);
  SST_LAUNCH_CHECK();
  return SST_OK;
}

}  // extern "C"
