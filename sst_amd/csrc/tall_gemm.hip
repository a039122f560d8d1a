// Tall fp32 linear layer on gfx950:  Y[M, 128] (+)= X[M, K] W^T + bias,  M ~ 1e5 tokens, K in {128, 256}: the
// projections and the FFN of an SRA encoder layer (mmdet3d/models/sst/sst_basic_block_v2.py:41-75, :104-126) and
// their data gradients (dX = dY W, the same product with the transposed weight).
//
// Why not the library: hipBLASLt's best solutions for these shapes reach 40 % (N = 128) to 56 % (N = 256) of the
// fp32 MFMA rate (tools/microbench.py gemm, profiles/): every workgroup re-stages W and the tiles are sized for
// square problems.  Here W (<= 128 KB) is staged ONCE per workgroup into LDS and stays there while the workgroup's
// waves stream 32-row tiles of X (persistent grid, one workgroup per CU, tiles dealt round-robin so every CU gets
// the same number +-1 whatever M is).
//
// Data path of X (the part that decides the speed):
//   * global -> VGPR in FULL 128-byte lines: one global_load_dwordx4 = 8 rows x 128 B (lanes 8r..8r+7 read row r).
//     The first version loaded MFMA-fragment-shaped pieces (32 rows x 32 B per instruction) straight into the
//     A-operand layout; PMC showed the L1/TA busy 83 % of the time and the waves 78 % of their life in
//     s_waitcnt (59 % MFMA utilisation): 64 partial-line requests per instruction saturate the texture path.
//   * VGPR -> LDS with ds_write_b128 into a per-wave [32 rows][8 x 16 B] image whose 16-byte slots are XOR-swizzled
//     (slot = row * 8 + (chunk ^ (row & 7))): both the line-shaped writes and the fragment-shaped reads are
//     bank-conflict free without padding.  No barrier: the image is private to the wave, LDS ops are in order.
//   * LDS -> A operand of v_mfma_f32_32x32x2_f32 with one ds_read_b128 per 4 MFMA steps: lane (i = l & 31,
//     h = l >> 5) reads X[row i][8j + 4h ..+3]; step t contracts k in {8j + t, 8j + 4 + t}, and the matching B operand
//     W[n][8j + 4h + t] is one ds_read_b128 of the weight image (same XOR swizzle, [N][K] row-major).
// A wave owns 32 rows x 128 columns (4 accumulator tiles).  The 16 line loads of a chunk (32 rows x 128 k) form a
// ring in time: as soon as a register group has been written to LDS it is refilled with the next chunk's data, so
// the prefetch distance is a whole chunk (~16 k MFMA cycles).  Loads, LDS traffic, waits and MFMAs are inline asm
// (volatile asm keeps program order; the compiler's own scheduler sinks prefetches next to their first use and its
// waitcnt insertion then waits where the load is issued — see wgrad.hip).  Every wait names the registers it
// guards as in/out operands, so consumers are ordered after it by data dependence and the registers cannot be
// copied / reused while a load is still in flight; tools/check_async_regs.py + tests/test_gpu_dense.py (stress
// test) guard the scheme.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kTgWaves = 8;     // per workgroup: 2 per SIMD, each with a 256-VGPR budget
constexpr int kTgGrid = 256;    // persistent: one workgroup per CU
constexpr int kTgImage = 4096;  // bytes of a wave's X image in LDS: 32 rows x 32 k

#define SST_TG_LOAD(DST, PTR, OFF) \
  asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(DST) : "v"(PTR), "n"(OFF))
#define SST_TG_MFMA(ACC, AV, BV) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(ACC) : "v"(AV), "v"(BV))
#define SST_TG_DSR(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF))
#define SST_TG_DSW(ADDR, SRC, OFF) \
  asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(ADDR), "v"(SRC), "n"(OFF) : "memory")

// EPI 0: Y = acc + bias (bias may be null).  EPI 1: Y = Y_old + acc (+ bias): accumulate into the destination.
// EPI 2: Y = acc + bias and AUX = gelu(Y) (erf form): linear1 + activation of the FFN in one pass (sst_basic_block_v2.py:116),
//        both tensors are kept for the backward pass.  EPI 3: Y = (acc + bias) * gelu'(AUX): the data gradient through
//        linear2 with the activation's derivative applied to the pre-activation AUX in the epilogue.
// TRANS_W 0: W is [N][K] row-major (forward: y = x W^T).  TRANS_W 1: W is [K][N] row-major (data gradient:
// dx = dy W with the layer's [out = K][in = N] weight) and is transposed while it is staged.
template <int K, int EPI, int TRANS_W>
__global__ __launch_bounds__(64 * kTgWaves) void tall_gemm_n128_k(const float* __restrict__ X, int64_t ldx,
                                                                   const float* __restrict__ W, int64_t ldw,
                                                                   const float* __restrict__ bias, int64_t m,
                                                                   float* __restrict__ Y, int64_t ldy,
                                                                   float* __restrict__ AUX) {
  constexpr int N = 128;
  constexpr int KH = K / 128;  // chunks per tile
  extern __shared__ __attribute__((aligned(16))) float Ws[];  // [N][K] swizzled, then kTgWaves X images
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int i = lane & 31, h = lane >> 5;
  const int n_tiles = (int)((m + 31) / 32);  // host side guarantees m < 2^31 - 32
  // tiles of this wave: t = blockIdx.x + gridDim.x * (wave + kTgWaves * q), q = 0, 1, ...
  const int t0 = blockIdx.x + gridDim.x * wave;
  const int tstep = gridDim.x * kTgWaves;
  int n_my = t0 < n_tiles ? (n_tiles - 1 - t0) / tstep + 1 : 0;
  n_my = __builtin_amdgcn_readfirstlane(n_my);
  const int n_chunks = n_my * KH;
  // line-shaped loads: lane L reads 16 B at (row 8 * q + L / 8, k = 4 * (L % 8)) of piece (chunk c, k-piece P)
  const int lr = lane >> 3, lc = lane & 7;
  auto chunk_ptr = [&](int c, int q) -> const float* {  // rows 8q..8q+7 of chunk c; past the end: the last chunk
    if (c >= n_chunks) c = n_chunks - 1;
    const int tile = t0 + tstep * (c / KH);
    int64_t arow = (int64_t)tile * 32 + 8 * q + lr;
    if (arow >= m) arow = m - 1;  // clamped: loads are unconditional, stores are guarded
    return X + arow * ldx + (c % KH) * 128 + 4 * lc;
  };
  f32x4 a[16];  // a[4 * P + q]: k-piece P (32 k), rows 8q..8q+7
  if (n_chunks > 0) {  // chunk 0 in flight while W is staged
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float* p0 = chunk_ptr(0, q);
#pragma unroll
      for (int P = 0; P < 4; ++P) SST_TG_LOAD(a[4 * P + q], p0, 128 * P);
    }
  }

  {  // stage W: all loads of a thread are issued before the first LDS write (one memory round trip, not PER)
    constexpr int PER = N * (K / 4) / (64 * kTgWaves);
    static_assert(PER * 64 * kTgWaves == N * (K / 4), "staging loop must be exact");
    float4 w[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int e = tid + q * 64 * kTgWaves;
      if (TRANS_W == 0) {
        const int n = e / (K / 4), k4 = e - n * (K / 4);
        w[q] = *(const float4*)(W + (int64_t)n * ldw + 4 * k4);
      } else {
        const int k = e / (N / 4), n4 = e - k * (N / 4);
        w[q] = *(const float4*)(W + (int64_t)k * ldw + 4 * n4);
      }
    }
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int e = tid + q * 64 * kTgWaves;
      if (TRANS_W == 0) {
        const int n = e / (K / 4), k4 = e - n * (K / 4);
        *(float4*)(Ws + n * K + 4 * (k4 ^ (n & 7))) = w[q];
      } else {
        const int k = e / (N / 4), n4 = e - k * (N / 4);
        const float wv[4] = {w[q].x, w[q].y, w[q].z, w[q].w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int n = 4 * n4 + u;
          Ws[n * K + 4 * ((k >> 2) ^ (n & 7)) + (k & 3)] = wv[u];
        }
      }
    }
  }
  __syncthreads();
  if (n_chunks == 0) return;
  float bv[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) bv[nt] = bias ? bias[32 * nt + i] : 0.f;
  typedef __attribute__((address_space(3))) float lds_float;
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_float*)Ws;
  const unsigned img = lds0 + (unsigned)(N * K * sizeof(float)) + (unsigned)(wave * kTgImage);
  const unsigned lw = img + (unsigned)(lr * 128 + 16 * (lc ^ lr));  // + 1024 * q: line-shaped write of rows 8q + lr
  unsigned sw[4];                                                   // 16 * ((2g + h) ^ (i & 7)): fragment-shaped reads
#pragma unroll
  for (int g = 0; g < 4; ++g) sw[g] = (unsigned)(16 * ((2 * g + h) ^ (i & 7)));
  const unsigned rd = img + (unsigned)(i * 128);
  unsigned wrow[4];  // this lane's row of W tile nt (+ the k-half of the current chunk)
  f32x16 acc[4];

  auto zero = [&]() {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
  };
  auto store = [&](int c) {
    // the last MFMAs (16 passes, opaque to the hazard recognizer) must have landed before acc is read
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
    const int64_t row0 = (int64_t)(t0 + tstep * (c / KH)) * 32;
    // D layout: lane (col = i, h), reg r -> row (r & 3) + 8 * (r >> 2) + 4 * h
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t row = row0 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (row < m) {
        // 32-bit byte offset from the (uniform, SGPR) base: one VGPR per row instead of a 64-bit pointer
        const unsigned yoff = (unsigned)((row * ldy + i) * (int64_t)sizeof(float));
        float v[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          v[nt] = acc[nt][r] + bv[nt];
          if (EPI == 1) v[nt] += *(const float*)((const char*)Y + yoff + 128 * nt);
          if (EPI == 3) {
            const float p = *(const float*)((const char*)AUX + yoff + 128 * nt);
            // d gelu(p) / dp = Phi(p) + p * phi(p)
            v[nt] *= 0.5f * (1.f + erff(p * 0.70710678118654752f)) + p * 0.39894228040143268f * __expf(-0.5f * p * p);
          }
        }
        // asm: EXACTLY 64 store instructions per full tile (the wait immediates below count them; EPI 2 issues 128,
        // which only makes vmcnt(63) wait for more of the - older - stores, never for less of the loads it guards)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
          asm volatile("global_store_dword %0, %1, %2 offset:%3" : : "v"(yoff), "v"(v[nt]), "s"(Y), "n"(128 * nt) : "memory");
        if (EPI == 2) {
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) {
            const float hgelu = 0.5f * v[nt] * (1.f + erff(v[nt] * 0.70710678118654752f));
            asm volatile("global_store_dword %0, %1, %2 offset:%3" : : "v"(yoff), "v"(hgelu), "s"(AUX), "n"(128 * nt) : "memory");
          }
        }
      }
    }
  };
  // ---- schedule -------------------------------------------------------------------------------------------
  // A chunk is 16 groups (4 k-pieces x 4 groups of 8 k); a group = 5 ds_read_b128 (A fragment + 4 B fragments)
  // + 16 MFMAs.  Software pipeline inside the wave: the operands of group G + 1 are requested before the MFMAs of
  // group G are issued (two operand buffers), so a wave never sits in an LDS wait with an empty MFMA queue - with
  // the simple read/wait/multiply order both waves of a SIMD fall into lock step and wait at the same time
  // (measured 73 % MFMA rate without the store path).  Before the first group of a piece can be requested the
  // piece must be in the image: PREP waits for its 4 line loads, writes them to LDS and immediately refills the
  // 4 register groups with the same piece of the NEXT chunk (ring distance = one chunk).
  //
  // PREP's wait immediate = number of VMEM ops younger than the 4 loads it needs: the 12 other loads of the ring,
  // plus the 64 stores of a tile whenever a store burst lies between their issue and now (12 + 64 > the 6-bit
  // maximum: 63 is used, weaker but sufficient - and it never stalls on the stores, which are ~a piece old by
  // then).  Which positions see a store burst is worked out per K below; the first tile is peeled (no stores yet).
  // No branch separates a refill from the wait that guards it (at a control-flow merge the register allocator
  // may copy or spill registers whose loads are still in flight).
  const float* pn[4];
  unsigned wrow_n[4];
  f32x4 af[2], b[2][4];
#define SST_TG_PTRS(C) _Pragma("unroll") for (int q = 0; q < 4; ++q) pn[q] = chunk_ptr((C), q);
#define SST_TG_WROW(DST, C)                        \
  _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) \
      DST[nt] = lds0 + (unsigned)(((32 * nt + i) * K + ((C) % KH) * 128) * sizeof(float));
#define SST_TG_PREP(WAITSTR, P)                                                                                 \
  asm volatile(WAITSTR : "+v"(a[4 * (P)]), "+v"(a[4 * (P) + 1]), "+v"(a[4 * (P) + 2]), "+v"(a[4 * (P) + 3]));   \
  _Pragma("unroll") for (int q = 0; q < 4; ++q) SST_TG_DSW(lw, a[4 * (P) + q], 1024 * q);                       \
  _Pragma("unroll") for (int q = 0; q < 4; ++q) SST_TG_LOAD(a[4 * (P) + q], pn[q], 128 * (P)); /* refill */
#define SST_TG_RD(BUF, WR, P, G)                     \
  {                                                  \
    const unsigned ra = rd + sw[G];                  \
    SST_TG_DSR(af[BUF], ra, 0);                      \
    _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) { \
      const unsigned rb = WR[nt] + sw[G];            \
      SST_TG_DSR(b[BUF][nt], rb, 128 * (P));         \
    }                                                \
  }
#define SST_TG_MM(LGKM, BUF)                                                                                   \
  asm volatile(LGKM : "+v"(af[BUF]), "+v"(b[BUF][0]), "+v"(b[BUF][1]), "+v"(b[BUF][2]), "+v"(b[BUF][3]));     \
  _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) SST_TG_MFMA(acc[nt], af[BUF].x, b[BUF][nt].x);             \
  _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) SST_TG_MFMA(acc[nt], af[BUF].y, b[BUF][nt].y);             \
  _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) SST_TG_MFMA(acc[nt], af[BUF].z, b[BUF][nt].z);             \
  _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) SST_TG_MFMA(acc[nt], af[BUF].w, b[BUF][nt].w);
  // groups 0..2 of piece P (operands of the next group requested first; 5 younger LDS ops)
#define SST_TG_G012(P)                                                       \
  SST_TG_RD(1, wrow, P, 1) SST_TG_MM("s_waitcnt lgkmcnt(5)", 0)              \
  SST_TG_RD(0, wrow, P, 2) SST_TG_MM("s_waitcnt lgkmcnt(5)", 1)              \
  SST_TG_RD(1, wrow, P, 3) SST_TG_MM("s_waitcnt lgkmcnt(5)", 0)
  // group 3 of piece P < 3: bring piece P + 1 into the image first (4 writes + 5 reads younger)
#define SST_TG_G3(WAITSTR, P) \
  SST_TG_PREP(WAITSTR, (P) + 1) SST_TG_RD(0, wrow, (P) + 1, 0) SST_TG_MM("s_waitcnt lgkmcnt(9)", 1)
  // group 3 of piece 3: the next chunk C1 takes over (its pointers for the refills, its W rows for the reads)
#define SST_TG_G3_NEXT(WAITSTR, C1)                                                                   \
  SST_TG_PTRS((C1) + 1) SST_TG_WROW(wrow_n, C1) SST_TG_PREP(WAITSTR, 0) SST_TG_RD(0, wrow_n, 0, 0)    \
  SST_TG_MM("s_waitcnt lgkmcnt(9)", 1)                                                                \
  _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) wrow[nt] = wrow_n[nt];
  // one chunk: W123 guards pieces 1..3 of this chunk, WN piece 0 of the next chunk C1
#define SST_TG_CHUNK(W123, WN, C1)                                                                   \
  SST_TG_G012(0) SST_TG_G3(W123, 0) SST_TG_G012(1) SST_TG_G3(W123, 1) SST_TG_G012(2) SST_TG_G3(W123, 2) \
  SST_TG_G012(3) SST_TG_G3_NEXT(WN, C1)

  // prologue: piece 0 of chunk 0 into the image, operands of its first group requested
  SST_TG_PTRS(1)
  SST_TG_WROW(wrow, 0)
  SST_TG_PREP("s_waitcnt vmcnt(12)", 0)
  SST_TG_RD(0, wrow, 0, 0)
  {  // tile 0: no stores in flight yet
    zero();
    if (KH == 1) {
      SST_TG_CHUNK("s_waitcnt vmcnt(12)", "s_waitcnt vmcnt(12)", 1)
    } else {
      SST_TG_CHUNK("s_waitcnt vmcnt(12)", "s_waitcnt vmcnt(12)", 1)
      SST_TG_CHUNK("s_waitcnt vmcnt(12)", "s_waitcnt vmcnt(12)", 2)
    }
    store(KH - 1);
  }
  for (int c = KH; c < n_chunks; c += KH) {
    zero();
    if (KH == 1) {
      // every load this tile waits for was issued before the previous tile's store burst
      SST_TG_CHUNK("s_waitcnt vmcnt(63)", "s_waitcnt vmcnt(63)", c + 1)
    } else {
      // first chunk of the tile: its pieces 1..3 and the second chunk's piece 0 were requested before the burst;
      // second chunk: everything it waits for was requested after it
      SST_TG_CHUNK("s_waitcnt vmcnt(63)", "s_waitcnt vmcnt(63)", c + 1)
      SST_TG_CHUNK("s_waitcnt vmcnt(12)", "s_waitcnt vmcnt(12)", c + 2)
    }
    store(c + KH - 1);
  }
#undef SST_TG_CHUNK
#undef SST_TG_G3_NEXT
#undef SST_TG_G3
#undef SST_TG_G012
#undef SST_TG_MM
#undef SST_TG_RD
#undef SST_TG_PREP
#undef SST_TG_WROW
#undef SST_TG_PTRS
  // operands requested for the (non-existent) group after the last one
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0]), "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[0][2]), "+v"(b[0][3]));
  // the redundant last refill must land before its registers are reused
  asm volatile("s_waitcnt vmcnt(0)"
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                 "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]),
                 "+v"(a[15]));
}

template <int K, int EPI, int TRANS_W>
int launch_n128(const float* X, int64_t ldx, const float* W, int64_t ldw, const float* bias, int64_t m, float* Y,
                int64_t ldy, hipStream_t st, float* aux = nullptr) {
  const size_t lds = (size_t)128 * K * sizeof(float) + (size_t)kTgWaves * kTgImage;
  static unsigned long long configured = 0;  // per instantiation; the attribute call costs tens of microseconds on the host
  if (sst_first_use_on_device(&configured)) {
    SST_HIP(hipFuncSetAttribute((const void*)tall_gemm_n128_k<K, EPI, TRANS_W>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    sst_mark_device(&configured);
  }
  const int64_t n_tiles = sst_div_up(m, 32);
  const int grid = (int)(n_tiles < kTgGrid ? n_tiles : kTgGrid);
  hipLaunchKernelGGL((tall_gemm_n128_k<K, EPI, TRANS_W>), dim3(grid), dim3(64 * kTgWaves), lds, st, X, ldx, W, ldw,
                     bias, m, Y, ldy, aux);
  SST_LAUNCH_CHECK();
  return SST_OK;
}

}  // namespace

extern "C" {

int sst_tall_linear_f32(const float* d_x, int64_t ldx, const float* d_w, int64_t ldw, const float* d_bias, int64_t m,
                        int n, int k, int trans_w, int accumulate, float* d_y, int64_t ldy, void* stream) {
  if (m < 0 || n != 128 || (k != 128 && k != 256)) return SST_ERR_UNSUPPORTED;
  if ((m + 32) * ldy >= ((int64_t)1 << 30) || (m + 32) * ldx >= ((int64_t)1 << 40)) return SST_ERR_UNSUPPORTED;  // 32-bit store offsets
  if (m == 0) return SST_OK;
  if (!d_x || !d_w || !d_y || ldx < k || ldy < n || (ldx & 3) || (ldw & 3) || ((uintptr_t)d_x & 15) ||
      ((uintptr_t)d_w & 15))
    return SST_ERR_ARG;
  if (ldw < (trans_w ? n : k)) return SST_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
#define SST_TG(KK, E, T) return launch_n128<KK, E, T>(d_x, ldx, d_w, ldw, d_bias, m, d_y, ldy, st)
  if (k == 128) {
    if (!accumulate && !trans_w) SST_TG(128, 0, 0);
    if (!accumulate && trans_w) SST_TG(128, 0, 1);
    if (accumulate && !trans_w) SST_TG(128, 1, 0);
    SST_TG(128, 1, 1);
  }
  if (!accumulate && !trans_w) SST_TG(256, 0, 0);
  if (!accumulate && trans_w) SST_TG(256, 0, 1);
  if (accumulate && !trans_w) SST_TG(256, 1, 0);
  SST_TG(256, 1, 1);
#undef SST_TG
}

int sst_tall_linear_gelu_f32(const float* d_x, int64_t ldx, const float* d_w, int64_t ldw, const float* d_bias,
                             int64_t m, int n, int k, int trans_w, int mode, float* d_aux, float* d_y, int64_t ldy,
                             void* stream) {
  // mode 0: y = x W^T + b, aux = gelu(y) (aux has the row stride of y);  mode 1: y = (x W) * gelu'(aux)
  if (m < 0 || n != 128 || k != 128 || (mode != 0 && mode != 1)) return SST_ERR_UNSUPPORTED;
  if ((m + 32) * ldy >= ((int64_t)1 << 30) || (m + 32) * ldx >= ((int64_t)1 << 40)) return SST_ERR_UNSUPPORTED;
  if (m == 0) return SST_OK;
  if (!d_x || !d_w || !d_y || !d_aux || ldx < k || ldy < n || (ldx & 3) || (ldw & 3) || ((uintptr_t)d_x & 15) ||
      ((uintptr_t)d_w & 15))
    return SST_ERR_ARG;
  if (ldw < (trans_w ? n : k)) return SST_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (mode == 0) {
    if (trans_w) return launch_n128<128, 2, 1>(d_x, ldx, d_w, ldw, d_bias, m, d_y, ldy, st, d_aux);
    return launch_n128<128, 2, 0>(d_x, ldx, d_w, ldw, d_bias, m, d_y, ldy, st, d_aux);
  }
  if (trans_w) return launch_n128<128, 3, 1>(d_x, ldx, d_w, ldw, d_bias, m, d_y, ldy, st, d_aux);
  return launch_n128<128, 3, 0>(d_x, ldx, d_w, ldw, d_bias, m, d_y, ldy, st, d_aux);
}

}  // extern "C"
