// Shared helpers for the gfx950 kernels of libsst_amd.  Wavefront = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/sst_amd.h"

#define SST_WAVE 64

#define SST_LAUNCH_CHECK()                          \
  do {                                              \
    hipError_t e__ = hipGetLastError();             \
    if (e__ != hipSuccess) return (int)e__;         \
  } while (0)

// hipFuncSetAttribute (and anything else that is per DEVICE) done once per device at a call site:
//   static unsigned long long done = 0;  if (sst_first_use_on_device(&done)) { ...; sst_mark_device(&done); }
// bit d of the mask = device d configured (a process-wide `static bool` left a second GPU of the process
// without the attribute).  Two threads racing on the first use both set the attribute: harmless.
static inline unsigned long long sst_device_bit(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  return 1ull << (dev & 63);
}
static inline bool sst_first_use_on_device(unsigned long long* mask) {
  return (__atomic_load_n(mask, __ATOMIC_ACQUIRE) & sst_device_bit()) == 0;
}
static inline void sst_mark_device(unsigned long long* mask) { __atomic_fetch_or(mask, sst_device_bit(), __ATOMIC_RELEASE); }

#define SST_HIP(call)                               \
  do {                                              \
    hipError_t e__ = (call);                        \
    if (e__ != hipSuccess) return (int)e__;         \
  } while (0)

static inline int64_t sst_div_up(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t sst_align_up(int64_t a, int64_t b) { return sst_div_up(a, b) * b; }

// Memory-bound 1-D launches: cap the grid at 256 CUs x 8 blocks and grid-stride the rest.
static inline int sst_grid_1d(int64_t work_items, int block) {
  int64_t g = sst_div_up(work_items, block);
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (int)g;
}

__device__ __forceinline__ int sst_lane() { return (int)(threadIdx.x & 63); }

// Inclusive scan across the 64 lanes of a wave.
__device__ __forceinline__ int sst_wave_incl_scan(int v) {
  const int lane = sst_lane();
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int t = __shfl_up(v, d, 64);
    if (lane >= d) v += t;
  }
  return v;
}

// Cross-file internals of the one-call layer executor (csrc/layer_exec.hip): the LayerNorm backward without its finishing launch
// (csrc/dense.hip) and the weight-gradient group whose reduction launch also sums those partials' columns (csrc/wgrad_x6.hip).
struct sst_colsum_rider {      // column sums of block partials [nb][width] -> out0 (columns < split) | out1
  const float* partials;
  int nb, width, split;
  float* out0;
  float* out1;
};
int sst_internal_add_layernorm_bwd2_partials_f32(const float* d_dy, const float* d_dy2, const float* d_sum, const float* d_stats,
                                                 const float* d_weight, int64_t m, int c, float* d_dx, void* d_workspace,
                                                 int* partial_rows, void* stream);
int sst_internal_add_layernorm_bwd_bf16_partials(const void* d_dy, const void* d_dy2, const void* d_sum, const float* d_stats,
                                                 const float* d_weight, int64_t m, int c, void* d_dx, void* d_workspace,
                                                 int* partial_rows, void* stream);
int sst_internal_wgrad_group_bf16(const sst_wgrad_problem_bf16* problems, int n, void* d_workspace, const sst_colsum_rider* riders,
                                  int n_riders, void* stream);
// csrc/layer_tail_x6.hip without the finishing launches of the LayerNorm parameter gradients: their per-workgroup partials stay
// in args->workspace as [*partial_rows][256] at *part2 (norm2) and *part1 (norm1); the dn* fields of args are not read.
int sst_internal_encoder_tail_bwd_f32x6(const sst_encoder_tail_bwd_args* args, float** part2, float** part1, int* partial_rows,
                                        void* stream);
int sst_internal_encoder_tail_bwd_bf16(const sst_encoder_tail_bwd_bf16_args* args, float** part2, float** part1, int* partial_rows,
                                       void* stream);
int sst_internal_weight_grad_group_f32x6(const sst_wgrad_problem_f32* problems, int n, void* d_workspace,
                                         const sst_colsum_rider* riders, int n_riders, void* stream);

// Workspace carving on the host side (256-byte aligned slices of one caller-owned buffer).
struct sst_carver {
  char* base;
  int64_t off;
  explicit sst_carver(void* p) : base((char*)p), off(0) {}
  template <typename T>
  T* take(int64_t count) {
    T* r = (T*)(base + off);
    off += sst_align_up((int64_t)sizeof(T) * (count > 0 ? count : 1), 256);
    return r;
  }
};
