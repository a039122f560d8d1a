// Dynamic voxelization for gfx950.
//
// Reference semantics: mmdet3d/ops/voxel/src/voxelization_cuda.cu:24-65 (kernel) and :332-375 (launcher);
// CPU twin voxelization_cpu.cpp:7-41.  Per point: c = (int)floor((p - min) / vsize) in fp32 (subtract,
// then IEEE divide), clamped to [0, grid-1] (this fork clamps instead of tagging -1), written as (z,y,x).
//
// HBM-bound: 12 B read + 12 B written per point (24 B/point).  One thread per point,
// grid-stride; the x,y,z reads of a wave cover 64 consecutive rows (row_stride floats apart), writes are
// 12 B/lane contiguous across the wave.  No device synchronisation (the reference's launcher calls
// cudaDeviceSynchronize, cuda.cu:371; nothing downstream needs it).
#include <math.h>
#include "common.h"

namespace {

struct vox_params {
  float vx, vy, vz;
  float x0, y0, z0;
  int gx, gy, gz;
};

__device__ __forceinline__ int vox_coord(float p, float lo, float v, int g) {
  // fp32 subtract then correctly rounded fp32 divide: no reciprocal, no contraction.
  const float q = __fdiv_rn(__fsub_rn(p, lo), v);
  int c = (int)floorf(q);  // v_cvt_i32_f32 saturates, NaN -> 0
  c = c < 0 ? 0 : c;
  c = c >= g ? g - 1 : c;
  return c;
}

__global__ __launch_bounds__(256) void dynamic_voxelize_k(const float* __restrict__ pts, int64_t n, int64_t row_stride,
                                                          vox_params vp, int32_t* __restrict__ coors,
                                                          int64_t coors_stride, int col0, int batch_idx) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float* p = pts + i * row_stride;
    const int cx = vox_coord(p[0], vp.x0, vp.vx, vp.gx);
    const int cy = vox_coord(p[1], vp.y0, vp.vy, vp.gy);
    const int cz = vox_coord(p[2], vp.z0, vp.vz, vp.gz);
    int32_t* o = coors + i * coors_stride;
    if (batch_idx >= 0 && col0 == 1) o[0] = batch_idx;
    o[col0 + 0] = cz;
    o[col0 + 1] = cy;
    o[col0 + 2] = cx;
  }
}

// Streaming form for the usual layout (points rows of STRIDE <= 8 floats, coors rows of 4 int32 with the batch index
// in column 0, 16-byte aligned): a workgroup copies the contiguous 256 x STRIDE floats of its points into LDS with
// fully coalesced dword loads (every fetched byte of a line is used once, instead of three strided dword loads per
// thread), each thread then takes its x, y, z from LDS and writes its row as ONE 16-byte store.  Same arithmetic.
template <int STRIDE>
__global__ __launch_bounds__(256) void dynamic_voxelize_rows_k(const float* __restrict__ pts, int64_t n, vox_params vp,
                                                               int4* __restrict__ coors, int batch_idx) {
  __shared__ float tile[256 * STRIDE];
  const int64_t nblk = (n + 255) / 256;
  for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const int64_t i0 = blk * 256;
    const int64_t rows = n - i0 < 256 ? n - i0 : 256;
    const float* src = pts + i0 * STRIDE;
    const int nflt = (int)rows * STRIDE;
#pragma unroll
    for (int k = 0; k < STRIDE; ++k) {
      const int e = k * 256 + threadIdx.x;
      if (e < nflt) tile[e] = src[e];
    }
    __syncthreads();
    if ((int64_t)threadIdx.x < rows) {
      const float* p = tile + threadIdx.x * STRIDE;
      const int cx = vox_coord(p[0], vp.x0, vp.vx, vp.gx);
      const int cy = vox_coord(p[1], vp.y0, vp.vy, vp.gy);
      const int cz = vox_coord(p[2], vp.z0, vp.vz, vp.gz);
      coors[i0 + threadIdx.x] = make_int4(batch_idx, cz, cy, cx);
    }
    __syncthreads();
  }
}

}  // namespace

extern "C" {

const char* sst_version(void) { return "sst_amd 0.1.0 gfx950"; }

void sst_dynamic_voxelize_grid(const float voxel_size[3], const float coors_range[6], int32_t grid_xyz[3]) {
  for (int d = 0; d < 3; ++d) {
    // fp32 arithmetic exactly as voxelization_cuda.cu:355-357: ceil((max - min) / voxel)
    volatile float span = coors_range[d + 3] - coors_range[d];
    volatile float q = span / voxel_size[d];
    grid_xyz[d] = (int32_t)ceilf(q);
  }
}

int sst_dynamic_voxelize_f32(const float* d_points, int64_t n, int64_t row_stride, const float voxel_size[3],
                             const float coors_range[6], int32_t* d_coors, int64_t coors_stride, int coors_col0,
                             int batch_idx, void* stream) {
  if (n < 0 || !voxel_size || !coors_range) return SST_ERR_ARG;
  if (n == 0) return SST_OK;
  if (!d_points || !d_coors || row_stride < 3 || coors_col0 < 0 || coors_stride < coors_col0 + 3) return SST_ERR_ARG;
  vox_params vp;
  vp.vx = voxel_size[0];
  vp.vy = voxel_size[1];
  vp.vz = voxel_size[2];
  vp.x0 = coors_range[0];
  vp.y0 = coors_range[1];
  vp.z0 = coors_range[2];
  int32_t g[3];
  sst_dynamic_voxelize_grid(voxel_size, coors_range, g);
  vp.gx = g[0];
  vp.gy = g[1];
  vp.gz = g[2];
  const int grid = sst_grid_1d(n, 256);
  const bool rows16 = coors_stride == 4 && coors_col0 == 1 && batch_idx >= 0 && (((uintptr_t)d_coors) & 15) == 0 &&
                      row_stride >= 3 && row_stride <= 8 && n >= 4096;
  hipStream_t st = (hipStream_t)stream;
#define SST_VOX_ROWS(S)                                                                                         \
  case S:                                                                                                       \
    hipLaunchKernelGGL(dynamic_voxelize_rows_k<S>, dim3(grid), dim3(256), 0, st, d_points, n, vp, (int4*)d_coors, \
                       batch_idx);                                                                              \
    break;
  if (rows16) {
    switch ((int)row_stride) {
      SST_VOX_ROWS(3)
      SST_VOX_ROWS(4)
      SST_VOX_ROWS(5)
      SST_VOX_ROWS(6)
      SST_VOX_ROWS(7)
      SST_VOX_ROWS(8)
    }
  } else {
    hipLaunchKernelGGL(dynamic_voxelize_k, dim3(grid), dim3(256), 0, st, d_points, n, row_stride, vp, d_coors,
                       coors_stride, coors_col0, batch_idx);
  }
#undef SST_VOX_ROWS
  SST_LAUNCH_CHECK();
  return SST_OK;
}

}  // extern "C"
