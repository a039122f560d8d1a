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
  hipLaunchKernelGGL(dynamic_voxelize_k, dim3(grid), dim3(256), 0, (hipStream_t)stream, d_points, n, row_stride, vp,
                     d_coors, coors_stride, coors_col0, batch_idx);
  SST_LAUNCH_CHECK();
  return SST_OK;
}

}  // extern "C"
