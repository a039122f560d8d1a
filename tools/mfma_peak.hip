// Sustained fp32 MFMA rate of the device (developer probe): every wave issues back-to-back
// v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32 on independent accumulators, no memory traffic.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 4; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
static void run(const char* name, F launch, double flop_per_wave_iter, int blocks, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  launch(iters / 10);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    launch(iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = flop_per_wave_iter * iters * blocks * 4.0;
    printf("%-34s blocks %5d  %8.3f ms  %7.1f TFLOP/s\n", name, blocks, ms, flop / ms / 1e9);
  }
}

int main() {
  float* out;
  hipMalloc(&out, sizeof(float) * 256 * 4096);
  const int iters = 20000;
  for (int blocks : {256, 512, 1024}) {
    run("32x32x2 f32, 4 acc/wave", [&](int it) { hipLaunchKernelGGL(k32<4>, dim3(blocks), dim3(256), 0, 0, out, it); },
        4 * 4096.0, blocks, iters);
    run("32x32x2 f32, 8 acc/wave", [&](int it) { hipLaunchKernelGGL(k32<8>, dim3(blocks), dim3(256), 0, 0, out, it); },
        8 * 4096.0, blocks, iters);
    run("16x16x4 f32, 8 acc/wave", [&](int it) { hipLaunchKernelGGL(k16<8>, dim3(blocks), dim3(256), 0, 0, out, it); },
        8 * 2048.0, blocks, iters);
  }
  return 0;
}
