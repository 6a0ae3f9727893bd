// Register-resident MFMA issue-rate probe: what the matrix pipe sustains on this MI355X when nothing else is in the way.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) (void)(x)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_f32_32x32x2(float* out, int iters, float a, float b) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_f32_16x16x4(float* out, int iters, float a, float b) {
  f32x4 acc[6];
  for (int i = 0; i < 6; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < 6; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 6; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_f64_16x16x4(float* out, int iters, double a, double b) {
  f64x4 acc[6];
  for (int i = 0; i < 6; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < 6; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0.0;
  for (int i = 0; i < 6; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j];
  out[blockIdx.x * 256 + threadIdx.x] = (float)s;
}

template <typename F> static double run(F launch) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); for (int i = 0; i < 5; ++i) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / 5;
}

int main(int argc, char** argv) {
  const int waves_per_simd = argc > 1 ? atoi(argv[1]) : 2;
  const int blocks = 256 * waves_per_simd, iters = 2000;
  float* out; CK(hipMalloc(&out, (size_t)blocks * 256 * 4));
  double ms = run([&] { hipLaunchKernelGGL(k_f32_32x32x2, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 1.0f); });
  printf("waves/SIMD %d  f32 32x32x2: %.3f ms  %.1f TFLOP/s\n", waves_per_simd, ms, (double)blocks * 4 * iters * 32 * 4096.0 / ms / 1e9);
  ms = run([&] { hipLaunchKernelGGL(k_f32_16x16x4, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 1.0f); });
  printf("waves/SIMD %d  f32 16x16x4: %.3f ms  %.1f TFLOP/s\n", waves_per_simd, ms, (double)blocks * 4 * iters * 48 * 2048.0 / ms / 1e9);
  ms = run([&] { hipLaunchKernelGGL(k_f64_16x16x4, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 1.0); });
  printf("waves/SIMD %d  f64 16x16x4: %.3f ms  %.1f TFLOP/s\n", waves_per_simd, ms, (double)blocks * 4 * iters * 48 * 2048.0 / ms / 1e9);
  return 0;
}
