// MFMA issue rate with RANDOM operands (mfma_peak.hip uses constant 1.0 operands): the chip clocks to its power
// budget, so what the f32 matrix pipe sustains on real data is lower than the constant-operand figure.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) (void)(x)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k16(const float* in, float* out, int iters) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  float av[8], bv[8];
  for (int i = 0; i < 8; ++i) {
    av[i] = in[(threadIdx.x * 8 + i) & 4095];
    bv[i] = in[(threadIdx.x * 8 + i + 2048) & 4095];
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[(u + i) & 7], bv[u], acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k32(const float* in, float* out, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  float av[8], bv[8];
  for (int i = 0; i < 8; ++i) {
    av[i] = in[(threadIdx.x * 8 + i) & 4095];
    bv[i] = in[(threadIdx.x * 8 + i + 2048) & 4095];
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[(u + i) & 7], bv[u], acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename F> static double run(F launch) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); for (int i = 0; i < 5; ++i) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / 5;
}
int main(int argc, char** argv) {
  const int wps = argc > 1 ? atoi(argv[1]) : 1;
  const int zero = argc > 2 ? atoi(argv[2]) : 0;
  const int blocks = 256 * wps, iters = 4000;
  float* out; CK(hipMalloc(&out, (size_t)blocks * 256 * 4));
  float h[4096];
  srand(1);
  for (int i = 0; i < 4096; ++i) h[i] = zero ? 1.0f : ((float)rand() / RAND_MAX * 2.f - 1.f) * 0.5f;
  float* in; CK(hipMalloc(&in, sizeof(h))); CK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
  double ms = run([&] { hipLaunchKernelGGL(k16<32>, dim3(blocks), dim3(256), 0, 0, in, out, iters); });
  printf("%s operands, waves/SIMD %d  f32 16x16x4 (32 acc): %.3f ms  %.1f TFLOP/s\n", zero ? "constant" : "random", wps, ms, (double)blocks * 4 * iters * 8 * 32 * 2048.0 / ms / 1e9);
  ms = run([&] { hipLaunchKernelGGL(k16<8>, dim3(blocks), dim3(256), 0, 0, in, out, iters); });
  printf("%s operands, waves/SIMD %d  f32 16x16x4 (8 acc): %.3f ms  %.1f TFLOP/s\n", zero ? "constant" : "random", wps, ms, (double)blocks * 4 * iters * 8 * 8 * 2048.0 / ms / 1e9);
  ms = run([&] { hipLaunchKernelGGL(k32, dim3(blocks), dim3(256), 0, 0, in, out, iters); });
  printf("%s operands, waves/SIMD %d  f32 32x32x2 (4 acc): %.3f ms  %.1f TFLOP/s\n", zero ? "constant" : "random", wps, ms, (double)blocks * 4 * iters * 8 * 4 * 4096.0 / ms / 1e9);
  return 0;
}
