// v_cvt_f64_f32 issue rate probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) (void)(x)
__global__ __launch_bounds__(256) void k_cvt(double* out, int iters, float seed) {
  float a[8]; double s[8];
  for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; s[i] = 0.0; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      double d;
      asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d) : "v"(a[i]));
      asm volatile("" : "+v"(d));
      s[i] = d;
    }
  }
  double t = 0; for (int i = 0; i < 8; ++i) t += s[i];
  out[blockIdx.x * 256 + threadIdx.x] = t;
}
__global__ __launch_bounds__(256) void k_add32(double* out, int iters, float seed) {
  float a[8];
  for (int i = 0; i < 8; ++i) a[i] = seed + i + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
  }
  float t = 0; for (int i = 0; i < 8; ++i) t += a[i];
  out[blockIdx.x * 256 + threadIdx.x] = t;
}
__global__ __launch_bounds__(256) void k_add64(double* out, int iters, float seed) {
  double a[8];
  for (int i = 0; i < 8; ++i) a[i] = seed + i + threadIdx.x;
  const double sd = seed;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(sd));
  }
  double t = 0; for (int i = 0; i < 8; ++i) t += a[i];
  out[blockIdx.x * 256 + threadIdx.x] = t;
}
__global__ __launch_bounds__(256) void k_pkadd32(double* out, int iters, float seed) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 a[8];
  for (int i = 0; i < 8; ++i) a[i] = (f2){seed + i + threadIdx.x, seed};
  const f2 sd = {seed, seed};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(sd));
  }
  float t = 0; for (int i = 0; i < 8; ++i) t += a[i].x + a[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = t;
}
int main() {
  const int blocks = 256 * 2, iters = 20000;
  double* out; CK(hipMalloc(&out, (size_t)blocks * 256 * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int which = 0; which < 4; ++which) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      if (which == 0) hipLaunchKernelGGL(k_cvt, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
      else if (which == 1) hipLaunchKernelGGL(k_add32, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
      else if (which == 2) hipLaunchKernelGGL(k_add64, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
      else hipLaunchKernelGGL(k_pkadd32, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    // per SIMD: 2 waves x iters x 8 ops
    const double ops = 2.0 * iters * 8;
    printf("%s: %.3f ms -> %.1f cycles per wave-instruction at 2.4 GHz\n", which == 0 ? "v_cvt_f64_f32" : which == 1 ? "v_add_f32" : which == 2 ? "v_add_f64" : "v_pk_add_f32", ms, ms * 1e-3 * 2.4e9 / ops);
  }
  return 0;
}
