// Issue rate of the packed-f32 VALU operations the text sweep is made of (qpg_text.hip, text_tile_dists):
//   (a) v_pk_add_f32 with both sources in VGPRs, (b) v_pk_add_f32 with one source an SGPR pair (how the sweep feeds
//   the wave-uniform query pair), (c) v_pk_mul_f32, (d) the sweep's own 3-op step (sgpr-sub, mul, add).
// hipcc -O3 --offload-arch=gfx950 pk_rate.hip -o pk_rate ; prints cycles per wave instruction per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define N_ACC 12
template <int WHICH>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed, f32x2 sq) {
  f32x2 a[N_ACC], x[N_ACC];
  for (int i = 0; i < N_ACC; ++i) { a[i] = f32x2{seed + i, seed - i}; x[i] = f32x2{0.5f * threadIdx.x, 0.25f * i}; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < N_ACC; ++i) {
      if (WHICH == 0) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(x[i]));
      if (WHICH == 1) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(a[i]) : "s"(sq));
      if (WHICH == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(x[i]));
      if (WHICH == 3) {
        f32x2 d;
        asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "s"(sq), "v"(x[i]));
        asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(d));
        asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(d));
      }
      if (WHICH == 4) {
        f32x2 d;
        asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a[(i + 1) % N_ACC]), "v"(x[i]));
        asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(d));
        asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(d));
      }
    }
  }
  f32x2 t{0.f, 0.f};
  for (int i = 0; i < N_ACC; ++i) t += a[i];
  out[blockIdx.x * 256 + threadIdx.x] = t.x + t.y;
}
int main() {
  const int iters = 4000;
  float* out; (void)hipMalloc(&out, (size_t)4096 * 256 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
  const double clk = p.clockRate * 1e3;
  const char* names[5] = {"v_pk_add_f32 v,v", "v_pk_add_f32 s,v", "v_pk_mul_f32 v,v", "sweep step (s-sub, mul, add)", "same step, all VGPR"};
  for (int wps = 1; wps <= 4; wps *= 2) {              // waves per SIMD
    const int blocks = p.multiProcessorCount * wps;    // 256 threads = 4 waves = one per SIMD
    for (int which = 0; which < 5; ++which) {
      float ms = 0;
      for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        const f32x2 sq{1.5f, 2.5f};
        if (which == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, sq);
        if (which == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, sq);
        if (which == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, sq);
        if (which == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, sq);
        if (which == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, sq);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
      }
      const double n_instr = (double)iters * N_ACC * (which >= 3 ? 3 : 1) * wps;   // wave instructions per SIMD
      printf("%d wave(s)/SIMD  %-30s %.2f cycles per wave instruction\n", wps, names[which], ms * 1e-3 * clk / n_instr);
    }
  }
  return 0;
}
