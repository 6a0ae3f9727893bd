// What the matrix pipe sustains for v_mfma_f32_16x16x32_f16 on this MI355X - and what the audio sweep's CHAIN SHAPES cost
// by themselves (round 5): register-resident operands, no memory traffic.
//   NC  accumulators interleaved per wave (independent chains)
//   LEN instructions per chain before it restarts from C = 0 (0: never - one endless accumulation per accumulator)
//   FL  1: a finished chain is added to f64 running sums (the sweep's flush: 4 cvt + 4 add per chain), 0: kept alive only
// Every launch also reads the shader clock (clock64) and the 100 MHz wall clock around its loop: the clock the matrix
// pipe actually ran at, so rates are reported in cycles per instruction per SIMD as well as in TFLOP/s.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_f16_chain mfma_f16_chain.hip ; run: ./mfma_f16_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) (void)(x)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int NC, int LEN, int FL>
__global__ __launch_bounds__(256) void k_chain(float* out, long long* clk, int iters, float av, float bv) {
  h8 a[2], b[NC];                                                     // every chain its own B fragment: nothing to merge
  for (int i = 0; i < 8; ++i) {
    a[0][i] = (_Float16)(av + i); a[1][i] = (_Float16)(av - i);
    for (int c = 0; c < NC; ++c) b[c][i] = (_Float16)(bv + (float)(threadIdx.x & 15) + 0.25f * c);
  }
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 d[NC];
  double acc[NC][4];
  for (int c = 0; c < NC; ++c) { d[c] = zero; for (int r = 0; r < 4; ++r) acc[c][r] = 0.0; }
  const long long t0 = clock64(), w0 = wall_clock64();
  constexpr int L = LEN > 0 ? LEN : 8;
  f32x4 dp[NC];                                                         // the previous trip's finished chains (flushed under this trip's)
  for (int c = 0; c < NC; ++c) dp[c] = zero;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < L; ++u) {
#pragma unroll
      for (int c = 0; c < NC; ++c)
        d[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[u & 1], b[c], (LEN > 0 && u == 0) ? zero : d[c], 0, 0, 0);
    }
    if (LEN > 0) {
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        if (FL) {
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[c][r] += (double)dp[c][r];
        } else {
          asm volatile("" ::"v"(dp[c]));
        }
        dp[c] = d[c];
      }
    }
  }
  for (int c = 0; c < NC; ++c) for (int r = 0; r < 4; ++r) acc[c][r] += (double)dp[c][r];
  const long long t1 = clock64(), w1 = wall_clock64();
  double s = 0.0;
  for (int c = 0; c < NC; ++c) for (int r = 0; r < 4; ++r) s += acc[c][r] + d[c][r];
  out[blockIdx.x * 256 + threadIdx.x] = (float)s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

template <int NC, int LEN, int FL> static void bench(const char* what, int wps, float* out, long long* clk) {
  const int blocks = 256 * wps, iters = 4000;
  constexpr int L = LEN > 0 ? LEN : 8;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto launch = [&] { hipLaunchKernelGGL((k_chain<NC, LEN, FL>), dim3(blocks), dim3(256), 0, 0, out, clk, iters, 1.0f, 0.5f); };
  launch(); launch(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); for (int i = 0; i < 5; ++i) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
  long long h[2]; CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
  const double ghz = (double)h[0] / ((double)h[1] * 10.0);            // shader cycles per ns (wall clock: 100 MHz)
  const double n_per_simd = (double)wps * iters * L * NC;              // instructions through one SIMD's pipe
  const double tf = (double)blocks * 4 * iters * L * NC * 16384.0 / ms / 1e9;
  printf("%-44s waves/SIMD %d: %7.3f ms  %7.1f TFLOP/s  clock %.2f GHz  %5.1f cycles per MFMA per SIMD (%.1f at 2.4 GHz)\n", what, wps,
         ms, tf, ghz, ms * 1e6 * ghz / n_per_simd, ms * 1e6 * 2.4 / n_per_simd);
}

int main() {
  float* out; CK(hipMalloc(&out, (size_t)256 * 4 * 256 * 4));
  long long* clk; CK(hipMalloc(&clk, 16));
  for (int wps = 1; wps <= 2; ++wps) {
    bench<8, 0, 0>("8 accumulators, endless", wps, out, clk);
    bench<4, 0, 0>("4 accumulators, endless", wps, out, clk);
    bench<2, 0, 0>("2 accumulators, endless", wps, out, clk);
    bench<1, 0, 0>("1 accumulator, endless", wps, out, clk);
    bench<2, 6, 0>("2 chains of 6 from C = 0 (sweep, f32 track)", wps, out, clk);
    bench<2, 6, 1>("2 chains of 6 from C = 0 + f64 flush", wps, out, clk);
    bench<2, 4, 1>("2 chains of 4 from C = 0 + f64 flush (f16 track)", wps, out, clk);
    bench<4, 6, 1>("4 chains of 6 from C = 0 + f64 flush", wps, out, clk);
    bench<4, 4, 1>("4 chains of 4 from C = 0 + f64 flush", wps, out, clk);
    bench<2, 8, 1>("2 chains of 8 from C = 0 + f64 flush", wps, out, clk);
    bench<1, 6, 1>("1 chain of 6 from C = 0 + f64 flush", wps, out, clk);
  }
  return 0;
}
