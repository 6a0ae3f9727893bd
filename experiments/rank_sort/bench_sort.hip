// Cycle count of block_sorted_ranks (csrc/qpg_common.h) for one block: hipcc --offload-arch=gfx950 -O3 -I. bench_sort.hip
#include "../../qpgesture_amd/csrc/qpg_common.h"
#include <vector>
#include <algorithm>
#include <cstdlib>
void qpg_set_error(const char*, ...) {}
template <typename T>
__global__ void k(const T* d, int K, short* out, long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned long long* skey = reinterpret_cast<unsigned long long*>(smem);
  int* scode = reinterpret_cast<int*>(skey + rank_sort_pow2(K));
  T* v = reinterpret_cast<T*>(scode + rank_sort_pow2(K));
  for (int i = threadIdx.x; i < K; i += blockDim.x) v[i] = d[i];
  __syncthreads();
  const long long t0 = clock64();
  block_sorted_ranks(v, K, skey, scode, [&](int kk, int r) { out[kk] = (short)r; });
  __syncthreads();
  const long long t1 = clock64();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  for (int K : {512, 500, 128, 2048})
    for (int threads : {256, 1024}) {
      std::vector<double> h(K);
      for (auto& x : h) x = (double)(rand() % 300);          // many exact ties
      double* d; short* o; long long* c;
      hipMalloc(&d, K * 8); hipMalloc(&o, K * 2); hipMalloc(&c, 8 * 64);
      hipMemcpy(d, h.data(), K * 8, hipMemcpyHostToDevice);
      const size_t sh = 12 * (size_t)rank_sort_pow2(K) + 8 * (size_t)K;
      for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k<double>, dim3(1), dim3(threads), sh, 0, d, K, o, c);
      hipDeviceSynchronize();
      std::vector<short> r(K); long long cy;
      hipMemcpy(r.data(), o, K * 2, hipMemcpyDeviceToHost); hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
      int bad = 0;
      for (int i = 0; i < K; ++i) {
        int want = 0;
        for (int j = 0; j < K; ++j) want += h[j] < h[i] || (h[j] == h[i] && j < i);
        bad += want != r[i];
      }
      printf("K=%4d threads=%4d  %lld cycles  %s\n", K, threads, cy, bad ? "MISMATCH" : "ok");
    }
  return 0;
}
