// What does it cost to reduce the audio sweep's Q x C distances by code straight from the sweep's registers (round 6)?
// The sweep's epilogue shape: 256 blocks x 8 waves, wave = one database window (26 candidates) x 48 queries, lane (cg, rg)
// holds 3 queries x 8 candidates.  Variants of what a lane does with each of its 24 values:
//   store   : the f32 matrix store of the product kernel (round 5)
//   dev32r  : device-scope returning atomicMax of the inverted 32-bit order key on table[Q][K]
//   dev32   : the same, result unused
//   dev64   : device-scope atomicMin of (key << 32 | candidate), result unused
//   xcd32r  : workgroup-scope (executed in THIS XCD's L2) returning atomicMax on table[xcc][Q][K]
//   xcd64   : workgroup-scope 64-bit atomicMin on table[xcc][Q][K]
//   xcd32r+list : xcd32r + the running-minimum records appended to a per-(block, query) segment through LDS counters
// Every variant's table is checked against a host reduction.
// build: hipcc --offload-arch=gfx950 -O3 -o epi_atomics epi_atomics.hip ; run: ./epi_atomics
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define Q 48
#define K 512
#define G 26
#define NW 2048

__host__ __device__ inline unsigned int hash32(unsigned int x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x;
}
__host__ __device__ inline float dist_of(int q, int c) {          // ~ cosine distances of random 6144-d rows: 1 +- 0.013
  const unsigned int h = hash32((unsigned)q * 2654435761u ^ hash32((unsigned)c + 12345u));
  const unsigned int h2 = hash32(h ^ 0x9e3779b9u);
  const float u = ((h >> 8) + (h2 >> 8)) * (1.0f / 16777216.0f) - 1.0f;   // triangular on [-1, 1]
  return 1.0f + 0.03f * u;
}
__host__ __device__ inline unsigned int order_key(float v) {
  unsigned int b;
#ifdef __HIP_DEVICE_COMPILE__
  b = __float_as_uint(v);
#else
  memcpy(&b, &v, 4);
#endif
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ inline int xcc_id() {
  int x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  return x & 7;
}

template <int MODE>
__global__ __launch_bounds__(512, 2) void epi_kernel(const short* __restrict__ code, float* __restrict__ D,
                                                     unsigned int* __restrict__ t32, unsigned long long* __restrict__ t64,
                                                     int* __restrict__ seg_cnt, unsigned long long* __restrict__ seg, int seg_cap,
                                                     float eps) {
  __shared__ int lcnt[Q];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int j = blockIdx.x * 8 + w;
  const int cg = lane & 15, rg = lane >> 4;
  if (MODE == 6) {
    if (tid < Q) lcnt[tid] = 0;
    __syncthreads();
  }
  int xcc = 0;
  if (MODE >= 4) xcc = xcc_id();
#pragma unroll
  for (int ct = 0; ct < 3; ++ct) {
    const int q = ct * 16 + cg;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int g0 = 16 * t + 4 * rg;
      short cd4[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) cd4[r] = (g0 + r < G) ? code[j * G + g0 + r] : (short)0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int g = g0 + r;
        if (g >= G) continue;
        const int c = j * G + g;
        const float dv = dist_of(q, c);
        const int cd = cd4[r];
        const unsigned int key = order_key(dv);
        if (MODE == 0) {
          D[(size_t)q * (NW * G) + c] = dv;
        } else if (MODE == 1) {
          const unsigned int old = __hip_atomic_fetch_max(&t32[q * K + cd], ~key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (old == 0x12345u) D[0] = 1.f;
        } else if (MODE == 2) {
          __hip_atomic_fetch_max(&t32[q * K + cd], ~key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (MODE == 3) {
          __hip_atomic_fetch_min(&t64[q * K + cd], ((unsigned long long)key << 32) | (unsigned)c, __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT);
        } else if (MODE == 4) {
          const unsigned int old =
              __hip_atomic_fetch_max(&t32[(xcc * Q + q) * K + cd], ~key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (old == 0x12345u) D[0] = 1.f;
        } else if (MODE == 5) {
          __hip_atomic_fetch_min(&t64[(xcc * Q + q) * K + cd], ((unsigned long long)key << 32) | (unsigned)c, __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (MODE == 6) {
          const unsigned int old =
              ~__hip_atomic_fetch_max(&t32[(xcc * Q + q) * K + cd], ~key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          const unsigned int lo = old < key ? old : key;
          unsigned int lb = (lo & 0x80000000u) ? (lo & 0x7fffffffu) : ~lo;
          if (dv <= __uint_as_float(lb) + eps) {
            const int pos = atomicAdd(&lcnt[q], 1);
            if (pos < seg_cap)
              seg[((size_t)blockIdx.x * Q + q) * seg_cap + pos] = ((unsigned long long)__float_as_uint(dv) << 32) | (unsigned)c;
          }
        }
      }
    }
  }
  if (MODE == 6) {
    __syncthreads();
    if (tid < Q) seg_cnt[blockIdx.x * Q + tid] = lcnt[tid];
  }
}

int main() {
  const int C = NW * G;
  std::vector<short> h_code(C);
  for (int c = 0; c < C; ++c) h_code[c] = (short)(hash32(c * 7919u + 17u) % K);
  std::vector<unsigned long long> ref((size_t)Q * K, ~0ull);
  for (int q = 0; q < Q; ++q)
    for (int c = 0; c < C; ++c) {
      const unsigned long long k = ((unsigned long long)order_key(dist_of(q, c)) << 32) | (unsigned)c;
      if (k < ref[(size_t)q * K + h_code[c]]) ref[(size_t)q * K + h_code[c]] = k;
    }
  short* d_code; float* D; unsigned int* t32; unsigned long long* t64; int* seg_cnt; unsigned long long* seg;
  const int seg_cap = 64;
  CK(hipMalloc(&d_code, C * 2)); CK(hipMalloc(&D, (size_t)Q * C * 4));
  CK(hipMalloc(&t32, 8 * Q * K * 4)); CK(hipMalloc(&t64, 8 * Q * K * 8));
  CK(hipMalloc(&seg_cnt, 256 * Q * 4)); CK(hipMalloc(&seg, (size_t)256 * Q * seg_cap * 8));
  CK(hipMemcpy(d_code, h_code.data(), C * 2, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const float eps = 2.73e-6f;
  auto time_it = [&](const char* name, auto launch, int mode) {
    double best = 1e30, sum = 0;
    const int reps = 20;
    for (int r = 0; r < reps + 2; ++r) {
      CK(hipMemsetAsync(t32, 0, 8 * Q * K * 4));
      CK(hipMemsetAsync(t64, 0xff, 8 * Q * K * 8));
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (r >= 2) { if (ms < best) best = ms; sum += ms; }
    }
    // check
    int bad = 0; long long npot = 0; int maxseg = 0;
    if (mode == 1 || mode == 2) {
      std::vector<unsigned int> h(Q * K); CK(hipMemcpy(h.data(), t32, Q * K * 4, hipMemcpyDeviceToHost));
      for (int i = 0; i < Q * K; ++i) bad += (~h[i]) != (unsigned)(ref[i] >> 32);
    } else if (mode == 3) {
      std::vector<unsigned long long> h(Q * K); CK(hipMemcpy(h.data(), t64, Q * K * 8, hipMemcpyDeviceToHost));
      for (int i = 0; i < Q * K; ++i) bad += h[i] != ref[i];
    } else if (mode == 4 || mode == 6) {
      std::vector<unsigned int> h(8 * Q * K); CK(hipMemcpy(h.data(), t32, 8 * Q * K * 4, hipMemcpyDeviceToHost));
      for (int i = 0; i < Q * K; ++i) {
        unsigned int m = 0;
        for (int x = 0; x < 8; ++x) m = h[x * Q * K + i] > m ? h[x * Q * K + i] : m;
        bad += (~m) != (unsigned)(ref[i] >> 32);
      }
      if (mode == 6) {
        std::vector<int> sc(256 * Q); CK(hipMemcpy(sc.data(), seg_cnt, 256 * Q * 4, hipMemcpyDeviceToHost));
        for (int i = 0; i < 256 * Q; ++i) { npot += sc[i]; maxseg = sc[i] > maxseg ? sc[i] : maxseg; }
      }
    } else if (mode == 5) {
      std::vector<unsigned long long> h(8 * Q * K); CK(hipMemcpy(h.data(), t64, 8 * Q * K * 8, hipMemcpyDeviceToHost));
      for (int i = 0; i < Q * K; ++i) {
        unsigned long long m = ~0ull;
        for (int x = 0; x < 8; ++x) m = h[x * Q * K + i] < m ? h[x * Q * K + i] : m;
        bad += m != ref[i];
      }
    }
    printf("%-14s min %7.1f us  mean %7.1f us  mismatches %d", name, best * 1e3, sum / reps * 1e3, bad);
    if (mode == 6) printf("  potentials %lld (%.0f per query), longest segment %d", npot, (double)npot / Q, maxseg);
    printf("\n");
  };
#define RUN(name, M) time_it(name, [&] { hipLaunchKernelGGL(epi_kernel<M>, dim3(256), dim3(512), 0, 0, d_code, D, t32, t64, seg_cnt, seg, seg_cap, eps); }, M)
  RUN("store", 0);
  RUN("dev32r", 1);
  RUN("dev32", 2);
  RUN("dev64", 3);
  RUN("xcd32r", 4);
  RUN("xcd64", 5);
  RUN("xcd32r+list", 6);
  RUN("store", 0);
  return 0;
}
