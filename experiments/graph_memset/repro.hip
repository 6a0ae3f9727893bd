// Minimal repro attempt for the round-1 observation "memset nodes of a captured hipGraph complete after the kernels
// that follow them once other work has run between replays" (csrc/qpg_select.hip, fill_ff_kernel).  Pattern of the
// original code: capture { hipMemsetAsync(table, 0xFF) ; atomicMin kernel on table ; read-out kernel }, replay, run eager
// work on the same stream between replays, check that every replay saw a freshly filled table.
//   hipcc --offload-arch=gfx950 -O2 -o repro repro.hip && ./repro [pool]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
__global__ void min_kernel(unsigned long long* t, const unsigned* v, int n, int k) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) atomicMin(&t[i % k], ((unsigned long long)v[i] << 32) | (unsigned)i);
}
__global__ void read_kernel(const unsigned long long* t, unsigned long long* out, int k) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < k) out[i] = t[i];
}
__global__ void busy_kernel(float* x, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { float a = x[i]; for (int j = 0; j < 200; ++j) a = a * 1.0001f + 0.5f; x[i] = a; }
}
int main(int argc, char** argv) {
  const bool pool = argc > 1 && argv[1][0] == 'p';   // tables from hipMallocAsync (stream-ordered pool) instead of hipMalloc
  const bool fork = argc > 1 && argv[1][0] == 'f';   // the capture forks a side stream (as CodeKNN's text side does)
  const int K = 48 * 512, N = 1 << 20;
  hipStream_t s; CK(hipStreamCreate(&s));
  unsigned long long *t, *out; unsigned* v; float* x;
  if (pool) { CK(hipMallocAsync(&t, K * 8, s)); CK(hipMallocAsync(&out, K * 8, s)); }
  else { CK(hipMalloc(&t, K * 8)); CK(hipMalloc(&out, K * 8)); }
  CK(hipMalloc(&v, N * 4)); CK(hipMalloc(&x, N * 4));
  unsigned* hv = (unsigned*)malloc(N * 4);
  for (int i = 0; i < N; ++i) hv[i] = 1000u + (unsigned)((i * 2654435761u) >> 12);
  CK(hipMemcpy(v, hv, N * 4, hipMemcpyHostToDevice)); CK(hipMemset(x, 0, N * 4));
  unsigned long long* want = (unsigned long long*)malloc(K * 8);
  for (int k = 0; k < K; ++k) want[k] = ~0ull;
  for (int i = 0; i < N; ++i) { unsigned long long key = ((unsigned long long)hv[i] << 32) | (unsigned)i; if (key < want[i % K]) want[i % K] = key; }
  hipGraph_t g; hipGraphExec_t ge;
  hipStream_t s2; CK(hipStreamCreate(&s2));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  if (fork) {
    CK(hipEventRecord(e0, s)); CK(hipStreamWaitEvent(s2, e0, 0));
    busy_kernel<<<N / 256, 256, 0, s2>>>(x, N);
    CK(hipEventRecord(e1, s2));
  }
  CK(hipMemsetAsync(t, 0xFF, K * 8, s));
  min_kernel<<<N / 256, 256, 0, s>>>(t, v, N, K);
  if (fork) CK(hipStreamWaitEvent(s, e1, 0));
  read_kernel<<<(K + 255) / 256, 256, 0, s>>>(t, out, K);
  CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  unsigned long long* got = (unsigned long long*)malloc(K * 8);
  int bad_replays = 0;
  for (int r = 0; r < 400; ++r) {
    if (r % 3 == 1) busy_kernel<<<N / 256, 256, 0, s>>>(x, N);                    // eager work between replays
    if (r % 5 == 2) { min_kernel<<<N / 256, 256, 0, s>>>(t, v, N / 2, K); }        // eager work that DIRTIES the table
    if (r % 7 == 3) CK(hipMemsetAsync(t, 0x00, K * 8, s));                          // ... or zeroes it
    CK(hipGraphLaunch(ge, s));
    CK(hipMemcpyAsync(got, out, K * 8, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
    int bad = 0, ff = 0;
    for (int k = 0; k < K; ++k) { bad += got[k] != want[k]; ff += got[k] == ~0ull; }
    if (bad) { if (bad_replays < 5) printf("replay %d: %d wrong entries (%d all-0xFF)\n", r, bad, ff); ++bad_replays; }
  }
  printf("%s: %d of 400 replays wrong\n", pool ? "pool tables" : (fork ? "forked capture" : "hipMalloc tables"), bad_replays);
  return bad_replays ? 1 : 0;
}
