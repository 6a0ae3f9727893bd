// Per-code segmented min/argmin over a distance row, and stable ranks.
//
// qpg_percode_argmin_*: the `if d < best[code]` update of CodeKNN.search_audio_cands /
// search_text_cands (GestureKNN.py:686-689, 717-720) for a whole query row at once.  The
// reference scans candidates in index order with a strict `<`, so the winner of a code is the
// candidate with the minimum distance and, among equals, the lowest index.  One block per
// query: distances are mapped to order-preserving unsigned keys and reduced with LDS atomics
// (ds_min_u64 / ds_min_u32); a second pass resolves the lowest index among the minima.
// HBM-bound: reads D once (Q*C*sizeof) plus the code column per candidate.
//
// qpg_rank_rows_*: np.argsort(np.argsort(x)) with a stable tie rule, by counting.
#include "qpg_common.h"

__device__ __forceinline__ unsigned long long order_key(double d) {
  unsigned long long b = (unsigned long long)__double_as_longlong(d);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_value(unsigned long long k, double) {
  unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  return __longlong_as_double((long long)b);
}
__device__ __forceinline__ unsigned int order_key(float d) {
  unsigned int b = __float_as_uint(d);
  return (b >> 31) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key_value(unsigned int k, float) {
  unsigned int b = (k >> 31) ? (k & 0x7fffffffu) : ~k;
  return __uint_as_float(b);
}

template <typename T, typename KeyT>
__global__ __launch_bounds__(1024) void percode_argmin_kernel(const T* __restrict__ D, int64_t ldD,
                                                              const int32_t* __restrict__ code, int code_ld, int N,
                                                              const int32_t* __restrict__ cand_cidx, int G, int K,
                                                              T absent, int32_t idx_base, T* __restrict__ out_dist,
                                                              int32_t* __restrict__ out_idx) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  KeyT* best = reinterpret_cast<KeyT*>(smem);
  unsigned int* besti = reinterpret_cast<unsigned int*>(smem + sizeof(KeyT) * K);

  const int q = blockIdx.x;
  const int64_t C = (int64_t)N * G;
  const T* row = D + (int64_t)q * ldD;
  const KeyT kmax = ~(KeyT)0;
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    best[k] = kmax;
    besti[k] = 0xffffffffu;
  }
  __syncthreads();
  for (int64_t c = threadIdx.x; c < C; c += blockDim.x) {
    const int j = (int)(c / G), g = (int)(c - (int64_t)j * G);
    const int cd = code[(int64_t)j * code_ld + cand_cidx[g]];
    if ((unsigned)cd < (unsigned)K) atomicMin(&best[cd], order_key(row[c]));
  }
  __syncthreads();
  for (int64_t c = threadIdx.x; c < C; c += blockDim.x) {
    const int j = (int)(c / G), g = (int)(c - (int64_t)j * G);
    const int cd = code[(int64_t)j * code_ld + cand_cidx[g]];
    if ((unsigned)cd < (unsigned)K && order_key(row[c]) == best[cd]) atomicMin(&besti[cd], (unsigned int)c);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    const bool have = besti[k] != 0xffffffffu;
    out_dist[(int64_t)q * K + k] = have ? key_value(best[k], T(0)) : absent;
    out_idx[(int64_t)q * K + k] = have ? (int32_t)besti[k] + idx_base : -1;
  }
}

template <typename T, typename KeyT>
static int percode_argmin(const char* name, qpg_ctx* ctx, void* stream, const T* D, int64_t ldD, int Q,
                          const int32_t* code, int code_ld, int N, const int32_t* cand_cidx, int G, int K, T absent,
                          int32_t idx_base, T* out_dist, int32_t* out_idx) {
  QPG_REQUIRE(ctx && D && code && cand_cidx && out_dist && out_idx, "%s: null pointer", name);
  QPG_REQUIRE(Q >= 0 && N >= 0 && G > 0 && K > 0 && K <= 4096 && code_ld > 0 && ldD >= (int64_t)N * G &&
                  (int64_t)N * G < 0x7fffffffll,
              "%s: bad size", name);
  if (Q == 0) return QPG_OK;
  size_t sh = (sizeof(KeyT) + sizeof(unsigned int)) * (size_t)K;
  hipLaunchKernelGGL((percode_argmin_kernel<T, KeyT>), dim3(Q), dim3(1024), sh, qpg_stream(stream), D, ldD, code,
                     code_ld, N, cand_cidx, G, K, absent, idx_base, out_dist, out_idx);
  QPG_LAUNCH_CHECK(name);
  return QPG_OK;
}

extern "C" int qpg_percode_argmin_f64(qpg_ctx* ctx, void* stream, const double* D, int64_t ldD, int Q,
                                      const int32_t* code, int code_ld, int N, const int32_t* cand_cidx, int G, int K,
                                      double absent, int32_t idx_base, double* out_dist, int32_t* out_idx) {
  return percode_argmin<double, unsigned long long>("qpg_percode_argmin_f64", ctx, stream, D, ldD, Q, code, code_ld,
                                                    N, cand_cidx, G, K, absent, idx_base, out_dist, out_idx);
}

extern "C" int qpg_percode_argmin_f32(qpg_ctx* ctx, void* stream, const float* D, int64_t ldD, int Q,
                                      const int32_t* code, int code_ld, int N, const int32_t* cand_cidx, int G, int K,
                                      float absent, int32_t idx_base, float* out_dist, int32_t* out_idx) {
  return percode_argmin<float, unsigned int>("qpg_percode_argmin_f32", ctx, stream, D, ldD, Q, code, code_ld, N,
                                             cand_cidx, G, K, absent, idx_base, out_dist, out_idx);
}

// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(1024) void rank_rows_kernel(const T* __restrict__ d, int K, int16_t* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* v = reinterpret_cast<T*>(smem);
  const int q = blockIdx.x;
  for (int k = threadIdx.x; k < K; k += blockDim.x) v[k] = d[(int64_t)q * K + k];
  __syncthreads();
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    const T x = v[k];
    int r = 0;
    for (int o = 0; o < K; ++o) {
      const T y = v[o];
      r += (y < x) || (y == x && o < k);
    }
    out[(int64_t)q * K + k] = (int16_t)r;
  }
}

template <typename T>
static int rank_rows(const char* name, qpg_ctx* ctx, void* stream, const T* d, int Q, int K, int16_t* out) {
  QPG_REQUIRE(ctx && d && out && Q >= 0 && K > 0 && K <= 8192, "%s: bad argument", name);
  if (Q == 0) return QPG_OK;
  int threads = K >= 1024 ? 1024 : ((K + 63) / 64) * 64;
  hipLaunchKernelGGL((rank_rows_kernel<T>), dim3(Q), dim3(threads), sizeof(T) * (size_t)K, qpg_stream(stream), d, K,
                     out);
  QPG_LAUNCH_CHECK(name);
  return QPG_OK;
}

extern "C" int qpg_rank_rows_f64(qpg_ctx* ctx, void* stream, const double* d, int Q, int K, int16_t* out) {
  return rank_rows<double>("qpg_rank_rows_f64", ctx, stream, d, Q, K, out);
}
extern "C" int qpg_rank_rows_f32(qpg_ctx* ctx, void* stream, const float* d, int Q, int K, int16_t* out) {
  return rank_rows<float>("qpg_rank_rows_f32", ctx, stream, d, Q, K, out);
}
