// Shared host-side plumbing for libqpg_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/qpg.h"

struct qpg_ctx {
  int device;
  int n_cu;
  float* zeros;   // 4 KB of device zeros: out-of-range tile loads are redirected here instead of being selected to 0
  bool select_lds_raised;   // percode_select_mixed_f64_kernel's dynamic-LDS limit has been raised on this device
  int opt[QPG_OPT_COUNT];   // qpg_ctx_set_option / qpg_ctx_get_option (include/qpg.h): per-context knobs, never process-wide
};

// Measurement knobs of the kernel experiments (tools/, experiments/): they exist only in a -DQPG_DEBUG_HOOKS build
// (tools/build_variant.sh ... "-DQPG_DEBUG_HOOKS"); in the product library each one is a compile-time constant and the
// qpg_debug_* setters are not exported (no global mutable state except the opaque context).
#ifdef QPG_DEBUG_HOOKS
#define QPG_HOOK_VAR(type, name, value) static type name = value
#else
#define QPG_HOOK_VAR(type, name, value) static constexpr type name = value
#endif

void qpg_set_error(const char* fmt, ...);

#define QPG_REQUIRE(cond, ...)      \
  do {                              \
    if (!(cond)) {                  \
      qpg_set_error(__VA_ARGS__);   \
      return QPG_EINVAL;            \
    }                               \
  } while (0)

#define QPG_LAUNCH_CHECK(name)                                                \
  do {                                                                        \
    hipError_t e_ = hipGetLastError();                                        \
    if (e_ != hipSuccess) {                                                   \
      qpg_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));    \
      return QPG_EHIP;                                                        \
    }                                                                         \
  } while (0)

static inline hipStream_t qpg_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// IEEE single-rounding float ops: the text and phase-gate distances must reproduce
// NumPy/scikit-learn float32 arithmetic bit for bit (separate multiply and add, correctly
// rounded sqrt and divide).  hipcc's __fmul_rn/__fsqrt_rn are plain `*` / native sqrt unless
// OCML_BASIC_ROUNDED_OPERATIONS is set, so exactness comes from the build flags instead:
// every file is compiled with -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt, and
// the pragma below repeats it where it matters.
#pragma clang fp contract(off)
__device__ __forceinline__ float f_mul(float a, float b) { return a * b; }
__device__ __forceinline__ float f_add(float a, float b) { return a + b; }
__device__ __forceinline__ float f_sub(float a, float b) { return a - b; }
__device__ __forceinline__ float f_div(float a, float b) { return a / b; }
__device__ __forceinline__ float f_sqrt(float a) { return __builtin_sqrtf(a); }
// the same for float64 (the near-tie guard re-evaluates audio distances in the reference's f64 arithmetic)
__device__ __forceinline__ double f_mul(double a, double b) { return a * b; }
__device__ __forceinline__ double f_add(double a, double b) { return a + b; }
__device__ __forceinline__ double f_sub(double a, double b) { return a - b; }
__device__ __forceinline__ double f_div(double a, double b) { return a / b; }

// sklearn semantics for degenerate rows: a row whose norm is < 10*eps is left unscaled by
// normalize(); for an all-zero row that gives 0.5*|other unit vector|^2 = 0.5 (0 if both are zero).
__device__ __forceinline__ double cosine_from_dot(double dot, double qn2, double cn2) {
  const double tiny = 10.0 * 2.220446049250313e-16;
  double nq = sqrt(qn2), nc = sqrt(cn2);
  bool zq = nq < tiny, zc = nc < tiny;
  if (zq || zc) {
    // unscaled row contributes its own squared norm; exact only for all-zero rows, which is
    // the case that occurs (zero padding); both-degenerate -> 0.5*(qn2 + cn2 - 2 dot)
    double a = zq ? qn2 : 1.0, b = zc ? cn2 : 1.0;
    double cross = dot / ((zq ? 1.0 : nq) * (zc ? 1.0 : nc));
    return 0.5 * (a + b - 2.0 * cross);
  }
  return 1.0 - dot / (nq * nc);
}

// A-priori error bound of the mixed-precision audio sweep (qpg_audio_cosine_mx, derivation in qpg_audio.hip):
// |D_mx[q][c] - D_f64[q][c]| <= gamma_32 (f32 FMA chains of 32 products) + f64 noise, for every pair.
// (the value is QPG_AUDIO_MX_ERR of include/qpg.h)

// ---- stable ranks of a table row by sorting (round 3) -----------------------------------------------------------------
// rank[k] = #{o : v[o] < v[k] or (v[o] == v[k] and o < k)} - what np.argsort(kind='stable') followed by argsort gives
// (GestureKNN.py:544-556 ranks its distance rows; the unstable-sort contract for exact ties is DESIGN.md §2).  Counting
// it is K^2 comparisons: 262 144 x ~7 VALU instructions = 14 us for one block at K = 512, more in f64 - a third of the
// audio select's launches was that.  A bitonic sort of the K (key, code) pairs in LDS is K/2 log^2 K / 2 compare-exchanges
// (11 520 at K = 512) and the sorted position IS the rank.
// skey: u64 [Kp] scratch, scode: i32 [Kp] scratch, Kp = rank_sort_pow2(K).  Every thread of the block calls it (it
// synchronises); emit(k, r) is called once per code k with its rank r.
__host__ __device__ __forceinline__ int rank_sort_pow2(int K) {
  int p = 2;
  while (p < K) p <<= 1;
  return p;
}
__device__ __forceinline__ unsigned long long rank_sort_key(double d) {
  const long long b = __double_as_longlong(d + 0.0);          // (-0.0 + 0.0 = +0.0: equal values get equal keys)
  return b < 0 ? ~(unsigned long long)b : ((unsigned long long)b | 0x8000000000000000ull);
}
__device__ __forceinline__ unsigned long long rank_sort_key(float d) {
  const unsigned int b = __float_as_uint(d + 0.0f);
  return (unsigned long long)((b >> 31) ? ~b : (b | 0x80000000u));
}
// Value of lane (l ^ PJ) of the wave, PJ a power of two < 64, without the LDS crossbar: DPP moves inside a row of 16
// (quad_perm for 1 and 2; row_shl:4 into banks 0, 2 and row_shr:4 into banks 1, 3; row_ror:8) and gfx950's
// v_permlane16_swap / v_permlane32_swap across rows.  A ds_bpermute_b32 costs a wave ~65 cycles and a wave's bpermutes do
// not overlap; these are ordinary VALU moves.
template <int PJ>
__device__ __forceinline__ int lane_xor(int x) {
  if (PJ == 1) return __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xf, 0xf, false);        // quad_perm:[1,0,3,2]
  if (PJ == 2) return __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xf, 0xf, false);        // quad_perm:[2,3,0,1]
  if (PJ == 4) {
    const int t = __builtin_amdgcn_update_dpp(x, x, 0x104, 0xf, 0x5, false);            // row_shl:4 -> lanes 0-3, 8-11
    return __builtin_amdgcn_update_dpp(t, x, 0x114, 0xf, 0xa, false);                   // row_shr:4 -> lanes 4-7, 12-15
  }
  if (PJ == 8) return __builtin_amdgcn_update_dpp(x, x, 0x128, 0xf, 0xf, false);       // row_ror:8
  if (PJ == 16) {
    const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);   // r[0]: rows (0,0,2,2) of x, r[1]: rows (1,1,3,3)
    return (threadIdx.x & 16) ? r[0] : r[1];
  }
  const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);     // r[0]: (lo, lo), r[1]: (hi, hi)
  return (threadIdx.x & 32) ? r[0] : r[1];
}
template <int PJ>
__device__ __forceinline__ void lane_xor_pair(unsigned long long k0, unsigned long long k1, int c0, int c1,
                                              unsigned long long& p0, unsigned long long& p1, int& q0, int& q1) {
  const unsigned int a0 = (unsigned int)lane_xor<PJ>((int)(unsigned int)k0);
  const unsigned int a1 = (unsigned int)lane_xor<PJ>((int)(unsigned int)(k0 >> 32));
  const unsigned int a2 = (unsigned int)lane_xor<PJ>((int)(unsigned int)k1);
  const unsigned int a3 = (unsigned int)lane_xor<PJ>((int)(unsigned int)(k1 >> 32));
  q0 = lane_xor<PJ>(c0);
  q1 = lane_xor<PJ>(c1);
  p0 = ((unsigned long long)a1 << 32) | a0;
  p1 = ((unsigned long long)a3 << 32) | a2;
}

// Kp / 2 threads (waves 0 .. Kp / 128 - 1) sort Kp pairs, two per thread (2t, 2t + 1), held in REGISTERS: a stage whose
// partner lives in the same wave exchanges through lane_xor (DPP / permlane swaps), only the stages that cross waves
// (j >= 128: 3 of the 45 at K = 512) go through LDS.  The j chain of a merge step is unrolled at compile time (template
// recursion; the k loop stays a loop: the fully unrolled network is 12 KB of code that runs once - instruction fetch then
// costs as much as the branches did).  Waves without elements only join the LDS stages' barriers.
// Measured at K = 512 (one block, 2.06 GHz): counting 14-17 us; bitonic over LDS with a block barrier per stage 14 us;
// this layout as a run-time loop on __shfl_xor (ds_bpermute) 14 us, bpermutes batched per stage 9.7 us, on DPP moves
// 12.5 us (~600 cycles per stage either way: a single wave spends them on the loop's branches and mask logic, not on the
// exchange); unrolled: DESIGN.md §4.3.
struct RankPair {
  unsigned long long k0, k1;
  int c0, c1;
};
__device__ __forceinline__ bool rank_gt(unsigned long long ka, int ca, unsigned long long kb, int cb) {
  return ka > kb || (ka == kb && ca > cb);
}
template <int J>
__device__ __forceinline__ void bitonic_chain(RankPair& e, int i0, int k, unsigned long long* skey, int* scode) {
  // the stages j = J, J / 2, .. 1 of merge step k (those with j < k), unrolled at compile time
  if constexpr (J == 0) {
    return;
  } else {
    if (J < k) {                                              // (uniform over the block: k is a loop counter)
      const bool up = (i0 & k) == 0;                          // (i0 and i0 + 1 differ in bit 0 only; k >= 2)
      if constexpr (J == 1) {
        if (rank_gt(e.k0, e.c0, e.k1, e.c1) == up) {
          const unsigned long long tk = e.k0; e.k0 = e.k1; e.k1 = tk;
          const int tc = e.c0; e.c0 = e.c1; e.c1 = tc;
        }
      } else {
        unsigned long long p0, p1;
        int q0, q1;
        if constexpr (J / 2 < 64) {
          lane_xor_pair<J / 2>(e.k0, e.k1, e.c0, e.c1, p0, p1, q0, q1);
        } else {
          __syncthreads();
          skey[i0] = e.k0; skey[i0 + 1] = e.k1; scode[i0] = e.c0; scode[i0 + 1] = e.c1;
          __syncthreads();
          const int o0 = i0 ^ J;
          p0 = skey[o0]; p1 = skey[o0 + 1]; q0 = scode[o0]; q1 = scode[o0 + 1];
        }
        const bool keep_min = ((i0 & J) == 0) == up;          // the lower index of an ascending pair keeps the smaller
        if (rank_gt(e.k0, e.c0, p0, q0) == keep_min) { e.k0 = p0; e.c0 = q0; }
        if (rank_gt(e.k1, e.c1, p1, q1) == keep_min) { e.k1 = p1; e.c1 = q1; }
      }
    }
    bitonic_chain<J / 2>(e, i0, k, skey, scode);
  }
}
template <int KP, typename T, typename F>
__device__ __forceinline__ void block_sorted_ranks_unrolled(const T* v, int K, unsigned long long* skey, int* scode,
                                                            F&& emit) {
  const int tid = threadIdx.x;
  if ((tid & ~63) * 2 >= KP) {                                // a wave without elements: the LDS stages' barriers only
    for (int k = 256; k <= KP; k <<= 1)
      for (int j = k >> 1; j >= 128; j >>= 1) {
        __syncthreads();
        __syncthreads();
      }
    return;
  }
  const int i0 = 2 * tid;
  RankPair e;
  e.k0 = i0 < K ? rank_sort_key(v[i0]) : ~0ull;
  e.k1 = i0 + 1 < K ? rank_sort_key(v[i0 + 1]) : ~0ull;
  e.c0 = i0;
  e.c1 = i0 + 1;
  for (int k = 2; k <= KP; k <<= 1) bitonic_chain<KP / 2>(e, i0, k, skey, scode);
  if (e.c0 < K) emit(e.c0, i0);
  if (e.c1 < K) emit(e.c1, i0 + 1);
}

template <typename T, typename F>
__device__ __forceinline__ void block_sorted_ranks(const T* v, int K, unsigned long long* skey, int* scode, F&& emit) {
  const int tid = threadIdx.x, nt = blockDim.x;
  const int Kp = rank_sort_pow2(K);
  if (Kp >= 128 && Kp <= 2 * nt && Kp <= 2048) {
    switch (Kp) {
      case 128: block_sorted_ranks_unrolled<128>(v, K, skey, scode, emit); break;
      case 256: block_sorted_ranks_unrolled<256>(v, K, skey, scode, emit); break;
      case 512: block_sorted_ranks_unrolled<512>(v, K, skey, scode, emit); break;
      case 1024: block_sorted_ranks_unrolled<1024>(v, K, skey, scode, emit); break;
      default: block_sorted_ranks_unrolled<2048>(v, K, skey, scode, emit); break;
    }
    return;
  }
  for (int i = tid; i < Kp; i += nt) {
    skey[i] = i < K ? rank_sort_key(v[i]) : ~0ull;
    scode[i] = i;
  }
  __syncthreads();
  for (int k = 2; k <= Kp; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < (Kp >> 1); i += nt) {
        const int lo = ((i & ~(j - 1)) << 1) | (i & (j - 1));       // the i-th pair of this stage: (lo, lo + j)
        const int hi = lo + j;
        const unsigned long long ka = skey[lo], kb = skey[hi];
        const int ca = scode[lo], cb = scode[hi];
        const bool gt = ka > kb || (ka == kb && ca > cb);
        if (gt == ((lo & k) == 0)) {
          skey[lo] = kb; skey[hi] = ka;
          scode[lo] = cb; scode[hi] = ca;
        }
      }
      __syncthreads();
    }
  for (int r = tid; r < K; r += nt) emit(scode[r], r);
}
