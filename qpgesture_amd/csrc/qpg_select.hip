// Per-code segmented min/argmin over a distance row, and stable ranks.
//
// qpg_percode_select_*: the `if d < best[code]` update of CodeKNN.search_audio_cands /
// search_text_cands (GestureKNN.py:686-689, 717-720) for a whole query row at once.  The
// reference scans candidates in index order with a strict `<`, so the winner of a code is the
// candidate with the minimum distance and, among equals, the lowest index.  Distances are mapped
// to order-preserving unsigned keys and reduced with LDS atomics (ds_min_u64 / ds_min_u32).
// (Rounds 1-2's stand-alone forms - qpg_percode_argmin_*, the resolve / finalize chain over global
// tables - left the library in round 5: no caller but tests.)
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

// ---------------------------------------------------------------------------------------------
// One-launch select (round 2): per query row, per-code minimum + first-wins index + distances + stable ranks in ONE
// kernel, one 1024-thread block per query (round 1 ran a chain of four launches over global tables, 624 blocks doing
// ~320 k global 64-bit atomics per clip): the whole row (C candidates,
// L2/MALL-resident: the sweep has just written it) is streamed by one block with 16-byte loads, the minima live in
// LDS only, and the candidates' codes come from a precomputed int16 array (no division by the grid size, no double
// indirection).  f64: two passes over the row (minimum of the ordered keys, then lowest index among the entries
// equal to it == the reference's strict-`<` scan); f32: one pass on (key << 32 | index).
// ---------------------------------------------------------------------------------------------
template <typename T, typename KeyT, bool PACKED>
__global__ __launch_bounds__(1024) void percode_select_kernel(const T* __restrict__ D, int64_t ldD,
                                                              const int16_t* __restrict__ cand_code, int64_t C, int K,
                                                              T absent, int32_t idx_base, T* __restrict__ out_dist,
                                                              int32_t* __restrict__ out_idx,
                                                              int16_t* __restrict__ out_rank, int q_block,
                                                              int64_t block_stride) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned long long* best = reinterpret_cast<unsigned long long*>(smem);
  unsigned int* besti = reinterpret_cast<unsigned int*>(smem + 8 * (size_t)K);
  T* v = reinterpret_cast<T*>(smem + 12 * (size_t)rank_sort_pow2(K));   // (the key tables double as the rank sort's scratch)
  constexpr int VEC = 16 / sizeof(T);                      // elements per 16-byte load
  typedef T vecT __attribute__((ext_vector_type(VEC)));
  typedef int16_t vecC __attribute__((ext_vector_type(VEC)));
  const int q = blockIdx.x;
  // exchange layout (sharded DB): row q lives in block q / q_block of a byte buffer whose blocks are block_stride
  // bytes apart (one block per destination rank, several arrays per block); q_block == 0: plain [Q][K] tables
  if (q_block > 0) {
    const int64_t shift = (int64_t)(q / q_block) * block_stride;
    const int64_t rowoff = (int64_t)(q % q_block) * K - (int64_t)q * K;
    out_dist = reinterpret_cast<T*>(reinterpret_cast<unsigned char*>(out_dist) + shift) + rowoff;
    out_idx = reinterpret_cast<int32_t*>(reinterpret_cast<unsigned char*>(out_idx) + shift) + rowoff;
  }
  const T* row = D + (int64_t)q * ldD;
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    best[k] = ~0ull;
    besti[k] = 0xffffffffu;
  }
  __syncthreads();
  const bool vec_ok = (ldD % VEC) == 0 && (reinterpret_cast<uintptr_t>(D) % 16) == 0 &&
                      (reinterpret_cast<uintptr_t>(cand_code) % (2 * VEC)) == 0;
  const int64_t Cv = vec_ok ? (C / VEC) * VEC : 0;
  for (int64_t c = (int64_t)threadIdx.x * VEC; c < Cv; c += (int64_t)blockDim.x * VEC) {
    const vecT d = *reinterpret_cast<const vecT*>(row + c);
    const vecC cd = *reinterpret_cast<const vecC*>(cand_code + c);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      if ((unsigned)cd[e] >= (unsigned)K) continue;
      const unsigned long long key = PACKED ? (((unsigned long long)order_key(d[e]) << 32) | (unsigned int)(c + e + idx_base))
                                            : (unsigned long long)order_key(d[e]);
      atomicMin(&best[cd[e]], key);
    }
  }
  for (int64_t c = Cv + threadIdx.x; c < C; c += blockDim.x) {
    const int cd = cand_code[c];
    if ((unsigned)cd >= (unsigned)K) continue;
    const unsigned long long key = PACKED ? (((unsigned long long)order_key(row[c]) << 32) | (unsigned int)(c + idx_base))
                                          : (unsigned long long)order_key(row[c]);
    atomicMin(&best[cd], key);
  }
  __syncthreads();
  if (!PACKED) {
    for (int64_t c = (int64_t)threadIdx.x * VEC; c < Cv; c += (int64_t)blockDim.x * VEC) {
      const vecT d = *reinterpret_cast<const vecT*>(row + c);
      const vecC cd = *reinterpret_cast<const vecC*>(cand_code + c);
#pragma unroll
      for (int e = 0; e < VEC; ++e)
        if ((unsigned)cd[e] < (unsigned)K && (unsigned long long)order_key(d[e]) == best[cd[e]])
          atomicMin(&besti[cd[e]], (unsigned int)(c + e + idx_base));
    }
    for (int64_t c = Cv + threadIdx.x; c < C; c += blockDim.x) {
      const int cd = cand_code[c];
      if ((unsigned)cd < (unsigned)K && (unsigned long long)order_key(row[c]) == best[cd])
        atomicMin(&besti[cd], (unsigned int)(c + idx_base));
    }
    __syncthreads();
  }
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    const unsigned long long kv = best[k];
    const bool have = kv != ~0ull;
    T d;
    int32_t ix;
    if (PACKED) {
      d = have ? key_value((KeyT)(kv >> 32), T(0)) : absent;
      ix = have ? (int32_t)(kv & 0xffffffffu) : -1;
    } else {
      d = have ? key_value((KeyT)kv, T(0)) : absent;
      ix = have ? (int32_t)besti[k] : -1;
    }
    v[k] = d;
    out_dist[(int64_t)q * K + k] = d;
    out_idx[(int64_t)q * K + k] = ix;
  }
  if (!out_rank) return;
  __syncthreads();
  const int Kp = rank_sort_pow2(K);
  block_sorted_ranks(v, K, best, reinterpret_cast<int*>(best + Kp),
                     [&](int k, int r) { out_rank[(int64_t)q * K + k] = (int16_t)r; });
}

// ---------------------------------------------------------------------------------------------
// Near-tie guard of the audio select (SURVEY.md §7 hard part 2).  The sweep computes 1 - q.c/(|q||c|) on the f64
// matrix cores; the reference computes 0.5*|q/|q| - c/|c||^2 with NumPy's einsum summation order
// (GestureKNN.py:685 -> sklearn paired_cosine_distances).  The two agree to ~1e-16, so they can only ORDER two
// distances differently when those are closer than that - rare, but "bit-exact indices" is the bar.  The guarded
// select therefore
//   1. counts, per (query, code), the candidates within `eps` of the code's minimum; where there are two or more
//      it re-evaluates exactly those candidates with the reference's own arithmetic (refine_pair_f64 below: sklearn's
//      normalise + einsum-order sum of squares, separately rounded multiplies and adds, IEEE sqrt and divide) and
//      picks the winner by (reference distance, index);
//   2. finds the codes whose minima lie within `eps` of another code's minimum and replaces those minima by the
//      reference-arithmetic value of their winner before ranking, so the rank order is the reference's.
// Nothing is flagged on ordinary data and the extra cost is one compare per candidate; flagged work happens inside
// the same launch (no host round trip).  stats[0] += re-evaluated pairs, stats[1] = 1 if a list overflowed.
// ---------------------------------------------------------------------------------------------
// Stable rank of every entry of the LDS row v[K] (value, then index): r[k] = #{o : v[o] < v[k] or (v[o] == v[k] and
// o < k)}.  The K x K count is VALU-bound (~35 cycles per comparison step per wave), so P = blockDim / K threads share
// an entry and add their partial counts in LDS (`cnt`, [K] ints).  Calls __syncthreads(); all threads must call.
template <typename T, typename F>
__device__ __forceinline__ void block_stable_ranks(const T* v, int K, int* cnt, F&& emit) {
  const int tid = threadIdx.x;
  if (K <= 512 && K >= 64 && (int)blockDim.x >= 256) {      // sorted (qpg_common.h) in 6 KB of static LDS: ~5 us, not 15
    __shared__ unsigned long long rank_skey[512];
    __shared__ int rank_scode[512];
    block_sorted_ranks(v, K, rank_skey, rank_scode, emit);
    return;
  }
  const int P = (int)blockDim.x >= 2 * K ? (int)blockDim.x / K : 1;
  if (P == 1) {
    for (int k = tid; k < K; k += blockDim.x) {
      const T x = v[k];
      int r = 0;
#pragma unroll 8
      for (int o = 0; o < K; ++o) {
        const T y = v[o];
        r += (y < x) || (y == x && o < k);
      }
      emit(k, r);
    }
    return;
  }
  for (int k = tid; k < K; k += blockDim.x) cnt[k] = 0;
  __syncthreads();
  if (tid < P * K) {
    const int k = tid % K, part = tid / K;
    const int o0 = (int)((int64_t)part * K / P), o1 = (int)((int64_t)(part + 1) * K / P);
    const T x = v[k];
    int r = 0;
#pragma unroll 8
    for (int o = o0; o < o1; ++o) {
      const T y = v[o];
      r += (y < x) || (y == x && o < k);
    }
    atomicAdd(&cnt[k], r);
  }
  __syncthreads();
  for (int k = tid; k < K; k += blockDim.x) emit(k, cnt[k]);
}

struct GuardArgs {
  const float* base;      // [N][T][F] interpolated WavLM frames of this shard (f32; IEEE f16 when `half`)
  int half;               // base is stored in f16: values are widened, i.e. the re-evaluation sees the rounded track
  const float* q32;       // [Q][n_taps*F] packed queries
  const int32_t* cand_t;  // [G] start frame of grid position g
  int T, F, G, n_taps, tap_stride;
  double eps;
  int32_t* stats;         // [2]
};
// Walk-relevance cut of the mixed select's tier-1 lists (round 4; pos_t == nullptr: off).  The walk reads, per step and
// per previous code p, only the code(s) with the smallest fused score  pos_rank[p][c] + 0.05 freq_rank[c] + rank(c)
// (GestureKNN.py:540-545, :574-576, order[0] - order[:2] without the text side, :593): a code whose rank is certainly
// above every step's winning score can not be read, so neither its rank among near-tied neighbours nor its winner among
// near-tied candidates has to be settled in f64.  See the list phase of percode_select_mixed_f64_kernel.
struct RankCut {
  const int16_t* pos_t;   // [K][K] TRANSPOSED pose ranks: pos_t[c * K + p] = rank of code c in previous code p's row
  const int16_t* freq;    // [K] frequency ranks
  int top_n;              // 1: the best fused score is read; 2: the best two
  int probe;              // number of best-ranked codes the bound on the winning score is taken over (<= K)
};
#define GUARD_LIST 256
typedef _Float16 g16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float guard_val(const GuardArgs& A, int64_t off) {       // base[off] as f32
  return A.half ? (float)reinterpret_cast<const _Float16*>(A.base)[off] : A.base[off];
}
__device__ __forceinline__ f32x4 guard_val4(const GuardArgs& A, int64_t off) {      // base[off .. off+3], off % 4 == 0
  if (A.half) {
    const g16x4 h = *reinterpret_cast<const g16x4*>(reinterpret_cast<const _Float16*>(A.base) + off);
    return (f32x4){(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
  }
  return *reinterpret_cast<const f32x4*>(A.base + off);
}

// One (query, candidate) distance in the reference's arithmetic, by FOUR cooperating lanes (an aligned quad):
// lanes 0/1 run the two einsum accumulator chains of the query's squared norm, lanes 2/3 those of the candidate's;
// then lanes 0/1 run the chains of the normalised difference.  NumPy's f64 einsum kernel: 2 SIMD lanes, groups of
// 8 elements visited as pairs 3,2,1,0, acc = v*v + acc with separate roundings (oracle/knn_oracle.py::einsum_sq).
__device__ __forceinline__ double refine_pair_f64(const GuardArgs& A, int q, int64_t c_local, int sub) {
  const int D = A.n_taps * A.F;
  const int j = (int)(c_local / A.G), g = (int)(c_local - (int64_t)j * A.G);
  const int t0 = A.cand_t[g];
  const float* qrow = A.q32 + (int64_t)q * D;
  const int64_t crow = (int64_t)j * A.T * A.F;
  auto cval = [&](int e) -> double {
    const int tap = e / A.F, f = e - tap * A.F;
    const int t = t0 + tap * A.tap_stride;
    return t < A.T ? (double)guard_val(A, crow + (int64_t)t * A.F + f) : 0.0;
  };
  const int l = sub & 1;
  const bool is_c = sub >= 2;
  auto chain = [&](auto val) -> double {
    double acc = 0.0;
    const int nfull = D / 8;
    for (int gq = 0; gq < nfull; ++gq) {
#pragma unroll
      for (int u = 3; u >= 0; --u) {
        const double v = val(8 * gq + 2 * u + l);
        acc = f_add(f_mul(v, v), acc);
      }
    }
    for (int i = nfull * 8; i < D; i += 2) {
      const double v = (i + l < D) ? val(i + l) : 0.0;
      acc = f_add(f_mul(v, v), acc);
    }
    return acc;
  };
  const double a1 = is_c ? chain(cval) : chain([&](int e) -> double { return (double)qrow[e]; });
  const double o1 = __shfl_xor(a1, 1, 64);
  double n = sqrt(l == 0 ? f_add(a1, o1) : f_add(o1, a1));             // acc[0] + acc[1]
  if (n < 10.0 * 2.220446049250313e-16) n = 1.0;                        // sklearn _handle_zeros_in_scale
  const double n_other = __shfl_xor(n, 2, 64);                          // quad: lanes 0,1 <-> 2,3
  const double nq = is_c ? n_other : n, nc = is_c ? n : n_other;
  const double a2 = chain([&](int e) -> double { return f_sub(f_div((double)qrow[e], nq), f_div(cval(e), nc)); });
  const double o2 = __shfl_xor(a2, 1, 64);
  return f_mul(0.5, l == 0 ? f_add(a2, o2) : f_add(o2, a2));
}

__global__ __launch_bounds__(1024) void percode_select_guarded_f64_kernel(
    const double* __restrict__ D, int64_t ldD, const int16_t* __restrict__ cand_code, int64_t C, int K, double absent,
    int32_t idx_base, double* __restrict__ out_dist, int32_t* __restrict__ out_idx, int16_t* __restrict__ out_rank,
    int q_block, int64_t block_stride, GuardArgs A, int sort_ranks) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned long long* best = reinterpret_cast<unsigned long long*>(smem);
  unsigned int* besti = reinterpret_cast<unsigned int*>(smem + 8 * (size_t)K);
  double* v = reinterpret_cast<double*>(smem + 12 * (size_t)K + 4 * (size_t)K);     // keep 8-byte alignment
  unsigned int* near_ = reinterpret_cast<unsigned int*>(smem + 12 * (size_t)K);
  unsigned char* tail = smem + 24 * (size_t)K;
  double* l_d = reinterpret_cast<double*>(tail);                                    // [GUARD_LIST]
  int* l_c = reinterpret_cast<int*>(tail + 8 * GUARD_LIST);                         // [GUARD_LIST] local candidate
  int* l_k = l_c + GUARD_LIST;                                                      // [GUARD_LIST] code
  int* ctl = l_k + GUARD_LIST;                                                      // [0] list length, [1] any flag
  const int q = blockIdx.x, tid = threadIdx.x;
  const double* row = D + (int64_t)q * ldD;
  if (q_block > 0) {
    const int64_t shift = (int64_t)(q / q_block) * block_stride;
    const int64_t rowoff = (int64_t)(q % q_block) * K - (int64_t)q * K;
    out_dist = reinterpret_cast<double*>(reinterpret_cast<unsigned char*>(out_dist) + shift) + rowoff;
    out_idx = reinterpret_cast<int32_t*>(reinterpret_cast<unsigned char*>(out_idx) + shift) + rowoff;
  }
  for (int k = tid; k < K; k += blockDim.x) {
    best[k] = ~0ull;
    besti[k] = 0xffffffffu;
    near_[k] = 0;
  }
  if (tid < 2) ctl[tid] = 0;
  __syncthreads();
  // the two streaming passes of percode_select_kernel (16-byte loads), pass 2 also counting the band population
  typedef double vecD __attribute__((ext_vector_type(2)));
  typedef int16_t vecC __attribute__((ext_vector_type(2)));
  const bool vec_ok = (ldD % 2) == 0 && (reinterpret_cast<uintptr_t>(D) % 16) == 0 &&
                      (reinterpret_cast<uintptr_t>(cand_code) % 4) == 0;
  const int64_t Cv = vec_ok ? (C / 2) * 2 : 0;
  for (int64_t c = (int64_t)tid * 2; c < Cv; c += (int64_t)blockDim.x * 2) {
    const vecD d = *reinterpret_cast<const vecD*>(row + c);
    const vecC cd = *reinterpret_cast<const vecC*>(cand_code + c);
#pragma unroll
    for (int e = 0; e < 2; ++e)
      if ((unsigned)cd[e] < (unsigned)K) atomicMin(&best[cd[e]], (unsigned long long)order_key(d[e]));
  }
  for (int64_t c = Cv + tid; c < C; c += blockDim.x) {
    const int cd = cand_code[c];
    if ((unsigned)cd < (unsigned)K) atomicMin(&best[cd], (unsigned long long)order_key(row[c]));
  }
  __syncthreads();
  auto pass2 = [&](int64_t c, double d, int cd) {
    if ((unsigned)cd >= (unsigned)K) return;
    const unsigned long long bk = best[cd];
    if ((unsigned long long)order_key(d) == bk) atomicMin(&besti[cd], (unsigned int)(c + idx_base));
    else if (d > key_value(bk, 0.0) + A.eps) return;
    if (atomicAdd(&near_[cd], 1u) == 1u) ctl[1] = 1;                  // a second candidate inside the band
  };
  for (int64_t c = (int64_t)tid * 2; c < Cv; c += (int64_t)blockDim.x * 2) {
    const vecD d = *reinterpret_cast<const vecD*>(row + c);
    const vecC cd = *reinterpret_cast<const vecC*>(cand_code + c);
    pass2(c, d[0], cd[0]);
    pass2(c + 1, d[1], cd[1]);
  }
  for (int64_t c = Cv + tid; c < C; c += blockDim.x) pass2(c, row[c], cand_code[c]);
  __syncthreads();
  // ---- 1. candidate-level near ties: re-evaluate the band of every flagged code in the reference's arithmetic
  if (ctl[1]) {
    for (int64_t c = tid; c < C; c += blockDim.x) {
      const int cd = cand_code[c];
      if ((unsigned)cd >= (unsigned)K || near_[cd] < 2) continue;
      if (row[c] <= key_value(best[cd], 0.0) + A.eps) {
        const int pos = atomicAdd(&ctl[0], 1);
        if (pos < GUARD_LIST) {
          l_c[pos] = (int)c;
          l_k[pos] = cd;
        }
      }
    }
    __syncthreads();
    int n = ctl[0];
    if (n > GUARD_LIST) {
      n = GUARD_LIST;
      if (tid == 0) A.stats[1] = 1;
    }
    for (int e0 = 0; e0 < n; e0 += blockDim.x / 4) {
      const int e = e0 + (tid >> 2);
      const double dr = refine_pair_f64(A, q, l_c[e < n ? e : 0], tid & 3);
      if (e < n && (tid & 3) == 0) l_d[e] = dr;
    }
    for (int k = tid; k < K; k += blockDim.x)
      if (near_[k] >= 2) {
        best[k] = ~0ull;
        besti[k] = 0xffffffffu;
      }
    __syncthreads();
    for (int e = tid; e < n; e += blockDim.x) atomicMin(&best[l_k[e]], (unsigned long long)order_key(l_d[e]));
    __syncthreads();
    for (int e = tid; e < n; e += blockDim.x)
      if ((unsigned long long)order_key(l_d[e]) == best[l_k[e]])
        atomicMin(&besti[l_k[e]], (unsigned int)(l_c[e] + idx_base));
    if (tid == 0) atomicAdd(&A.stats[0], n);
    __syncthreads();
  }
  for (int k = tid; k < K; k += blockDim.x) {
    const bool have = besti[k] != 0xffffffffu;
    v[k] = have ? key_value(best[k], 0.0) : absent;
  }
  if (tid < 2) ctl[tid] = 0;
  __syncthreads();
  // ---- 2. ranks; rank-level near ties: minima of DIFFERENT codes closer than eps are compared in reference
  // arithmetic.  Two values are that close iff they are neighbours in the sorted order, so the check is one look at
  // each rank neighbour (the sorted copy is scattered by rank), not another K x K sweep.
  for (int k = tid; k < K; k += blockDim.x) {
    const bool have = besti[k] != 0xffffffffu;
    out_dist[(int64_t)q * K + k] = v[k];
    out_idx[(int64_t)q * K + k] = have ? (int32_t)besti[k] : -1;
  }
  if (!out_rank) return;
  int* s_code = reinterpret_cast<int*>(best);                 // the key table is no longer needed: [K] code at rank r
  int* rcnt = s_code + K;                                     // second half of the key table: rank counters
  // (sorted when the launch left room for the sort's scratch behind the lists - K <= 1024 -, counted otherwise)
  unsigned long long* skey = reinterpret_cast<unsigned long long*>(ctl + 4);
  auto rank_pass = [&]() {
    auto emit = [&](int k, int r) {
      out_rank[(int64_t)q * K + k] = (int16_t)r;
      s_code[r] = k;
    };
    if (sort_ranks) block_sorted_ranks(v, K, skey, reinterpret_cast<int*>(skey + rank_sort_pow2(K)), emit);
    else block_stable_ranks(v, K, rcnt, emit);
  };
  rank_pass();
  __syncthreads();
  for (int r = tid; r + 1 < K; r += blockDim.x) {
    const int ka = s_code[r], kb = s_code[r + 1];
    if (besti[ka] == 0xffffffffu || besti[kb] == 0xffffffffu) continue;      // absent codes tie at `absent` by design
    if (v[kb] - v[ka] < A.eps) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int k = h ? kb : ka;
        if (near_[k] >= 2) continue;                                          // already a reference-arithmetic value
        if (atomicExch(&near_[k], 2u) >= 2u) continue;                        // listed once
        const int pos = atomicAdd(&ctl[0], 1);
        if (pos < GUARD_LIST) {
          l_c[pos] = (int)(besti[k] - (unsigned int)idx_base);
          l_k[pos] = k;
        }
      }
    }
  }
  __syncthreads();
  int n2 = ctl[0];
  if (n2 == 0) return;
  if (n2 > GUARD_LIST) {
    n2 = GUARD_LIST;
    if (tid == 0) A.stats[1] = 1;
  }
  for (int e0 = 0; e0 < n2; e0 += blockDim.x / 4) {
    const int e = e0 + (tid >> 2);
    const double dr = refine_pair_f64(A, q, l_c[e < n2 ? e : 0], tid & 3);
    if (e < n2 && (tid & 3) == 0) {
      v[l_k[e]] = dr;
      out_dist[(int64_t)q * K + l_k[e]] = dr;
    }
  }
  if (tid == 0) atomicAdd(&A.stats[0], n2);
  __syncthreads();
  rank_pass();
}

extern "C" int qpg_percode_select_guarded_f64(qpg_ctx* ctx, void* stream, const double* D, int64_t ldD, int Q,
                                              const int16_t* cand_code, int64_t C, int K, double absent,
                                              int32_t idx_base, double* out_dist, int32_t* out_idx, int16_t* out_rank,
                                              int q_block, int64_t block_stride, const float* base, int T, int F,
                                              const int32_t* cand_t, int G, int n_taps, int tap_stride,
                                              const float* q32, double eps, int32_t* stats, int base_is_f16) {
  QPG_REQUIRE(ctx && D && (cand_code || C == 0) && out_dist && out_idx && base && cand_t && q32 && stats,
              "qpg_percode_select_guarded_f64: null pointer");
  QPG_REQUIRE(Q >= 0 && C >= 0 && K > 0 && K <= 2048 && ldD >= C && C + (int64_t)idx_base < 0x7fffffffll && T > 0 &&
                  F > 0 && G > 0 && n_taps > 0 && tap_stride > 0 && eps >= 0.0 && C % G == 0,
              "qpg_percode_select_guarded_f64: bad size");
  QPG_REQUIRE(q_block >= 0 && (q_block == 0 || (!out_rank && Q % q_block == 0 && block_stride % 8 == 0)),
              "qpg_percode_select_guarded_f64: block layout needs Q %% q_block == 0 and no rank output");
  if (Q == 0) return QPG_OK;
  GuardArgs A;
  A.base = base; A.half = base_is_f16; A.q32 = q32; A.cand_t = cand_t; A.T = T; A.F = F; A.G = G; A.n_taps = n_taps;
  A.tap_stride = tap_stride; A.eps = eps; A.stats = stats;
  size_t sh = 24 * (size_t)K + GUARD_LIST * 16 + 16;
  const int sort_ranks = sh + 12 * (size_t)rank_sort_pow2(K) <= 64 * 1024;          // room for the rank sort's scratch
  if (sort_ranks) sh += 12 * (size_t)rank_sort_pow2(K);
  hipLaunchKernelGGL(percode_select_guarded_f64_kernel, dim3(Q), dim3(1024), sh, qpg_stream(stream), D, ldD, cand_code,
                     C, K, absent, idx_base, out_dist, out_idx, out_rank, q_block, block_stride, A, sort_ranks);
  QPG_LAUNCH_CHECK("percode_select_guarded_f64_kernel");
  return QPG_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Select for the MIXED-PRECISION sweep (qpg_audio_cosine_mx): the matrix holds distances with a guaranteed error
// <= E = QPG_AUDIO_MX_ERR.  Two values further apart than eps1 = 2E (+ margin) are ordered like the exact values, so
// only comparisons inside an eps1 band can be wrong, and those are re-evaluated before they are used:
//   tier 1  every candidate within eps1 of its code's minimum (when there are two or more) and the winner of every
//           code whose minimum lies within eps1 of its rank neighbour gets an f64 dot product (one wave per pair,
//           accurate to ~1e-15 like the f64 sweep's value);
//   tier 2  the near-tie guard of qpg_percode_select_guarded_f64 on those f64 values (band eps2, the reference's own
//           arithmetic) — anything closer than eps2 in exact terms was inside the eps1 band and is in the list.
// A tier-1 value and an untouched sweep value are always >= eps1 - E apart, so mixing them in one ranking is safe.
// out_dist therefore carries the sweep value (error <= E) for untouched codes and the refined value otherwise;
// out_idx / out_rank are what the f64 path returns.
// ---------------------------------------------------------------------------------------------------------------
#define MIX_LIST 2048
#define MIX_LIST2 256
#ifndef MIX_POT
#define MIX_POT 6144    // candidates that were within the band of their code's minimum SO FAR when pass 1 visited them
#endif                  // (-DMIX_POT=64 makes every test take the overflow path: verified once, experiments/audio_mx/README.md)

// f64 dot product of (query q, local candidate c) by one wave; every lane returns the sum.  F % 256 == 0 (the WavLM
// width) and n_taps == 6: a lane owns the 16-byte piece lane + 64*j of every tap; ALL candidate loads of the pair
// (24 at F = 1024) are issued before the first use — one wave per pair is latency-bound otherwise (measured: 18 us per
// pair with a tap's loads at a time) — and the query row is read from LDS (`qlds`, staged once per block).
template <int NPER>
__device__ __forceinline__ double pair_dot_fast_f64(const GuardArgs& A, const float* qlds, int64_t c_local, int lane) {
  const int j = (int)(c_local / A.G), g = (int)(c_local - (int64_t)j * A.G);
  const int t0 = A.cand_t[g];
  const int64_t crow = (int64_t)j * A.T * A.F + lane * 4;
  f32x4 cv[6][NPER];
#pragma unroll
  for (int tap = 0; tap < 6; ++tap) {
    const int t = t0 + tap * A.tap_stride;
    const bool ok = t < A.T;
    const int64_t cp = crow + (int64_t)(ok ? t : t0) * A.F;
#pragma unroll
    for (int u = 0; u < NPER; ++u) {
      cv[tap][u] = guard_val4(A, cp + 256 * u);
      if (!ok) cv[tap][u] = (f32x4){0.f, 0.f, 0.f, 0.f};      // zero padding past the end of the window
    }
  }
  double s0 = 0.0, s1 = 0.0;
#pragma unroll
  for (int tap = 0; tap < 6; ++tap)
#pragma unroll
    for (int u = 0; u < NPER; ++u) {
      const f32x4 qv = *reinterpret_cast<const f32x4*>(qlds + tap * A.F + 256 * u + lane * 4);
      s0 += (double)qv.x * (double)cv[tap][u].x;
      s1 += (double)qv.y * (double)cv[tap][u].y;
      s0 += (double)qv.z * (double)cv[tap][u].z;
      s1 += (double)qv.w * (double)cv[tap][u].w;
    }
  double s = s0 + s1;
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  return s;
}

__device__ __forceinline__ double pair_dot_wave_f64(const GuardArgs& A, int q, int64_t c_local, int lane) {
  const int D = A.n_taps * A.F;
  const int j = (int)(c_local / A.G), g = (int)(c_local - (int64_t)j * A.G);
  const int t0 = A.cand_t[g];
  const float* qrow = A.q32 + (int64_t)q * D;
  const int64_t crow = (int64_t)j * A.T * A.F;
  double s0 = 0.0, s1 = 0.0;
  for (int e = lane * 4; e < D; e += 256) {                   // F % 4 == 0: a 16-byte piece never straddles a tap
    const int tap = e / A.F, f = e - tap * A.F;
    const int t = t0 + tap * A.tap_stride;
    const f32x4 qv = *reinterpret_cast<const f32x4*>(qrow + e);
    f32x4 cv = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (t < A.T) cv = guard_val4(A, crow + (int64_t)t * A.F + f);
    s0 += (double)qv.x * (double)cv.x;
    s1 += (double)qv.y * (double)cv.y;
    s0 += (double)qv.z * (double)cv.z;
    s1 += (double)qv.w * (double)cv.w;
  }
  double s = s0 + s1;
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  return s;
}

// Workspace of the cut launch: per query [parking space | streamed state].
//   parking space   best u64 K | v f64 K | besti u32 K | near u32 K | n, order-valid, pad (16 B) | l_c i32 L | l_k i32 L | l_d f64 L | order i16 K
//   streamed state  inv u32 [MIX_SPLIT][K] (NOT of each slice's minimum's order key, 0 = the slice has no candidate of the
//                   code; every slice stores its whole row - round 6: 196 K device-scope atomicMax per clip into one
//                   shared row were the slowest thing the streaming pass did, experiments/epilogue_atomics) | survivors of
//                   each slice i32 [MIX_SPLIT] (-1: its list overflowed) | pot_c i32 P | pot_d f32 P | pot_k i16 P
//                   (P = MIX_SPLIT x MIX_SPOT: slice s owns entries [s MIX_SPOT, (s + 1) MIX_SPOT) - no global counter)
// The streamed states are ALL-ZERO between launches (the list kernel resets what it consumes; offsets do not depend on
// Q): the caller zero-fills the workspace once.
#ifndef MIX_INV_ATOMIC
#define MIX_INV_ATOMIC 0  // 1 (A/B builds only): round 5's merge of the slices' minima - atomicMax into one shared row
#endif
#define MIX_SPLIT 8       // blocks per query streaming the row
#define MIX_SPOT 3072     // potential band members a slice can hold in LDS
#define MIX_GPOT (MIX_SPLIT * MIX_SPOT)
__host__ __device__ __forceinline__ size_t mix_park_bytes(int K) {      // (... | order i16 K: the cut's sorted order, round 5)
  return 24 * (size_t)K + 16 + 8 * (size_t)MIX_LIST + 8 * (size_t)MIX_LIST + ((2 * (size_t)K + 15) & ~(size_t)15);
}
__host__ __device__ __forceinline__ size_t mix_park_order_off(int K) {
  return 24 * (size_t)K + 16 + 8 * (size_t)MIX_LIST + 8 * (size_t)MIX_LIST;
}
__host__ __device__ __forceinline__ size_t mix_ws_stride(int K) {          // bytes of one query's space (multiple of 16)
  return ((mix_park_bytes(K) + 4 * (size_t)K * MIX_SPLIT + 4 * MIX_SPLIT + 10 * (size_t)MIX_GPOT) + 15) & ~(size_t)15;
}
struct MixStream {        // one query's streamed state
  unsigned int* inv;
  int* cnt;               // [MIX_SPLIT] survivors of each slice (-1: its LDS list overflowed)
  int* pot_c;
  float* pot_d;
  int16_t* pot_k;
};
__host__ __device__ __forceinline__ unsigned char* mix_query_base(unsigned char* ws, int q, int K) {
  return ws + (size_t)q * mix_ws_stride(K);
}
__host__ __device__ __forceinline__ MixStream mix_stream_of(unsigned char* ws, int q, int K) {
  unsigned char* b = mix_query_base(ws, q, K) + mix_park_bytes(K);
  MixStream m;
  m.inv = reinterpret_cast<unsigned int*>(b);
  m.cnt = reinterpret_cast<int*>(b + 4 * (size_t)K * MIX_SPLIT);
  m.pot_c = m.cnt + MIX_SPLIT;
  m.pot_d = reinterpret_cast<float*>(m.pot_c + MIX_GPOT);
  m.pot_k = reinterpret_cast<int16_t*>(m.pot_d + MIX_GPOT);
  return m;
}


// -DQPG_SELECT_PROF (experiments/select_prof): block 0 stamps the 100 MHz wall clock at the section boundaries of the
// mixed select; qpg_debug_select_prof copies the stamps out.  Not in the product build (-DQPG_SELECT_PROF implies a
// hooks build: experiments/select_prof/build.sh).
#ifdef QPG_SELECT_PROF
__device__ long long qpg_select_prof_buf[6][16];      // [phase]: wall clock; [3 + phase]: shader clock (s_memtime)
#define SEL_STAMP(i)                                                                          \
  do {                                                                                        \
    __syncthreads();                                                                          \
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {                             \
      qpg_select_prof_buf[phase][i] = wall_clock64();                                         \
      qpg_select_prof_buf[3 + phase][i] = clock64();                                          \
    }                                                                                         \
  } while (0)
extern "C" int qpg_debug_select_prof(long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(qpg_select_prof_buf), sizeof(long long) * 96) == hipSuccess ? 0 : -1;
}
#else
#define SEL_STAMP(i)
#endif

// Tier-1 dot products of every query's list, on the whole GPU: grid (Q, RB), 4 waves per block, wave g of the 4 RB of a
// query takes list entries g, g + 4 RB, ...  RB = 64 at Q = 48: 256 waves per query, one pair each for any list up to 256
// entries (round 2 gave a query 64 waves: the longest list, ~150 entries, took three rounds of ~12 us - a pair is 24 KB of
// gathered rows; the 3 400 pairs of a clip move 83 MB in those 12 us, the HBM rate).  Waves beyond the list leave at once.
// (A flat list over all queries with one global counter was tried: 4 096 waves reading ONE counter address spend 13 us
// queueing on its L2 channel.)
template <int NPER>
__global__ __launch_bounds__(256) void select_refine_kernel(GuardArgs A, int K, const double* __restrict__ cn2,
                                                            const double* __restrict__ qn2, unsigned char* __restrict__ ws,
                                                            int fast) {
  const int q = blockIdx.x, lane = threadIdx.x & 63;
  const int g = blockIdx.y * 4 + (threadIdx.x >> 6), ng = gridDim.y * 4;
  unsigned char* wq = mix_query_base(ws, q, K);
  const int* w_n = reinterpret_cast<const int*>(wq + 24 * (size_t)K);
  const int* w_lc = w_n + 4;
  double* w_ld = reinterpret_cast<double*>(const_cast<int*>(w_lc) + 2 * MIX_LIST);
  const int n = w_n[0];
  if (g >= n) return;
  const double qq = qn2[q];
  const float* qrow = A.q32 + (int64_t)q * A.n_taps * A.F;
  for (int e = g; e < n; e += ng) {
    const int c = w_lc[e];
    const double dot = fast ? pair_dot_fast_f64<NPER>(A, qrow, c, lane) : pair_dot_wave_f64(A, q, c, lane);
    if (lane == 0) w_ld[e] = cosine_from_dot(dot, qq, cn2[c]);
  }
}

// The streaming pass of the select, MIX_SPLIT blocks per query (a single CU pulls ~25 GB/s: one block per query spent
// 21 us on the row's 320 KB, with 48 of the 256 CUs busy).  Block (q, s) streams slice s of row q: per-code minimum in
// LDS (4-byte keys), the candidates within eps1 of their code's minimum SO FAR remembered; at the end those still within
// eps1 of the slice's FINAL minimum are appended to the query's list in the workspace (candidate, value, code) and the
// slice's minima merged into the query's table (atomicMax of the inverted key).  The list kernel (phase 1 of
// percode_select_mixed_f64_kernel, `pre` set) starts from that state instead of streaming.
__global__ __launch_bounds__(1024) void mixed_stream_kernel(const float* __restrict__ D, int64_t ldD,
                                                           const int16_t* __restrict__ cand_code, int64_t C, int K,
                                                           double eps1, unsigned char* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned int* best32 = reinterpret_cast<unsigned int*>(smem);                         // [K]
  int* pc = reinterpret_cast<int*>(best32 + K);                                         // [MIX_SPOT]
  float* pd = reinterpret_cast<float*>(pc + MIX_SPOT);                                  // [MIX_SPOT]
  int16_t* pk = reinterpret_cast<int16_t*>(pd + MIX_SPOT);                              // [MIX_SPOT]
  __shared__ int n_pot, n_keep;
  const int q = blockIdx.x, sl = blockIdx.y, tid = threadIdx.x;
#ifdef QPG_SELECT_PROF
  const int phase = 0;
#endif
  const float* row = D + (int64_t)q * ldD;
  const int64_t chunk = (((C + MIX_SPLIT - 1) / MIX_SPLIT) + 63) & ~(int64_t)63;
  const int64_t c0 = (int64_t)sl * chunk, c1 = c0 + chunk < C ? c0 + chunk : C;
  for (int k = tid; k < K; k += blockDim.x) best32[k] = 0xffffffffu;
  if (tid == 0) {
    n_pot = 0;
    n_keep = 0;
  }
  __syncthreads();
  SEL_STAMP(0);
  auto visit = [&](int64_t c, float dv, int cd) {
    if ((unsigned)cd >= (unsigned)K) return;
    const unsigned int key = order_key(dv);
    unsigned int old = *reinterpret_cast<volatile unsigned int*>(&best32[cd]);
    if (key < old) old = atomicMin(&best32[cd], key);
    if ((double)dv <= (double)key_value(old < key ? old : key, 0.f) + eps1) {
      const int pp = atomicAdd(&n_pot, 1);
      if (pp < MIX_SPOT) {
        pc[pp] = (int)c;
        pd[pp] = dv;
        pk[pp] = (int16_t)cd;
      }
    }
  };
  typedef int16_t c16x4 __attribute__((ext_vector_type(4)));
  const bool vec_ok = (ldD % 4) == 0 && (reinterpret_cast<uintptr_t>(D) % 16) == 0 &&
                      (reinterpret_cast<uintptr_t>(cand_code) % 8) == 0;
  const int64_t cv = vec_ok ? c0 + ((c1 - c0) / 4) * 4 : c0;
  // What bounds a slice is the number of load ROUND TRIPS (~1 us each under load), not its bytes: 256 threads x 4
  // iterations in flight took 7 round trips and 10 us for 6 656 candidates.  1024 threads x UN = 2: the whole slice of the
  // bench geometry is in flight at once.  A batch then goes through in two sweeps - all table reads (LDS round trips,
  // issued back to back), then the compares.
  constexpr int UN = 2;
  for (int64_t b = c0 + (int64_t)tid * 4; b < cv; b += (int64_t)blockDim.x * 4 * UN) {
    f32x4 d[UN];
    c16x4 cd[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int64_t c = b + (int64_t)u * blockDim.x * 4;
      d[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
      cd[u] = (c16x4){-1, -1, -1, -1};
      if (c < cv) {
        d[u] = *reinterpret_cast<const f32x4*>(row + c);
        cd[u] = *reinterpret_cast<const c16x4*>(cand_code + c);
      }
    }
    unsigned int key[UN][4], old[UN][4];
#pragma unroll
    for (int u = 0; u < UN; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int code = cd[u][e];
        key[u][e] = order_key(d[u][e]);
        old[u][e] = (unsigned)code < (unsigned)K ? *reinterpret_cast<volatile unsigned int*>(&best32[code]) : 0u;
      }
#pragma unroll
    for (int u = 0; u < UN; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int code = cd[u][e];
        if ((unsigned)code >= (unsigned)K) continue;
        unsigned int o = old[u][e];
        if (key[u][e] < o) o = atomicMin(&best32[code], key[u][e]);
        const unsigned int lo = o < key[u][e] ? o : key[u][e];
        if ((double)d[u][e] <= (double)key_value(lo, 0.f) + eps1) {
          const int pp = atomicAdd(&n_pot, 1);
          if (pp < MIX_SPOT) {
            pc[pp] = (int)(b + (int64_t)u * blockDim.x * 4 + e);
            pd[pp] = d[u][e];
            pk[pp] = (int16_t)code;
          }
        }
      }
  }
  for (int64_t c = cv + tid; c < c1; c += blockDim.x) visit(c, row[c], cand_code[c]);
  __syncthreads();
  SEL_STAMP(1);
  MixStream m = mix_stream_of(ws, q, K);
  if (MIX_INV_ATOMIC) {
    for (int k = tid; k < K; k += blockDim.x)
      if (best32[k] != 0xffffffffu) atomicMax(&m.inv[k], ~best32[k]);
  } else {
    for (int k = tid; k < K; k += blockDim.x) m.inv[(size_t)sl * K + k] = ~best32[k];     // (0: no candidate in this slice)
  }
  SEL_STAMP(2);
  const int np = n_pot;
  if (np > MIX_SPOT) {                        // the list kernel then streams the row itself (its own fallback)
    if (tid == 0) m.cnt[sl] = -1;
    return;
  }
  // survivors: within eps1 of the slice's final minimum (a superset of what the row's final minimum keeps), into this
  // slice's own segment of the query's list
  const int seg = sl * MIX_SPOT;
  for (int e = tid; e < np; e += blockDim.x) {
    if (!((double)pd[e] <= (double)key_value(best32[pk[e]], 0.f) + eps1)) continue;
    const int pos = seg + atomicAdd(&n_keep, 1);
    m.pot_c[pos] = pc[e];
    m.pot_d[pos] = pd[e];
    m.pot_k[pos] = pk[e];
  }
  __syncthreads();
  if (tid == 0) m.cnt[sl] = n_keep;
  SEL_STAMP(3);
}

template <typename DT>
__global__ __launch_bounds__(1024) void percode_select_mixed_f64_kernel(
    const DT* __restrict__ D, int64_t ldD, const int16_t* __restrict__ cand_code, int64_t C, int K, double absent,
    int32_t idx_base, double* __restrict__ out_dist, int32_t* __restrict__ out_idx, int16_t* __restrict__ out_rank,
    int q_block, int64_t block_stride, GuardArgs A, double eps1, const double* __restrict__ cn2,
    const double* __restrict__ qn2, int use_qlds, int phase, unsigned char* __restrict__ ws, int pre, RankCut RC) {
  // phase 0: everything in this launch (tier-1 dot products by this block's 16 waves: one CU per query).
  // pre (phase 1, f32 matrix): mixed_stream_kernel has streamed the row - per-code minima and the potential band members
  // are in the workspace; this launch starts at pass 2.
  // phase 1 / 2: the launch is cut at the tier-1 list — phase 1 parks its state in `ws`, select_refine_kernel computes
  // the listed dot products on ALL CUs (a single CU pulls ~26 GB/s; the list of a 48-query clip is ~110 MB of rows),
  // phase 2 picks the state up again.
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned long long* best = reinterpret_cast<unsigned long long*>(smem);               // [K] order key of the minimum
  double* v = reinterpret_cast<double*>(smem + 8 * (size_t)K);                          // [K] value table being ranked
  unsigned int* besti = reinterpret_cast<unsigned int*>(smem + 16 * (size_t)K);         // [K] global candidate index
  unsigned int* near_ = besti + K;            // [K] band population; >= 2: "touched" (all band members are listed)
  int* s_code = reinterpret_cast<int*>(near_ + K);                                      // [K] code at rank r / scratch
  unsigned char* refd = reinterpret_cast<unsigned char*>(s_code + K);                   // [K] value is reference-arithmetic
  unsigned char* tail = smem + 32 * (size_t)K;
  double* l_d = reinterpret_cast<double*>(tail);                                        // [MIX_LIST]
  int* l_c = reinterpret_cast<int*>(tail + 8 * MIX_LIST);                               // [MIX_LIST] local candidate
  int* l_k = l_c + MIX_LIST;                                                            // [MIX_LIST] code
  int* l2 = l_k + MIX_LIST;                                                             // [MIX_LIST2] tier-2 entries
  int* ctl = l2 + MIX_LIST2;     // [0] list length, [1] multi-member band seen, [2] tier-2 n, [3] band size, [4] potentials
  int* p_c = ctl + 8;                                        // [MIX_LIST] every band member seen by pass 2 (candidate)
  int16_t* p_k = reinterpret_cast<int16_t*>(p_c + MIX_LIST);                            // [MIX_LIST] its code
  int* rk = reinterpret_cast<int*>(p_k + MIX_LIST);          // [K] rank counters
  int* pot_c = rk + K;                                       // [MIX_POT] pass 1's potential band members (candidate)
  int16_t* pot_k = reinterpret_cast<int16_t*>(pot_c + MIX_POT);                         // [MIX_POT] their codes
  float* qlds = reinterpret_cast<float*>(pot_k + MIX_POT);   // [n_taps*F] this query's row (fast tier-1 path only)
  const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nwv = blockDim.x >> 6;
  const DT* row = D + (int64_t)q * ldD;                       // DT = float: the sweep stored its matrix in f32 (half the bytes
  if (q_block > 0) {                                            // of the two streaming passes, +1.2e-7 inside the bound)
    const int64_t shift = (int64_t)(q / q_block) * block_stride;
    const int64_t rowoff = (int64_t)(q % q_block) * K - (int64_t)q * K;
    out_dist = reinterpret_cast<double*>(reinterpret_cast<unsigned char*>(out_dist) + shift) + rowoff;
    out_idx = reinterpret_cast<int32_t*>(reinterpret_cast<unsigned char*>(out_idx) + shift) + rowoff;
  }
  unsigned char* wq = ws ? mix_query_base(ws, q, K) : nullptr;                // this query's parking space
  for (int k = tid; k < K; k += blockDim.x) {
    best[k] = ~0ull;
    besti[k] = 0xffffffffu;
    near_[k] = 0;
    refd[k] = 0;
  }
  if (tid < 8) ctl[tid] = 0;
  __syncthreads();
  SEL_STAMP(0);
  // rank of every code in the value table v (stable: value, then code); s_code[r] = code at rank r
  // (sorted, not counted: block_sorted_ranks; its scratch aliases p_c / p_k, which are dead once list (a) is built)
  unsigned long long* skey = reinterpret_cast<unsigned long long*>(p_c);      // [512] (K <= 512: 6 KB of p_c | p_k's 12)
  int* scode = reinterpret_cast<int*>(skey + 512);
  auto rank_pass = [&](bool store) {
    block_sorted_ranks(v, K, skey, scode, [&](int k, int r) {
      if (store && out_rank) out_rank[(int64_t)q * K + k] = (int16_t)r;
      s_code[r] = k;
    });
  };
  int n = 0;
  bool order_parked = false;                                   // (the cut's sorted order went to the parking space)
  if (phase != 2) {
  constexpr int VE = 16 / (int)sizeof(DT);                    // matrix elements per 16-byte load
  typedef DT vecD __attribute__((ext_vector_type(VE)));
  typedef int16_t vecC __attribute__((ext_vector_type(VE)));
  const bool vec_ok = (ldD % VE) == 0 && (reinterpret_cast<uintptr_t>(D) % 16) == 0 &&
                      (reinterpret_cast<uintptr_t>(cand_code) % (2 * VE)) == 0;
  const int64_t Cv = vec_ok ? (C / VE) * VE : 0;
  // Pass 1: per-code minimum (ds_min_rtn_u64).  The value the atomic returns is the code's minimum SO FAR, which is
  // never below the final one: a candidate further than eps1 above it cannot be in the final band, so only the others
  // - the successive minima of a code (~ln(n) of its n candidates in random order) and what lies within eps1 of them -
  // are remembered, and "pass 2" below visits that list instead of streaming the row a second time.  A list that
  // overflows (adversarially ordered or crowded rows) falls back to the second pass over the row.
  // (the table is READ first and the atomic issued only for a new minimum - a code's i-th candidate is one with
  // probability 1/i; a stale read is never below the true running minimum, so the test below only gets more inclusive.
  // The block's LDS pipe is what bounds this pass: 53 248 gathers into the key table.  With an f32 matrix the pass keeps
  // 4-byte keys in `s_code` (free until the ranks) - half the LDS words per gather - and widens them afterwards.)
  constexpr bool K32 = sizeof(DT) == 4;
  unsigned int* best32 = reinterpret_cast<unsigned int*>(s_code);
  const bool streamed = K32 && pre;
  MixStream ms = {};
  if (streamed) ms = mix_stream_of(ws, q, K);
  if (K32 && !streamed) {
    for (int k = tid; k < K; k += blockDim.x) best32[k] = 0xffffffffu;
    __syncthreads();
  }
  auto pass1 = [&](int64_t c, DT dv, int cd) {
    if ((unsigned)cd >= (unsigned)K) return;
    const double d = (double)dv;
    double lo;
    if constexpr (K32) {
      const unsigned int key = order_key((float)dv);
      unsigned int old = *reinterpret_cast<volatile unsigned int*>(&best32[cd]);
      if (key < old) old = atomicMin(&best32[cd], key);
      lo = (double)key_value(old < key ? old : key, 0.f);
    } else {
      const unsigned long long key = (unsigned long long)order_key(d);
      unsigned long long old = *reinterpret_cast<volatile unsigned long long*>(&best[cd]);
      if (key < old) old = atomicMin(&best[cd], key);
      lo = key_value(old < key ? old : key, 0.0);
    }
    if (d <= lo + eps1) {
      const int pp = atomicAdd(&ctl[4], 1);
      if (pp < MIX_POT) {
        pot_c[pp] = (int)c;
        pot_k[pp] = (int16_t)cd;
      }
    }
  };
  if (!streamed) {
#pragma unroll 4
    for (int64_t c = (int64_t)tid * VE; c < Cv; c += (int64_t)blockDim.x * VE) {
      const vecD d = *reinterpret_cast<const vecD*>(row + c);
      const vecC cd = *reinterpret_cast<const vecC*>(cand_code + c);
#pragma unroll
      for (int e = 0; e < VE; ++e) pass1(c + e, d[e], cd[e]);
    }
    for (int64_t c = Cv + tid; c < C; c += blockDim.x) pass1(c, row[c], cand_code[c]);
    __syncthreads();
  }
  int g_pot = 0;                              // streamed: entries of the workspace list (or the overflow mark)
  if (K32) {
    for (int k = tid; k < K; k += blockDim.x) {
      unsigned int b;
      if (streamed && MIX_INV_ATOMIC) {
        b = ~ms.inv[k];
        ms.inv[k] = 0;                        // (left all-zero for the next launch)
      } else if (streamed) {
        unsigned int iv[MIX_SPLIT], mx = 0u;  // the slices' rows: independent loads, one round trip
#pragma unroll
        for (int sl = 0; sl < MIX_SPLIT; ++sl) iv[sl] = ms.inv[(size_t)sl * K + k];
#pragma unroll
        for (int sl = 0; sl < MIX_SPLIT; ++sl) mx = iv[sl] > mx ? iv[sl] : mx;
        b = ~mx;
      } else {
        b = best32[k];
      }
      best[k] = b == 0xffffffffu ? ~0ull : (unsigned long long)order_key((double)key_value(b, 0.f));
    }
    if (streamed) {
#pragma unroll
      for (int sl = 0; sl < MIX_SPLIT; ++sl) {
        const int ns = ms.cnt[sl];
        g_pot = (ns < 0 || g_pot < 0) ? -1 : g_pot + ns;     // a slice's list overflowed: pass 2 streams the row
      }
    }
    __syncthreads();
  }
  SEL_STAMP(1);
  auto pass2 = [&](int64_t c, double d, int cd) {
    if ((unsigned)cd >= (unsigned)K) return;
    const unsigned long long bk = best[cd];
    if ((unsigned long long)order_key(d) == bk) atomicMin(&besti[cd], (unsigned int)(c + idx_base));
    else if (d > key_value(bk, 0.0) + eps1) return;
    if (atomicAdd(&near_[cd], 1u) == 1u) ctl[1] = 1;
    const int pp = atomicAdd(&ctl[3], 1);                               // remember the band member (K winners + the rest)
    if (pp < MIX_LIST) {
      p_c[pp] = (int)c;
      p_k[pp] = (int16_t)cd;
    }
  };
  if (streamed && g_pot >= 0) {
    // the slices' segments as one index space: thread t takes entries t, t + blockDim, ... of the concatenation; all of a
    // thread's entries (<= 4 at the usual ~4 000 survivors) are loaded before the first is used
    int seg_end[MIX_SPLIT];
    {
      int acc = 0;
#pragma unroll
      for (int sl = 0; sl < MIX_SPLIT; ++sl) {
        acc += ms.cnt[sl];
        seg_end[sl] = acc;
      }
    }
    constexpr int UE = 4;
    for (int e0 = tid; e0 < g_pot; e0 += (int)blockDim.x * UE) {
      int pc_[UE], pk_[UE];
      float pd_[UE];
#pragma unroll
      for (int u = 0; u < UE; ++u) {
        const int e = e0 + u * (int)blockDim.x;
        pk_[u] = -1;
        if (e < g_pot) {
          int sl = 0, first = 0;                              // the segment e falls into and its first index
#pragma unroll
          for (int i = 0; i < MIX_SPLIT - 1; ++i)
            if (e >= seg_end[i]) {
              sl = i + 1;
              first = seg_end[i];
            }
          const int pos = sl * MIX_SPOT + e - first;
          pc_[u] = ms.pot_c[pos];
          pd_[u] = ms.pot_d[pos];
          pk_[u] = ms.pot_k[pos];
        }
      }
#pragma unroll
      for (int u = 0; u < UE; ++u)
        if (pk_[u] >= 0) pass2(pc_[u], (double)pd_[u], pk_[u]);
    }
  } else if (!streamed && ctl[4] <= MIX_POT) {
    const int npot = ctl[4];
    for (int e = tid; e < npot; e += blockDim.x) pass2(pot_c[e], (double)row[pot_c[e]], pot_k[e]);
  } else {
#pragma unroll 4
    for (int64_t c = (int64_t)tid * VE; c < Cv; c += (int64_t)blockDim.x * VE) {
      const vecD d = *reinterpret_cast<const vecD*>(row + c);
      const vecC cd = *reinterpret_cast<const vecC*>(cand_code + c);
#pragma unroll
      for (int e = 0; e < VE; ++e) pass2(c + e, (double)d[e], cd[e]);
    }
    for (int64_t c = Cv + tid; c < C; c += blockDim.x) pass2(c, (double)row[c], cand_code[c]);
  }
  __syncthreads();
  if (streamed && tid < MIX_SPLIT) ms.cnt[tid] = 0;            // (left all-zero for the next launch)
  SEL_STAMP(2);
  for (int k = tid; k < K; k += blockDim.x) v[k] = besti[k] != 0xffffffffu ? key_value(best[k], 0.0) : absent;
  __syncthreads();
  // ---- walk-relevance cut (RankCut; streamed f32 path with ranks only).  With E the sweep's bound (eps1 >= 2 E), a
  // code's true rank lies in [lo, hi] = its rank in the table of sweep values -/+ the number of codes within eps1 below /
  // above it.  U(p) = min over the `probe` best-ranked codes c of  pos_rank[p][c] + 0.05 freq_rank[c] + hi(c)  is an upper
  // bound on the smallest fused score of previous code p (the top_n-th smallest for top_n = 2), Umax its maximum over p.
  // A code with lo > Umax scores above the winner(s) of EVERY previous code whatever its exact rank in [lo, hi] is, so
  // the walk never reads its row: it is left out of both lists (its table entries keep their sweep values, its rank is
  // its position among those) unless a code that CAN be read lies within eps1 of it (whose rank needs their order).
  const bool cut = RC.pos_t != nullptr && streamed && out_rank != nullptr && phase == 1;
  unsigned char* need2 = nullptr;
  if (cut) {
    unsigned char* cs = reinterpret_cast<unsigned char*>(pot_c);      // (the potential lists of the unstreamed path: free)
    const int Kp = rank_sort_pow2(K);
    unsigned long long* skey2 = reinterpret_cast<unsigned long long*>(cs);
    int* scode2 = reinterpret_cast<int*>(skey2 + Kp);
    int16_t* lo_ = reinterpret_cast<int16_t*>(scode2 + Kp);
    int16_t* hi_ = lo_ + K;
    unsigned char* need = reinterpret_cast<unsigned char*>(hi_ + K);
    need2 = need + K;
    double* red = reinterpret_cast<double*>(cs + 12 * (size_t)Kp + 8 * (size_t)((K + 7) / 8 * 8));   // [17]
    const double f05 = tid < K ? (double)RC.freq[tid] * 0.05 : 0.0;   // (in flight under the sort)
    block_sorted_ranks(v, K, skey2, scode2, [&](int k, int r) {
      s_code[r] = k;
      rk[k] = r;
    });
    __syncthreads();
    // (parked for the merge launch: the refined table differs from this one in a few near-tied entries, so its ranks are
    // this order + a local repair instead of a second sort of the 512 values)
    for (int r = tid; r < K; r += blockDim.x) reinterpret_cast<int16_t*>(wq + mix_park_order_off(K))[r] = (int16_t)s_code[r];
    order_parked = true;
    SEL_STAMP(13);
    for (int r = tid; r < K; r += blockDim.x) {
      const int k = s_code[r];
      int b = 0, a = 0;
      if (besti[k] != 0xffffffffu) {
        while (r - 1 - b >= 0 && v[k] - v[s_code[r - 1 - b]] <= eps1) ++b;
        while (r + 1 + a < K && besti[s_code[r + 1 + a]] != 0xffffffffu && v[s_code[r + 1 + a]] - v[k] <= eps1) ++a;
      }
      lo_[k] = (int16_t)(r - b);
      hi_[k] = (int16_t)(r + a);
    }
    __syncthreads();
    SEL_STAMP(14);
    // the probe codes' constants in LDS, then every previous code p takes its column entries of the probe codes: thread
    // (p, half) reads 32 of the (at most 64) entries with 32 INDEPENDENT loads in flight - a loop of dependent round trips,
    // one per probe code, cost 28 us here - and the two halves meet in LDS
    int np = RC.probe < K ? RC.probe : K;
    np = np < 64 ? np : 64;
    double* pf_ = red + 17;                                            // [64] 0.05 x the probe code's frequency rank
    double* ph_ = pf_ + 64;                                            // [64] hi of its rank
    int* pk_ = reinterpret_cast<int*>(ph_ + 64);                       // [64] probe code (-1: absent / beyond the probe)
    double* pm_ = reinterpret_cast<double*>(pk_ + 64);                 // [K][2] the second half's two smallest scores
    if (tid < K) pm_[tid] = f05;                                       // (staged through pm_: free until the halves meet)
    __syncthreads();
    for (int r = tid; r < 64; r += blockDim.x) {
      const int k = r < np ? s_code[r] : -1;
      const bool ok = k >= 0 && besti[k] != 0xffffffffu;
      pk_[r] = ok ? k : -1;
      pf_[r] = ok ? pm_[k] : 0.0;
      ph_[r] = ok ? (double)hi_[k] : 0.0;
    }
    __syncthreads();
    double u = -1.0, m1 = __builtin_inf(), m2 = __builtin_inf();
    const int pp = tid < K ? tid : tid - K, half = tid < K ? 0 : 1;
    if (tid < 2 * K) {
      const int r0 = 32 * half;
      int kk[32];
      int16_t pv[32];
#pragma unroll
      for (int i = 0; i < 32; ++i) kk[i] = pk_[r0 + i];
#pragma unroll
      for (int i = 0; i < 32; ++i)                                      // UNCONDITIONAL (row 0 for a masked entry)
        pv[i] = RC.pos_t[(int64_t)(kk[i] < 0 ? 0 : kk[i]) * K + pp];
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        if (kk[i] < 0) continue;
        const double sc = ((double)pv[i] + pf_[r0 + i]) + ph_[r0 + i];
        if (sc < m1) {
          m2 = m1;
          m1 = sc;
        } else if (sc < m2) {
          m2 = sc;
        }
      }
      if (half) {
        pm_[2 * pp] = m1;
        pm_[2 * pp + 1] = m2;
      }
    }
    __syncthreads();
    if (tid < K) {
      const double a1 = pm_[2 * pp], a2 = pm_[2 * pp + 1];
      if (blockDim.x >= 2 * (unsigned)K) {                              // (a block too small for two halves: none here)
        if (a1 < m1) {
          m2 = m1 < a2 ? m1 : a2;
          m1 = a1;
        } else if (a1 < m2) {
          m2 = a1;
        }
      }
      u = RC.top_n >= 2 ? m2 : m1;
    }
    for (int o = 32; o > 0; o >>= 1) {
      const double t = __shfl_xor(u, o, 64);
      u = t > u ? t : u;
    }
    if (lane == 0) red[wv] = u;
    __syncthreads();
    if (tid == 0) {
      double m = red[0];
      for (int i = 1; i < nwv; ++i) m = red[i] > m ? red[i] : m;
      red[16] = m;
    }
    __syncthreads();
    const double umax = red[16];
    SEL_STAMP(15);
    for (int k = tid; k < K; k += blockDim.x) need[k] = (besti[k] != 0xffffffffu && (double)lo_[k] <= umax) ? 1 : 0;
    __syncthreads();
    for (int r = tid; r < K; r += blockDim.x) {
      const int k = s_code[r];
      unsigned char nd = need[k];
      if (!nd && besti[k] != 0xffffffffu)
        for (int rr = lo_[k]; rr <= hi_[k] && !nd; ++rr) nd = need[s_code[rr]];
      need2[k] = nd;
    }
    __syncthreads();
  }
  // ---- list (a): every member of a band with two or more members — from the members pass 2 remembered, or, if there
  // were more than it could hold, by a third pass over the row
  if (ctl[1] && ctl[3] <= MIX_LIST) {
    const int np = ctl[3];
    for (int e = tid; e < np; e += blockDim.x) {
      const int cd = p_k[e];
      if (near_[cd] < 2 || (need2 && !need2[cd])) continue;
      const int pos = atomicAdd(&ctl[0], 1);
      if (pos < MIX_LIST) {
        l_c[pos] = p_c[e];
        l_k[pos] = cd;
      }
    }
  } else if (ctl[1]) {
    for (int64_t c = tid; c < C; c += blockDim.x) {
      const int cd = cand_code[c];
      if ((unsigned)cd >= (unsigned)K || near_[cd] < 2 || (need2 && !need2[cd])) continue;
      if ((double)row[c] <= key_value(best[cd], 0.0) + eps1) {
        const int pos = atomicAdd(&ctl[0], 1);
        if (pos < MIX_LIST) {
          l_c[pos] = (int)c;
          l_k[pos] = cd;
        }
      }
    }
  }
  __syncthreads();
  SEL_STAMP(3);
  // ---- list (b): winners of codes whose minima lie within eps1 of ANOTHER code's (only needed when ranks are wanted:
  // without them the minima of different codes are never compared here).  Two values that close are rank neighbours, but
  // finding them does not need the ranks (a sort: 14 us here): every present code drops its value into a grid of cells
  // a little wider than eps1 (hashed into 2048 LDS counters that hold population and code sum); a code looks at its own
  // and the two adjacent cells: a single other occupant is tested directly (|dv| < eps1: a hash collision is then harmless -
  // without the test half of the codes were listed, 273 entries per query instead of ~70), two or more list the code
  // without a test (it then gets an f64 value like any other listed code: never wrong, only work).
  if (out_rank && cut) {
    // (the cut has sorted the table: a code has company iff its [lo, hi] is more than its own rank - neighbours within
    // eps1 INCLUSIVE, a superset of the grid's test below)
    const int16_t* lo_ = reinterpret_cast<const int16_t*>(reinterpret_cast<unsigned char*>(pot_c) + 12 * (size_t)rank_sort_pow2(K));
    const int16_t* hi_ = lo_ + K;
    for (int k = tid; k < K; k += blockDim.x) {
      if (besti[k] == 0xffffffffu || lo_[k] == hi_[k] || !need2[k]) continue;
      if (atomicMax(&near_[k], 2u) >= 2u) continue;                    // band-listed already
      const int pos = atomicAdd(&ctl[0], 1);
      if (pos < MIX_LIST) {
        l_c[pos] = (int)(besti[k] - (unsigned int)idx_base);
        l_k[pos] = k;
      }
    }
    __syncthreads();
    SEL_STAMP(4);
  } else if (out_rank) {
    unsigned int* cell = reinterpret_cast<unsigned int*>(p_c);          // [2048] (p_c / p_k are dead: list (a) is built)
    const double inv_w = 1.0 / (eps1 * 1.000001);
    auto slot = [](long long b) { return (unsigned int)(((unsigned long long)b * 0x9E3779B97F4A7C15ull) >> 53); };
    for (int i = tid; i < 2048; i += blockDim.x) cell[i] = 0;
    __syncthreads();
    // a counter holds (population << 16) + the sum of its codes: with one occupant, the occupant
    for (int k = tid; k < K; k += blockDim.x)
      if (besti[k] != 0xffffffffu) atomicAdd(&cell[slot((long long)floor(v[k] * inv_w))], 0x10000u + (unsigned int)k);
    __syncthreads();
    for (int k = tid; k < K; k += blockDim.x) {
      if (besti[k] == 0xffffffffu) continue;
      const long long b = (long long)floor(v[k] * inv_w);
      const unsigned int s0 = slot(b);
      bool company = false;
#pragma unroll
      for (int o = -1; o <= 1; ++o) {
        const unsigned int si = slot(b + o);
        if (o != 0 && si == s0) continue;                               // (the own counter is looked at once)
        unsigned int c = cell[si];
        if (si == s0) c -= 0x10000u + (unsigned int)k;                  // without this code itself
        const unsigned int pop = c >> 16;
        if (pop == 1) {                                                 // one other code (of this cell, or one that
          const int other = (int)(c & 0xffffu);                         // hashed here): the test itself
          company |= fabs(v[other] - v[k]) < eps1;
        } else if (pop > 1) {
          company = true;
        }
      }
      if (!company || (need2 && !need2[k])) continue;                  // (cut: the walk can not read this code's row)
      if (atomicMax(&near_[k], 2u) >= 2u) continue;                    // band-listed already
      const int pos = atomicAdd(&ctl[0], 1);
      if (pos < MIX_LIST) {
        l_c[pos] = (int)(besti[k] - (unsigned int)idx_base);
        l_k[pos] = k;
      }
    }
    __syncthreads();
    SEL_STAMP(4);
  }
  n = ctl[0];
  if (n > MIX_LIST) {
    n = MIX_LIST;
    if (tid == 0) atomicOr(&A.stats[1], 1);
  }
  }   // phase != 2
  SEL_STAMP(5);
  if (phase == 1) {            // park: [best u64 K][v f64 K][besti u32 K][near u32 K][n, pad][l_c i32 L][l_k i32 L][l_d f64 L]
    unsigned long long* w_best = reinterpret_cast<unsigned long long*>(wq);
    double* w_v = reinterpret_cast<double*>(wq + 8 * (size_t)K);
    unsigned int* w_bi = reinterpret_cast<unsigned int*>(wq + 16 * (size_t)K);
    unsigned int* w_nr = w_bi + K;
    int* w_n = reinterpret_cast<int*>(wq + 24 * (size_t)K);
    int* w_lc = w_n + 4;
    int* w_lk = w_lc + MIX_LIST;
    for (int k = tid; k < K; k += blockDim.x) {
      w_best[k] = best[k];
      w_v[k] = v[k];
      w_bi[k] = besti[k];
      w_nr[k] = near_[k];
    }
    for (int e = tid; e < n; e += blockDim.x) {
      w_lc[e] = l_c[e];
      w_lk[e] = l_k[e];
    }
    if (tid == 0) {
      w_n[0] = n;
      w_n[1] = order_parked ? 1 : 0;         // the parked order is this launch's
    }
    SEL_STAMP(6);
    return;
  }
  if (phase == 2) {
    const unsigned long long* w_best = reinterpret_cast<const unsigned long long*>(wq);
    const double* w_v = reinterpret_cast<const double*>(wq + 8 * (size_t)K);
    const unsigned int* w_bi = reinterpret_cast<const unsigned int*>(wq + 16 * (size_t)K);
    const unsigned int* w_nr = w_bi + K;
    const int* w_n = reinterpret_cast<const int*>(wq + 24 * (size_t)K);
    const int* w_lc = w_n + 4;
    const int* w_lk = w_lc + MIX_LIST;
    const double* w_ld = reinterpret_cast<const double*>(w_lk + MIX_LIST);
    n = w_n[0];
    for (int k = tid; k < K; k += blockDim.x) {
      best[k] = w_best[k];
      v[k] = w_v[k];
      besti[k] = w_bi[k];
      near_[k] = w_nr[k];
    }
    for (int e = tid; e < n; e += blockDim.x) {
      l_c[e] = w_lc[e];
      l_k[e] = w_lk[e];
      l_d[e] = w_ld[e];
    }
    if (RC.pos_t) {                 // cut lists: which codes have f64 values at all (the rank-level tier 2 asks)
      unsigned char* isref = reinterpret_cast<unsigned char*>(pot_c);
      for (int k = tid; k < K; k += blockDim.x) isref[k] = 0;
      __syncthreads();
      for (int e = tid; e < n; e += blockDim.x) isref[w_lk[e]] = 1;
    }
    __syncthreads();
  }
  SEL_STAMP(7);
  if (n > 0) {
    // ---- tier 1: f64 dot products, one wave per listed pair (phase 2: already done by select_refine_kernel)
    const double qq = qn2[q];
    const bool fast = use_qlds != 0;
    if (phase == 0 && fast) {
      const int D4 = (A.n_taps * A.F) >> 2;
      const f32x4* src = reinterpret_cast<const f32x4*>(A.q32 + (int64_t)q * A.n_taps * A.F);
      for (int i = tid; i < D4; i += blockDim.x) reinterpret_cast<f32x4*>(qlds)[i] = src[i];
      __syncthreads();
    }
    for (int e = wv; e < n && phase == 0; e += nwv) {
      const int c = l_c[e];
      const double dot = fast ? pair_dot_fast_f64<4>(A, qlds, c, lane) : pair_dot_wave_f64(A, q, c, lane);
      if (lane == 0) l_d[e] = cosine_from_dot(dot, qq, cn2[c]);
    }
    for (int k = tid; k < K; k += blockDim.x)
      if (near_[k] >= 2) {
        rk[k] = (int)besti[k];                                 // kept for codes a TRUNCATED list has no entry of (below)
        best[k] = ~0ull;
        besti[k] = 0xffffffffu;
        s_code[k] = 0;                                         // scratch: members within eps2 of the refined minimum
      }
    __syncthreads();
    for (int e = tid; e < n; e += blockDim.x) atomicMin(&best[l_k[e]], (unsigned long long)order_key(l_d[e]));
    __syncthreads();
    // ---- tier 2 (candidate level): two or more members of one code within eps2 of its refined minimum
    if (A.eps > 0.0) {
      for (int e = tid; e < n; e += blockDim.x)
        if (l_d[e] <= key_value(best[l_k[e]], 0.0) + A.eps) atomicAdd(&s_code[l_k[e]], 1);
      __syncthreads();
      for (int e = tid; e < n; e += blockDim.x)
        if (s_code[l_k[e]] >= 2 && l_d[e] <= key_value(best[l_k[e]], 0.0) + A.eps) {
          const int pos = atomicAdd(&ctl[2], 1);
          if (pos < MIX_LIST2) l2[pos] = e;
        }
      __syncthreads();
      int n2 = ctl[2];
      if (n2 > MIX_LIST2) {
        n2 = MIX_LIST2;
        if (tid == 0) atomicOr(&A.stats[1], 1);
      }
      if (n2 > 0) {
        for (int i0 = 0; i0 < n2; i0 += blockDim.x / 4) {
          const int i = i0 + (tid >> 2);
          const int e = l2[i < n2 ? i : 0];
          const double dr = refine_pair_f64(A, q, l_c[e], tid & 3);
          if (i < n2 && (tid & 3) == 0) {
            l_d[e] = dr;
            refd[l_k[e]] = 1;
          }
        }
        __syncthreads();
        for (int k = tid; k < K; k += blockDim.x)
          if (refd[k]) best[k] = ~0ull;
        __syncthreads();
        // only the re-evaluated members can win such a code (the rest are > eps2 above them)
        for (int i = tid; i < n2; i += blockDim.x)
          atomicMin(&best[l_k[l2[i]]], (unsigned long long)order_key(l_d[l2[i]]));
        __syncthreads();
        if (tid == 0) atomicAdd(&A.stats[0], n2);
      }
    }
    for (int e = tid; e < n; e += blockDim.x)
      if ((unsigned long long)order_key(l_d[e]) == best[l_k[e]])
        atomicMin(&besti[l_k[e]], (unsigned int)(l_c[e] + idx_base));
    __syncthreads();
    // A touched code with no list entry exists only after a list overflow (flagged: the host re-matches the clip):
    // it keeps its sweep value and candidate, so that nothing below ranks a NaN or dereferences an empty slot
    // (round 2 did, and a crowded row - 3 000 near-copies under one code - faulted in the rank-level refine).
    for (int k = tid; k < K; k += blockDim.x)
      if (near_[k] >= 2) {
        if (best[k] != ~0ull) v[k] = key_value(best[k], 0.0);
        else besti[k] = (unsigned int)rk[k];
      }
    if (tid == 0) atomicAdd(&A.stats[2], n);
    __syncthreads();
  }
  for (int k = tid; k < K; k += blockDim.x) {
    const bool have = besti[k] != 0xffffffffu;
    out_dist[(int64_t)q * K + k] = v[k];
    out_idx[(int64_t)q * K + k] = have ? (int32_t)besti[k] : -1;
  }
  SEL_STAMP(8);
  if (!out_rank) return;
  if (tid == 0) ctl[2] = 0;
  __syncthreads();
  // Merge launch behind a cut list launch: the order of the sweep values is parked; the refined values differ from them
  // in the few re-evaluated codes, each by less than the band - the (value, code) order is restored by odd-even
  // transposition passes over the parked order (a pass = 2 barriers; one or two rounds on a typical row against the 45
  // stages of the sort), and a row that is not settled after 24 rounds (a flagged, crowded one) is sorted the ordinary way.
  bool repaired = false;
  if (phase == 2 && RC.pos_t && reinterpret_cast<const int*>(wq + 24 * (size_t)K)[1] == 1) {
    const int16_t* w_order = reinterpret_cast<const int16_t*>(wq + mix_park_order_off(K));
    for (int r = tid; r < K; r += blockDim.x) s_code[r] = w_order[r];
    __syncthreads();
    for (int round = 0; round < 24 && !repaired; ++round) {
      if (tid == 0) ctl[5] = 0;
      __syncthreads();
#pragma unroll
      for (int parity = 0; parity < 2; ++parity) {
        for (int r = 2 * tid + parity; r + 1 < K; r += 2 * blockDim.x) {
          const int ka = s_code[r], kb = s_code[r + 1];
          if (rank_gt(rank_sort_key(v[ka]), ka, rank_sort_key(v[kb]), kb)) {
            s_code[r] = kb;
            s_code[r + 1] = ka;
            ctl[5] = 1;
          }
        }
        __syncthreads();
      }
      repaired = ctl[5] == 0;
      __syncthreads();
    }
    if (repaired)
      for (int r = tid; r < K; r += blockDim.x) out_rank[(int64_t)q * K + s_code[r]] = (int16_t)r;
  }
  if (!repaired) rank_pass(true);
  SEL_STAMP(9);
  if (A.eps <= 0.0) return;
  __syncthreads();
  // ---- tier 2 (rank level): refined minima of different codes within eps2 -> reference arithmetic for their winners
  for (int r = tid; r + 1 < K; r += blockDim.x) {
    const int ka = s_code[r], kb = s_code[r + 1];
    if (besti[ka] == 0xffffffffu || besti[kb] == 0xffffffffu) continue;
    // (cut lists: two sweep values that happen to coincide belong to rows the walk can not read - nothing to settle)
    if (phase == 2 && RC.pos_t &&
        !(reinterpret_cast<const unsigned char*>(pot_c)[ka] && reinterpret_cast<const unsigned char*>(pot_c)[kb]))
      continue;
    if (v[kb] - v[ka] < A.eps) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int k = h ? kb : ka;
        if (atomicExch(&near_[k], 0xffffffffu) == 0xffffffffu || refd[k]) continue;
        const int pos = atomicAdd(&ctl[2], 1);
        if (pos < MIX_LIST2) {
          l2[pos] = k;
        }
      }
    }
  }
  __syncthreads();
  int n3 = ctl[2];
  SEL_STAMP(10);
  if (n3 == 0) return;
  if (n3 > MIX_LIST2) {
    n3 = MIX_LIST2;
    if (tid == 0) atomicOr(&A.stats[1], 1);
  }
  for (int i0 = 0; i0 < n3; i0 += blockDim.x / 4) {
    const int i = i0 + (tid >> 2);
    const int k = l2[i < n3 ? i : 0];
    const double dr = refine_pair_f64(A, q, (int64_t)(besti[k] - (unsigned int)idx_base), tid & 3);
    if (i < n3 && (tid & 3) == 0) {
      v[k] = dr;
      out_dist[(int64_t)q * K + k] = dr;
    }
  }
  if (tid == 0) atomicAdd(&A.stats[0], n3);
  __syncthreads();
  SEL_STAMP(11);
  rank_pass(true);
  SEL_STAMP(12);
}

extern "C" int64_t qpg_percode_select_mixed_ws_bytes(int Q, int K) {
  return (Q <= 0 || K <= 0) ? 0 : (int64_t)((size_t)Q * mix_ws_stride(K));
}
extern "C" int64_t qpg_percode_select_mixed_ws_stride(int K) {       // bytes per query; query q's space starts at
  return K <= 0 ? 0 : (int64_t)mix_ws_stride(K);                     // q x stride, its list length is the i32 at + 24 K
}

static int select_mixed_impl(const char* name, qpg_ctx* ctx, void* stream, const void* D, int d_is_f32, int64_t ldD, int Q,
                             const int16_t* cand_code, int64_t C, int K, double absent,
                             int32_t idx_base, double* out_dist, int32_t* out_idx, int16_t* out_rank,
                             int q_block, int64_t block_stride, const float* base, int T, int F,
                             const int32_t* cand_t, int G, int n_taps, int tap_stride,
                             const float* q32, const double* qn2, const double* cn2, double eps1,
                             double eps2, int32_t* stats, void* ws, int64_t ws_bytes, int base_is_f16,
                             const RankCut* cut = nullptr) {
  QPG_REQUIRE(ctx && D && (cand_code || C == 0) && out_dist && out_idx && base && cand_t && q32 && qn2 && cn2 && stats,
              "%s: null pointer", name);
  QPG_REQUIRE(Q >= 0 && C >= 0 && K > 0 && K <= 512 && ldD >= C && C + (int64_t)idx_base < 0x7fffffffll,
              "%s: bad size (K <= 512)", name);
  QPG_REQUIRE(T > 0 && F > 0 && (F % 4) == 0 && G > 0 && n_taps > 0 && tap_stride > 0 && C % G == 0 &&
                  eps1 >= 2.0 * QPG_AUDIO_HL_ERR && eps2 >= 0.0 && eps2 < eps1,
              "%s: bad geometry (F %% 4 == 0) or eps (eps1 >= 2 x the sweep's error bound - at least %g -, 0 <= eps2 < eps1)",
              name, 2.0 * (double)QPG_AUDIO_HL_ERR);
  QPG_REQUIRE(q_block >= 0 && (q_block == 0 || (!out_rank && Q % q_block == 0 && block_stride % 8 == 0)),
              "%s: block layout needs Q %% q_block == 0, an 8-byte multiple stride and no rank output", name);
  if (Q == 0) return QPG_OK;
  GuardArgs A;
  A.base = base; A.half = base_is_f16; A.q32 = q32; A.cand_t = cand_t; A.T = T; A.F = F; A.G = G; A.n_taps = n_taps;
  A.tap_stride = tap_stride; A.eps = eps2; A.stats = stats;
  // fast tier-1 path (WavLM geometry: 6 taps x 1024 features): the query row is staged in LDS (+24 KB: 123 KB in all, one block per CU)
  const int use_qlds = (n_taps == 6 && F == 1024) ? 1 : 0;
  const size_t sh2 = 32 * (size_t)K + 16 * MIX_LIST + 4 * MIX_LIST2 + 32 + 6 * MIX_LIST + 4 * (size_t)K;   // merge phase
  size_t sh2c = sh2;                                                                                       // ... + the cut's scratch
  const RankCut RCnone = {nullptr, nullptr, 1, 0};
  const size_t sh1 = sh2 + 6 * MIX_POT;                                                                    // list phase
  const size_t sh = sh1 + (use_qlds ? (size_t)n_taps * F * 4 : 0);                                         // one launch
  if (!ctx->select_lds_raised) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(percode_select_mixed_f64_kernel<double>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(percode_select_mixed_f64_kernel<float>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess) {
      qpg_set_error("%s: cannot raise the dynamic LDS limit", name);
      return QPG_EHIP;
    }
    ctx->select_lds_raised = true;
  }
  unsigned char* w = static_cast<unsigned char*>(ws);
  if (ws)
    QPG_REQUIRE(ws_bytes >= qpg_percode_select_mixed_ws_bytes(Q, K) && (reinterpret_cast<uintptr_t>(ws) % 16) == 0,
                "%s: workspace too small or misaligned (qpg_percode_select_mixed_ws_bytes)", name);
#define SEL_MIX_LAUNCH(DT, SH, PHASE, PRE)                                                                                  \
  hipLaunchKernelGGL((percode_select_mixed_f64_kernel<DT>), dim3(Q), dim3(1024), SH, qpg_stream(stream),               \
                     static_cast<const DT*>(D), ldD, cand_code, C, K, absent, idx_base, out_dist, out_idx, out_rank,   \
                     q_block, block_stride, A, eps1, cn2, qn2, use_qlds, PHASE, w, PRE, RCv)
  RankCut RCv = RCnone;
  if (!ws) {
    if (d_is_f32) SEL_MIX_LAUNCH(float, sh, 0, 0); else SEL_MIX_LAUNCH(double, sh, 0, 0);
    QPG_LAUNCH_CHECK("percode_select_mixed_f64_kernel");
    return QPG_OK;
  }
  if (d_is_f32) {
    // the row is streamed by MIX_SPLIT blocks per query; the list kernel starts from their state
    const size_t shs = 4 * (size_t)K + 10 * (size_t)MIX_SPOT;
    hipLaunchKernelGGL(mixed_stream_kernel, dim3(Q, MIX_SPLIT), dim3(1024), shs, qpg_stream(stream),
                       static_cast<const float*>(D), ldD, cand_code, C, K, eps1, w);
    QPG_LAUNCH_CHECK("mixed_stream_kernel");
    if (cut && cut->pos_t && out_rank) {          // walk-relevance cut: the list and merge launches get its scratch
      RCv = *cut;
      sh2c = sh2 + 12 * (size_t)rank_sort_pow2(K) + 8 * (size_t)((K + 7) / 8 * 8) + 17 * 8 + 64 * (8 + 8 + 4) +
             16 * (size_t)K;
    }
    SEL_MIX_LAUNCH(float, sh2c, 1, 1);
  } else {
    SEL_MIX_LAUNCH(double, sh1, 1, 0);
  }
  QPG_LAUNCH_CHECK("percode_select_mixed_f64_kernel (lists)");
  const int rb = Q >= 512 ? 8 : (Q >= 128 ? 16 : 64);         // waves per query = 4 rb
  hipLaunchKernelGGL((select_refine_kernel<4>), dim3(Q, rb), dim3(256), 0, qpg_stream(stream), A, K, cn2, qn2, w, use_qlds);
  QPG_LAUNCH_CHECK("select_refine_kernel");
  if (d_is_f32) SEL_MIX_LAUNCH(float, sh2c, 2, 0); else SEL_MIX_LAUNCH(double, sh2, 2, 0);
  QPG_LAUNCH_CHECK("percode_select_mixed_f64_kernel (merge)");
#undef SEL_MIX_LAUNCH
  return QPG_OK;
}

extern "C" int qpg_percode_select_mixed_f64(qpg_ctx* ctx, void* stream, const void* D, int d_is_f32, int64_t ldD, int Q,
                                            const int16_t* cand_code, int64_t C, int K, double absent,
                                            int32_t idx_base, double* out_dist, int32_t* out_idx, int16_t* out_rank,
                                            int q_block, int64_t block_stride, const float* base, int T, int F,
                                            const int32_t* cand_t, int G, int n_taps, int tap_stride,
                                            const float* q32, const double* qn2, const double* cn2, double eps1,
                                            double eps2, int32_t* stats, void* ws, int64_t ws_bytes, int base_is_f16) {
  return select_mixed_impl("qpg_percode_select_mixed_f64", ctx, stream, D, d_is_f32, ldD, Q, cand_code, C, K, absent, idx_base,
                           out_dist, out_idx, out_rank, q_block, block_stride, base, T, F, cand_t, G, n_taps, tap_stride, q32,
                           qn2, cn2, eps1, eps2, stats, ws, ws_bytes, base_is_f16);
}

// The four-launch form with the walk-relevance cut (RankCut above): pos_rank_t [dev] i16 [K][K] = the TRANSPOSE of
// qpg_match_steps' pos_rank, freq_rank [dev] i16 [K], top_n = how many of a step's best fused scores the walk reads (1 with
// the text side, 2 without: GestureKNN.py:593, :627-657), probe = the number of best-ranked codes the bound on the winning
// score is taken over (0: 64).  out_rank is required,
// an f32 matrix and a workspace too.
extern "C" int qpg_percode_select_mixed_f64_cut(qpg_ctx* ctx, void* stream, const void* D, int d_is_f32, int64_t ldD, int Q,
                                                const int16_t* cand_code, int64_t C, int K, double absent,
                                                int32_t idx_base, double* out_dist, int32_t* out_idx, int16_t* out_rank,
                                                int q_block, int64_t block_stride, const float* base, int T, int F,
                                                const int32_t* cand_t, int G, int n_taps, int tap_stride,
                                                const float* q32, const double* qn2, const double* cn2, double eps1,
                                                double eps2, int32_t* stats, void* ws, int64_t ws_bytes, int base_is_f16,
                                                const int16_t* pos_rank_t, const int16_t* freq_rank, int top_n, int probe) {
  const char* name = "qpg_percode_select_mixed_f64_cut";
  QPG_REQUIRE(pos_rank_t && freq_rank && out_rank && ws && d_is_f32 && q_block == 0 && (top_n == 1 || top_n == 2) &&
                  probe >= 0,
              "%s: needs the rank tables, out_rank, an f32 matrix, a workspace, no block layout, top_n 1 or 2", name);
  RankCut rc = {pos_rank_t, freq_rank, top_n, probe > 0 ? probe : 64};
  return select_mixed_impl(name, ctx, stream, D, d_is_f32, ldD, Q, cand_code, C, K, absent, idx_base, out_dist, out_idx,
                           out_rank, q_block, block_stride, base, T, F, cand_t, G, n_taps, tap_stride, q32, qn2, cn2, eps1, eps2,
                           stats, ws, ws_bytes, base_is_f16, &rc);
}

template <typename T, typename KeyT, bool PACKED>
static int percode_select(const char* name, qpg_ctx* ctx, void* stream, const T* D, int64_t ldD, int Q,
                          const int16_t* cand_code, int64_t C, int K, T absent, int32_t idx_base, T* out_dist,
                          int32_t* out_idx, int16_t* out_rank, int q_block, int64_t block_stride) {
  QPG_REQUIRE(ctx && D && (cand_code || C == 0) && out_dist && out_idx, "%s: null pointer", name);
  QPG_REQUIRE(Q >= 0 && C >= 0 && K > 0 && K <= 2048 && ldD >= C && C + (int64_t)idx_base < 0x7fffffffll,
              "%s: bad size (K <= 2048, candidate indices must stay below 2^31)", name);
  QPG_REQUIRE(q_block >= 0 && (q_block == 0 || (!out_rank && Q % q_block == 0 && block_stride % 8 == 0)),
              "%s: block layout needs Q %% q_block == 0, an 8-byte multiple stride and no rank output", name);
  if (Q == 0) return QPG_OK;
  const size_t sh = 12 * (size_t)rank_sort_pow2(K) + (size_t)K * sizeof(T);
  hipLaunchKernelGGL((percode_select_kernel<T, KeyT, PACKED>), dim3(Q), dim3(1024), sh, qpg_stream(stream), D, ldD,
                     cand_code, C, K, absent, idx_base, out_dist, out_idx, out_rank, q_block, block_stride);
  QPG_LAUNCH_CHECK(name);
  return QPG_OK;
}

extern "C" int qpg_percode_select_f64(qpg_ctx* ctx, void* stream, const double* D, int64_t ldD, int Q,
                                      const int16_t* cand_code, int64_t C, int K, double absent, int32_t idx_base,
                                      double* out_dist, int32_t* out_idx, int16_t* out_rank, int q_block,
                                      int64_t block_stride) {
  return percode_select<double, unsigned long long, false>("qpg_percode_select_f64", ctx, stream, D, ldD, Q, cand_code,
                                                           C, K, absent, idx_base, out_dist, out_idx, out_rank, q_block,
                                                           block_stride);
}
extern "C" int qpg_percode_select_f32(qpg_ctx* ctx, void* stream, const float* D, int64_t ldD, int Q,
                                      const int16_t* cand_code, int64_t C, int K, float absent, int32_t idx_base,
                                      float* out_dist, int32_t* out_idx, int16_t* out_rank, int q_block,
                                      int64_t block_stride) {
  return percode_select<float, unsigned int, true>("qpg_percode_select_f32", ctx, stream, D, ldD, Q, cand_code, C, K,
                                                   absent, idx_base, out_dist, out_idx, out_rank, q_block, block_stride);
}

// ---------------------------------------------------------------------------------------------
// UNCAPPED near-tie guard (round 3): qpg_percode_select_exact_f64.  Same contract as the guarded select above, but
// every list lives in a global workspace sized for the worst case (every candidate of a row inside the band), so NO
// population of near-ties can overflow it: this is the path a clip is re-matched on when a capped select (guarded:
// 256 entries, mixed: 2048 / 256) or the mixed-precision sweep's norm check raised its flag — the reference's scan
// (GestureKNN.py:685-689) visits every candidate with a strict `<` and has no cap either.  Launches:
//   zero | per-code minimum + first index (percode_select_kernel) | band members -> list | reference-arithmetic
//   re-evaluation of the members of every band with two or more (all CUs, one quad per pair) | per-query merge:
//   winners by (reference distance, index), ranks, rank-neighbour check (at most K entries: cannot overflow either).
// Cost on ordinary data: one more streaming pass than the guarded select (~+30 us at Q = 48); with all 53 248
// candidates of every row in one band: 2.5 M reference-arithmetic pairs, a few ms.
// ---------------------------------------------------------------------------------------------
struct ExactWs {
  double* wmin;          // [Q][K] per-code minimum of the sweep's values
  double* list_d;        // [Q][C] reference-arithmetic distance of list entry e
  int32_t* widx;         // [Q][K] its first-wins candidate (global index, -1 absent)
  unsigned int* near_;   // [Q][K] band population
  int* list_c;           // [Q][C] band members (local candidate)
  int* cnt;              // [Q] list length
};
__host__ static size_t exact_ws_layout(int Q, int64_t C, int K, unsigned char* b, ExactWs* w) {
  size_t o = 0;
  const size_t QK = (size_t)Q * K, QC = (size_t)Q * (size_t)C;
  if (w) w->wmin = reinterpret_cast<double*>(b + o);
  o += QK * 8;
  if (w) w->list_d = reinterpret_cast<double*>(b + o);
  o += QC * 8;
  if (w) w->widx = reinterpret_cast<int32_t*>(b + o);
  o += QK * 4;
  if (w) w->near_ = reinterpret_cast<unsigned int*>(b + o);
  o += QK * 4;
  if (w) w->list_c = reinterpret_cast<int*>(b + o);
  o += QC * 4;
  if (w) w->cnt = reinterpret_cast<int*>(b + o);
  o += ((size_t)Q * 4 + 15) / 16 * 16;
  return o;
}

__global__ void exact_zero_kernel(unsigned int* __restrict__ near_, int64_t n, int* __restrict__ cnt, int Q) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) near_[i] = 0;
  if (i < Q) cnt[i] = 0;
}

__global__ __launch_bounds__(256) void exact_band_kernel(const double* __restrict__ D, int64_t ldD,
                                                         const int16_t* __restrict__ cand_code, int64_t C, int K,
                                                         double eps, ExactWs W) {
  const int q = blockIdx.x;
  const double* row = D + (int64_t)q * ldD;
  for (int64_t c = (int64_t)blockIdx.y * blockDim.x + threadIdx.x; c < C; c += (int64_t)gridDim.y * blockDim.x) {
    const int cd = cand_code[c];
    if ((unsigned)cd >= (unsigned)K) continue;
    if (row[c] <= W.wmin[(int64_t)q * K + cd] + eps) {
      atomicAdd(&W.near_[(int64_t)q * K + cd], 1u);
      const int pos = atomicAdd(&W.cnt[q], 1);                       // pos < C: every candidate is listed at most once
      W.list_c[(int64_t)q * C + pos] = (int)c;
    }
  }
}

__global__ __launch_bounds__(256) void exact_refine_kernel(GuardArgs A, const int16_t* __restrict__ cand_code, int64_t C,
                                                           int K, ExactWs W) {
  const int q = blockIdx.x, tid = threadIdx.x;
  const int n = W.cnt[q];
  for (int e0 = blockIdx.y * 64; e0 < n; e0 += gridDim.y * 64) {      // 64 quads per block
    const int e = e0 + (tid >> 2);
    int c = 0;
    bool need = false;
    if (e < n) {
      c = W.list_c[(int64_t)q * C + e];
      need = W.near_[(int64_t)q * K + cand_code[c]] >= 2;
    }
    if (__ballot(need) == 0ull) continue;                             // wave-uniform: nothing to re-evaluate here
    const double dr = refine_pair_f64(A, q, need ? c : 0, tid & 3);   // (idle quads run along: uniform control flow)
    if (need && (tid & 3) == 0) W.list_d[(int64_t)q * C + e] = dr;
  }
}

__global__ __launch_bounds__(1024) void exact_merge_kernel(const int16_t* __restrict__ cand_code, int64_t C, int K,
                                                           double absent, int32_t idx_base, double* __restrict__ out_dist,
                                                           int32_t* __restrict__ out_idx, int16_t* __restrict__ out_rank,
                                                           int q_block, int64_t block_stride, GuardArgs A, ExactWs W) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned long long* best = reinterpret_cast<unsigned long long*>(smem);          // [K]
  double* v = reinterpret_cast<double*>(smem + 8 * (size_t)K);                     // [K]
  unsigned int* besti = reinterpret_cast<unsigned int*>(smem + 16 * (size_t)K);    // [K]
  unsigned int* nearl = besti + K;                                                 // [K]
  int* s_code = reinterpret_cast<int*>(nearl + K);                                 // [K]
  int* rkc = s_code + K;                                                           // [K]
  int* l2 = rkc + K;                                                               // [K] rank-level list (codes)
  int* ctl = l2 + K;                                                               // [2]
  const int q = blockIdx.x, tid = threadIdx.x;
  if (q_block > 0) {
    const int64_t shift = (int64_t)(q / q_block) * block_stride;
    const int64_t rowoff = (int64_t)(q % q_block) * K - (int64_t)q * K;
    out_dist = reinterpret_cast<double*>(reinterpret_cast<unsigned char*>(out_dist) + shift) + rowoff;
    out_idx = reinterpret_cast<int32_t*>(reinterpret_cast<unsigned char*>(out_idx) + shift) + rowoff;
  }
  for (int k = tid; k < K; k += blockDim.x) {
    const unsigned int nr = W.near_[(int64_t)q * K + k];
    const int32_t wi = W.widx[(int64_t)q * K + k];
    nearl[k] = nr;
    best[k] = ~0ull;
    besti[k] = nr >= 2 ? 0xffffffffu : (unsigned int)wi;                           // -1 -> 0xffffffff (absent)
    v[k] = wi >= 0 ? W.wmin[(int64_t)q * K + k] : absent;
  }
  if (tid < 2) ctl[tid] = 0;
  __syncthreads();
  const int n = W.cnt[q];
  const int* lc = W.list_c + (int64_t)q * C;
  const double* ld = W.list_d + (int64_t)q * C;
  int mine = 0;
  for (int e = tid; e < n; e += blockDim.x) {
    const int cd = cand_code[lc[e]];
    if (nearl[cd] >= 2) {
      atomicMin(&best[cd], (unsigned long long)order_key(ld[e]));
      ++mine;
    }
  }
  if (mine) atomicAdd(&ctl[1], mine);
  __syncthreads();
  for (int e = tid; e < n; e += blockDim.x) {
    const int c = lc[e], cd = cand_code[c];
    if (nearl[cd] >= 2 && (unsigned long long)order_key(ld[e]) == best[cd])
      atomicMin(&besti[cd], (unsigned int)(c + idx_base));
  }
  __syncthreads();
  for (int k = tid; k < K; k += blockDim.x) {
    if (nearl[k] >= 2) v[k] = key_value(best[k], 0.0);
    out_dist[(int64_t)q * K + k] = v[k];
    out_idx[(int64_t)q * K + k] = besti[k] != 0xffffffffu ? (int32_t)besti[k] : -1;
  }
  if (tid == 0 && ctl[1]) atomicAdd(&A.stats[0], ctl[1]);
  if (!out_rank) return;
  __syncthreads();
  auto rank_pass = [&]() {
    block_stable_ranks(v, K, rkc, [&](int k, int r) {
      out_rank[(int64_t)q * K + k] = (int16_t)r;
      s_code[r] = k;
    });
  };
  rank_pass();
  __syncthreads();
  for (int r = tid; r + 1 < K; r += blockDim.x) {
    const int ka = s_code[r], kb = s_code[r + 1];
    if (besti[ka] == 0xffffffffu || besti[kb] == 0xffffffffu) continue;
    if (v[kb] - v[ka] < A.eps) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int k = h ? kb : ka;
        if (atomicMax(&nearl[k], 2u) >= 2u) continue;                 // a reference-arithmetic value already / listed once
        l2[atomicAdd(&ctl[0], 1)] = k;                                // at most K entries
      }
    }
  }
  __syncthreads();
  const int n2 = ctl[0];
  if (n2 == 0) return;
  for (int i0 = 0; i0 < n2; i0 += blockDim.x / 4) {
    const int i = i0 + (tid >> 2);
    if (i0 + ((tid >> 6) << 4) >= n2) continue;                       // the whole wave is past the list
    const int k = l2[i < n2 ? i : 0];
    const double dr = refine_pair_f64(A, q, (int64_t)(besti[k] - (unsigned int)idx_base), tid & 3);
    if (i < n2 && (tid & 3) == 0) {
      v[k] = dr;
      out_dist[(int64_t)q * K + k] = dr;
    }
  }
  if (tid == 0) atomicAdd(&A.stats[0], n2);
  __syncthreads();
  rank_pass();
}

extern "C" int64_t qpg_percode_select_exact_ws_bytes(int Q, int64_t C, int K) {
  return (Q <= 0 || K <= 0 || C < 0) ? 0 : (int64_t)exact_ws_layout(Q, C, K, nullptr, nullptr);
}

extern "C" int qpg_percode_select_exact_f64(qpg_ctx* ctx, void* stream, const double* D, int64_t ldD, int Q,
                                            const int16_t* cand_code, int64_t C, int K, double absent, int32_t idx_base,
                                            double* out_dist, int32_t* out_idx, int16_t* out_rank, int q_block,
                                            int64_t block_stride, const float* base, int T, int F, const int32_t* cand_t,
                                            int G, int n_taps, int tap_stride, const float* q32, double eps,
                                            int32_t* stats, int base_is_f16, void* ws, int64_t ws_bytes) {
  const char* name = "qpg_percode_select_exact_f64";
  QPG_REQUIRE(ctx && D && (cand_code || C == 0) && out_dist && out_idx && base && cand_t && q32 && stats && ws,
              "%s: null pointer", name);
  QPG_REQUIRE(Q >= 0 && C >= 0 && K > 0 && K <= 1024 && ldD >= C && C + (int64_t)idx_base < 0x7fffffffll && T > 0 &&
                  F > 0 && G > 0 && n_taps > 0 && tap_stride > 0 && eps >= 0.0 && C % G == 0,
              "%s: bad size (K <= 1024)", name);
  QPG_REQUIRE(q_block >= 0 && (q_block == 0 || (!out_rank && Q % q_block == 0 && block_stride % 8 == 0)),
              "%s: block layout needs Q %% q_block == 0 and no rank output", name);
  QPG_REQUIRE(ws_bytes >= qpg_percode_select_exact_ws_bytes(Q, C, K) && (reinterpret_cast<uintptr_t>(ws) % 16) == 0,
              "%s: workspace too small or misaligned (qpg_percode_select_exact_ws_bytes)", name);
  if (Q == 0) return QPG_OK;
  ExactWs W;
  exact_ws_layout(Q, C, K, static_cast<unsigned char*>(ws), &W);
  GuardArgs A;
  A.base = base; A.half = base_is_f16; A.q32 = q32; A.cand_t = cand_t; A.T = T; A.F = F; A.G = G; A.n_taps = n_taps;
  A.tap_stride = tap_stride; A.eps = eps; A.stats = stats;
  hipStream_t st = qpg_stream(stream);
  const int64_t QK = (int64_t)Q * K;
  hipLaunchKernelGGL(exact_zero_kernel, dim3((unsigned)((QK + 255) / 256)), dim3(256), 0, st, W.near_, QK, W.cnt, Q);
  QPG_LAUNCH_CHECK("exact_zero_kernel");
  const int rc = percode_select<double, unsigned long long, false>(name, ctx, stream, D, ldD, Q, cand_code, C, K, absent,
                                                                    idx_base, W.wmin, W.widx, nullptr, 0, 0);
  if (rc != QPG_OK) return rc;
  int by = (int)((C + 255) / 256);
  const int by_cap = Q >= 256 ? 8 : (Q >= 32 ? 32 : 128);
  if (by > by_cap) by = by_cap;
  if (by < 1) by = 1;
  hipLaunchKernelGGL(exact_band_kernel, dim3(Q, by), dim3(256), 0, st, D, ldD, cand_code, C, K, eps, W);
  QPG_LAUNCH_CHECK("exact_band_kernel");
  hipLaunchKernelGGL(exact_refine_kernel, dim3(Q, by_cap), dim3(256), 0, st, A, cand_code, C, K, W);
  QPG_LAUNCH_CHECK("exact_refine_kernel");
  const size_t sh = 36 * (size_t)K + 16;
  hipLaunchKernelGGL(exact_merge_kernel, dim3(Q), dim3(1024), sh, st, cand_code, C, K, absent, idx_base, out_dist,
                     out_idx, out_rank, q_block, block_stride, A, W);
  QPG_LAUNCH_CHECK("exact_merge_kernel");
  return QPG_OK;
}

// ---------------------------------------------------------------------------------------------
// Cross-shard merge (the reduction is an associative min-with-index per (query, code)): after the
// exchange (RCCL all-gather or all-to-all of the ranks' byte buffers) row q's W candidate (distance, index) pairs
// sit W blocks apart; the winner is the minimum distance and, among equal distances, the lowest GLOBAL candidate
// index (shards are ascending row blocks, so that is the reference's first-wins scan); -1 marks "code absent in
// that shard".  One block per query row; the stable ranks of the merged row are produced in the same launch.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(1024) void merge_select_kernel(const unsigned char* __restrict__ recv, int W,
                                                           int64_t src_stride, int64_t dist_off, int64_t idx_off, int K,
                                                           T absent, T* __restrict__ out_dist,
                                                           int32_t* __restrict__ out_idx,
                                                           int16_t* __restrict__ out_rank, T eps2,
                                                           int32_t* __restrict__ stats) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* v = reinterpret_cast<T*>(smem);
  int* cnt = reinterpret_cast<int*>(v + K);
  int* s_code = cnt + K;
  int* have = s_code + K;
  const int q = blockIdx.x;
  const bool guard = stats != nullptr && eps2 > T(0);
  int trouble = 0;
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    T bd = absent;
    int32_t bi = -1;
    for (int w = 0; w < W; ++w) {
      const unsigned char* src = recv + (int64_t)w * src_stride;
      const T d = reinterpret_cast<const T*>(src + dist_off)[(int64_t)q * K + k];
      const int32_t i = reinterpret_cast<const int32_t*>(src + idx_off)[(int64_t)q * K + k];
      if (i < 0) continue;
      if (bi < 0 || d < bd || (d == bd && i < bi)) {
        bd = d;
        bi = i;
      }
    }
    if (guard && bi >= 0) {
      // two shards' minima of one code closer than eps2: which one the reference's own arithmetic prefers is not
      // decided by these values (each shard settled its near-ties inside the shard only)
      int near = 0;
      for (int w = 0; w < W; ++w) {
        const unsigned char* src = recv + (int64_t)w * src_stride;
        const int32_t i = reinterpret_cast<const int32_t*>(src + idx_off)[(int64_t)q * K + k];
        near += i >= 0 && reinterpret_cast<const T*>(src + dist_off)[(int64_t)q * K + k] <= bd + eps2;
      }
      trouble |= near >= 2;
    }
    v[k] = bd;
    have[k] = bi >= 0;
    out_dist[(int64_t)q * K + k] = bd;
    out_idx[(int64_t)q * K + k] = bi;
  }
  if (trouble) atomicOr(&stats[1], 8);
  if (!out_rank) return;
  __syncthreads();
  block_stable_ranks(v, K, cnt, [&](int k, int r) {
    out_rank[(int64_t)q * K + k] = (int16_t)r;
    s_code[r] = k;
  });
  if (!guard) return;
  __syncthreads();
  // minima of DIFFERENT codes closer than eps2 (rank neighbours): the shards' selects do not compare codes with each
  // other when the ranks are taken after the merge, so this is the only place that sees them
  trouble = 0;
  for (int r = threadIdx.x; r + 1 < K; r += blockDim.x) {
    const int ka = s_code[r], kb = s_code[r + 1];
    trouble |= have[ka] && have[kb] && v[kb] - v[ka] < eps2;
  }
  if (trouble) atomicOr(&stats[1], 8);
}

template <typename T>
static int merge_select(const char* name, qpg_ctx* ctx, void* stream, const void* recv, int W, int64_t src_stride,
                        int64_t dist_off, int64_t idx_off, int Q, int K, T absent, T* out_dist, int32_t* out_idx,
                        int16_t* out_rank, T eps2, int32_t* stats) {
  QPG_REQUIRE(ctx && recv && out_dist && out_idx, "%s: null pointer", name);
  QPG_REQUIRE(W > 0 && Q >= 0 && K > 0 && K <= 2048 && src_stride >= 0 && dist_off >= 0 && idx_off >= 0 &&
                  dist_off % (int64_t)sizeof(T) == 0 && idx_off % 4 == 0 && src_stride % 8 == 0,
              "%s: bad size / alignment", name);
  if (Q == 0) return QPG_OK;
  hipLaunchKernelGGL((merge_select_kernel<T>), dim3(Q), dim3(1024), (sizeof(T) + 12) * (size_t)K, qpg_stream(stream),
                     static_cast<const unsigned char*>(recv), W, src_stride, dist_off, idx_off, K, absent, out_dist,
                     out_idx, out_rank, eps2, stats);
  QPG_LAUNCH_CHECK(name);
  return QPG_OK;
}

extern "C" int qpg_merge_select_f64(qpg_ctx* ctx, void* stream, const void* recv, int W, int64_t src_stride,
                                    int64_t dist_off, int64_t idx_off, int Q, int K, double absent, double* out_dist,
                                    int32_t* out_idx, int16_t* out_rank, double eps2, int32_t* stats) {
  return merge_select<double>("qpg_merge_select_f64", ctx, stream, recv, W, src_stride, dist_off, idx_off, Q, K, absent,
                              out_dist, out_idx, out_rank, eps2, stats);
}
extern "C" int qpg_merge_select_f32(qpg_ctx* ctx, void* stream, const void* recv, int W, int64_t src_stride,
                                    int64_t dist_off, int64_t idx_off, int Q, int K, float absent, float* out_dist,
                                    int32_t* out_idx, int16_t* out_rank) {
  return merge_select<float>("qpg_merge_select_f32", ctx, stream, recv, W, src_stride, dist_off, idx_off, Q, K, absent,
                             out_dist, out_idx, out_rank, 0.f, nullptr);
}

// ---------------------------------------------------------------------------------------------------------------
// Cross-shard merge for the MIXED-PRECISION sweep (row-sharded DB).  Every shard contributes per-(query, code) minima
// that are only accurate to E = QPG_AUDIO_MX_ERR (its own select has already settled the near-ties INSIDE the shard),
// so the owner of a query block cannot simply take the minimum over shards.  Same rule as on one GPU, with the
// re-evaluation done where the rows live:
//   phase 1 (owner)   approximate merge; every shard within eps1 of a code's merged minimum (when there are two or
//                     more) and the winners of codes whose merged minima are rank neighbours within eps1 become
//                     REQUESTS (query, code, candidate) to the shard that holds the candidate;
//   refine  (shards)  exact f64 distance of every requested pair (one wave per pair), sent back;
//   phase 2 (owner)   winners among the re-evaluated contenders by (exact value, index), ranks over exact + untouched.
// Message layout of one (owner -> shard) request block: [i64 count][R x u64 (q_local << 48 | code << 32 | candidate)];
// response block: [R x f64].  Both travel through the same byte exchange as the tables (parallel.exchange_bytes).
// ---------------------------------------------------------------------------------------------------------------
// (flag-list capacity per query: `fl_cap`, a call argument - 1024 on the normal path, K*W on the uncapped one)

// Trouble bits travel WITH the exchanges (round 4; they used to take their own 4-byte all-reduce): every exchanged block
// carries a flag word - table blocks at `flag_off`, request / response blocks in the high half of their 8-byte header -
// holding the sender's stats[1] at send time; the consumer ORs what it received into its own stats[1].
//   qpg_flags_stamp   (sender)   flag word of each of nblk blocks = stats[1]
//   qpg_flags_gather  (receiver) stats[1] |= OR of the nblk received flag words
// The mixed merge folds both into its own kernels: its prologue gathers the table blocks' words and seeds the request
// headers, the shard refine gathers the request headers and seeds the response headers, phase 2 gathers those.
__global__ void flags_stamp_kernel(unsigned char* __restrict__ buf, int nblk, int64_t stride, int64_t off,
                                   const int32_t* __restrict__ stats) {
  const int f = stats[1];
  for (int b = threadIdx.x; b < nblk; b += blockDim.x) *reinterpret_cast<int32_t*>(buf + (int64_t)b * stride + off) = f;
}
__global__ void flags_gather_kernel(const unsigned char* __restrict__ buf, int nblk, int64_t stride, int64_t off,
                                    int32_t* __restrict__ stats) {
  int f = 0;
  for (int b = threadIdx.x; b < nblk; b += blockDim.x) f |= *reinterpret_cast<const int32_t*>(buf + (int64_t)b * stride + off);
  if (f) atomicOr(&stats[1], f);
}
extern "C" int qpg_flags_stamp(qpg_ctx* ctx, void* stream, void* buf, int nblk, int64_t stride, int64_t off,
                               const int32_t* stats) {
  QPG_REQUIRE(ctx && buf && stats && nblk > 0 && off >= 0 && off % 4 == 0, "qpg_flags_stamp: bad argument");
  hipLaunchKernelGGL(flags_stamp_kernel, dim3(1), dim3(64), 0, qpg_stream(stream), static_cast<unsigned char*>(buf), nblk,
                     stride, off, stats);
  QPG_LAUNCH_CHECK("flags_stamp_kernel");
  return QPG_OK;
}
extern "C" int qpg_flags_gather(qpg_ctx* ctx, void* stream, const void* buf, int nblk, int64_t stride, int64_t off,
                                int32_t* stats) {
  QPG_REQUIRE(ctx && buf && stats && nblk > 0 && off >= 0 && off % 4 == 0, "qpg_flags_gather: bad argument");
  hipLaunchKernelGGL(flags_gather_kernel, dim3(1), dim3(64), 0, qpg_stream(stream), static_cast<const unsigned char*>(buf),
                     nblk, stride, off, stats);
  QPG_LAUNCH_CHECK("flags_gather_kernel");
  return QPG_OK;
}

// request block header: [i32 count (diagnostics) | i32 flags]; the prologue zeroes the counts, gathers the W received table
// blocks' flag words (flag_off >= 0) and seeds every header's flag word with this rank's trouble word so far
__global__ void merge_mixed_prologue_kernel(unsigned char* __restrict__ req, int64_t req_stride, int W,
                                            const unsigned char* __restrict__ recv, int64_t src_stride, int64_t flag_off,
                                            int32_t* __restrict__ stats) {
  __shared__ int f_s;
  if (threadIdx.x == 0) f_s = stats[1];
  __syncthreads();
  if (flag_off >= 0) {
    int f = 0;
    for (int w = threadIdx.x; w < W; w += blockDim.x) f |= *reinterpret_cast<const int32_t*>(recv + (int64_t)w * src_stride + flag_off);
    if (f) atomicOr(&f_s, f);
  }
  __syncthreads();
  const int f = f_s;
  if (threadIdx.x == 0 && f) atomicOr(&stats[1], f);
  for (int w = threadIdx.x; w < W; w += blockDim.x) {
    int32_t* h = reinterpret_cast<int32_t*>(req + (int64_t)w * req_stride);
    h[0] = 0;
    h[1] = f;
  }
}

__global__ __launch_bounds__(1024) void merge_mixed_phase1_kernel(
    const unsigned char* __restrict__ recv, int W, int64_t src_stride, int64_t dist_off, int64_t idx_off, int K,
    double absent, double eps1, int R, unsigned char* __restrict__ req, int64_t req_stride,
    double* __restrict__ prov_d, int32_t* __restrict__ prov_i, unsigned long long* __restrict__ fl,
    int32_t* __restrict__ fl_cnt, int32_t* __restrict__ stats, int MM_FL) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double* v = reinterpret_cast<double*>(smem);                   // [K] merged approximate minimum
  int* bi = reinterpret_cast<int*>(v + K);                       // [K] its candidate
  int* bs = bi + K;                                              // [K] its shard
  int* cnt = bs + K;                                             // [K] shards within eps1 of the merged minimum
  int* s_code = cnt + K;                                         // [K] code at rank r
  int* flg = s_code + K;                                         // [K] rank-level flag
  int* rkc = flg + K;                                            // [K] rank counters
  __shared__ int n_fl;
  const int q = blockIdx.x;
  if (threadIdx.x == 0) n_fl = 0;
  auto dval = [&](int w, int k) { return reinterpret_cast<const double*>(recv + (int64_t)w * src_stride + dist_off)[(int64_t)q * K + k]; };
  auto ival = [&](int w, int k) { return reinterpret_cast<const int32_t*>(recv + (int64_t)w * src_stride + idx_off)[(int64_t)q * K + k]; };
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    double bd = absent;
    int b = -1, sh = -1;
    for (int w = 0; w < W; ++w) {
      const int i = ival(w, k);
      if (i < 0) continue;
      const double d = dval(w, k);
      if (b < 0 || d < bd || (d == bd && i < b)) {
        bd = d;
        b = i;
        sh = w;
      }
    }
    int c = 0;
    if (b >= 0)
      for (int w = 0; w < W; ++w) c += ival(w, k) >= 0 && dval(w, k) <= bd + eps1;
    v[k] = bd;
    bi[k] = b;
    bs[k] = sh;
    cnt[k] = c;
    flg[k] = 0;
    prov_d[(int64_t)q * K + k] = bd;
    prov_i[(int64_t)q * K + k] = b;
  }
  __syncthreads();
  block_stable_ranks(v, K, rkc, [&](int k, int r) { s_code[r] = k; });
  __syncthreads();
  for (int r = threadIdx.x; r + 1 < K; r += blockDim.x) {
    const int ka = s_code[r], kb = s_code[r + 1];
    if (bi[ka] < 0 || bi[kb] < 0) continue;
    if (v[kb] - v[ka] < eps1) {
      flg[ka] = 1;                                               // (benign races: only ever set to 1)
      flg[kb] = 1;
    }
  }
  __syncthreads();
  // Slots are DETERMINISTIC (round 4): the request of (query q, code k) to shard w sits at q * Rq + its position among
  // q's requests to w in code order - Rq = R / Q slots per (query, shard), unused ones hold ~0 and the shard skips them.
  // Every rank that runs this kernel on the same tables produces the same request blocks, so in the all-gather form of the
  // exchange (every rank holds every shard's tables) nobody has to SEND requests: a shard takes its own block of its own
  // run.  Positions come from block-wide prefix sums over the codes, four shards per pass in 16-bit fields (a query sends
  // a shard at most one request per code, K <= 2048).
  const int Rq = R / (int)gridDim.x;
  __shared__ unsigned long long wtot[17];
  __shared__ int carry[4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = (blockDim.x + 63) >> 6;
  for (int i = threadIdx.x; i < W * Rq; i += blockDim.x)
    reinterpret_cast<unsigned long long*>(req + (int64_t)(i / Rq) * req_stride + 8)[(int64_t)q * Rq + (i % Rq)] = ~0ull;
  auto wants = [&](int k, int w) -> int {          // candidate this query requests from shard w for code k, or -1
    if (k >= K || bi[k] < 0) return -1;
    if (cnt[k] >= 2) {
      const int i = ival(w, k);
      return (i >= 0 && dval(w, k) <= v[k] + eps1) ? i : -1;
    }
    return (flg[k] && bs[k] == w) ? bi[k] : -1;
  };
  __syncthreads();
  for (int w0 = 0; w0 < W; w0 += 4) {
    if (threadIdx.x < 4) carry[threadIdx.x] = 0;
    __syncthreads();
    for (int kb = 0; kb < K; kb += blockDim.x) {
      const int k = kb + threadIdx.x;
      int cand[4];
      unsigned long long mine = 0;
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        cand[f] = (w0 + f < W) ? wants(k, w0 + f) : -1;
        mine |= (unsigned long long)(cand[f] >= 0) << (16 * f);
      }
      unsigned long long inc = mine;               // inclusive scan inside the wave
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
      }
      if (lane == 63) wtot[wv] = inc;
      __syncthreads();
      if (threadIdx.x == 0) {
        unsigned long long run = 0;
        for (int i = 0; i < nwv; ++i) {
          const unsigned long long t = wtot[i];
          wtot[i] = run;
          run += t;
        }
        wtot[16] = run;
      }
      __syncthreads();
      const unsigned long long excl = inc - mine + wtot[wv];
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        if (cand[f] < 0) continue;
        const int w = w0 + f;
        const int pos = carry[f] + (int)((excl >> (16 * f)) & 0xffff);
        if (pos >= Rq) {                           // request overflow: the clip is re-matched; every shard hears of it
          atomicOr(&stats[1], 4);
          for (int x = 0; x < W; ++x) atomicOr(reinterpret_cast<int*>(req + (int64_t)x * req_stride) + 1, 4);
          continue;
        }
        const int j = q * Rq + pos;
        reinterpret_cast<unsigned long long*>(req + (int64_t)w * req_stride + 8)[j] =
            ((unsigned long long)q << 48) | ((unsigned long long)k << 32) | (unsigned int)cand[f];
        const int fp = atomicAdd(&n_fl, 1);        // (order inside the flag list is irrelevant to phase 2)
        if (fp >= MM_FL) {
          atomicOr(&stats[1], 4);
          for (int x = 0; x < W; ++x) atomicOr(reinterpret_cast<int*>(req + (int64_t)x * req_stride) + 1, 4);
          continue;
        }
        fl[(int64_t)q * MM_FL + fp] = ((unsigned long long)k << 40) | ((unsigned long long)w << 32) | (unsigned int)j;
      }
      __syncthreads();
      if (threadIdx.x < 4) carry[threadIdx.x] += (int)((wtot[16] >> (16 * threadIdx.x)) & 0xffff);
      __syncthreads();
    }
    if (threadIdx.x < 4 && w0 + (int)threadIdx.x < W) {
      const int c = carry[threadIdx.x] < Rq ? carry[threadIdx.x] : Rq;
      if (c) atomicAdd(reinterpret_cast<int*>(req + (int64_t)(w0 + threadIdx.x) * req_stride), c);   // (diagnostics)
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) fl_cnt[q] = n_fl < MM_FL ? n_fl : MM_FL;
}

// shards: exact distance of every requested (query, candidate); one wave per request.  All R slots of a block are visited
// (deterministic slots: unused ones hold ~0).  Block (o, 0) carries the trouble bits on: what the W received request headers
// hold is ORed into stats[1] and, with this shard's own word, into the header of response block o.
__device__ __forceinline__ void refine_carry_flags(const unsigned char* req_recv, int64_t req_stride, int W,
                                                   unsigned char* resp, int64_t resp_stride, int o, int32_t* stats) {
  if (blockIdx.y != 0 || threadIdx.x != 0) return;
  int f = stats ? stats[1] : 0;
  for (int w = 0; w < W; ++w) f |= reinterpret_cast<const int32_t*>(req_recv + (int64_t)w * req_stride)[1];
  if (stats && f) atomicOr(&stats[1], f);
  int32_t* h = reinterpret_cast<int32_t*>(resp + (int64_t)o * resp_stride);
  h[0] = 0;
  h[1] = f;
}

__global__ __launch_bounds__(256) void shard_refine_kernel(GuardArgs A, const unsigned char* __restrict__ req_recv,
                                                           int64_t req_stride, int R, int q_stride, int64_t cand_base,
                                                           const double* __restrict__ cn2, const double* __restrict__ qn2,
                                                           unsigned char* __restrict__ resp, int64_t resp_stride, int fast,
                                                           int W, int32_t* __restrict__ stats, int Rq) {
  const int o = blockIdx.x, lane = threadIdx.x & 63;
  const int g = blockIdx.y * 4 + (threadIdx.x >> 6), ng = gridDim.y * 4;
  const unsigned char* blk = req_recv + (int64_t)o * req_stride;
  refine_carry_flags(req_recv, req_stride, W, resp, resp_stride, o, stats);
  const unsigned long long* ent = reinterpret_cast<const unsigned long long*>(blk + 8);
  double* out = reinterpret_cast<double*>(resp + (int64_t)o * resp_stride + 8);
  // slots are visited POSITION-major (all queries' first slots, then their second ones, ...): a query's requests sit at the
  // front of its Rq-slot segment, so this spreads the live slots over the waves of the first round
  const int Qn = R / Rq;
  for (int t = g; t < R; t += ng) {
    const int e = (t % Qn) * Rq + t / Qn;
    const unsigned long long x = ent[e];
    if (x == ~0ull) continue;
    const int q = o * q_stride + (int)(x >> 48);
    const int64_t c = (int64_t)(unsigned int)(x & 0xffffffffu) - cand_base;        // local candidate
    const float* qrow = A.q32 + (int64_t)q * A.n_taps * A.F;
    const double dot = fast ? pair_dot_fast_f64<4>(A, qrow, c, lane) : pair_dot_wave_f64(A, q, c, lane);
    if (lane == 0) out[e] = cosine_from_dot(dot, qn2[q], cn2[c]);
  }
}

__global__ __launch_bounds__(1024) void merge_mixed_phase2_kernel(
    const unsigned char* __restrict__ recv, int W, int64_t src_stride, int64_t idx_off, int K, double absent,
    const double* __restrict__ prov_d, const int32_t* __restrict__ prov_i, const unsigned long long* __restrict__ fl,
    const int32_t* __restrict__ fl_cnt, const unsigned char* __restrict__ resp_recv, int64_t resp_stride,
    double* __restrict__ out_dist, int32_t* __restrict__ out_idx, int16_t* __restrict__ out_rank,
    int32_t* __restrict__ stats, int MM_FL, double eps2) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double* v = reinterpret_cast<double*>(smem);                                       // [K]
  unsigned long long* best = reinterpret_cast<unsigned long long*>(v + K);           // [K] key of the refined minimum
  unsigned int* besti = reinterpret_cast<unsigned int*>(best + K);                   // [K]
  int* touched = reinterpret_cast<int*>(besti + K);                                  // [K]
  const int q = blockIdx.x;
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    v[k] = prov_d[(int64_t)q * K + k];
    besti[k] = (unsigned int)prov_i[(int64_t)q * K + k];
    best[k] = ~0ull;
    touched[k] = 0;
  }
  __syncthreads();
  const int n = fl_cnt[q];
  auto entry = [&](int e, int& k, double& d, unsigned int& cand) {
    const unsigned long long x = fl[(int64_t)q * MM_FL + e];
    k = (int)(x >> 40);
    const int w = (int)((x >> 32) & 0xff), j = (int)(x & 0xffffffffu);
    d = reinterpret_cast<const double*>(resp_recv + (int64_t)w * resp_stride + 8)[j];
    cand = (unsigned int)reinterpret_cast<const int32_t*>(recv + (int64_t)w * src_stride + idx_off)[(int64_t)q * K + k];
  };
  if (q == 0) {                                          // the response headers' trouble bits (see qpg_flags_stamp)
    int f = 0;
    for (int w = threadIdx.x; w < W; w += blockDim.x) f |= reinterpret_cast<const int32_t*>(resp_recv + (int64_t)w * resp_stride)[1];
    if (f) atomicOr(&stats[1], f);
  }
  for (int e = threadIdx.x; e < n; e += blockDim.x) {
    int k;
    double d;
    unsigned int c;
    entry(e, k, d, c);
    touched[k] = 1;
    atomicMin(&best[k], (unsigned long long)order_key(d));
  }
  __syncthreads();
  for (int k = threadIdx.x; k < K; k += blockDim.x)
    if (touched[k]) besti[k] = 0xffffffffu;
  __syncthreads();
  for (int e = threadIdx.x; e < n; e += blockDim.x) {
    int k;
    double d;
    unsigned int c;
    entry(e, k, d, c);
    if ((unsigned long long)order_key(d) == best[k]) atomicMin(&besti[k], c);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    if (touched[k]) v[k] = key_value(best[k], 0.0);
    out_dist[(int64_t)q * K + k] = v[k];
    out_idx[(int64_t)q * K + k] = (int32_t)besti[k];
  }
  if (threadIdx.x == 0 && n > 0) atomicAdd(&stats[3], n);
  // eps2 > 0 (the responses are f64 dot-product distances): contenders of one code from different shards closer than
  // eps2 — below the resolution at which this form and the reference's own arithmetic order two distances alike — are
  // flagged (stats[1] |= 8) and the host re-matches the clip on the path whose responses ARE reference arithmetic.
  int trouble = 0;
  if (eps2 > 0.0)
    for (int e = threadIdx.x; e < n; e += blockDim.x) {
      int k;
      double d;
      unsigned int c;
      entry(e, k, d, c);
      trouble |= c != besti[k] && d <= key_value(best[k], 0.0) + eps2;
    }
  if (trouble) atomicOr(&stats[1], 8);
  if (!out_rank) return;
  __syncthreads();
  int* rkc = touched;                                                               // rank counters, then reused
  int* s_code = reinterpret_cast<int*>(best);                                       // [K] code at rank r (keys are done)
  block_stable_ranks(v, K, rkc, [&](int k, int r) {
    out_rank[(int64_t)q * K + k] = (int16_t)r;
    s_code[r] = k;
  });
  if (eps2 <= 0.0) return;
  __syncthreads();
  trouble = 0;
  for (int r = threadIdx.x; r + 1 < K; r += blockDim.x) {
    const int ka = s_code[r], kb = s_code[r + 1];
    trouble |= besti[ka] != 0xffffffffu && besti[kb] != 0xffffffffu && v[kb] - v[ka] < eps2;
  }
  if (trouble) atomicOr(&stats[1], 8);
}

// the same in the reference's own arithmetic (refine_pair_f64: one quad per request) - the uncapped sharded path
__global__ __launch_bounds__(256) void shard_refine_ref_kernel(GuardArgs A, const unsigned char* __restrict__ req_recv,
                                                               int64_t req_stride, int R, int q_stride, int64_t cand_base,
                                                               unsigned char* __restrict__ resp, int64_t resp_stride,
                                                               int W, int32_t* __restrict__ stats, int Rq) {
  const int o = blockIdx.x, tid = threadIdx.x;
  const unsigned char* blk = req_recv + (int64_t)o * req_stride;
  refine_carry_flags(req_recv, req_stride, W, resp, resp_stride, o, stats);
  const unsigned long long* ent = reinterpret_cast<const unsigned long long*>(blk + 8);
  double* out = reinterpret_cast<double*>(resp + (int64_t)o * resp_stride + 8);
  const int Qn = R / Rq;
  for (int e0 = blockIdx.y * 64; e0 < R; e0 += gridDim.y * 64) {
    const int t = e0 + (tid >> 2);                                    // position-major slot order (see shard_refine_kernel)
    const int e = t < R ? (t % Qn) * Rq + t / Qn : 0;
    unsigned long long x = ent[e];
    const bool live = t < R && x != ~0ull;
    // (a wave whose 16 slots are all unused skips the evaluation; otherwise idle quads run along on slot data that is
    // valid for SOME request - uniform control flow inside refine_pair_f64's quad shuffles)
    const unsigned long long m = __ballot(live);                      // (wave-uniform: taken before any divergence)
    if (m == 0) continue;
    const unsigned long long xl = __shfl(x, __ffsll((long long)m) - 1, 64);   // a live lane's request
    if (!live) x = xl;
    const int q = o * q_stride + (int)(x >> 48);
    const int64_t c = (int64_t)(unsigned int)(x & 0xffffffffu) - cand_base;
    const double dr = refine_pair_f64(A, q, c, tid & 3);
    if (live && (tid & 3) == 0) out[e] = dr;
  }
}

extern "C" int64_t qpg_merge_mixed_ws_bytes(int Q, int K, int fl_cap) {          // prov_d | prov_i | fl | fl_cnt
  return (Q <= 0 || K <= 0 || fl_cap <= 0) ? 0 : (int64_t)Q * K * 12 + (int64_t)Q * fl_cap * 8 + (int64_t)Q * 4 + 64;
}

extern "C" int qpg_merge_mixed_phase1_f64(qpg_ctx* ctx, void* stream, const void* recv, int W, int64_t src_stride,
                                          int64_t dist_off, int64_t idx_off, int Q, int K, double absent, double eps1,
                                          int R, void* req, int64_t req_stride, void* ws, int64_t ws_bytes,
                                          int32_t* stats, int fl_cap, int64_t flag_off) {
  const int MM_FL = fl_cap;
  const char* name = "qpg_merge_mixed_phase1_f64";
  QPG_REQUIRE(ctx && recv && req && ws && stats, "%s: null pointer", name);
  QPG_REQUIRE(Q == 0 || (R % Q) == 0, "%s: R must be a multiple of Q (R / Q slots per query and shard)", name);
  QPG_REQUIRE(flag_off < 0 || flag_off % 4 == 0, "%s: misaligned flag word", name);
  QPG_REQUIRE(W > 0 && W <= 255 && Q >= 0 && Q < 65536 && K > 0 && K <= 2048 && R > 0 && src_stride % 8 == 0 &&
                  dist_off % 8 == 0 && idx_off % 4 == 0 && req_stride >= 8 + 8 * (int64_t)R && req_stride % 8 == 0 &&
                  eps1 > 0.0 && fl_cap > 0 && ws_bytes >= qpg_merge_mixed_ws_bytes(Q, K, fl_cap) &&
                  (reinterpret_cast<uintptr_t>(ws) % 8) == 0 && (reinterpret_cast<uintptr_t>(req) % 8) == 0,
              "%s: bad size / alignment (W <= 255, Q < 65536, K <= 2048, eps1 > 0)", name);
  if (Q == 0) return QPG_OK;
  unsigned char* w = static_cast<unsigned char*>(ws);
  double* prov_d = reinterpret_cast<double*>(w);
  int32_t* prov_i = reinterpret_cast<int32_t*>(w + (size_t)Q * K * 8);
  unsigned long long* fl = reinterpret_cast<unsigned long long*>(w + (((size_t)Q * K * 12 + 7) / 8) * 8);
  int32_t* fl_cnt = reinterpret_cast<int32_t*>(reinterpret_cast<unsigned char*>(fl) + (size_t)Q * MM_FL * 8);
  hipLaunchKernelGGL(merge_mixed_prologue_kernel, dim3(1), dim3(256), 0, qpg_stream(stream),
                     static_cast<unsigned char*>(req), req_stride, W, static_cast<const unsigned char*>(recv), src_stride,
                     flag_off, stats);
  QPG_LAUNCH_CHECK("merge_mixed_prologue_kernel");
  hipLaunchKernelGGL(merge_mixed_phase1_kernel, dim3(Q), dim3(1024), (size_t)K * 32 + 8,
                     qpg_stream(stream),
                     static_cast<const unsigned char*>(recv), W, src_stride, dist_off, idx_off, K, absent, eps1, R,
                     static_cast<unsigned char*>(req), req_stride, prov_d, prov_i, fl, fl_cnt, stats, MM_FL);
  QPG_LAUNCH_CHECK("merge_mixed_phase1_kernel");
  return QPG_OK;
}

extern "C" int qpg_shard_refine_f64(qpg_ctx* ctx, void* stream, const void* req_recv, int W, int64_t req_stride, int R,
                                    int q_stride, int64_t cand_base, const float* base, int base_is_f16, int T, int F,
                                    const int32_t* cand_t, int G, int n_taps, int tap_stride, const float* q32,
                                    const double* qn2, const double* cn2, void* resp, int64_t resp_stride,
                                    int reference_arithmetic, int32_t* stats, int Rq) {
  const char* name = "qpg_shard_refine_f64";
  QPG_REQUIRE(Rq > 0 && (R % Rq) == 0, "%s: R must be a multiple of Rq (slots per query)", name);
  QPG_REQUIRE(ctx && req_recv && base && cand_t && q32 && qn2 && cn2 && resp, "%s: null pointer", name);
  QPG_REQUIRE(W > 0 && R > 0 && req_stride >= 8 + 8 * (int64_t)R && resp_stride >= 8 + 8 * (int64_t)R && T > 0 && F > 0 &&
                  (F % 4) == 0 && G > 0 && n_taps > 0 && tap_stride > 0 && q_stride >= 0,
              "%s: bad size", name);
  GuardArgs A;
  A.base = base; A.half = base_is_f16; A.q32 = q32; A.cand_t = cand_t; A.T = T; A.F = F; A.G = G; A.n_taps = n_taps;
  A.tap_stride = tap_stride; A.eps = 0.0; A.stats = nullptr;
  if (reference_arithmetic) {
    hipLaunchKernelGGL(shard_refine_ref_kernel, dim3(W, 64), dim3(256), 0, qpg_stream(stream), A,
                       static_cast<const unsigned char*>(req_recv), req_stride, R, q_stride, cand_base,
                       static_cast<unsigned char*>(resp), resp_stride, W, stats, Rq);
    QPG_LAUNCH_CHECK("shard_refine_ref_kernel");
    return QPG_OK;
  }
  const int fast = (n_taps == 6 && F == 1024) ? 1 : 0;
  // a request is ~12 us of dependent gathers whatever the wave count: enough waves for ONE round at the usual ~3 300
  // requests per owner and step (4 W ry >= 4096; W = 1 with 256 waves took 13 rounds, 120 us); the rest leave at once
  int ry = 1024 / W;
  ry = ry < 64 ? 64 : ry;
  hipLaunchKernelGGL(shard_refine_kernel, dim3(W, ry), dim3(256), 0, qpg_stream(stream), A,
                     static_cast<const unsigned char*>(req_recv), req_stride, R, q_stride, cand_base, cn2, qn2,
                     static_cast<unsigned char*>(resp), resp_stride, fast, W, stats, Rq);
  QPG_LAUNCH_CHECK("shard_refine_kernel");
  return QPG_OK;
}

extern "C" int qpg_merge_mixed_phase2_f64(qpg_ctx* ctx, void* stream, const void* recv, int W, int64_t src_stride,
                                          int64_t idx_off, int Q, int K, double absent, const void* ws, int64_t ws_bytes,
                                          const void* resp_recv, int64_t resp_stride, double* out_dist, int32_t* out_idx,
                                          int16_t* out_rank, int32_t* stats, int fl_cap, double eps2) {
  const int MM_FL = fl_cap;
  const char* name = "qpg_merge_mixed_phase2_f64";
  QPG_REQUIRE(ctx && recv && ws && resp_recv && out_dist && out_idx && stats, "%s: null pointer", name);
  QPG_REQUIRE(W > 0 && Q >= 0 && K > 0 && K <= 2048 && fl_cap > 0 && eps2 >= 0.0 &&
                  ws_bytes >= qpg_merge_mixed_ws_bytes(Q, K, fl_cap) && resp_stride % 8 == 0,
              "%s: bad size", name);
  if (Q == 0) return QPG_OK;
  const unsigned char* w = static_cast<const unsigned char*>(ws);
  const double* prov_d = reinterpret_cast<const double*>(w);
  const int32_t* prov_i = reinterpret_cast<const int32_t*>(w + (size_t)Q * K * 8);
  const unsigned long long* fl = reinterpret_cast<const unsigned long long*>(w + (((size_t)Q * K * 12 + 7) / 8) * 8);
  const int32_t* fl_cnt = reinterpret_cast<const int32_t*>(reinterpret_cast<const unsigned char*>(fl) + (size_t)Q * MM_FL * 8);
  hipLaunchKernelGGL(merge_mixed_phase2_kernel, dim3(Q), dim3(1024), (size_t)K * 24, qpg_stream(stream),
                     static_cast<const unsigned char*>(recv), W, src_stride, idx_off, K, absent, prov_d, prov_i, fl, fl_cnt,
                     static_cast<const unsigned char*>(resp_recv), resp_stride, out_dist, out_idx, out_rank, stats, MM_FL,
                     eps2);
  QPG_LAUNCH_CHECK("merge_mixed_phase2_kernel");
  return QPG_OK;
}

// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(1024) void rank_rows_kernel(const T* __restrict__ d, int K, int16_t* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned long long* skey = reinterpret_cast<unsigned long long*>(smem);
  int* scode = reinterpret_cast<int*>(skey + rank_sort_pow2(K));
  T* v = reinterpret_cast<T*>(scode + rank_sort_pow2(K));
  const int q = blockIdx.x;
  for (int k = threadIdx.x; k < K; k += blockDim.x) v[k] = d[(int64_t)q * K + k];
  __syncthreads();
  block_sorted_ranks(v, K, skey, scode, [&](int k, int r) { out[(int64_t)q * K + k] = (int16_t)r; });
}

template <typename T>
static int rank_rows(const char* name, qpg_ctx* ctx, void* stream, const T* d, int Q, int K, int16_t* out) {
  QPG_REQUIRE(ctx && d && out && Q >= 0 && K > 0 && K <= 4096, "%s: bad argument (K <= 4096)", name);
  if (Q == 0) return QPG_OK;
  const size_t sh = 12 * (size_t)rank_sort_pow2(K) + sizeof(T) * (size_t)K;        // K <= 4096: 80 KB at most
  if (sh > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(rank_rows_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)sh) != hipSuccess) {
    qpg_set_error("%s: cannot raise the dynamic LDS limit", name);
    return QPG_EHIP;
  }
  const int threads = rank_sort_pow2(K) >= 2048 ? 1024 : (rank_sort_pow2(K) >= 128 ? rank_sort_pow2(K) / 2 : 64);
  hipLaunchKernelGGL((rank_rows_kernel<T>), dim3(Q), dim3(threads), sh, qpg_stream(stream), d, K, out);
  QPG_LAUNCH_CHECK(name);
  return QPG_OK;
}

extern "C" int qpg_rank_rows_f64(qpg_ctx* ctx, void* stream, const double* d, int Q, int K, int16_t* out) {
  return rank_rows<double>("qpg_rank_rows_f64", ctx, stream, d, Q, K, out);
}
extern "C" int qpg_rank_rows_f32(qpg_ctx* ctx, void* stream, const float* d, int Q, int K, int16_t* out) {
  return rank_rows<float>("qpg_rank_rows_f32", ctx, stream, d, Q, K, out);
}
