// The state-dependent tail of CodeKNN.search_code_knn (GestureKNN.py:501-664) for a whole clip:
// pose-signature term, rank fusion, phase gate, window chaining.  Everything that depends only
// on the query position (the candidate sweeps and their ranks) has already been computed for all
// Q steps; what is left is sequential in the running (last code, last phase block) state and is
// O(K) per step, so it runs as ONE workgroup that walks the M*steps chain without returning to
// the host.  Float paths reproduce the reference's arithmetic: combined scores in float64 in the
// reference's operation order, the 128-d phase-gate cosine in scikit-learn's float32 order.
#include "qpg_common.h"
#include <type_traits>

// ---------------------------------------------------------------------------------------------
// pose-signature distance table: out[p][c] = |sig[p] - sig[c]|_2 (f32), +inf on the diagonal
// (GestureKNN.py:531-536).  The difference is taken in f32 like the reference; the sum of
// squares is accumulated in f64 and rounded once (the reference's np.linalg.norm sums in f32 in
// a BLAS-dependent order; see DESIGN.md "Tie contract").
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void l2_table_kernel(const float* __restrict__ sig, int K, int Dm,
                                                       float* __restrict__ out) {
  const int p = blockIdx.x;
  for (int c = threadIdx.x; c < K; c += blockDim.x) {
    double s = 0.0;
    for (int e = 0; e < Dm; ++e) {
      const float d = f_sub(sig[(int64_t)p * Dm + e], sig[(int64_t)c * Dm + e]);
      s += (double)d * (double)d;
    }
    out[(int64_t)p * K + c] = (c == p) ? __builtin_inff() : (float)sqrt(s);
  }
}

extern "C" int qpg_l2_table_f32(qpg_ctx* ctx, void* stream, const float* sig, int K, int Dm, float* out) {
  QPG_REQUIRE(ctx && sig && out && K > 0 && Dm > 0, "qpg_l2_table_f32: bad argument");
  hipLaunchKernelGGL(l2_table_kernel, dim3(K), dim3(256), 0, qpg_stream(stream), sig, K, Dm, out);
  QPG_LAUNCH_CHECK("l2_table_kernel");
  return QPG_OK;
}

// ---------------------------------------------------------------------------------------------
// sklearn-exact f32 cosine of two 128-d vectors, cooperatively by 4 lanes (lane l = einsum lane).
// `a`/`b` are LDS arrays of 128 floats.  All 4 lanes return the same value.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float lane4_sum(float v, int l) {
  // (l0 + l1) + (l2 + l3) in exactly that order, via shuffles within the aligned group of 4
  const float o1 = __shfl_xor(v, 1, 64);
  const float pair = (l & 1) ? f_add(o1, v) : f_add(v, o1);  // lanes 0,1 -> l0+l1 ; lanes 2,3 -> l2+l3
  const float o2 = __shfl_xor(pair, 2, 64);
  return (l & 2) ? f_add(o2, pair) : f_add(pair, o2);         // (l0+l1)+(l2+l3)
}

__device__ __forceinline__ float einsum_sq_128(const float* v, int l) {
  float a = 0.f;
#pragma unroll
  for (int g = 0; g < 8; ++g) {
#pragma unroll
    for (int u = 3; u >= 0; --u) {
      const float x = v[g * 16 + u * 4 + l];
      a = f_add(f_mul(x, x), a);
    }
  }
  return lane4_sum(a, l);
}

struct ArgMin {
  double v;
  int i;
};
__device__ __forceinline__ ArgMin amin(ArgMin a, ArgMin b) {
  return (b.v < a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}
__device__ __forceinline__ ArgMin wave_argmin(ArgMin m) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    ArgMin t{__shfl_xor(m.v, o, 64), __shfl_xor(m.i, o, 64)};
    m = amin(m, t);
  }
  return m;
}

// ---------------------------------------------------------------------------------------------
// Rank fusion for EVERY possible previous code, in parallel (GestureKNN.py:540-545, 553-555, 574-576).
// The fused score of step q depends on the running state only through the previous code p, and p
// takes K values, so the argmin over codes is tabulated for all (q, p) up front: one wave per (q, p),
// 8 codes per lane, float64 in the reference's operation order
//     combined[c] = (pos_rank[p][c] + freq_rank[c]*0.05) + rank[q][c],   argmin = lowest index.
// What is stored is the winning CANDIDATE index (idx[q][argmin]), so the sequential walk needs a
// single LDS lookup per step instead of an O(K) reduction and a dependent global load.
// ---------------------------------------------------------------------------------------------
#define QPG_KMAX_PER_LANE 16  // K <= 1024

__global__ __launch_bounds__(256) void fuse_best_kernel(const int16_t* __restrict__ rank0,
                                                        const int32_t* __restrict__ idx0,
                                                        const int16_t* __restrict__ rank1,
                                                        const int32_t* __restrict__ idx1,
                                                        const int16_t* __restrict__ pos_rank,
                                                        const int16_t* __restrict__ freq_rank, int Q, int K, int mode,
                                                        int32_t* __restrict__ T0, int32_t* __restrict__ T1) {
  const int lane = threadIdx.x & 63;
  const int64_t task = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);   // (q, p)
  if (task >= (int64_t)Q * K) return;
  const int q = (int)(task / K), p = (int)(task - (int64_t)q * K);
  if (mode == 0) {
    // both modalities in one wave: the pose / frequency part of the score is shared, only the last addend differs
    const int16_t* ra = rank0 + (int64_t)q * K;
    const int16_t* rt = rank1 + (int64_t)q * K;
    ArgMin ma{__builtin_inf(), 0x7fffffff}, mt{__builtin_inf(), 0x7fffffff};
#pragma unroll
    for (int i = 0; i < QPG_KMAX_PER_LANE; ++i) {
      const int c = lane + 64 * i;
      if (c < K) {
        const double pos_score = (double)pos_rank[(int64_t)p * K + c] + (double)freq_rank[c] * 0.05;
        ma = amin(ma, ArgMin{pos_score + (double)ra[c], c});
        mt = amin(mt, ArgMin{pos_score + (double)rt[c], c});
      }
    }
    ma = wave_argmin(ma);
    mt = wave_argmin(mt);
    if (lane == 0) {
      T0[task] = idx0[(int64_t)q * K + ma.i];
      T1[task] = idx1[(int64_t)q * K + mt.i];
    }
    return;
  }
  const int16_t* rk = (mode == 1 ? rank0 : rank1) + (int64_t)q * K;
  const int32_t* ix = (mode == 1 ? idx0 : idx1) + (int64_t)q * K;
  ArgMin m{__builtin_inf(), 0x7fffffff};
  double vals[QPG_KMAX_PER_LANE];
#pragma unroll
  for (int i = 0; i < QPG_KMAX_PER_LANE; ++i) {
    const int c = lane + 64 * i;
    vals[i] = __builtin_inf();
    if (c < K) {
      const double pos_score = (double)pos_rank[(int64_t)p * K + c] + (double)freq_rank[c] * 0.05;
      vals[i] = pos_score + (double)rk[c];
      m = amin(m, ArgMin{vals[i], c});
    }
  }
  m = wave_argmin(m);
  // single-modality modes: the two best codes go through the phase gate (GestureKNN.py:596, 613)
  ArgMin m2{__builtin_inf(), 0x7fffffff};
#pragma unroll
  for (int i = 0; i < QPG_KMAX_PER_LANE; ++i) {
    const int c = lane + 64 * i;
    if (c < K && c != m.i) m2 = amin(m2, ArgMin{vals[i], c});
  }
  m2 = wave_argmin(m2);
  if (lane == 0) {
    T0[task] = ix[m.i];
    T1[task] = ix[m2.i];
  }
}

// The same tables for the two-modality mode, by BRANCH AND BOUND over the ranks (round 3; round 2's version gave every
// 16-lane group one previous code and all K codes, 8 per lane per 16-byte load: 12.6 M f64 score evaluations, 15 us).  A rank row is a permutation of 0..K-1 and the
// other two addends are non-negative, so the fused score of the code at rank r is >= r: scanning the codes of step q in
// rank order (inv[r] = code at rank r, rebuilt in LDS per block), a 16-lane group can stop as soon as the next chunk's
// first rank exceeds the best score so far - about 2 x sqrt(K) ranks instead of K codes (three chunks of 16 at K = 512
// instead of 32 codes per lane; scanned four chunks at a time).  Same f64 operations per visited code, `(pos + freq * 0.05) + rank`, and the lowest code
// index among equal scores (a later rank r == best with a zero pose / frequency part can still tie: the scan continues
// while r <= best).  Block = the 16 previous codes p0..p0+15 of one step q (K % 16 == 0).  A row that is not a permutation
// (never produced by the rank kernels; checked anyway) makes its block scan every code.
__global__ __launch_bounds__(256) void fuse_best_ranked_kernel(const int16_t* __restrict__ rank0,
                                                               const int32_t* __restrict__ idx0,
                                                               const int16_t* __restrict__ rank1,
                                                               const int32_t* __restrict__ idx1,
                                                               const int16_t* __restrict__ pos_rank,
                                                               const int16_t* __restrict__ freq_rank, int Q, int K,
                                                               int32_t* __restrict__ T0, int32_t* __restrict__ T1) {
  extern __shared__ __attribute__((aligned(16))) int16_t inv[];          // [2][K]
  __shared__ int bad;
  const int tid = threadIdx.x, l16 = tid & 15;
  const int64_t task = (int64_t)blockIdx.x * 16 + (tid >> 4);            // (q, p): one per 16-lane group
  const int q = (int)(((int64_t)blockIdx.x * 16) / K);
  const int p = (int)(task - (int64_t)q * K);
  if (tid == 0) bad = 0;
  for (int c = tid; c < 2 * K; c += blockDim.x) inv[c] = -1;
  __syncthreads();
  for (int c = tid; c < K; c += blockDim.x) {
    const int ra = rank0[(int64_t)q * K + c], rt = rank1[(int64_t)q * K + c];
    if ((unsigned)ra < (unsigned)K) inv[ra] = (int16_t)c; else bad = 1;
    if ((unsigned)rt < (unsigned)K) inv[K + rt] = (int16_t)c; else bad = 1;
  }
  __syncthreads();
  for (int c = tid; c < 2 * K; c += blockDim.x)
    if (inv[c] < 0) bad = 1;
  __syncthreads();
  const bool full = bad != 0;
  const int16_t* pr = pos_rank + (int64_t)p * K;
  auto xchg = [](ArgMin m, auto tag) {
    constexpr int PJ = decltype(tag)::value;
    const unsigned long long b = (unsigned long long)__double_as_longlong(m.v);
    const unsigned int lo = (unsigned int)lane_xor<PJ>((int)(unsigned int)b);
    const unsigned int hi = (unsigned int)lane_xor<PJ>((int)(unsigned int)(b >> 32));
    return ArgMin{__longlong_as_double((long long)(((unsigned long long)hi << 32) | lo)), lane_xor<PJ>(m.i)};
  };
  // (four chunks of 16 ranks per round, their gathers in flight together: a task needs ~3 chunks, and a round is one
  // dependent gather latency either way)
  auto scan = [&](const int16_t* iv, const int16_t* rk) {
    ArgMin m{__builtin_inf(), 0x7fffffff};
    for (int base = 0; base < K; base += 64) {
      if (!full && (double)base > m.v) break;                 // (m is uniform over the group after the reduction)
      int c[4];
      double pv[4], fv[4], rr[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r = base + 16 * u + l16;
        const bool ok = r < K;
        c[u] = ok ? (full ? r : (int)iv[r]) : -1;
        pv[u] = ok ? (double)pr[c[u]] : 0.0;
        fv[u] = ok ? (double)freq_rank[c[u]] : 0.0;
        rr[u] = ok ? (full ? (double)rk[c[u]] : (double)r) : 0.0;
      }
      ArgMin x{__builtin_inf(), 0x7fffffff};
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (c[u] >= 0) x = amin(x, ArgMin{(pv[u] + fv[u] * 0.05) + rr[u], c[u]});
      x = amin(x, xchg(x, std::integral_constant<int, 8>{}));
      x = amin(x, xchg(x, std::integral_constant<int, 4>{}));
      x = amin(x, xchg(x, std::integral_constant<int, 2>{}));
      x = amin(x, xchg(x, std::integral_constant<int, 1>{}));
      m = amin(m, x);
    }
    return m;
  };
  const ArgMin ma = scan(inv, rank0 + (int64_t)q * K);
  const ArgMin mt = scan(inv + K, rank1 + (int64_t)q * K);
  if (l16 == 0) {
    T0[task] = idx0[(int64_t)q * K + ma.i];
    T1[task] = idx1[(int64_t)q * K + mt.i];
  }
}

// One modality's half of fuse_best_ranked_kernel (round 5): T[task] = candidate of the code with the smallest fused score
// pos_rank[p][c] + 0.05 freq_rank[c] + rank[q][c] - the audio order's and the text order's winners are independent
// (GestureKNN.py:574-576 / :553-555: two separate argsorts), so each side's launch can follow its own select on its own
// stream and the join moves behind them.  Same scan, same operations, same tie rule as the two-table kernel.
__global__ __launch_bounds__(256) void fuse_best_ranked_one_kernel(const int16_t* __restrict__ rank,
                                                                   const int32_t* __restrict__ idx,
                                                                   const int16_t* __restrict__ pos_rank,
                                                                   const int16_t* __restrict__ freq_rank, int Q, int K,
                                                                   int32_t* __restrict__ T) {
  extern __shared__ __attribute__((aligned(16))) int16_t inv[];          // [K]
  __shared__ int bad;
  const int tid = threadIdx.x, l16 = tid & 15;
  const int64_t task = (int64_t)blockIdx.x * 16 + (tid >> 4);            // (q, p): one per 16-lane group
  const int q = (int)(((int64_t)blockIdx.x * 16) / K);
  const int p = (int)(task - (int64_t)q * K);
  if (tid == 0) bad = 0;
  for (int c = tid; c < K; c += blockDim.x) inv[c] = -1;
  __syncthreads();
  for (int c = tid; c < K; c += blockDim.x) {
    const int r = rank[(int64_t)q * K + c];
    if ((unsigned)r < (unsigned)K) inv[r] = (int16_t)c; else bad = 1;
  }
  __syncthreads();
  for (int c = tid; c < K; c += blockDim.x)
    if (inv[c] < 0) bad = 1;
  __syncthreads();
  const bool full = bad != 0;
  const int16_t* pr = pos_rank + (int64_t)p * K;
  const int16_t* rk = rank + (int64_t)q * K;
  auto xchg = [](ArgMin m, auto tag) {
    constexpr int PJ = decltype(tag)::value;
    const unsigned long long b = (unsigned long long)__double_as_longlong(m.v);
    const unsigned int lo = (unsigned int)lane_xor<PJ>((int)(unsigned int)b);
    const unsigned int hi = (unsigned int)lane_xor<PJ>((int)(unsigned int)(b >> 32));
    return ArgMin{__longlong_as_double((long long)(((unsigned long long)hi << 32) | lo)), lane_xor<PJ>(m.i)};
  };
  ArgMin m{__builtin_inf(), 0x7fffffff};
  for (int base = 0; base < K; base += 64) {
    if (!full && (double)base > m.v) break;                   // (m is uniform over the group after the reduction)
    int c[4];
    double pv[4], fv[4], rr[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = base + 16 * u + l16;
      const bool ok = r < K;
      c[u] = ok ? (full ? r : (int)inv[r]) : -1;
      pv[u] = ok ? (double)pr[c[u]] : 0.0;
      fv[u] = ok ? (double)freq_rank[c[u]] : 0.0;
      rr[u] = ok ? (full ? (double)rk[c[u]] : (double)r) : 0.0;
    }
    ArgMin x{__builtin_inf(), 0x7fffffff};
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (c[u] >= 0) x = amin(x, ArgMin{(pv[u] + fv[u] * 0.05) + rr[u], c[u]});
    x = amin(x, xchg(x, std::integral_constant<int, 8>{}));
    x = amin(x, xchg(x, std::integral_constant<int, 4>{}));
    x = amin(x, xchg(x, std::integral_constant<int, 2>{}));
    x = amin(x, xchg(x, std::integral_constant<int, 1>{}));
    m = amin(m, x);
  }
  if (l16 == 0) T[task] = idx[(int64_t)q * K + m.i];
}

struct TailArgs {
  const int32_t* T0;         // [Q][K] candidate index of the first gate candidate given previous code p
  const int32_t* T1;         // [Q][K] second gate candidate
  const int32_t* code;       // [N][code_ld]
  int code_ld;
  const int32_t* cidx0;      // [G0] code column of grid position (source of T0)
  const int32_t* pslot0;     // [G0] phase start frame int(k/398*240)
  int G0;
  const int32_t* cidx1;
  const int32_t* pslot1;
  int G1;
  const float* phase;        // [N][Tp][2][8]  (p, a): one frame = 16 contiguous floats [phase | amp]
  int Tp;
  int M, steps, step_codes, codes_per_window;
  int K;
  int seed_code;             // seed of chain 0 when seed_codes is NULL
  const float* seed_phase;   // [n_chains][8][16]
  const int32_t* seed_codes; // [n_chains] (device) or NULL: several independent clips of M windows each in one launch
  int n_chains;              // chain c owns steps [c M steps, (c + 1) M steps) of the tables and of the outputs
  int64_t status_stride;     // ints between the chains' status pairs
  int32_t* out_codes;        // [M][codes_per_window]
  float* out_phase;          // [M][steps][8][16]
  int32_t* out_vote;         // [M][steps]
  int32_t* out_status;       // [2] [0]: 0 ok, 1 = an absent code won a rank fusion (reference would raise IndexError);
                             //     [1]: copy of *guard_flags (0 without it)
  const int32_t* guard_flags;  // the sweeps' / selects' trouble word (stats[1]), or NULL: rides out with the results
};

// ---------------------------------------------------------------------------------------------
// The sequential walk: ONE wave.  A workgroup of one wave makes __syncthreads() a plain LDS fence, so the
// chain has no multi-wave barrier latency.  Per step: two LDS lookups (the gate candidates for the current
// previous code), the 128-d phase-gate cosine in scikit-learn's f32 arithmetic, the state update.
//
// The global loads are taken OFF the dependent chain by speculation: the next step's previous code can
// only be the last payload code of one of the two current gate candidates, so as soon as their payloads
// are known the phase blocks / payloads of all four possible next candidates are requested, and they
// arrive while the current step's gate arithmetic runs (r01: 3.2 us/step with dependent loads).
// ---------------------------------------------------------------------------------------------
struct CandRegs {
  float2 head, tail;   // this lane's 2 floats of the first / last 8-frame block (8 x 16 floats each)
  int pay;             // lanes 0..3: the candidate's 4 codes
  int absent;          // the table entry was -1 (code absent from the DB): only an error if this candidate is USED
};

__device__ __forceinline__ CandRegs load_cand(const TailArgs& A, int which, int ci, const int* s_cidx,
                                              const int* s_pslot, int lane) {
  // which: 0 -> table T0's grid, 1 -> table T1's grid.  s_cidx/s_pslot: LDS copies [2][64].
  const int G = which ? A.G1 : A.G0;
  const int cc = ci < 0 ? 0 : ci;
  const int j = cc / G, g = cc - j * G;
  const int ps = s_pslot[which * 64 + g];
  const float* b = A.phase + ((int64_t)j * A.Tp + ps) * 16;   // rows [ps, ps+8) and [ps+24, ps+32): 128 floats each
  CandRegs r;
  r.absent = ci < 0;
  r.head = reinterpret_cast<const float2*>(b)[lane];
  r.tail = reinterpret_cast<const float2*>(b + 384)[lane];
  r.pay = A.code[(int64_t)j * A.code_ld + s_cidx[which * 64 + g] + (lane & 3)];
  return r;
}

// einsum_sq of a 128-vector stored TRANSPOSED in LDS (vt[l*32 + i] = v[4*i + l]): lane l's chain elements
// are contiguous, read as 8 x 16 B.  Order inside the chain: 16-element groups g = i>>2 ascending, within a
// group u = i&3 visited 3,2,1,0 (NumPy einsum's unrolled order).
__device__ __forceinline__ float einsum_sq_128_t(const float* vt, int l) {
  const f32x4* p = reinterpret_cast<const f32x4*>(vt + l * 32);
  float a = 0.f;
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    const f32x4 x = p[g];                       // u = 0..3 of group g
    a = f_add(f_mul(x.w, x.w), a);
    a = f_add(f_mul(x.z, x.z), a);
    a = f_add(f_mul(x.y, x.y), a);
    a = f_add(f_mul(x.x, x.x), a);
  }
  return lane4_sum(a, l);
}

__device__ __forceinline__ int tpos(int e) { return (e & 3) * 32 + (e >> 2); }   // transposed slot of element e

__global__ __launch_bounds__(64) void match_walk_kernel(TailArgs A) {
  extern __shared__ __attribute__((aligned(16))) int32_t tab[];   // [2][steps][K] gate candidates of this window
  __shared__ __attribute__((aligned(16))) float va[2][128], vb[2][128];   // gate vectors, transposed layout
  __shared__ float nrm[4];
  __shared__ float score[2];
  __shared__ int wincodes[64];
  __shared__ int s_cidx[128], s_pslot[128];

  const int lane = threadIdx.x, K = A.K;
  // the running phase block (8 frames x [8 phase | 8 amp] = 128 floats) lives in registers: 2 floats per lane
  float2 prev = reinterpret_cast<const float2*>(A.seed_phase)[lane];
  if (lane < A.G0) {
    s_cidx[lane] = A.cidx0[lane];
    s_pslot[lane] = A.pslot0[lane];
  }
  if (lane < A.G1) {
    s_cidx[64 + lane] = A.cidx1[lane];
    s_pslot[64 + lane] = A.pslot1[lane];
  }
  int prev_code = A.seed_codes ? A.seed_codes[0] : A.seed_code;
  int bad = 0;
  const float eps10 = 10.f * 1.1920928955078125e-07f;
  const int last_idx = A.codes_per_window - 1;                 // the next window is seeded by this kept code
  const int e0 = 2 * lane;                                      // this lane's two block elements

  for (int w = 0; w < A.M; ++w) {
    // this window's gate tables -> LDS (steps*K*2 i32, 16-B loads)
    const int n4 = A.steps * K / 4;
    __syncthreads();
    for (int t = 0; t < 2; ++t) {
      const int4* src = reinterpret_cast<const int4*>((t ? A.T1 : A.T0) + (int64_t)w * A.steps * K);
      int4* dst = reinterpret_cast<int4*>(tab + t * A.steps * K);
      for (int v = lane; v < n4; v += 64) dst[v] = src[v];
    }
    __syncthreads();
    CandRegs cur[2];
    cur[0] = load_cand(A, 0, tab[prev_code], s_cidx, s_pslot, lane);
    cur[1] = load_cand(A, 1, tab[A.steps * K + prev_code], s_cidx, s_pslot, lane);

    for (int s = 0; s < A.steps; ++s) {
      bad |= cur[0].absent | cur[1].absent;
      // gate vectors straight from registers (GestureKNN.py:636):
      //   a = [prev[-5:], head[:3]] -> a[e] = prev[48+e] (e < 80), head[e-80] (e >= 80)
      //   b = [prev[-3:], head[:5]] -> b[e] = prev[80+e] (e < 48), head[e-48] (e >= 48)
      // lane owns block elements e0, e0+1: prev element e0 lands at a[e0-48] / b[e0-80], head at a[e0+80] / b[e0+48]
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        if (e0 >= 48) {
          va[k][tpos(e0 - 48)] = prev.x;
          va[k][tpos(e0 - 47)] = prev.y;
        }
        if (e0 >= 80) {
          vb[k][tpos(e0 - 80)] = prev.x;
          vb[k][tpos(e0 - 79)] = prev.y;
        }
        if (e0 < 48) {
          va[k][tpos(e0 + 80)] = cur[k].head.x;
          va[k][tpos(e0 + 81)] = cur[k].head.y;
        }
        if (e0 < 80) {
          vb[k][tpos(e0 + 48)] = cur[k].head.x;
          vb[k][tpos(e0 + 49)] = cur[k].head.y;
        }
      }
      // speculation: the last payload code of either candidate is the next step's previous code
      CandRegs nxt[2][2];
      const bool spec = s + 1 < A.steps;
      if (spec) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int pn = __shfl(cur[k].pay, 3, 64);
          nxt[k][0] = load_cand(A, 0, tab[(s + 1) * K + pn], s_cidx, s_pslot, lane);
          nxt[k][1] = load_cand(A, 1, tab[A.steps * K + (s + 1) * K + pn], s_cidx, s_pslot, lane);
        }
      }
      __syncthreads();
      // norms: lanes 0-3 |a0|, 4-7 |b0|, 8-11 |a1|, 12-15 |b1|
      if (lane < 16) {
        const int l = lane & 3, grp = lane >> 2;
        const float* v = (grp & 1) ? vb[grp >> 1] : va[grp >> 1];
        float n = f_sqrt(einsum_sq_128_t(v, l));
        if (n < eps10) n = 1.f;
        if (l == 0) nrm[grp] = n;
      }
      __syncthreads();
      // normalised difference, every lane 2 slots of each vector (slot order is irrelevant here), in place
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const float na = nrm[2 * k], nb = nrm[2 * k + 1];
        const float2 xa = reinterpret_cast<const float2*>(va[k])[lane], xb = reinterpret_cast<const float2*>(vb[k])[lane];
        float2 d;
        d.x = f_sub(f_div(xa.x, na), f_div(xb.x, nb));
        d.y = f_sub(f_div(xa.y, na), f_div(xb.y, nb));
        reinterpret_cast<float2*>(va[k])[lane] = d;
      }
      __syncthreads();
      if (lane < 8) {
        const int l = lane & 3, k = lane >> 2;
        const float sc = f_mul(0.5f, einsum_sq_128_t(va[k], l));
        if (l == 0) score[k] = sc;
      }
      __syncthreads();
      const int fi = (score[1] < score[0]) ? 1 : 0;            // list.index(min): first on ties
      // append the winner's 4 codes, carry its last-8-frame block (GestureKNN.py:648-657)
      prev = fi ? cur[1].tail : cur[0].tail;
      reinterpret_cast<float2*>(A.out_phase + ((int64_t)w * A.steps + s) * 128)[lane] = prev;
      const int wpay = fi ? cur[1].pay : cur[0].pay;
      if (lane < A.step_codes) wincodes[s * A.step_codes + lane] = wpay;
      if (lane == 0) A.out_vote[w * A.steps + s] = fi;
      prev_code = __shfl(wpay, A.step_codes - 1, 64);
      if (spec) {
        cur[0] = fi ? nxt[1][0] : nxt[0][0];
        cur[1] = fi ? nxt[1][1] : nxt[0][1];
      }
    }
    __syncthreads();
    // window result = first codes_per_window codes; the next window is seeded by the LAST KEPT code
    // (motion_output[-1][-1], GestureKNN.py:800) and the last phase block.
    if (lane < A.codes_per_window) A.out_codes[(int64_t)w * A.codes_per_window + lane] = wincodes[lane];
    prev_code = wincodes[last_idx];
  }
  __threadfence_system();          // (status pair last, behind a system-scope fence: see gate_chase_kernel)
  __syncthreads();
  if (lane == 0) {
    A.out_status[0] = bad;
    __threadfence_system();
    A.out_status[1] = A.guard_flags ? A.guard_flags[0] : 0;
  }
}


// ---------------------------------------------------------------------------------------------
// Tabulated walk.  The running state entering step q is (previous code, previous phase block), and BOTH are
// functions of the previous step's winning candidate, which is one of the two gate candidates of the previous
// code before it: at most 2K states sigma = (p', vote) per step.  So the phase gate is evaluated for EVERY
// reachable state of EVERY step in parallel (gate_table_kernel: 8 lanes per evaluation, same f32 arithmetic in
// the same order as match_walk_kernel), and the sequential part shrinks to Q dependent 2-byte LDS lookups
// (gate_chase_kernel), followed by a parallel gather of the winners' codes / phase blocks.
//   G[q][sigma] = (p << 1) | vote, p = the previous code seen by step q in state sigma; G[0][0] = the seed's step.
// ---------------------------------------------------------------------------------------------
struct GateGeom {
  int s_last, off_last;   // the kept code that seeds the next window: step and offset inside its payload
};

__device__ __forceinline__ const float* cand_block(const TailArgs& A, int which, int ci, int* pay_base) {
  const int G = which ? A.G1 : A.G0;
  const int cc = ci < 0 ? 0 : ci;
  const int j = cc / G, g = cc - j * G;
  const int ps = (which ? A.pslot1 : A.pslot0)[g];
  *pay_base = j * A.code_ld + (which ? A.cidx1 : A.cidx0)[g];
  return A.phase + ((int64_t)j * A.Tp + ps) * 16;
}

// One gate evaluation by 8 lanes (k = which candidate of the pair, l = einsum lane): the state's previous phase block
// `prev` (128 floats) and previous code p -> (p << 1) | vote.  Every lane of the group returns the same value.
__device__ __forceinline__ unsigned int gate_eval(const TailArgs& A, int q, int p, const float* prev, int lane) {
  const int K = A.K;
  const int k = (lane >> 2) & 1, l = lane & 3;
  const int ck = (k ? A.T1 : A.T0)[(int64_t)q * K + p];
  int pb_unused;
  const float* head = cand_block(A, k, ck, &pb_unused);
  // a = [prev[48:], head[:80]], b = [prev[80:], head[:48]]  (GestureKNN.py:636); lane l owns e = 16g + 4u + l
  // (Reading the vectors as 16-byte pieces, 8 per lane, and transposing 4 x 4 inside the quad with DPP moves - 16 vector
  // loads per lane instead of 128 scalar ones - measured SLOWER, 20.5 us against 17.5: the registers of the staged pieces
  // cost more occupancy than the load instructions cost issue slots.)
  float xa[32], xb[32];
  float sa = 0.f, sb = 0.f;
#pragma unroll
  for (int g = 0; g < 8; ++g) {
#pragma unroll
    for (int u = 3; u >= 0; --u) {
      const int e = g * 16 + u * 4 + l;
      const float va = e < 80 ? prev[48 + e] : head[e - 80];
      const float vb = e < 48 ? prev[80 + e] : head[e - 48];
      xa[g * 4 + u] = va;
      xb[g * 4 + u] = vb;
      sa = f_add(f_mul(va, va), sa);
      sb = f_add(f_mul(vb, vb), sb);
    }
  }
  const float eps10 = 10.f * 1.1920928955078125e-07f;
  float na = f_sqrt(lane4_sum(sa, l)), nb = f_sqrt(lane4_sum(sb, l));
  if (na < eps10) na = 1.f;
  if (nb < eps10) nb = 1.f;
  float sd = 0.f;
#pragma unroll
  for (int g = 0; g < 8; ++g) {
#pragma unroll
    for (int u = 3; u >= 0; --u) {
      const float d = f_sub(f_div(xa[g * 4 + u], na), f_div(xb[g * 4 + u], nb));
      sd = f_add(f_mul(d, d), sd);
    }
  }
  const float score = f_mul(0.5f, lane4_sum(sd, l));
  const float other = __shfl_xor(score, 4, 64);
  const float s0 = k ? other : score, s1 = k ? score : other;
  const int fi = (s1 < s0) ? 1 : 0;                     // list.index(min): first on ties
  return (unsigned int)((p << 1) | fi);
}

// previous phase block and previous code of the state "candidate ci of table kp won step q - 1" (s: step inside the window)
__device__ __forceinline__ const float* gate_prev(const TailArgs& A, const GateGeom& geo, int s, int kp, int ci, int* p) {
  int pb;
  const float* prev = cand_block(A, kp, ci, &pb) + 384;   // rows [ps+24, ps+32): the winner's last 8 frames
  *p = A.code[pb + (s == 0 ? geo.off_last : A.step_codes - 1)];
  return prev;
}

__global__ __launch_bounds__(256) void gate_table_kernel(TailArgs A, GateGeom geo, uint16_t* __restrict__ Gt) {
  const int K = A.K, Qc = A.M * A.steps, Q = Qc * A.n_chains;
  const int lane = threadIdx.x & 63;
  const int64_t task = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 3;          // (q, sigma)
  const int64_t n_task = (int64_t)Q * 2 * K;
  const bool live = task < n_task;
  const int q = live ? (int)(task / (2 * K)) : 0;
  const int sigma = live ? (int)(task - (int64_t)q * 2 * K) : 0;
  const int chain = q / Qc;
  const bool first = q == chain * Qc;                   // a chain's first step: the seed's state only
  if (first && sigma != 0) return;                      // uniform per 8-lane group; the shuffles below are group-local
  const int s = q % A.steps;
  int p;
  const float* prev;                                    // 128 floats: the previous phase block
  if (first) {
    p = A.seed_codes ? A.seed_codes[chain] : A.seed_code;
    prev = A.seed_phase + (int64_t)chain * 128;
  } else {
    const int pp = sigma >> 1, kp = sigma & 1;
    const int ci = (kp ? A.T1 : A.T0)[(int64_t)(q - 1) * K + pp];
    prev = gate_prev(A, geo, s, kp, ci, &p);
  }
  const unsigned int g = gate_eval(A, q, p, prev, lane);
  if (live && (lane & 7) == 0) Gt[task] = (uint16_t)g;
}

// The same table for MANY chains per launch (round 5; 16 clips: 24 576 blocks of the kernel above fill the chip and the
// table takes 190-200 us).  A state's gate depends on (previous code pp, vote kp) only through the candidate that won step
// q - 1, ci = T_kp[q - 1][pp] - and the 2 K states of a step share a few dozen to a few hundred distinct winners.  Block =
// one step q: the states' winners go into an LDS hash table (key = (candidate, table)), every DISTINCT key is evaluated
// once (8 lanes each, the code above), every state copies its key's result.  Same table, bit for bit
// (tests/test_gpu_fullsize.py runs the batched walk on both).  16 clips: 203 -> 30 us; ONE clip (48 blocks): 17.9 -> 13.3 us
// (round 4's deduplicated gate - a marked-winner table and a second lookup per chase step - measured 20-21 us; this one
// leaves the table's format and the chase alone).
#define GD_SLOTS 2048
#define GD_THREADS 1024
__global__ __launch_bounds__(GD_THREADS) void gate_table_dedup_kernel(TailArgs A, GateGeom geo, uint16_t* __restrict__ Gt) {
  __shared__ int key[GD_SLOTS];                 // ((candidate << 1) | table) + 1; 0: empty
  __shared__ unsigned short val[GD_SLOTS];
  __shared__ unsigned short ulist[GD_SLOTS];    // slots of the distinct keys
  __shared__ int n_u;
  const int K = A.K, Qc = A.M * A.steps, tid = threadIdx.x, lane = tid & 63;
  const int q = blockIdx.x, chain = q / Qc, s = q % A.steps;
  uint16_t* out = Gt + (int64_t)q * 2 * K;
  if (q == chain * Qc) {                        // a chain's first step: the seed's state only
    if (tid < 8) {
      const int p = A.seed_codes ? A.seed_codes[chain] : A.seed_code;
      const unsigned int g = gate_eval(A, q, p, A.seed_phase + (int64_t)chain * 128, lane);
      if (tid == 0) out[0] = (uint16_t)g;
    }
    return;
  }
  for (int i = tid; i < GD_SLOTS; i += GD_THREADS) key[i] = 0;
  if (tid == 0) n_u = 0;
  __syncthreads();
  constexpr int PER = 2 * 512 / GD_THREADS;     // states per thread at K = 512 (the launcher's bound)
  int slot[PER];
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int sigma = u * GD_THREADS + tid;
    slot[u] = -1;
    if (sigma < 2 * K) {
      const int pp = sigma >> 1, kp = sigma & 1;
      const int ci = (kp ? A.T1 : A.T0)[(int64_t)(q - 1) * K + pp];
      const int kv = (((ci < 0 ? 0 : ci) << 1) | kp) + 1;          // (an absent winner reads candidate 0, as cand_block does)
      unsigned int h = ((unsigned int)kv * 2654435761u) >> 21;     // 11 bits
      for (;;) {
        const int old = atomicCAS(&key[h], 0, kv);
        if (old == 0) {
          ulist[atomicAdd(&n_u, 1)] = (unsigned short)h;
          break;
        }
        if (old == kv) break;
        h = (h + 1) & (GD_SLOTS - 1);
      }
      slot[u] = (int)h;
    }
  }
  __syncthreads();
  const int nu = n_u;
  for (int u = tid >> 3; u < nu; u += GD_THREADS >> 3) {           // (uniform per 8-lane group)
    const int h = ulist[u], kv = key[h] - 1;
    int p;
    const float* prev = gate_prev(A, geo, s, kv & 1, kv >> 1, &p);
    const unsigned int g = gate_eval(A, q, p, prev, lane);
    if ((lane & 7) == 0) val[h] = (unsigned short)g;
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < PER; ++u)
    if (slot[u] >= 0) out[u * GD_THREADS + tid] = val[slot[u]];
}

#define QPG_CHASE_QMAX 2048
__global__ __launch_bounds__(1024) void gate_chase_kernel(TailArgs A, const uint16_t* __restrict__ Gt) {
  extern __shared__ __attribute__((aligned(16))) uint16_t gl[];     // 2 x [steps][2K]: current window + the next being staged
  __shared__ uint16_t sig[QPG_CHASE_QMAX];
  __shared__ int bad_s;
  const int K = A.K, Q = A.M * A.steps, tid = threadIdx.x, nt = blockDim.x;
  const int per_w = A.steps * 2 * K;                                // u16 per window
  // one block per chain (clip): everything below is the chain's own slice of the tables and of the outputs
  const int chain = blockIdx.x;
  const int64_t q0 = (int64_t)chain * Q;
  Gt += q0 * 2 * K;
  A.T0 += q0 * K;
  A.T1 += q0 * K;
  A.out_phase += q0 * 128;
  A.out_vote += q0;
  A.out_codes += (int64_t)chain * A.M * A.codes_per_window;
  A.out_status += (int64_t)chain * A.status_stride;
  if (tid == 0) bad_s = 0;
  int sigma = 0;
  // two LDS buffers: the other waves stage window w+1's table while lane 0 of wave 0 chases window w
  auto stage = [&](int w, int first, int step) {
    const int4* src = reinterpret_cast<const int4*>(Gt + (int64_t)w * per_w);
    int4* dst = reinterpret_cast<int4*>(gl + (size_t)(w & 1) * per_w);
    for (int v = first; v < per_w / 8; v += step) dst[v] = src[v];
  };
  stage(0, tid, nt);
  __syncthreads();
  for (int w = 0; w < A.M; ++w) {
    if (tid >= 64 && w + 1 < A.M) stage(w + 1, tid - 64, nt - 64);
    if (tid == 0) {
      const uint16_t* g = gl + (size_t)(w & 1) * per_w;
      for (int s = 0; s < A.steps; ++s) {
        sigma = (w == 0 && s == 0) ? g[0] : g[s * 2 * K + sigma];
        sig[w * A.steps + s] = (uint16_t)sigma;
      }
    }
    __syncthreads();
  }
  // parallel epilogue: the winners' phase blocks, votes, codes; absent-candidate check of every visited gate
  for (int i = tid; i < Q * 32; i += nt) {                         // 32 x 16 B per phase block
    const int q = i >> 5, v = i & 31;
    const int sg = sig[q], p = sg >> 1, fi = sg & 1;
    const int ci = (fi ? A.T1 : A.T0)[(int64_t)q * K + p];
    int pb;
    const float* blk = cand_block(A, fi, ci, &pb) + 384;
    reinterpret_cast<f32x4*>(A.out_phase + (int64_t)q * 128)[v] = reinterpret_cast<const f32x4*>(blk)[v];
    if (v == 0) {
      A.out_vote[q] = fi;
      if (A.T0[(int64_t)q * K + p] < 0 || A.T1[(int64_t)q * K + p] < 0) bad_s = 1;
    }
  }
  for (int i = tid; i < A.M * A.codes_per_window; i += nt) {
    const int w = i / A.codes_per_window, c = i - w * A.codes_per_window;
    const int q = w * A.steps + c / A.step_codes;
    const int sg = sig[q], p = sg >> 1, fi = sg & 1;
    const int ci = (fi ? A.T1 : A.T0)[(int64_t)q * K + p];
    int pb;
    cand_block(A, fi, ci, &pb);
    A.out_codes[i] = A.code[pb + c % A.step_codes];
  }
  // The status pair is written LAST and behind a system-scope fence: when the outputs live in pinned host memory (the
  // matcher's zero-copy results) a host that sees the pair also sees the codes and votes of every thread above.
  __threadfence_system();
  __syncthreads();
  if (tid == 0) {
    A.out_status[0] = bad_s;
    __threadfence_system();
    A.out_status[1] = A.guard_flags ? A.guard_flags[0] : 0;
  }
}

__global__ void status_only_kernel(int32_t* out_status, const int32_t* guard_flags) {
  out_status[0] = 0;
  out_status[1] = guard_flags ? guard_flags[0] : 0;
}

// One modality's gate-candidate table, launched behind that modality's select on ITS stream (sweep_tables for the walk):
// rank i16 [Q][K] (a permutation per row), idx i32 [Q][K], T i32 [Q][K] = the T0 (audio) or T1 (text) region of the walk's
// gate_tables.  qpg_match_steps* with QPG_MODE_PREFUSED then starts at the gate table.  K % 16 == 0, K <= 4096.
extern "C" int qpg_fuse_best_ranked(qpg_ctx* ctx, void* stream, const int16_t* rank, const int32_t* idx,
                                    const int16_t* pos_rank, const int16_t* freq_rank, int Q, int K, int32_t* T) {
  QPG_REQUIRE(ctx && rank && idx && pos_rank && freq_rank && T, "qpg_fuse_best_ranked: null pointer");
  QPG_REQUIRE(Q > 0 && K > 0 && (K % 16) == 0 && K <= 4096 && ((int64_t)Q * K) / 16 < 0x7fffffffll,
              "qpg_fuse_best_ranked: K %% 16 == 0, K <= 4096");
  hipLaunchKernelGGL(fuse_best_ranked_one_kernel, dim3((unsigned)(((int64_t)Q * K) / 16)), dim3(256), 2 * (size_t)K,
                     qpg_stream(stream), rank, idx, pos_rank, freq_rank, Q, K, T);
  QPG_LAUNCH_CHECK("fuse_best_ranked_one_kernel");
  return QPG_OK;
}

// From how many chains per launch the gate table is deduplicated by the previous winner (gate_table_dedup_kernel; 0 =
// never): the context's QPG_OPT_GATE_DEDUP_FROM_CHAINS (default 1; the tests set 0 to walk on round 4's plain table).

static int match_steps_impl(qpg_ctx* ctx, void* stream, const int16_t* aud_rank, const int32_t* aud_idx,
                            const int16_t* txt_rank, const int32_t* txt_idx, const int16_t* pos_rank,
                            const int16_t* freq_rank, const int32_t* code, int code_ld, const int32_t* aud_cidx,
                            const int32_t* aud_pslot, int Ga, const int32_t* txt_cidx, const int32_t* txt_pslot,
                            int Gt, const float* phase, int Tp, int mode, int M, int steps, int K, int seed_code,
                            const float* seed_phase, int32_t* gate_tables, int32_t* out_codes, float* out_phase,
                            int32_t* out_vote, int32_t* out_status, const int32_t* guard_flags, int n_chains,
                            const int32_t* seed_codes, int64_t status_stride) {
  QPG_REQUIRE(ctx && pos_rank && freq_rank && code && phase && seed_phase && gate_tables && out_codes && out_phase &&
                  out_vote && out_status,
              "qpg_match_steps: null pointer");
  QPG_REQUIRE(n_chains >= 1 && (n_chains == 1 || (seed_codes && status_stride >= 2)),
              "qpg_match_steps_batch: n_chains >= 1, device seed codes and a status stride >= 2");
  const bool serial_walk = (mode & QPG_MODE_SERIAL_WALK) != 0;
  const bool prefused = (mode & QPG_MODE_PREFUSED) != 0;     // T0 | T1 of gate_tables were filled by qpg_fuse_best_ranked
  mode &= ~(QPG_MODE_SERIAL_WALK | QPG_MODE_PREFUSED);
  QPG_REQUIRE(!prefused || mode == 0, "qpg_match_steps: QPG_MODE_PREFUSED goes with the two-modality mode");
  QPG_REQUIRE(mode >= 0 && mode <= 2, "qpg_match_steps: bad mode %d", mode);
  QPG_REQUIRE(mode == QPG_MODE_TXT || (aud_rank && aud_idx && aud_cidx && aud_pslot && Ga > 0),
              "qpg_match_steps: audio tables missing");
  QPG_REQUIRE(mode == QPG_MODE_AUD || (txt_rank && txt_idx && txt_cidx && txt_pslot && Gt > 0),
              "qpg_match_steps: text tables missing");
  QPG_REQUIRE(M >= 0 && steps > 0 && steps * 4 <= 64 && K > 0 && K <= 64 * QPG_KMAX_PER_LANE && (K % 4) == 0 &&
                  seed_code >= 0 && seed_code < K && Ga <= 64 && Gt <= 64,
              "qpg_match_steps: bad size");
  const size_t lds = (size_t)2 * steps * K * sizeof(int32_t);
  QPG_REQUIRE(lds <= 96 * 1024, "qpg_match_steps: steps*K too large for the LDS gate tables");
  if (M == 0) {              // an empty clip still gets a defined status word (the host reads it with the results)
    hipLaunchKernelGGL(status_only_kernel, dim3(1), dim3(1), 0, qpg_stream(stream), out_status, guard_flags);
    QPG_LAUNCH_CHECK("status_only_kernel");
    return QPG_OK;
  }
  const int Qc = M * steps;                  // steps of one chain
  const int Q = Qc * n_chains;
  int32_t* T0 = gate_tables;
  int32_t* T1 = gate_tables + (int64_t)Q * K;
  if (prefused) {
    // (nothing: both tables are there)
  } else if (mode == 0 && (K % 16) == 0 && K <= 4096) {
    hipLaunchKernelGGL(fuse_best_ranked_kernel, dim3((unsigned)(((int64_t)Q * K) / 16)), dim3(256), 4 * (size_t)K,
                       qpg_stream(stream), aud_rank, aud_idx, txt_rank, txt_idx, pos_rank, freq_rank, Q, K, T0, T1);
  } else {
    dim3 grid((unsigned)(((int64_t)Q * K + 3) / 4), 1);
    hipLaunchKernelGGL(fuse_best_kernel, grid, dim3(256), 0, qpg_stream(stream), aud_rank, aud_idx, txt_rank, txt_idx,
                       pos_rank, freq_rank, Q, K, mode, T0, T1);
  }
  QPG_LAUNCH_CHECK("fuse_best_kernel");
  TailArgs A;
  A.T0 = T0; A.T1 = T1; A.code = code; A.code_ld = code_ld;
  const bool txt0 = (mode == QPG_MODE_TXT), txt1 = (mode != QPG_MODE_AUD);
  A.cidx0 = txt0 ? txt_cidx : aud_cidx; A.pslot0 = txt0 ? txt_pslot : aud_pslot; A.G0 = txt0 ? Gt : Ga;
  A.cidx1 = txt1 ? txt_cidx : aud_cidx; A.pslot1 = txt1 ? txt_pslot : aud_pslot; A.G1 = txt1 ? Gt : Ga;
  A.phase = phase; A.Tp = Tp; A.M = M; A.steps = steps; A.step_codes = 4;
  A.codes_per_window = (steps * 4 < 30) ? steps * 4 : 30;
  A.K = K; A.seed_code = seed_code; A.seed_phase = seed_phase; A.seed_codes = seed_codes; A.n_chains = n_chains;
  A.status_stride = status_stride;
  A.out_codes = out_codes; A.out_phase = out_phase; A.out_vote = out_vote; A.out_status = out_status;
  A.guard_flags = guard_flags;
  // tabulated walk when the code that seeds the next window comes from the window's LAST step (always true for the
  // reference's grids: 8 steps x 4 codes, 30 kept) and the state fits 16 bits; the one-wave sequential walk otherwise
  const int last_idx = A.codes_per_window - 1;
  GateGeom geo{last_idx / A.step_codes, last_idx % A.step_codes};
  const size_t lds_g = (size_t)2 * steps * 2 * K * sizeof(uint16_t);     // two window tables (double buffer)
  const bool tabulated = !serial_walk && geo.s_last == steps - 1 && 2 * K <= 65536 && Qc <= QPG_CHASE_QMAX &&
                         lds_g <= 64 * 1024 && ((steps * 2 * K) % 8) == 0;
  if (!tabulated) {
    QPG_REQUIRE(n_chains == 1, "qpg_match_steps_batch: the sequential walk takes one chain per call");
    hipLaunchKernelGGL(match_walk_kernel, dim3(1), dim3(64), lds, qpg_stream(stream), A);
    QPG_LAUNCH_CHECK("match_walk_kernel");
    return QPG_OK;
  }
  uint16_t* gtab = reinterpret_cast<uint16_t*>(gate_tables + (int64_t)2 * Q * K);     // third [Q][K] i32 region
  const int64_t lanes = (int64_t)Q * 2 * K * 8;
  const int dedup_from = ctx->opt[QPG_OPT_GATE_DEDUP_FROM_CHAINS];      // (qpg_ctx_set_option; 0: never)
  if (dedup_from > 0 && n_chains >= dedup_from && K <= 512) {
    hipLaunchKernelGGL(gate_table_dedup_kernel, dim3((unsigned)Q), dim3(GD_THREADS), 0, qpg_stream(stream), A, geo, gtab);
    QPG_LAUNCH_CHECK("gate_table_dedup_kernel");
  } else {
    hipLaunchKernelGGL(gate_table_kernel, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, qpg_stream(stream), A,
                       geo, gtab);
    QPG_LAUNCH_CHECK("gate_table_kernel");
  }
  hipLaunchKernelGGL(gate_chase_kernel, dim3(n_chains), dim3(1024), lds_g, qpg_stream(stream), A, (const uint16_t*)gtab);
  QPG_LAUNCH_CHECK("gate_chase_kernel");
  return QPG_OK;
}

extern "C" int qpg_match_steps(qpg_ctx* ctx, void* stream, const int16_t* aud_rank, const int32_t* aud_idx,
                               const int16_t* txt_rank, const int32_t* txt_idx, const int16_t* pos_rank,
                               const int16_t* freq_rank, const int32_t* code, int code_ld, const int32_t* aud_cidx,
                               const int32_t* aud_pslot, int Ga, const int32_t* txt_cidx, const int32_t* txt_pslot,
                               int Gt, const float* phase, int Tp, int mode, int M, int steps, int K, int seed_code,
                               const float* seed_phase, int32_t* gate_tables, int32_t* out_codes, float* out_phase,
                               int32_t* out_vote, int32_t* out_status, const int32_t* guard_flags) {
  return match_steps_impl(ctx, stream, aud_rank, aud_idx, txt_rank, txt_idx, pos_rank, freq_rank, code, code_ld, aud_cidx,
                          aud_pslot, Ga, txt_cidx, txt_pslot, Gt, phase, Tp, mode, M, steps, K, seed_code, seed_phase,
                          gate_tables, out_codes, out_phase, out_vote, out_status, guard_flags, 1, nullptr, 2);
}

// Several INDEPENDENT clips (chains) of M windows each in one set of launches: the tables hold the chains' steps back to
// back ([n_chains M steps][K]); seed_codes [dev] i32 [n_chains], seed_phase [dev] f32 [n_chains][8][16]; outputs
// [n_chains][...] in the single-clip shapes; out_status [dev] i32: chain c's pair at c x status_stride.  gate_tables:
// 3 x n_chains x M x steps x K i32.  (bench.py --clips 16 walked its clips one after the other: 16 x 3 launches.)
extern "C" int qpg_match_steps_batch(qpg_ctx* ctx, void* stream, const int16_t* aud_rank, const int32_t* aud_idx,
                                     const int16_t* txt_rank, const int32_t* txt_idx, const int16_t* pos_rank,
                                     const int16_t* freq_rank, const int32_t* code, int code_ld, const int32_t* aud_cidx,
                                     const int32_t* aud_pslot, int Ga, const int32_t* txt_cidx, const int32_t* txt_pslot,
                                     int Gt, const float* phase, int Tp, int mode, int M, int steps, int K,
                                     int n_chains, const int32_t* seed_codes, const float* seed_phase,
                                     int32_t* gate_tables, int32_t* out_codes, float* out_phase, int32_t* out_vote,
                                     int32_t* out_status, int64_t status_stride, const int32_t* guard_flags) {
  QPG_REQUIRE(M > 0 && seed_codes, "qpg_match_steps_batch: M > 0 and device seed codes");
  return match_steps_impl(ctx, stream, aud_rank, aud_idx, txt_rank, txt_idx, pos_rank, freq_rank, code, code_ld, aud_cidx,
                          aud_pslot, Ga, txt_cidx, txt_pslot, Gt, phase, Tp, mode, M, steps, K, 0, seed_phase, gate_tables,
                          out_codes, out_phase, out_vote, out_status, guard_flags, n_chains, seed_codes, status_stride);
}
