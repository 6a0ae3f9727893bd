// Per-code minimum of the exact-f32 cosine family from a BOUNDED prefilter (round 3; BASELINE.json configs[2]).
//
// The reference's text distance (GestureKNN.py:716 -> sklearn paired_distances(metric='cosine') on float32) is defined
// by its arithmetic: normalise, subtract, square and sum in NumPy's einsum order, three separately rounded f32
// operations per element pair - which is why the sweeps of qpg_text.hip are VALU-bound (2.7 ms for 1 000 queries x
// 100 000 rows).  But only the per-code MINIMUM and its first-wins candidate are wanted, and both are decided by
// comparisons: a prefilter whose error against the sklearn value is bounded a priori leaves, per (query, code), a BAND of
// candidates that can be the minimum; only those are evaluated in the exact order.
//   prefilter  qpg_hl_gemm_distance (qpg_audio_hl.hip): d~ = 1 - <x^, q^> on the f16 matrix cores (split operands,
//              f64 block sums), rows SORTED BY CODE (stable: original order inside a code) and padded to 16 per code;
//   bound      |d~ - d_sklearn| <= E = E_pre + E_sk:  E_pre = 1.3e-6 (the GEMM, unit-norm operands) + 2 eps1 (x^, q^ are
//              the f32-normalised rows, off the true unit vectors by eps1 = ((D/4 + 2)/2 + 2) u each);
//              E_sk = 0.5 [ 8 eps1 + 4 (D/4 + 3) u ]  (sklearn's own f32 rounding against the real-number value:
//              normalisation errors through the difference, Cauchy-Schwarz with |delta| <= 2, then the 4-lane chains of
//              D/4 squares) - 4.3e-5 at D = 512; `band` = 2.1 E is passed by the caller;
//   select     one block per query: (1) per-code minimum of d~ over the sorted row (segmented minimum over each wave:
//              runs of a code are contiguous), (2) rows within `band` of their code's minimum are listed, (3) the listed
//              (query, row) pairs are evaluated in sklearn's exact order from the f32 rows, (4) per code the minimum exact
//              distance and, among equals, the lowest ORIGINAL index (first-wins); tables and nearest neighbours.
// The tables are bit-identical to qpg_text_percode_f32's.  A list that overflows raises stats[1] |= 1 (the host then
// runs the exact VALU sweep): real text embeddings repeat (silence), and thousands of exact ties in one code are then
// all inside the band.
#include "qpg_common.h"

__device__ __forceinline__ unsigned int okey32(float d) {
  const unsigned int b = __float_as_uint(d);
  return (b >> 31) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float okey32_value(unsigned int k) {
  return __uint_as_float((k >> 31) ? (k & 0x7fffffffu) : ~k);
}

#define SORT_LIST 8192

__global__ __launch_bounds__(1024, 8) void percode_select_sorted_kernel(
    const float* __restrict__ Dm, int64_t ldD, int64_t R, const int16_t* __restrict__ row_code,
    const int32_t* __restrict__ row_index, int K, float band, const float* __restrict__ qn, const float* __restrict__ xs,
    int Dd, float absent, float* __restrict__ out_dist, int32_t* __restrict__ out_idx, int32_t* __restrict__ out_nn,
    int32_t* __restrict__ stats) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned long long* best = reinterpret_cast<unsigned long long*>(smem);            // [K] approx key << 32 | row
  unsigned long long* ebest = best + K;                                               // [K] exact key << 32 | original index
  int* list = reinterpret_cast<int*>(ebest + K);                                      // [SORT_LIST] rows in a band
  float* qrow = reinterpret_cast<float*>(list + SORT_LIST);                           // [Dd]
  __shared__ int n_list;
  __shared__ unsigned long long nn_key;
  const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const float* row = Dm + (int64_t)q * ldD;
  for (int k = tid; k < K; k += blockDim.x) {
    best[k] = ~0ull;
    ebest[k] = ~0ull;
  }
  for (int i = tid; i < Dd / 4; i += blockDim.x)
    reinterpret_cast<f32x4*>(qrow)[i] = reinterpret_cast<const f32x4*>(qn + (int64_t)q * Dd)[i];
  if (tid == 0) {
    n_list = 0;
    nn_key = ~0ull;
  }
  __syncthreads();
  // (1) per-code minimum of the prefilter values.  Rows are sorted by code in segments padded to 16 rows (a padding row
  // carries its segment's code with bit 14 set): a lane's 4 consecutive rows are ONE code, runs of a code are contiguous
  // across lanes.  Segmented minimum over the wave (6 shuffle steps), then one LDS atomic per run: a sorted row would
  // otherwise send all 64 lanes of an instruction to the same LDS word.
  typedef int16_t c16x4 __attribute__((ext_vector_type(4)));
  const int wv = tid >> 6, nwv = blockDim.x >> 6;
  constexpr int UN = 4;                                           // 256-row pieces in flight per wave (latency: the
  for (int64_t base0 = (int64_t)wv * 256 * UN; base0 < R; base0 += (int64_t)nwv * 256 * UN) {     // row comes from HBM)
    f32x4 dv[UN];
    c16x4 cv[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int64_t r0 = base0 + u * 256 + lane * 4;
      dv[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
      cv[u] = (c16x4){-1, -1, -1, -1};
      if (r0 < R) {
        dv[u] = *reinterpret_cast<const f32x4*>(row + r0);
        cv[u] = *reinterpret_cast<const c16x4*>(row_code + r0);
      }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int64_t r0 = base0 + u * 256 + lane * 4;
      int code = -1 - lane;                                       // (out of range: a run of its own)
      unsigned long long m = ~0ull;
      if (r0 < R) {
        code = cv[u][0] & 0x3fff;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const unsigned long long k = ((unsigned long long)okey32(dv[u][e]) << 32) | (unsigned int)(r0 + e);
          if (!(cv[u][e] & 0x4000) && k < m) m = k;
        }
      }
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long om = __shfl_down(m, off, 64);
        const int oc = __shfl_down(code, off, 64);
        if (lane + off < 64 && oc == code && om < m) m = om;
      }
      const int pc = __shfl_up(code, 1, 64);
      if ((lane == 0 || pc != code) && m != ~0ull && (unsigned)code < (unsigned)K) atomicMin(&best[code], m);
    }
  }
  __syncthreads();
  // (2) the band of every code   (band < 0: timing diagnostics - nothing is listed, the tables come out empty)
  if (band >= 0.f)
#pragma unroll 4
  for (int64_t r0 = (int64_t)tid * 4; r0 < R; r0 += (int64_t)blockDim.x * 4) {
    const f32x4 d = *reinterpret_cast<const f32x4*>(row + r0);
    const c16x4 cd = *reinterpret_cast<const c16x4*>(row_code + r0);
    const int code = cd[0] & 0x3fff;
    if ((unsigned)code >= (unsigned)K) continue;
    const float lim = okey32_value((unsigned int)(best[code] >> 32)) + band;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if ((cd[e] & 0x4000) || !(d[e] <= lim)) continue;
      const int pos = atomicAdd(&n_list, 1);
      if (pos < SORT_LIST) list[pos] = (int)(r0 + e);
    }
  }
  __syncthreads();
  int n = n_list;
  if (n > SORT_LIST) {
    n = SORT_LIST;
    if (tid == 0 && stats) atomicOr(&stats[1], 1);
  }
  // (3) exact sklearn-order distance of every listed (query, row) pair: 0.5 * einsum_sq(qn - xn), four lane chains,
  // 16-element groups visited u = 3,2,1,0, separate multiply and add, (l0 + l1) + (l2 + l3).  One thread per pair, the
  // query row in LDS, the candidate row gathered (8 loads in flight per thread).  Measured alternatives: evaluating per
  // CODE instead (buckets of (query, row) pairs, one block per code, the rows of a code read once for all queries) is
  // SLOWER - 465 us against ~200: every lane then gathers BOTH operands in 16-byte pieces of 128-byte lines.
  for (int e = tid; e < n; e += blockDim.x) {
    const int r = list[e];
    const f32x4* xp = reinterpret_cast<const f32x4*>(xs + (int64_t)r * Dd);
    const f32x4* qp = reinterpret_cast<const f32x4*>(qrow);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int g0 = 0; g0 < Dd / 16; g0 += 2) {
      f32x4 xv[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) xv[i] = (g0 * 4 + i) < Dd / 4 ? xp[g0 * 4 + i] : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int gg = 0; gg < 2; ++gg) {
        if (g0 + gg >= Dd / 16) break;
#pragma unroll
        for (int u = 3; u >= 0; --u) {
          const f32x4 x = xv[gg * 4 + u], qq = qp[(g0 + gg) * 4 + u];
          const float d0 = f_sub(qq.x, x.x), d1 = f_sub(qq.y, x.y), d2 = f_sub(qq.z, x.z), d3 = f_sub(qq.w, x.w);
          a0 = f_add(f_mul(d0, d0), a0);
          a1 = f_add(f_mul(d1, d1), a1);
          a2 = f_add(f_mul(d2, d2), a2);
          a3 = f_add(f_mul(d3, d3), a3);
        }
      }
    }
    const float dist = f_mul(0.5f, f_add(f_add(a0, a1), f_add(a2, a3)));
    const int cd = row_code[r] & 0x3fff;
    atomicMin(&ebest[cd], ((unsigned long long)okey32(dist) << 32) | (unsigned int)row_index[r]);
  }
  __syncthreads();
  // (4) tables + the query's global nearest neighbour
  unsigned long long mine = ~0ull;
  for (int k = tid; k < K; k += blockDim.x) {
    const unsigned long long kv = ebest[k];
    const bool have = kv != ~0ull;
    out_dist[(int64_t)q * K + k] = have ? okey32_value((unsigned int)(kv >> 32)) : absent;
    out_idx[(int64_t)q * K + k] = have ? (int32_t)(kv & 0xffffffffu) : -1;
    if (have && kv < mine) mine = kv;
  }
  if (out_nn) {
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long other = __shfl_xor(mine, o, 64);
      mine = other < mine ? other : mine;
    }
    if (lane == 0 && mine != ~0ull) atomicMin(&nn_key, mine);
    __syncthreads();
    if (tid == 0) out_nn[q] = nn_key != ~0ull ? (int32_t)(nn_key & 0xffffffffu) : -1;
  }
}

extern "C" int qpg_percode_select_sorted_f32(qpg_ctx* ctx, void* stream, const float* Dm, int64_t ldD, int Q, int64_t R,
                                             const int16_t* row_code, const int32_t* row_index, int K, float band,
                                             const float* qn, const float* xs, int Dd, float absent, float* out_dist,
                                             int32_t* out_idx, int32_t* out_nn, int32_t* stats) {
  const char* name = "qpg_percode_select_sorted_f32";
  QPG_REQUIRE(ctx && Dm && row_code && row_index && qn && xs && out_dist && out_idx, "%s: null pointer", name);
  QPG_REQUIRE(Q >= 0 && R > 0 && (R % 4) == 0 && R < 0x7fffffffll && ldD >= R && (ldD % 4) == 0 && K > 0 && K <= 2048 &&
                  K <= 0x3fff && Dd > 0 && (Dd % 16) == 0 && (reinterpret_cast<uintptr_t>(Dm) % 16) == 0 &&
                  (reinterpret_cast<uintptr_t>(xs) % 16) == 0 && (reinterpret_cast<uintptr_t>(qn) % 16) == 0 &&
                  (reinterpret_cast<uintptr_t>(row_code) % 8) == 0,
              "%s: bad size / alignment (R %% 4 == 0, D %% 16 == 0, K <= 2048)", name);
  if (Q == 0) return QPG_OK;
  const size_t sh = 16 * (size_t)K + 4 * (size_t)SORT_LIST + 4 * (size_t)Dd;
  QPG_REQUIRE(sh <= 64 * 1024, "%s: K / D too large for the LDS tables", name);
  hipLaunchKernelGGL(percode_select_sorted_kernel, dim3(Q), dim3(1024), sh, qpg_stream(stream), Dm, ldD, R, row_code,
                     row_index, K, band, qn, xs, Dd, absent, out_dist, out_idx, out_nn, stats);
  QPG_LAUNCH_CHECK("percode_select_sorted_kernel");
  return QPG_OK;
}
