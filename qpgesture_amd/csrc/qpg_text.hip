// Text (sentence-embedding) candidate sweep with scikit-learn's float32 arithmetic, bit-exact.
//
// Replaces CodeKNN.search_text_cands (GestureKNN.py:708-721): cosine distance of the query
// context vector against context_train[j, k//8], k = 0,8,..,200.  The reference keeps float32
// end to end (sklearn does not promote f32), so two candidates whose true distances differ by
// less than f32 rounding are ordered by the *arithmetic*, not the mathematics.  To return the
// reference's indices — not merely close distances — this kernel reproduces that arithmetic:
//     d = 0.5 * sum_e (qn[e] - xn[e])^2        with NumPy-einsum summation order
// (4 lane accumulators, separate multiply and add, 16-element groups visited u = 3,2,1,0,
// horizontal (l0+l1)+(l2+l3)); xn / qn are sklearn-normalised rows.
//
// The summation order is fixed per (query, candidate) pair, so the parallelism is across pairs:
// lane = candidate, QB queries per lane, 4 accumulators each.  Data layout is chosen for that:
//   * candidates are stored pre-normalised and TILED by qpg_text_pack_candidates_f32 as
//     xt[tile][Dm/4][64][4]: the 64 lanes of a wave read 64 consecutive 16-B pieces (1 KiB, fully
//     coalesced), straight into registers — no LDS, no barriers, loads pipelined by the unrolled loop;
//   * the QB query rows of a wave are the same for all lanes and arrive through scalar loads (SGPRs).
// VALU-bound by construction: 3 separately rounded ops per element pair, FMA contraction forbidden.
// (r01 history: LDS-staged tiles with per-chunk barriers ran 196-224 us at N_db=2048, latency-bound.)
#include "qpg_common.h"

// ---------------------------------------------------------------------------------------------
// DB preparation: normalise the grid rows of the context array (sklearn-exact, 4 threads per row as
// in l2_normalize_rows_kernel) and write them in the tiled layout.  Candidate c = j*G + g.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void text_pack_candidates_kernel(const float* __restrict__ x, int N, int R, int Dm,
                                                                   const int32_t* __restrict__ cand_r, int G,
                                                                   float* __restrict__ xt) {
  const int64_t C = (int64_t)N * G;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t c = t >> 2;
  const int l = (int)(t & 3);
  const bool live = c < C;
  if (!live) c = C - 1;
  const int j = (int)(c / G), g = (int)(c - (int64_t)j * G);
  const float* p = x + ((int64_t)j * R + cand_r[g]) * Dm;
  float a = 0.f;
  for (int k = 0; k < (Dm >> 4); ++k) {
#pragma unroll
    for (int u = 3; u >= 0; --u) {
      const float v = p[k * 16 + u * 4 + l];
      a = f_add(f_mul(v, v), a);
    }
  }
  const float o1 = __shfl_xor(a, 1, 64);
  const float pair = f_add(a, o1);
  const float o2 = __shfl_xor(pair, 2, 64);
  float n = f_sqrt(f_add(pair, o2));
  if (n < 10.f * 1.1920928955078125e-07f) n = 1.f;
  if (live) {
    const int64_t tile = c >> 6;
    const int lane = (int)(c & 63);
    float* o = xt + tile * (int64_t)Dm * 64;
    for (int e = l; e < Dm; e += 4) o[((int64_t)(e >> 2) * 64 + lane) * 4 + (e & 3)] = f_div(p[e], n);
  }
}

extern "C" int qpg_text_pack_candidates_f32(qpg_ctx* ctx, void* stream, const float* x, int N, int R, int Dm,
                                            const int32_t* cand_r, int G, float* xt) {
  QPG_REQUIRE(ctx && x && cand_r && xt && N >= 0 && R > 0 && G > 0, "qpg_text_pack_candidates_f32: bad argument");
  if (Dm <= 0 || (Dm % 16) != 0) {
    qpg_set_error("qpg_text_pack_candidates_f32: compiled for Dm %% 16 == 0 (got %d)", Dm);
    return QPG_EUNSUP;
  }
  const int64_t C = (int64_t)N * G;
  if (C == 0) return QPG_OK;
  hipLaunchKernelGGL(text_pack_candidates_kernel, dim3((unsigned)((C * 4 + 255) / 256)), dim3(256), 0,
                     qpg_stream(stream), x, N, R, Dm, cand_r, G, xt);
  QPG_LAUNCH_CHECK("text_pack_candidates_kernel");
  return QPG_OK;
}

// fp16-STORAGE variant (BASELINE.json configs[2]/[4] "fp16 features"): the raw rows are stored in IEEE f16, tiled
// xh[tile][Dm/8][64][8] (a lane's 16-byte piece = 8 consecutive features of its candidate), plus one f32 norm per
// candidate computed — in the same sklearn/einsum order — from the ROUNDED values.  The sweep widens f16 -> f32,
// divides by the norm (sklearn's `X /= norms`, a correctly rounded f32 division) and continues with the same
// arithmetic, so its results are the reference's on the f16-rounded database, bit for bit.
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void text_pack_candidates_h_kernel(const float* __restrict__ x, int N, int R, int Dm,
                                                                     const int32_t* __restrict__ cand_r, int G,
                                                                     _Float16* __restrict__ xh, float* __restrict__ nrm) {
  const int64_t C = (int64_t)N * G;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t c = t >> 2;
  const int l = (int)(t & 3);
  const bool live = c < C;
  if (!live) c = C - 1;
  const int j = (int)(c / G), g = (int)(c - (int64_t)j * G);
  const float* p = x + ((int64_t)j * R + cand_r[g]) * Dm;
  float a = 0.f;
  for (int k = 0; k < (Dm >> 4); ++k) {
#pragma unroll
    for (int u = 3; u >= 0; --u) {
      const float v = (float)(_Float16)p[k * 16 + u * 4 + l];
      a = f_add(f_mul(v, v), a);
    }
  }
  const float o1 = __shfl_xor(a, 1, 64);
  const float pair = f_add(a, o1);
  const float o2 = __shfl_xor(pair, 2, 64);
  float n = f_sqrt(f_add(pair, o2));
  if (n < 10.f * 1.1920928955078125e-07f) n = 1.f;
  if (live) {
    const int64_t tile = c >> 6;
    const int lane = (int)(c & 63);
    _Float16* o = xh + tile * (int64_t)Dm * 64;
    for (int e = l; e < Dm; e += 4) o[((int64_t)(e >> 3) * 64 + lane) * 8 + (e & 7)] = (_Float16)p[e];
    if (l == 0) nrm[c] = n;
  }
}

extern "C" int qpg_text_pack_candidates_f16(qpg_ctx* ctx, void* stream, const float* x, int N, int R, int Dm,
                                            const int32_t* cand_r, int G, void* xh, float* nrm) {
  QPG_REQUIRE(ctx && x && cand_r && xh && nrm && N >= 0 && R > 0 && G > 0, "qpg_text_pack_candidates_f16: bad argument");
  if (Dm <= 0 || (Dm % 16) != 0) {
    qpg_set_error("qpg_text_pack_candidates_f16: compiled for Dm %% 16 == 0 (got %d)", Dm);
    return QPG_EUNSUP;
  }
  const int64_t C = (int64_t)N * G;
  if (C == 0) return QPG_OK;
  hipLaunchKernelGGL(text_pack_candidates_h_kernel, dim3((unsigned)((C * 4 + 255) / 256)), dim3(256), 0,
                     qpg_stream(stream), x, N, R, Dm, cand_r, G, static_cast<_Float16*>(xh), nrm);
  QPG_LAUNCH_CHECK("text_pack_candidates_h_kernel");
  return QPG_OK;
}

// QB queries per lane, NG waves per block (all on the same 64-candidate tile, different query groups).
typedef float f32x2 __attribute__((ext_vector_type(2)));
// q - x on a pair: one v_pk_add_f32 with the (wave-uniform) query pair read directly from an SGPR pair and the
// candidate pair negated by the instruction's source modifiers — IEEE add of q and -x, i.e. exactly q - x.
__device__ __forceinline__ f32x2 pk_sub_sv(f32x2 q, f32x2 x) {
  f32x2 d;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "s"(q), "v"(x));
  return d;
}

// Distances of ONE 64-candidate tile (lane = candidate) against QB wave-uniform query rows, sklearn/einsum f32 order.
// HALF: xp points at the lane's 16-byte f16 pieces (two per 16-element group, 64 pieces apart) and `nrm` is the
// candidate's norm: the group is widened and divided element by element before use (sklearn's normalisation).
template <int QB, bool HALF = false>
__device__ __forceinline__ void text_tile_dists(const f32x4* __restrict__ xp, const float* const (&qrow)[QB], int nk,
                                                float (&dist)[QB], float nrm = 1.f) {
  // accumulators as two packed pairs (einsum lanes 0,1 and 2,3): every step is 3 packed VALU ops per 2 elements
  // (v_pk_add_f32 with the query pair straight from SGPRs and a negated candidate pair, v_pk_mul_f32, v_pk_add_f32)
  f32x2 acc[QB][2];
#pragma unroll
  for (int i = 0; i < QB; ++i) acc[i][0] = acc[i][1] = f32x2{0.f, 0.f};

  // Software pipeline: the 64-B scalar load of the NEXT query row segment is issued before the 48 VALU ops
  // of the current one (two 16-SGPR buffers); element order per accumulator stays u = 3,2,1,0.
  f32x16 qv = *reinterpret_cast<const f32x16*>(qrow[0]);
  f32x4 xnext[4];
  constexpr int NL = HALF ? 2 : 4;                // 16-byte loads per 16-element group
#pragma unroll
  for (int u = 0; u < NL; ++u) xnext[u] = xp[u * 64];
  for (int k = 0; k < nk; ++k) {
    f32x4 x[4];
    const int kx = (k + 1 < nk) ? k + 1 : 0;      // candidate tile: prefetch the next 16-element group too
    if (HALF) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const h16x8 h = __builtin_bit_cast(h16x8, xnext[u]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          x[2 * u][e] = f_div((float)h[e], nrm);
          x[2 * u + 1][e] = f_div((float)h[4 + e], nrm);
        }
        xnext[u] = xp[(kx * 2 + u) * 64];
      }
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        x[u] = xnext[u];
        xnext[u] = xp[(kx * 4 + u) * 64];
      }
    }
#pragma unroll
    for (int i = 0; i < QB; ++i) {
      const int kn = (i + 1 < QB) ? k : (k + 1 < nk ? k + 1 : 0);
      const f32x16 qnext = *reinterpret_cast<const f32x16*>(qrow[(i + 1) % QB] + kn * 16);
#define QPG_TEXT_STEP(U)                                                                              \
  {                                                                                                   \
    const f32x2 d01 = pk_sub_sv(__builtin_shufflevector(qv, qv, (U) * 4 + 0, (U) * 4 + 1),            \
                                __builtin_shufflevector(x[U], x[U], 0, 1));                           \
    const f32x2 d23 = pk_sub_sv(__builtin_shufflevector(qv, qv, (U) * 4 + 2, (U) * 4 + 3),            \
                                __builtin_shufflevector(x[U], x[U], 2, 3));                           \
    acc[i][0] = d01 * d01 + acc[i][0];                                                                \
    acc[i][1] = d23 * d23 + acc[i][1];                                                                \
  }
      QPG_TEXT_STEP(3) QPG_TEXT_STEP(2) QPG_TEXT_STEP(1) QPG_TEXT_STEP(0)
#undef QPG_TEXT_STEP
      qv = qnext;
    }
  }
#pragma unroll
  for (int i = 0; i < QB; ++i)
    dist[i] = f_mul(0.5f, f_add(f_add(acc[i][0].x, acc[i][0].y), f_add(acc[i][1].x, acc[i][1].y)));
}

template <int QB, int NG>
__global__ __launch_bounds__(64 * NG) void text_cosine_f32_kernel(const float* __restrict__ xt, int64_t C, int Dm,
                                                                  const float* __restrict__ qn, int Q,
                                                                  float* __restrict__ D, int64_t ldD) {
  const int lane = threadIdx.x & 63;
  const int qg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform
  const int64_t tile = blockIdx.x;
  const int q0 = blockIdx.y * (NG * QB) + qg * QB;
  if (q0 >= Q) return;
  const float* qrow[QB];
#pragma unroll
  for (int i = 0; i < QB; ++i) {
    int q = q0 + i;
    if (q >= Q) q = Q - 1;
    qrow[i] = qn + (int64_t)q * Dm;
  }
  const f32x4* xp = reinterpret_cast<const f32x4*>(xt + tile * (int64_t)Dm * 64) + lane;
  float dist[QB];
  text_tile_dists<QB>(xp, qrow, Dm >> 4, dist);
  const int64_t c = tile * 64 + lane;
  if (c < C) {
#pragma unroll
    for (int i = 0; i < QB; ++i)
      if (q0 + i < Q) D[(int64_t)(q0 + i) * ldD + c] = dist[i];
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Sweep with the per-code minimum FUSED (large query counts: BASELINE.json configs[2], 1 000 queries x 100 000
// candidates).  The Q x C distance matrix (400 MB there) never reaches HBM: a block = NG waves on the SAME QB queries,
// each wave walking every NG-th 64-candidate tile of the block's chunk; every lane folds its QB distances into an
// LDS table [QB][K] of packed (ordered distance << 32 | candidate index) with ds_min_u64 (first-wins = lowest index
// among equal distances, as the reference's strict `<` scan), and the block folds its table into ONE global [Q][K]
// table (filtered atomicMin, see the flush); a last launch decodes it.  min is associative and commutative, so the
// result does not depend on the order the blocks arrive in.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned int text_order_key(float d) {
  const unsigned int b = __float_as_uint(d);
  return (b >> 31) ? ~b : (b | 0x80000000u);
}

template <int QB, int NG>
__global__ __launch_bounds__(64 * NG) void text_cosine_percode_f32_kernel(
    const float* __restrict__ xt, int64_t C, int Dm, const int16_t* __restrict__ cand_code, int K,
    const float* __restrict__ qn, int Q, int tiles_per_chunk, int32_t idx_base, unsigned long long* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned long long* tab = reinterpret_cast<unsigned long long*>(smem);          // [QB][K]
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int chunk = blockIdx.x;
  const int q0 = blockIdx.y * QB;
  const float* qrow[QB];
#pragma unroll
  for (int i = 0; i < QB; ++i) {
    int q = q0 + i;
    if (q >= Q) q = Q - 1;
    qrow[i] = qn + (int64_t)q * Dm;
  }
  for (int i = threadIdx.x; i < QB * K; i += 64 * NG) tab[i] = ~0ull;
  __syncthreads();
  const int64_t ntile = (C + 63) / 64;
  const int64_t t0 = (int64_t)chunk * tiles_per_chunk;
  const int64_t t1 = t0 + tiles_per_chunk < ntile ? t0 + tiles_per_chunk : ntile;
  for (int64_t tile = t0 + wv; tile < t1; tile += NG) {
    const f32x4* xp = reinterpret_cast<const f32x4*>(xt + tile * (int64_t)Dm * 64) + lane;
    float dist[QB];
    text_tile_dists<QB>(xp, qrow, Dm >> 4, dist);
    const int64_t c = tile * 64 + lane;
    const int cd = c < C ? cand_code[c] : -1;
    if ((unsigned)cd < (unsigned)K) {
      const unsigned int ci = (unsigned int)(c + idx_base);
#pragma unroll
      for (int i = 0; i < QB; ++i)
        atomicMin(&tab[i * K + cd], ((unsigned long long)text_order_key(dist[i]) << 32) | ci);
    }
  }
  __syncthreads();
  // Flush into the ONE global [Q][K] table (4 MB, L2-resident) with atomicMin - but look first: the running minimum of
  // a (query, code) pair is improved by the i-th chunk with probability ~1/i, so nearly all atomics are skipped after
  // the first few chunks (a stale, i.e. larger, value read through L1 only costs a redundant atomic: values only fall).
  for (int i = threadIdx.x; i < QB * K; i += 64 * NG) {
    const int q = q0 + i / K;
    const unsigned long long key = tab[i];
    if (q < Q && key != ~0ull) {
      unsigned long long* g = partial + (int64_t)q * K + (i % K);
      if (key < *g) atomicMin(g, key);
    }
  }
}

// Variant without any LDS: the NG waves of a block take DIFFERENT query groups on the SAME candidate tile (as the
// plain sweep does, so a tile is fetched once per NG*QB queries instead of once per QB) and every lane folds its QB
// distances straight into the global [Q][K] table with the same look-first atomicMin.  The look is an L1-bypassing
// load (a stale value could only be larger, i.e. cost a redundant atomic).  Per (query, code) the n-th candidate is a
// new minimum with probability 1/n, so the atomics are ~H(C/K) per pair (about 6 of 195 here), not one per candidate.
template <int QB, int NG, bool HALF>
__global__ __launch_bounds__(64 * NG) void text_cosine_gmin_f32_kernel(const float* __restrict__ xt, int64_t C, int Dm,
                                                                       const int16_t* __restrict__ cand_code, int K,
                                                                       const float* __restrict__ qn, int Q,
                                                                       int32_t idx_base,
                                                                       unsigned long long* __restrict__ table,
                                                                       const float* __restrict__ nrm) {
  const int lane = threadIdx.x & 63;
  const int qg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // XCD-aware work mapping (blocks are dealt round-robin to the 8 XCDs, each with its own L2): all the query-group
  // blocks of one candidate tile get ids that are equal mod 8, i.e. run on ONE XCD at about the same time, so the
  // tile (64 x Dm floats) crosses the fabric once and is served to the other query groups from that XCD's L2.
  // Placement is a speed matter only; any mapping gives the same table.
  const int nqb = (Q + NG * QB - 1) / (NG * QB);
  const int64_t slot = blockIdx.x >> 3;
  const int64_t tile = (slot / nqb) * 8 + (blockIdx.x & 7);
  const int q0 = (int)(slot % nqb) * (NG * QB) + qg * QB;
  if (tile * 64 >= C || q0 >= Q) return;
  const float* qrow[QB];
#pragma unroll
  for (int i = 0; i < QB; ++i) {
    int q = q0 + i;
    if (q >= Q) q = Q - 1;
    qrow[i] = qn + (int64_t)q * Dm;
  }
  // (HALF: xt is the f16 image, half as many bytes per tile)
  const f32x4* xp = reinterpret_cast<const f32x4*>(xt + tile * (int64_t)Dm * (HALF ? 32 : 64)) + lane;
  float dist[QB];
  const int64_t c = tile * 64 + lane;
  text_tile_dists<QB, HALF>(xp, qrow, Dm >> 4, dist, HALF ? nrm[c < C ? c : C - 1] : 1.f);
  const int cd = c < C ? cand_code[c] : -1;
  if ((unsigned)cd >= (unsigned)K) return;
  const unsigned int ci = (unsigned int)(c + idx_base);
#pragma unroll
  for (int i = 0; i < QB; ++i) {
    if (q0 + i >= Q) break;
    unsigned long long* g = table + (int64_t)(q0 + i) * K + cd;
    const unsigned long long key = ((unsigned long long)text_order_key(dist[i]) << 32) | ci;
    const unsigned long long cur = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (key < cur) atomicMin(g, key);
  }
}

__global__ __launch_bounds__(256) void text_fill_ff_kernel(unsigned long long* __restrict__ a, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) a[i] = ~0ull;
}

// min over the chunk partials + decode: one block per query row (optionally the stable ranks too)
__global__ __launch_bounds__(512) void text_percode_merge_kernel(const unsigned long long* __restrict__ partial,
                                                                 int nchunk, int Q, int K, float absent,
                                                                 float* __restrict__ out_dist,
                                                                 int32_t* __restrict__ out_idx,
                                                                 int16_t* __restrict__ out_rank,
                                                                 int32_t* __restrict__ out_nn) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* v = reinterpret_cast<float*>(smem);
  __shared__ unsigned long long gbest;
  const int q = blockIdx.x;
  if (threadIdx.x == 0) gbest = ~0ull;
  __syncthreads();
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    unsigned long long m = ~0ull;
    for (int ch = 0; ch < nchunk; ++ch) {
      const unsigned long long p = partial[((int64_t)ch * Q + q) * K + k];
      m = p < m ? p : m;
    }
    const bool have = m != ~0ull;
    const unsigned int kb = (unsigned int)(m >> 32);
    const unsigned int fb = (kb >> 31) ? (kb & 0x7fffffffu) : ~kb;
    const float d = have ? __uint_as_float(fb) : absent;
    v[k] = d;
    out_dist[(int64_t)q * K + k] = d;
    out_idx[(int64_t)q * K + k] = have ? (int32_t)(m & 0xffffffffu) : -1;
    if (out_nn && have) atomicMin(&gbest, m);
  }
  __syncthreads();
  // the query's global nearest neighbour over all codes (minimum distance, lowest candidate index among equals)
  if (out_nn && threadIdx.x == 0) out_nn[q] = gbest != ~0ull ? (int32_t)(gbest & 0xffffffffu) : -1;
  if (!out_rank) return;
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    const float x = v[k];
    int r = 0;
    for (int o = 0; o < K; ++o) {
      const float y = v[o];
      r += (y < x) || (y == x && o < k);
    }
    out_rank[(int64_t)q * K + k] = (int16_t)r;
  }
}

extern "C" int64_t qpg_text_percode_ws_bytes(int64_t C, int Q, int K, int tiles_per_chunk) {
  if (C < 0 || Q < 0 || K <= 0 || tiles_per_chunk <= 0) return -1;
  return (int64_t)Q * K * 8;
}

static int text_percode(qpg_ctx* ctx, void* stream, const float* xt, const float* nrm, int64_t C, int Dm,
                        const int16_t* cand_code, int K, const float* qn, int Q, int tiles_per_chunk,
                        int32_t idx_base, float absent, void* ws, int64_t ws_bytes, float* out_dist,
                        int32_t* out_idx, int16_t* out_rank, int32_t* out_nn) {
  QPG_REQUIRE(ctx && xt && (cand_code || C == 0) && qn && ws && out_dist && out_idx, "qpg_text_percode_f32: null pointer");
  QPG_REQUIRE(!nrm || tiles_per_chunk == 1, "qpg_text_percode_f16: the f16 image is swept by the LDS-free organisation only "
                                            "(tiles_per_chunk == 1)");
  QPG_REQUIRE(C >= 0 && Q >= 0 && K > 0 && K <= 1024 && tiles_per_chunk > 0 && C + (int64_t)idx_base < 0xffffffffll,
              "qpg_text_percode_f32: bad size (K <= 1024)");
  if (Dm <= 0 || (Dm % 16) != 0) {
    qpg_set_error("qpg_text_percode_f32: compiled for Dm %% 16 == 0 (got %d)", Dm);
    return QPG_EUNSUP;
  }
  QPG_REQUIRE(ws_bytes >= qpg_text_percode_ws_bytes(C, Q, K, tiles_per_chunk), "qpg_text_percode_f32: workspace too small");
  if (Q == 0) return QPG_OK;
  constexpr int QB = 12, NG = 8;      // 8 waves share one 48 KB table: 3 blocks = 24 waves per CU
  const int64_t ntile = (C + 63) / 64;
  const int nchunk = (int)((ntile + tiles_per_chunk - 1) / tiles_per_chunk);
  unsigned long long* partial = static_cast<unsigned long long*>(ws);
  const size_t sh = (size_t)QB * K * 8;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(text_cosine_percode_f32_kernel<QB, NG>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 12 * 1024 * 8) != hipSuccess) {
      qpg_set_error("qpg_text_percode_f32: cannot reserve LDS");
      return QPG_EHIP;
    }
    attr_set = true;
  }
  {
    const int64_t n = (int64_t)Q * K;
    hipLaunchKernelGGL(text_fill_ff_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, qpg_stream(stream), partial, n);
    QPG_LAUNCH_CHECK("text_fill_ff_kernel");
  }
  if (tiles_per_chunk == 1) {      // LDS-free organisation: one tile per block, NG query groups share it
    constexpr int QG = 12, NW = 8;
    const int64_t nqb = (Q + QG * NW - 1) / (QG * NW);
    const dim3 grid((unsigned)(((ntile + 7) / 8) * nqb * 8));
    if (nrm)
      hipLaunchKernelGGL((text_cosine_gmin_f32_kernel<QG, NW, true>), grid, dim3(64 * NW), 0, qpg_stream(stream), xt, C, Dm,
                         cand_code, K, qn, Q, idx_base, partial, nrm);
    else
      hipLaunchKernelGGL((text_cosine_gmin_f32_kernel<QG, NW, false>), grid, dim3(64 * NW), 0, qpg_stream(stream), xt, C, Dm,
                         cand_code, K, qn, Q, idx_base, partial, nrm);
    QPG_LAUNCH_CHECK("text_cosine_gmin_f32_kernel");
  } else if (nchunk > 0) {
    hipLaunchKernelGGL((text_cosine_percode_f32_kernel<QB, NG>), dim3((unsigned)nchunk, (unsigned)((Q + QB - 1) / QB)),
                       dim3(64 * NG), sh, qpg_stream(stream), xt, C, Dm, cand_code, K, qn, Q, tiles_per_chunk, idx_base,
                       partial);
    QPG_LAUNCH_CHECK("text_cosine_percode_f32_kernel");
  }
  hipLaunchKernelGGL(text_percode_merge_kernel, dim3(Q), dim3(512), sizeof(float) * (size_t)K, qpg_stream(stream),
                     partial, 1, Q, K, absent, out_dist, out_idx, out_rank, out_nn);
  QPG_LAUNCH_CHECK("text_percode_merge_kernel");
  return QPG_OK;
}

extern "C" int qpg_text_percode_f32(qpg_ctx* ctx, void* stream, const float* xt, int64_t C, int Dm,
                                    const int16_t* cand_code, int K, const float* qn, int Q, int tiles_per_chunk,
                                    int32_t idx_base, float absent, void* ws, int64_t ws_bytes, float* out_dist,
                                    int32_t* out_idx, int16_t* out_rank, int32_t* out_nn) {
  return text_percode(ctx, stream, xt, nullptr, C, Dm, cand_code, K, qn, Q, tiles_per_chunk, idx_base, absent, ws, ws_bytes,
                      out_dist, out_idx, out_rank, out_nn);
}

extern "C" int qpg_text_percode_f16(qpg_ctx* ctx, void* stream, const void* xh, const float* nrm, int64_t C, int Dm,
                                    const int16_t* cand_code, int K, const float* qn, int Q, int32_t idx_base,
                                    float absent, void* ws, int64_t ws_bytes, float* out_dist, int32_t* out_idx,
                                    int16_t* out_rank, int32_t* out_nn) {
  QPG_REQUIRE(nrm && (reinterpret_cast<uintptr_t>(xh) % 16) == 0, "qpg_text_percode_f16: norms missing or image misaligned");
  return text_percode(ctx, stream, static_cast<const float*>(xh), nrm, C, Dm, cand_code, K, qn, Q, 1, idx_base, absent, ws,
                      ws_bytes, out_dist, out_idx, out_rank, out_nn);
}

template <int QB, int NG>
static int launch_text(void* stream, const float* xt, int64_t C, int Dm, const float* qn, int Q, float* D,
                       int64_t ldD) {
  dim3 grid((unsigned)((C + 63) / 64), (unsigned)((Q + NG * QB - 1) / (NG * QB)));
  hipLaunchKernelGGL((text_cosine_f32_kernel<QB, NG>), grid, dim3(64 * NG), 0, qpg_stream(stream), xt, C, Dm, qn, Q,
                     D, ldD);
  QPG_LAUNCH_CHECK("text_cosine_f32_kernel");
  return QPG_OK;
}

extern "C" int qpg_text_cosine_f32(qpg_ctx* ctx, void* stream, const float* xt, int64_t C, int Dm, const float* qn,
                                   int Q, float* D, int64_t ldD) {
  QPG_REQUIRE(ctx && xt && qn && D, "qpg_text_cosine_f32: null pointer");
  QPG_REQUIRE(C >= 0 && Q >= 0 && ldD >= C, "qpg_text_cosine_f32: bad size");
  if (Dm <= 0 || (Dm % 16) != 0) {
    qpg_set_error("qpg_text_cosine_f32: compiled for Dm %% 16 == 0 (got %d)", Dm);
    return QPG_EUNSUP;
  }
  if (C == 0 || Q == 0) return QPG_OK;
  if (Q > 96) return launch_text<12, 4>(stream, xt, C, Dm, qn, Q, D, ldD);
  // one clip (48 steps): 4 queries per lane = three times the waves of <12, 4>; 69.7 -> 59.6 us at 53 248 candidates
  // (experiments/text_shape: <6,4> 61.5, <6,8> 63.7, <4,12> 64.5, <3,8> 68.4; the packed-VALU floor is 37.5 us)
  if (Q > 24) return launch_text<4, 4>(stream, xt, C, Dm, qn, Q, D, ldD);
  if (Q > 8) return launch_text<6, 4>(stream, xt, C, Dm, qn, Q, D, ldD);
  if (Q > 2) return launch_text<2, 4>(stream, xt, C, Dm, qn, Q, D, ldD);
  return launch_text<1, 2>(stream, xt, C, Dm, qn, Q, D, ldD);
}
