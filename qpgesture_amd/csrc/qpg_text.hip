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

// QB queries per lane, NG waves per block (all on the same 64-candidate tile, different query groups).
typedef float f32x2 __attribute__((ext_vector_type(2)));
// q - x on a pair: one v_pk_add_f32 with the (wave-uniform) query pair read directly from an SGPR pair and the
// candidate pair negated by the instruction's source modifiers — IEEE add of q and -x, i.e. exactly q - x.
__device__ __forceinline__ f32x2 pk_sub_sv(f32x2 q, f32x2 x) {
  f32x2 d;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "s"(q), "v"(x));
  return d;
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

  // accumulators as two packed pairs (einsum lanes 0,1 and 2,3): every step is 3 packed VALU ops per 2 elements
  // (v_pk_add_f32 with the query pair straight from SGPRs and a negated candidate pair, v_pk_mul_f32, v_pk_add_f32)
  f32x2 acc[QB][2];
#pragma unroll
  for (int i = 0; i < QB; ++i) acc[i][0] = acc[i][1] = f32x2{0.f, 0.f};

  // Software pipeline: the 64-B scalar load of the NEXT query row segment is issued before the 48 VALU ops
  // of the current one (two 16-SGPR buffers); element order per accumulator stays u = 3,2,1,0.
  const int nk = Dm >> 4;
  f32x16 qv = *reinterpret_cast<const f32x16*>(qrow[0]);
  f32x4 xnext[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) xnext[u] = xp[u * 64];
  for (int k = 0; k < nk; ++k) {
    f32x4 x[4];
    const int kx = (k + 1 < nk) ? k + 1 : 0;      // candidate tile: prefetch the next 16-element group too
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      x[u] = xnext[u];
      xnext[u] = xp[(kx * 4 + u) * 64];
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

  const int64_t c = tile * 64 + lane;
  if (c < C) {
#pragma unroll
    for (int i = 0; i < QB; ++i) {
      if (q0 + i < Q) {
        const float s = f_add(f_add(acc[i][0].x, acc[i][0].y), f_add(acc[i][1].x, acc[i][1].y));
        D[(int64_t)(q0 + i) * ldD + c] = f_mul(0.5f, s);
      }
    }
  }
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
  if (Q > 24) return launch_text<12, 4>(stream, xt, C, Dm, qn, Q, D, ldD);
  if (Q > 8) return launch_text<6, 4>(stream, xt, C, Dm, qn, Q, D, ldD);
  if (Q > 2) return launch_text<2, 4>(stream, xt, C, Dm, qn, Q, D, ldD);
  return launch_text<1, 2>(stream, xt, C, Dm, qn, Q, D, ldD);
}
