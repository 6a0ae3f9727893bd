// Text (sentence-embedding) candidate sweep with scikit-learn's float32 arithmetic, bit-exact.
//
// Replaces CodeKNN.search_text_cands (GestureKNN.py:708-721): cosine distance of the query
// context vector against context_train[j, k//8], k = 0,8,..,200.  The reference keeps float32
// end to end (sklearn does not promote f32), so two candidates whose true distances differ by
// less than f32 rounding are ordered by the *arithmetic*, not the mathematics.  To return the
// reference's indices — not merely close distances — this kernel reproduces that arithmetic:
//     d = 0.5 * sum_e (qn[e] - xn[e])^2        with NumPy-einsum summation order
// (4 lane accumulators, separate multiply and add, 16-element groups visited u = 3,2,1,0,
// horizontal (l0+l1)+(l2+l3)); xn / qn are rows normalised by qpg_l2_normalize_rows_f32.
// The order is fixed per (query, candidate) pair, so the parallelism is across pairs:
//   block = 256 threads = 64 candidates x 4 query groups; each thread owns one candidate row and
//   QB queries, 4 lane-accumulators each; the candidate tile and the query tile are staged in
//   LDS in 64-element chunks (b128 reads, row stride 68 floats = conflict-free for ds_read_b128).
// VALU-bound by construction (3 dependent-rounding ops per element pair, no FMA allowed).
#include "qpg_common.h"

#define TX_CH 64       // elements per LDS chunk (4 einsum groups of 16)
#define TX_LD 68       // padded row stride in floats (16 B aligned, conflict-free b128)

template <int QB>
__global__ __launch_bounds__(256) void text_cosine_f32_kernel(const float* __restrict__ xn, int N, int R, int Dm,
                                                              const int32_t* __restrict__ cand_r, int G,
                                                              const float* __restrict__ qn, int Q,
                                                              float* __restrict__ D, int64_t ldD) {
  __shared__ __attribute__((aligned(16))) float xs[64 * TX_LD];
  __shared__ __attribute__((aligned(16))) float qs[4 * QB * TX_CH];

  const int64_t C = (int64_t)N * G;
  const int lc = threadIdx.x & 63, qg = threadIdx.x >> 6;
  const int64_t c0 = (int64_t)blockIdx.x * 64;
  const int q0 = blockIdx.y * (4 * QB);

  float acc[QB][4];
#pragma unroll
  for (int i = 0; i < QB; ++i) acc[i][0] = acc[i][1] = acc[i][2] = acc[i][3] = 0.f;

  for (int ch = 0; ch < Dm; ch += TX_CH) {
    // stage 64 candidate rows x 64 floats: 1024 float4, 4 per thread, coalesced along the row
    for (int v = threadIdx.x; v < 64 * (TX_CH / 4); v += 256) {
      const int r = v >> 4, e4 = (v & 15) << 2;
      int64_t c = c0 + r;
      if (c >= C) c = C - 1;
      const int j = (int)(c / G), g = (int)(c - (int64_t)j * G);
      const float* src = xn + ((int64_t)j * R + cand_r[g]) * Dm + ch + e4;
      *reinterpret_cast<f32x4*>(&xs[r * TX_LD + e4]) = *reinterpret_cast<const f32x4*>(src);
    }
    for (int v = threadIdx.x; v < 4 * QB * (TX_CH / 4); v += 256) {
      const int r = v >> 4, e4 = (v & 15) << 2;
      int q = q0 + r;
      if (q >= Q) q = Q - 1;
      *reinterpret_cast<f32x4*>(&qs[r * TX_CH + e4]) = *reinterpret_cast<const f32x4*>(qn + (int64_t)q * Dm + ch + e4);
    }
    __syncthreads();
#pragma unroll
    for (int g16 = 0; g16 < TX_CH / 16; ++g16) {
#pragma unroll
      for (int u = 3; u >= 0; --u) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(&xs[lc * TX_LD + g16 * 16 + u * 4]);
#pragma unroll
        for (int i = 0; i < QB; ++i) {
          const f32x4 qv = *reinterpret_cast<const f32x4*>(&qs[(qg * QB + i) * TX_CH + g16 * 16 + u * 4]);
          float d0 = f_sub(qv.x, x.x), d1 = f_sub(qv.y, x.y), d2 = f_sub(qv.z, x.z), d3 = f_sub(qv.w, x.w);
          acc[i][0] = f_add(f_mul(d0, d0), acc[i][0]);
          acc[i][1] = f_add(f_mul(d1, d1), acc[i][1]);
          acc[i][2] = f_add(f_mul(d2, d2), acc[i][2]);
          acc[i][3] = f_add(f_mul(d3, d3), acc[i][3]);
        }
      }
    }
    __syncthreads();
  }

  const int64_t c = c0 + lc;
  if (c < C) {
#pragma unroll
    for (int i = 0; i < QB; ++i) {
      const int q = q0 + qg * QB + i;
      if (q < Q) {
        const float s = f_add(f_add(acc[i][0], acc[i][1]), f_add(acc[i][2], acc[i][3]));
        D[(int64_t)q * ldD + c] = f_mul(0.5f, s);
      }
    }
  }
}

template <int QB>
static int launch_text(void* stream, const float* xn, int N, int R, int Dm, const int32_t* cand_r, int G,
                       const float* qn, int Q, float* D, int64_t ldD) {
  const int64_t C = (int64_t)N * G;
  dim3 grid((unsigned)((C + 63) / 64), (unsigned)((Q + 4 * QB - 1) / (4 * QB)));
  hipLaunchKernelGGL((text_cosine_f32_kernel<QB>), grid, dim3(256), 0, qpg_stream(stream), xn, N, R, Dm, cand_r, G,
                     qn, Q, D, ldD);
  QPG_LAUNCH_CHECK("text_cosine_f32_kernel");
  return QPG_OK;
}

extern "C" int qpg_text_cosine_f32(qpg_ctx* ctx, void* stream, const float* xn, int N, int R, int Dm,
                                   const int32_t* cand_r, int G, const float* qn, int Q, float* D, int64_t ldD) {
  QPG_REQUIRE(ctx && xn && cand_r && qn && D, "qpg_text_cosine_f32: null pointer");
  QPG_REQUIRE(N >= 0 && R > 0 && G > 0 && Q >= 0 && ldD >= (int64_t)N * G, "qpg_text_cosine_f32: bad size");
  if (Dm <= 0 || (Dm % TX_CH) != 0) {
    qpg_set_error("qpg_text_cosine_f32: compiled for Dm %% 64 == 0 (got %d)", Dm);
    return QPG_EUNSUP;
  }
  if (N == 0 || Q == 0) return QPG_OK;
  if (Q > 16) return launch_text<12>(stream, xn, N, R, Dm, cand_r, G, qn, Q, D, ldD);
  if (Q > 4) return launch_text<4>(stream, xn, N, R, Dm, cand_r, G, qn, Q, D, ldD);
  return launch_text<1>(stream, xn, N, R, Dm, cand_r, G, qn, Q, D, ldD);
}
