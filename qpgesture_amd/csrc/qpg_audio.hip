// Audio (WavLM) candidate sweep: float64 cosine distance of every query step against every
// database candidate, on the f64 matrix cores (v_mfma_f64_16x16x4_f64).
//
// Replaces CodeKNN.search_audio_cands(mode='wavlm_feat') (GestureKNN.py:666-691) and the
// feature stacking of data_processing.py:264-268.  A candidate is six frames of the
// interpolated WavLM track, two frames apart; the reference materialises each candidate as a
// 6144-d float64 row (8.85 MB per DB window).  Here candidates are *addressed*, never stored:
// a block gathers its 16 candidates' frames straight from the (N,180,1024) f32 base, so HBM
// traffic is the base array once (frames shared by two neighbouring candidates are re-touched
// within three loop iterations of the same wave and hit L1/L2).
//
// Tiling (gfx950, wave64):
//   block  = 256 threads = 4 waves (one per SIMD) -> 16 consecutive candidates x NT*16 queries
//   wave w = K-slice e in [w*F/4, (w+1)*F/4) of every tap  (split-K over the feature axis)
//   MFMA   A = candidates (row = lane&15, k = lane>>4), B = queries (col = lane&15), f64 acc
//   each lane loads 16 B of its candidate row per (e0, tap) and feeds four MFMA k-steps from it
//   partial sums of the 4 waves are reduced through LDS, turned into distances and stored as
//   128-B runs along the candidate axis of D[q][c].
#include "qpg_common.h"

__global__ __launch_bounds__(256) void audio_pack_queries_kernel(const float* __restrict__ qbase, int M, int T, int F,
                                                                 const int32_t* __restrict__ q_win,
                                                                 const int32_t* __restrict__ q_t, int n_taps,
                                                                 int tap_stride, double* __restrict__ q64,
                                                                 double* __restrict__ qn2) {
  const int q = blockIdx.x;
  const int w = q_win[q], t0 = q_t[q];
  const int K = n_taps * F;
  double s = 0.0;
  for (int i = threadIdx.x; i < K; i += blockDim.x) {
    int tap = i / F, e = i - tap * F;
    int t = t0 + tap * tap_stride;
    double v = (t < T) ? (double)qbase[((int64_t)w * T + t) * F + e] : 0.0;
    q64[(int64_t)q * K + i] = v;
    s += v * v;
  }
  __shared__ double red[4];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) qn2[q] = red[0] + red[1] + red[2] + red[3];
}

extern "C" int qpg_audio_pack_queries(qpg_ctx* ctx, void* stream, const float* qbase, int M, int T, int F,
                                      const int32_t* q_win, const int32_t* q_t, int Q, int n_taps, int tap_stride,
                                      double* q64, double* qn2) {
  QPG_REQUIRE(ctx && qbase && q_win && q_t && q64 && qn2 && M > 0 && T > 0 && F > 0 && Q >= 0 && n_taps > 0 &&
                  tap_stride > 0,
              "qpg_audio_pack_queries: bad argument");
  if (Q == 0) return QPG_OK;
  hipLaunchKernelGGL(audio_pack_queries_kernel, dim3(Q), dim3(256), 0, qpg_stream(stream), qbase, M, T, F, q_win,
                     q_t, n_taps, tap_stride, q64, qn2);
  QPG_LAUNCH_CHECK("audio_pack_queries_kernel");
  return QPG_OK;
}

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

template <int NT, int NTAPS>
__global__ __launch_bounds__(256) void audio_cosine_f64_kernel(const float* __restrict__ base, int N, int T, int F,
                                                               const int32_t* __restrict__ cand_t, int G,
                                                               int tap_stride, const double* __restrict__ cn2,
                                                               const double* __restrict__ q64,
                                                               const double* __restrict__ qn2, int Q,
                                                               double* __restrict__ D, int64_t ldD) {
  __shared__ double red[4][NT][4][64];  // [wave][query tile][acc reg][lane]

  const int64_t C = (int64_t)N * G;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int row = lane & 15, kq = lane >> 4;
  const int64_t c0 = (int64_t)blockIdx.x * 16;
  const int q0 = blockIdx.y * (NT * 16);

  // A side: this lane's candidate row
  int64_t c = c0 + row;
  if (c >= C) c = C - 1;
  const int j = (int)(c / G), g = (int)(c - (int64_t)j * G);
  const int t0 = cand_t[g];
  const float* arow = base + ((int64_t)j * T + t0) * F + 4 * kq;
  // B side: this lane's query column in each of the NT tiles
  const int KQ = NTAPS * F;
  const double* brow[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    int q = q0 + nt * 16 + row;
    if (q >= Q) q = Q - 1;
    brow[nt] = q64 + (int64_t)q * KQ + 4 * kq;
  }

  f64x4 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt] = (f64x4){0.0, 0.0, 0.0, 0.0};

  const int eBeg = w * (F >> 2), eEnd = eBeg + (F >> 2);
  for (int e0 = eBeg; e0 < eEnd; e0 += 16) {
#pragma unroll
    for (int tap = 0; tap < NTAPS; ++tap) {
      const int t = t0 + tap * tap_stride;
      f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (t < T) a = *reinterpret_cast<const f32x4*>(arow + (int64_t)tap * tap_stride * F + e0);
      const double a0 = a.x, a1 = a.y, a2 = a.z, a3 = a.w;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const f64x4 b = *reinterpret_cast<const f64x4*>(brow[nt] + tap * F + e0);
        acc[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b.x, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b.y, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b.z, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, b.w, acc[nt], 0, 0, 0);
      }
    }
  }

#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    red[w][nt][0][lane] = acc[nt].x;
    red[w][nt][1][lane] = acc[nt].y;
    red[w][nt][2][lane] = acc[nt].z;
    red[w][nt][3][lane] = acc[nt].w;
  }
  __syncthreads();

  // f64 C/D layout of v_mfma_f64_16x16x4_f64: lane l, reg r holds (cand row = (l>>4) + 4r, query col = l&15).
  // Output element o = ql*16 + cr (query-local, candidate row): consecutive threads walk the
  // candidate axis, so each query row gets one 128-B store run.
  for (int o = threadIdx.x; o < NT * 256; o += 256) {
    const int ql = o >> 4, cr = o & 15;
    const int nt = ql >> 4, qc = ql & 15;
    const int r = cr >> 2, l = ((cr & 3) << 4) | qc;
    const double dot = red[0][nt][r][l] + red[1][nt][r][l] + red[2][nt][r][l] + red[3][nt][r][l];
    const int q = q0 + ql;
    const int64_t cc = c0 + cr;
    if (q < Q && cc < C) D[(int64_t)q * ldD + cc] = cosine_from_dot(dot, qn2[q], cn2[cc]);
  }
}

template <int NT>
static int launch_audio(qpg_ctx* ctx, void* stream, const float* base, int N, int T, int F, const int32_t* cand_t,
                        int G, int tap_stride, const double* cn2, const double* q64, const double* qn2, int Q,
                        int qtiles_y, double* D, int64_t ldD) {
  int64_t C = (int64_t)N * G;
  dim3 grid((unsigned)((C + 15) / 16), (unsigned)qtiles_y);
  hipLaunchKernelGGL((audio_cosine_f64_kernel<NT, 6>), grid, dim3(256), 0, qpg_stream(stream), base, N, T, F, cand_t,
                     G, tap_stride, cn2, q64, qn2, Q, D, ldD);
  QPG_LAUNCH_CHECK("audio_cosine_f64_kernel");
  return QPG_OK;
}

extern "C" int qpg_audio_cosine_f64(qpg_ctx* ctx, void* stream, const float* base, int N, int T, int F,
                                    const int32_t* cand_t, int G, int n_taps, int tap_stride, const double* cn2,
                                    const double* q64, const double* qn2, int Q, double* D, int64_t ldD) {
  QPG_REQUIRE(ctx && base && cand_t && cn2 && q64 && qn2 && D, "qpg_audio_cosine_f64: null pointer");
  QPG_REQUIRE(N >= 0 && T > 0 && G > 0 && Q >= 0 && tap_stride > 0 && ldD >= (int64_t)N * G,
              "qpg_audio_cosine_f64: bad size");
  if (n_taps != 6 || F <= 0 || (F % 64) != 0) {
    qpg_set_error("qpg_audio_cosine_f64: compiled for n_taps=6 and F %% 64 == 0 (got n_taps=%d F=%d)", n_taps, F);
    return QPG_EUNSUP;
  }
  if (N == 0 || Q == 0) return QPG_OK;
  const int qt = (Q + 15) / 16;  // 16-query tiles
  // widest tile that divides the work without an empty tail: prefer 3 (a 24 s clip is 48 queries)
  if (qt % 3 == 0)
    return launch_audio<3>(ctx, stream, base, N, T, F, cand_t, G, tap_stride, cn2, q64, qn2, Q, qt / 3, D, ldD);
  if (qt % 4 == 0)
    return launch_audio<4>(ctx, stream, base, N, T, F, cand_t, G, tap_stride, cn2, q64, qn2, Q, qt / 4, D, ldD);
  if (qt % 2 == 0)
    return launch_audio<2>(ctx, stream, base, N, T, F, cand_t, G, tap_stride, cn2, q64, qn2, Q, qt / 2, D, ldD);
  return launch_audio<1>(ctx, stream, base, N, T, F, cand_t, G, tap_stride, cn2, q64, qn2, Q, qt, D, ldD);
}
