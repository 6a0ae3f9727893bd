// vq-wav2vec audio sweep: Levenshtein distance between 11-symbol strings (the mode the paper describes;
// CodeKNN.search_audio_cands(mode='wavvq_feat') + wavvq_distances(mode='combine'), GestureKNN.py:57-67,
// 676-677).  A feature row is 6 backward taps and 5 forward taps of the (g1,g2) vq-wav2vec index pair,
// zero padded at the window ends (data_processing.py:297-335); symbol = g1*320 + g2.  Like the WavLM
// stack, the (N,398,22) feature array is never materialised: rows are gathered from the (N,398) symbol
// track with the tap offsets.  Integer work: one thread per candidate, the DP row (12 ints) lives in
// registers, the query strings in LDS; exact small integers are stored as f32 so the per-code argmin /
// rank kernels are shared with the text path.
#include "qpg_common.h"

#define LEV_L 11

struct LevTaps {
  int off[LEV_L];   // signed frame offset of each of the 11 taps (back taps negative)
};

__device__ __forceinline__ void gather_symbols(const int32_t* __restrict__ track, int T, int t, const LevTaps& taps,
                                               int* out) {
#pragma unroll
  for (int i = 0; i < LEV_L; ++i) {
    const int tt = t + taps.off[i];
    out[i] = (tt >= 0 && tt < T) ? track[tt] : 0;       // zero padding: (g1,g2) = (0,0) -> symbol 0
  }
}

__global__ __launch_bounds__(256) void wavvq_lev_kernel(const int32_t* __restrict__ sym_db, int N, int T,
                                                        const int32_t* __restrict__ cand_t, int G, LevTaps taps,
                                                        const int32_t* __restrict__ sym_q, int Tq,
                                                        const int32_t* __restrict__ q_win,
                                                        const int32_t* __restrict__ q_t, int Q,
                                                        float* __restrict__ D, int64_t ldD) {
  extern __shared__ int qs[];   // [Q][11]
  for (int i = threadIdx.x; i < Q * LEV_L; i += blockDim.x) {
    const int q = i / LEV_L, k = i - q * LEV_L;
    const int tt = q_t[q] + taps.off[k];
    qs[i] = (tt >= 0 && tt < Tq) ? sym_q[(int64_t)q_win[q] * Tq + tt] : 0;
  }
  __syncthreads();
  const int64_t C = (int64_t)N * G;
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const int j = (int)(c / G), g = (int)(c - (int64_t)j * G);
  int b[LEV_L];
  gather_symbols(sym_db + (int64_t)j * T, T, cand_t[g], taps, b);
  for (int q = 0; q < Q; ++q) {
    const int* a = qs + q * LEV_L;
    int row[LEV_L + 1];
#pragma unroll
    for (int x = 0; x <= LEV_L; ++x) row[x] = x;
#pragma unroll
    for (int i = 1; i <= LEV_L; ++i) {
      const int ai = a[i - 1];
      int diag = row[0];
      row[0] = i;
#pragma unroll
      for (int x = 1; x <= LEV_L; ++x) {
        const int up = row[x];
        const int v = min(min(up + 1, row[x - 1] + 1), diag + (ai != b[x - 1]));
        diag = up;
        row[x] = v;
      }
    }
    D[(int64_t)q * ldD + c] = (float)row[LEV_L];
  }
}

extern "C" int qpg_wavvq_lev_f32(qpg_ctx* ctx, void* stream, const int32_t* sym_db, int N, int T,
                                 const int32_t* cand_t, int G, const int32_t* tap_off, int n_taps,
                                 const int32_t* sym_q, int Mq, int Tq, const int32_t* q_win, const int32_t* q_t,
                                 int Q, float* D, int64_t ldD) {
  QPG_REQUIRE(ctx && sym_db && cand_t && tap_off && sym_q && q_win && q_t && D, "qpg_wavvq_lev_f32: null pointer");
  QPG_REQUIRE(N >= 0 && T > 0 && G > 0 && Mq > 0 && Tq > 0 && Q >= 0 && ldD >= (int64_t)N * G,
              "qpg_wavvq_lev_f32: bad size");
  if (n_taps != LEV_L) {
    qpg_set_error("qpg_wavvq_lev_f32: compiled for %d-symbol strings (got %d)", LEV_L, n_taps);
    return QPG_EUNSUP;
  }
  if (N == 0 || Q == 0) return QPG_OK;
  LevTaps taps;   // tap_off is a HOST array of 11 ints (tiny, part of the call like the scalar arguments)
  for (int i = 0; i < LEV_L; ++i) taps.off[i] = tap_off[i];
  const int64_t C = (int64_t)N * G;
  // the query strings of a launch live in LDS (44 B each): long clips go in chunks of 1024 queries (45 KB)
  constexpr int QCH = 1024;
  for (int q0 = 0; q0 < Q; q0 += QCH) {
    const int qn = Q - q0 < QCH ? Q - q0 : QCH;
    hipLaunchKernelGGL(wavvq_lev_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), sizeof(int) * (size_t)qn * LEV_L,
                       qpg_stream(stream), sym_db, N, T, cand_t, G, taps, sym_q, Tq, q_win + q0, q_t + q0, qn,
                       D + (int64_t)q0 * ldD, ldD);
    QPG_LAUNCH_CHECK("wavvq_lev_kernel");
  }
  return QPG_OK;
}
