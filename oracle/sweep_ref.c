/*
 * ORACLE (test infrastructure, not product): plain-C restatement of the two candidate scans of the
 * reference's CodeKNN — search_audio_cands(mode='wavlm_feat') and search_text_cands
 * (codebook/Speech2GestureMatching/GestureKNN.py:666-691, 708-721) — with the distance
 * sklearn.metrics.pairwise.paired_distances(metric='cosine') computes, in the SAME floating-point
 * operation order (NumPy einsum, SSE baseline: 2 f64 / 4 f32 lane accumulators, separate mul and
 * add, unrolled groups visited 3,2,1,0; see oracle/knn_oracle.py).  Results are bit-identical to
 * the reference's; tests/test_oracle_golden.py checks that against the reference-generated goldens.
 *
 * Used only as (a) a checker in tests and (b) the `cpu_baseline` leg of bench.py ("port":
 * a competent scalar/SSE CPU implementation of the same scan, OpenMP over DB windows).
 * Build: gcc -O2 -fopenmp -ffp-contract=off -shared -fPIC (oracle/Makefile).  Never linked into
 * the product.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static double einsum_sq_f64(const double* x, int64_t n) {
  double a0 = 0.0, a1 = 0.0;
  int64_t i = 0;
  for (; n - i >= 8; i += 8) { /* 4 vectors of 2 lanes, visited 3,2,1,0 */
    a0 = x[i + 6] * x[i + 6] + a0; a1 = x[i + 7] * x[i + 7] + a1;
    a0 = x[i + 4] * x[i + 4] + a0; a1 = x[i + 5] * x[i + 5] + a1;
    a0 = x[i + 2] * x[i + 2] + a0; a1 = x[i + 3] * x[i + 3] + a1;
    a0 = x[i + 0] * x[i + 0] + a0; a1 = x[i + 1] * x[i + 1] + a1;
  }
  for (; i < n; i += 2) {
    double v0 = x[i], v1 = (i + 1 < n) ? x[i + 1] : 0.0;
    a0 = v0 * v0 + a0; a1 = v1 * v1 + a1;
  }
  return a0 + a1;
}

static double einsum_sqdiff_f64(const double* x, const double* y, int64_t n) {
  double a0 = 0.0, a1 = 0.0, d;
  int64_t i = 0;
#define STEP64(o, acc) d = x[i + (o)] - y[i + (o)]; acc = d * d + acc;
  for (; n - i >= 8; i += 8) {
    STEP64(6, a0) STEP64(7, a1) STEP64(4, a0) STEP64(5, a1) STEP64(2, a0) STEP64(3, a1) STEP64(0, a0) STEP64(1, a1)
  }
  for (; i < n; i += 2) {
    d = x[i] - y[i]; a0 = d * d + a0;
    if (i + 1 < n) { d = x[i + 1] - y[i + 1]; a1 = d * d + a1; }
  }
#undef STEP64
  return a0 + a1;
}

static float einsum_sq_f32(const float* x, int64_t n) {
  float a[4] = {0.f, 0.f, 0.f, 0.f};
  int64_t i = 0;
  for (; n - i >= 16; i += 16)
    for (int u = 3; u >= 0; --u)
      for (int l = 0; l < 4; ++l) { float v = x[i + 4 * u + l]; a[l] = v * v + a[l]; }
  for (; i < n; i += 4)
    for (int l = 0; l < 4; ++l) { float v = (i + l < n) ? x[i + l] : 0.f; a[l] = v * v + a[l]; }
  return (a[0] + a[1]) + (a[2] + a[3]);
}

static float einsum_sqdiff_f32(const float* x, const float* y, int64_t n) {
  float a[4] = {0.f, 0.f, 0.f, 0.f};
  int64_t i = 0;
  for (; n - i >= 16; i += 16)
    for (int u = 3; u >= 0; --u)
      for (int l = 0; l < 4; ++l) { float d = x[i + 4 * u + l] - y[i + 4 * u + l]; a[l] = d * d + a[l]; }
  for (; i < n; i += 4)
    for (int l = 0; l < 4; ++l) { float d = (i + l < n) ? x[i + l] - y[i + l] : 0.f; a[l] = d * d + a[l]; }
  return (a[0] + a[1]) + (a[2] + a[3]);
}

static void normalize_f64(double* x, int64_t n) {
  double nr = sqrt(einsum_sq_f64(x, n));
  if (nr < 10.0 * 2.220446049250313e-16) nr = 1.0;
  for (int64_t i = 0; i < n; ++i) x[i] = x[i] / nr;
}

static void normalize_f32(float* x, int64_t n) {
  float nr = sqrtf(einsum_sq_f32(x, n));
  if (nr < 10.f * 1.1920928955078125e-07f) nr = 1.f;
  for (int64_t i = 0; i < n; ++i) x[i] = x[i] / nr;
}

/* Merge helper: best[] per code with first-wins (strict <) in candidate order. */
typedef struct { double d; int32_t idx; } best_t;

/*
 * Audio scan for Q queries.  base f32 [N][T][F]; candidate (j,g) = concat_i base[j][cand_t[g]+i*stride]
 * promoted to f64 (zeros past T); q f64 [Q][ntaps*F] raw (un-normalised) query rows.
 * out_dist f64 [Q][K] (1e3 where absent), out_idx i32 [Q][K] = j*G+g or -1.
 */
void qpg_ref_audio_scan(const float* base, int N, int T, int F, const int32_t* cand_t, int G, int ntaps, int stride,
                        const int32_t* code, int code_ld, const int32_t* cand_cidx, const double* q, int Q, int K,
                        double* out_dist, int32_t* out_idx, int n_threads) {
  const int64_t D = (int64_t)ntaps * F;
  double* qn = (double*)malloc(sizeof(double) * Q * D);
  memcpy(qn, q, sizeof(double) * Q * D);
  for (int i = 0; i < Q; ++i) normalize_f64(qn + i * D, D);
  double* dist = (double*)malloc(sizeof(double) * (size_t)Q * N * G);
  (void)n_threads;
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads)
  for (int j = 0; j < N; ++j) {
    double* c = (double*)malloc(sizeof(double) * D);
    for (int g = 0; g < G; ++g) {
      for (int i = 0; i < ntaps; ++i) {
        int t = cand_t[g] + i * stride;
        for (int e = 0; e < F; ++e) c[(int64_t)i * F + e] = (t < T) ? (double)base[((int64_t)j * T + t) * F + e] : 0.0;
      }
      normalize_f64(c, D);
      for (int qi = 0; qi < Q; ++qi)
        dist[(size_t)qi * N * G + (size_t)j * G + g] = 0.5 * einsum_sqdiff_f64(qn + qi * D, c, D);
    }
    free(c);
  }
  for (int qi = 0; qi < Q; ++qi) {
    for (int k = 0; k < K; ++k) { out_dist[(size_t)qi * K + k] = 1e+3; out_idx[(size_t)qi * K + k] = -1; }
    for (int j = 0; j < N; ++j)
      for (int g = 0; g < G; ++g) {
        int cd = code[(int64_t)j * code_ld + cand_cidx[g]];
        if (cd < 0 || cd >= K) continue;   /* masked / invalid candidate (the product kernels skip these too) */
        double d = dist[(size_t)qi * N * G + (size_t)j * G + g];
        if (d < out_dist[(size_t)qi * K + cd]) { out_dist[(size_t)qi * K + cd] = d; out_idx[(size_t)qi * K + cd] = j * G + g; }
      }
  }
  free(dist);
  free(qn);
}

/* Text scan.  ctx f32 [N][R][Dm]; candidate (j,g) = ctx[j][cand_r[g]]; q f32 [Q][Dm] raw. */
void qpg_ref_text_scan(const float* ctx, int N, int R, int Dm, const int32_t* cand_r, int G, const int32_t* code,
                       int code_ld, const int32_t* cand_cidx, const float* q, int Q, int K, float* out_dist,
                       int32_t* out_idx, int n_threads) {
  float* qn = (float*)malloc(sizeof(float) * Q * Dm);
  memcpy(qn, q, sizeof(float) * Q * Dm);
  for (int i = 0; i < Q; ++i) normalize_f32(qn + (int64_t)i * Dm, Dm);
  float* dist = (float*)malloc(sizeof(float) * (size_t)Q * N * G);
  (void)n_threads;
#pragma omp parallel for schedule(dynamic, 4) num_threads(n_threads)
  for (int j = 0; j < N; ++j) {
    float* c = (float*)malloc(sizeof(float) * Dm);
    for (int g = 0; g < G; ++g) {
      memcpy(c, ctx + ((int64_t)j * R + cand_r[g]) * Dm, sizeof(float) * Dm);
      normalize_f32(c, Dm);
      for (int qi = 0; qi < Q; ++qi)
        dist[(size_t)qi * N * G + (size_t)j * G + g] = 0.5f * einsum_sqdiff_f32(qn + (int64_t)qi * Dm, c, Dm);
    }
    free(c);
  }
  for (int qi = 0; qi < Q; ++qi) {
    for (int k = 0; k < K; ++k) { out_dist[(size_t)qi * K + k] = 1e+3f; out_idx[(size_t)qi * K + k] = -1; }
    for (int j = 0; j < N; ++j)
      for (int g = 0; g < G; ++g) {
        int cd = code[(int64_t)j * code_ld + cand_cidx[g]];
        if (cd < 0 || cd >= K) continue;   /* masked / invalid candidate */
        float d = dist[(size_t)qi * N * G + (size_t)j * G + g];
        if (d < out_dist[(size_t)qi * K + cd]) { out_dist[(size_t)qi * K + cd] = d; out_idx[(size_t)qi * K + cd] = j * G + g; }
      }
  }
  free(dist);
  free(qn);
}
