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

__global__ __launch_bounds__(1024) void audio_pack_queries_kernel(const float* __restrict__ qbase, int M, int T, int F,
                                                                  const int32_t* __restrict__ q_win,
                                                                  const int32_t* __restrict__ q_t, int n_taps,
                                                                  int tap_stride, float* __restrict__ q32,
                                                                  double* __restrict__ qn2) {
  // one block of 1024 threads per query (this kernel is on the critical path in front of the sweep: 6 iterations
  // per thread instead of 24); 16-B copies when F % 4 == 0; squared norm in f64, fixed reduction order
  const int q = blockIdx.x;
  const int w = q_win[q], t0 = q_t[q];
  const int K = n_taps * F;
  double s = 0.0;
  if ((F & 3) == 0) {
    const int K4 = K >> 2, F4 = F >> 2;
    for (int i = threadIdx.x; i < K4; i += blockDim.x) {
      const int tap = i / F4, e4 = i - tap * F4;
      const int t = t0 + tap * tap_stride;
      f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (t < T) v = reinterpret_cast<const f32x4*>(qbase + ((int64_t)w * T + t) * F)[e4];
      reinterpret_cast<f32x4*>(q32 + (int64_t)q * K)[i] = v;
      s += (double)v.x * (double)v.x;
      s += (double)v.y * (double)v.y;
      s += (double)v.z * (double)v.z;
      s += (double)v.w * (double)v.w;
    }
  } else {
    for (int i = threadIdx.x; i < K; i += blockDim.x) {
      const int tap = i / F, e = i - tap * F;
      const int t = t0 + tap * tap_stride;
      const float vf = (t < T) ? qbase[((int64_t)w * T + t) * F + e] : 0.f;
      q32[(int64_t)q * K + i] = vf;
      s += (double)vf * (double)vf;
    }
  }
  __shared__ double red[16];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i];
    qn2[q] = t;
  }
}

extern "C" int qpg_audio_pack_queries(qpg_ctx* ctx, void* stream, const float* qbase, int M, int T, int F,
                                      const int32_t* q_win, const int32_t* q_t, int Q, int n_taps, int tap_stride,
                                      float* q32, double* qn2) {
  QPG_REQUIRE(ctx && qbase && q_win && q_t && q32 && qn2 && M > 0 && T > 0 && F > 0 && Q >= 0 && n_taps > 0 &&
                  tap_stride > 0,
              "qpg_audio_pack_queries: bad argument");
  if (Q == 0) return QPG_OK;
  hipLaunchKernelGGL(audio_pack_queries_kernel, dim3(Q), dim3(1024), 0, qpg_stream(stream), qbase, M, T, F, q_win,
                     q_t, n_taps, tap_stride, q32, qn2);
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

// Wave tile = MT*16 candidates x NT*16 queries.  Per (e0, tap) a lane loads MT + NT float4 (16 B each;
// the query operand is kept in f32 in memory — WavLM values are f32 — and widened in registers) and
// issues 4*MT*NT MFMAs, i.e. (MT+NT)*16 B of L1/L2 traffic per 4*MT*NT*64 matrix-pipe cycles.  At
// MT=1, NT=3 with f64 queries the kernel was L2->L1 bound (37 B/clk/CU, r01 v1 profile: 31 TF);
// MT=2, NT=3 with f32 queries needs 13 B/clk/CU.
// HALF: the base track is stored in f16 (BASELINE.json configs[4] "fp16 features"): one 16-byte load brings the lane's 8
// features, widened f16 -> f32 -> f64 in registers; the arithmetic is the same f64 as for an f32 base, applied to the
// f16-rounded values (so results match the oracle run on the rounded track, not on the original one).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int MT, int NT, int NTAPS, int KS, bool HALF>
__global__ __launch_bounds__(64 * KS) void audio_cosine_f64_kernel(const void* __restrict__ base_, int N, int T, int F,
                                                               const int32_t* __restrict__ cand_t, int G,
                                                               int tap_stride, const double* __restrict__ cn2,
                                                               const float* __restrict__ q32,
                                                               const double* __restrict__ qn2, int Q,
                                                               double* __restrict__ D, int64_t ldD,
                                                               const float* __restrict__ zeros) {
  __shared__ double red[KS][MT * NT][4][64];  // [wave][tile][acc reg][lane]

  const int64_t C = (int64_t)N * G;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int row = lane & 15, kq = lane >> 4;
  const int64_t c0 = (int64_t)blockIdx.x * (16 * MT);
  const int q0 = blockIdx.y * (NT * 16);

  // A side: this lane's candidate row in each of the MT tiles (element offsets into the base track)
  const float* base = static_cast<const float*>(base_);
  const _Float16* baseh = static_cast<const _Float16*>(base_);
  int64_t aoff[MT];
  int at0[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    int64_t c = c0 + mt * 16 + row;
    if (c >= C) c = C - 1;
    const int j = (int)(c / G), g = (int)(c - (int64_t)j * G);
    at0[mt] = cand_t[g];
    aoff[mt] = ((int64_t)j * T + at0[mt]) * F + 8 * kq;
  }
  // B side: this lane's query column in each of the NT tiles
  const int KQ = NTAPS * F;
  const float* brow[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    int q = q0 + nt * 16 + row;
    if (q >= Q) q = Q - 1;
    brow[nt] = q32 + (int64_t)q * KQ + 8 * kq;
  }

  f64x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f64x4){0.0, 0.0, 0.0, 0.0};

  // One operand buffer = 8 consecutive features per lane (two 16-B loads: a wave instruction pair
  // consumes whole 128-B lines of 16 rows) for each of the MT candidate and NT query tiles = 8 MFMA
  // k-steps.  Two buffers alternate so the loads of the next (tap, e) group are in flight while the
  // matrix pipe works on the current one (r01 PMC: 31 % of wave time was s_waitcnt before this).
  struct Buf {
    f32x4 a[MT][2], b[NT][2];
  };
  auto load = [&](Buf& u, int e0, int tap) {
#if defined(QPG_AUDIO_PROBE)     // experiments/conv_probe: operands from constants, no loads
    for (int mt = 0; mt < MT; ++mt) u.a[mt][0] = u.a[mt][1] = (f32x4){1.f, 1.f, 1.f, 1.f};
    for (int nt = 0; nt < NT; ++nt) u.b[nt][0] = u.b[nt][1] = (f32x4){1.f, 1.f, 1.f, 1.f};
    return;
#endif
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      // taps past the end of the window are zero padding (data_processing.py:266): the load is unconditional (a
      // conditional load makes hipcc branch around it and drain vmcnt(0), which serialises the prefetch) and a
      // padded tap reads the context's zero page instead of being selected to zero afterwards
      const bool ok = at0[mt] + tap * tap_stride < T;
      const int64_t o = aoff[mt] + (int64_t)tap * tap_stride * F + e0;
      if (HALF) {
        const f16x8 h = *reinterpret_cast<const f16x8*>(ok ? reinterpret_cast<const void*>(baseh + o)
                                                           : reinterpret_cast<const void*>(zeros));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          u.a[mt][0][i] = (float)h[i];
          u.a[mt][1][i] = (float)h[4 + i];
        }
      } else {
        const float* p = ok ? base + o : zeros;
        u.a[mt][0] = *reinterpret_cast<const f32x4*>(p);
        u.a[mt][1] = *reinterpret_cast<const f32x4*>(p + 4);
      }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const float* p = brow[nt] + tap * F + e0;
      u.b[nt][0] = *reinterpret_cast<const f32x4*>(p);
      u.b[nt][1] = *reinterpret_cast<const f32x4*>(p + 4);
    }
  };
  auto mma = [&](const Buf& u) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        double bd[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bd[nt] = (double)u.b[nt][h][i];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const double ad = (double)u.a[mt][h][i];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(ad, bd[nt], acc[mt][nt], 0, 0, 0);
        }
      }
  };

  static_assert(NTAPS == 6, "tap schedule below is written for 6 taps");
  const int eBeg = w * (F / KS), eEnd = eBeg + (F / KS);
  Buf u0, u1;
  load(u0, eBeg, 0);
  // Issue pattern of one (load next group, 8*MT*NT MFMAs of the current group) stage: the (MT+NT)*2 16-B loads are
  // spread between the MFMAs — 2 MFMAs, 1 load, ... — instead of being issued as one burst in front of them, so the
  // matrix pipe is never left waiting behind a queue of address computations and the loads still lead their use by
  // a whole stage.  sched_barrier(0) closes the region.
#ifndef QPG_MPL
#define QPG_MPL 2
#endif
#define QPG_AUDIO_STAGE(LOADSTMT, MMASTMT)                                   \
  LOADSTMT;                                                                  \
  MMASTMT;                                                                   \
  _Pragma("unroll") for (int sg = 0; sg < (MT + NT) * 2; ++sg) {             \
    __builtin_amdgcn_sched_group_barrier(0x008, QPG_MPL, 0);                 \
    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                       \
  }                                                                          \
  __builtin_amdgcn_sched_group_barrier(0x008, 8 * MT * NT - QPG_MPL * 2 * (MT + NT), 0); \
  __builtin_amdgcn_sched_barrier(0);
  for (int e0 = eBeg; e0 < eEnd; e0 += 32) {
    QPG_AUDIO_STAGE(load(u1, e0, 1), mma(u0))
    QPG_AUDIO_STAGE(load(u0, e0, 2), mma(u1))
    QPG_AUDIO_STAGE(load(u1, e0, 3), mma(u0))
    QPG_AUDIO_STAGE(load(u0, e0, 4), mma(u1))
    QPG_AUDIO_STAGE(load(u1, e0, 5), mma(u0))
    const int en = (e0 + 32 < eEnd) ? e0 + 32 : eBeg;   // last prefetch wraps to a valid address, unused
    QPG_AUDIO_STAGE(load(u0, en, 0), mma(u1))
  }
#undef QPG_AUDIO_STAGE

#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      red[w][mt * NT + nt][0][lane] = acc[mt][nt].x;
      red[w][mt * NT + nt][1][lane] = acc[mt][nt].y;
      red[w][mt * NT + nt][2][lane] = acc[mt][nt].z;
      red[w][mt * NT + nt][3][lane] = acc[mt][nt].w;
    }
  __syncthreads();

  // f64 C/D layout of v_mfma_f64_16x16x4_f64: lane l, reg r holds (cand row = (l>>4) + 4r, query col = l&15).
  // Output element o = ql*(16*MT) + cr (query-local, candidate row): consecutive threads walk the
  // candidate axis, so each query row gets one (128*MT)-B store run.
  constexpr int CR = 16 * MT;
  for (int o = threadIdx.x; o < NT * 16 * CR; o += 64 * KS) {
    const int ql = o / CR, cr = o - ql * CR;
    const int nt = ql >> 4, qc = ql & 15;
    const int mt = cr >> 4, crr = cr & 15;
    const int r = crr >> 2, l = ((crr & 3) << 4) | qc;
    const int t = mt * NT + nt;
    double dot = red[0][t][r][l];
#pragma unroll
    for (int k = 1; k < KS; ++k) dot += red[k][t][r][l];
    const int q = q0 + ql;
    const int64_t cc = c0 + cr;
    if (q < Q && cc < C) {
      const double d = cosine_from_dot(dot, qn2[q], cn2[cc]);
      D[(int64_t)q * ldD + cc] = d;
    }
  }
}

template <int MT, int NT>
static int launch_audio(qpg_ctx* ctx, void* stream, const void* base, bool half, int N, int T, int F,
                        const int32_t* cand_t, int G, int tap_stride, const double* cn2, const float* q32,
                        const double* qn2, int Q, int qtiles_y, double* D, int64_t ldD) {
  int64_t C = (int64_t)N * G;
  dim3 grid((unsigned)((C + 16 * MT - 1) / (16 * MT)), (unsigned)qtiles_y);
  // 4 waves (one per SIMD) split the feature axis.  An 8-wave split (finer work units, 6.5 instead of
  // 3.25 rounds of blocks at N_db=2048) measured slower on MI355X: 703 vs 629 us (r01 notes).
  if (half)
    hipLaunchKernelGGL((audio_cosine_f64_kernel<MT, NT, 6, 4, true>), grid, dim3(256), 0, qpg_stream(stream), base, N, T,
                       F, cand_t, G, tap_stride, cn2, q32, qn2, Q, D, ldD, (const float*)ctx->zeros);
  else
    hipLaunchKernelGGL((audio_cosine_f64_kernel<MT, NT, 6, 4, false>), grid, dim3(256), 0, qpg_stream(stream), base, N, T,
                       F, cand_t, G, tap_stride, cn2, q32, qn2, Q, D, ldD, (const float*)ctx->zeros);
  QPG_LAUNCH_CHECK("audio_cosine_f64_kernel");
  return QPG_OK;
}

static int audio_cosine(const char* name, qpg_ctx* ctx, void* stream, const void* base, bool half, int N, int T, int F,
                        const int32_t* cand_t, int G, int n_taps, int tap_stride, const double* cn2, const float* q32,
                        const double* qn2, int Q, double* D, int64_t ldD) {
  QPG_REQUIRE(ctx && base && cand_t && cn2 && q32 && qn2 && D, "%s: null pointer", name);
  QPG_REQUIRE(N >= 0 && T > 0 && G > 0 && Q >= 0 && tap_stride > 0 && ldD >= (int64_t)N * G, "%s: bad size", name);
  if (n_taps != 6 || F <= 0 || (F % 128) != 0) {
    qpg_set_error("%s: compiled for n_taps=6 and F %% 128 == 0 (got n_taps=%d F=%d)", name, n_taps, F);
    return QPG_EUNSUP;
  }
  if (N == 0 || Q == 0) return QPG_OK;
  const int qt = (Q + 15) / 16;  // 16-query tiles
  // widest query tile that divides the work without an empty tail: prefer 3 (a 24 s clip is 48 queries)
#define QPG_AUDIO_ARGS ctx, stream, base, half, N, T, F, cand_t, G, tap_stride, cn2, q32, qn2, Q
#define QPG_AUDIO_TAIL D, ldD
  if (qt % 3 == 0) return launch_audio<2, 3>(QPG_AUDIO_ARGS, qt / 3, QPG_AUDIO_TAIL);
  if (qt % 4 == 0) return launch_audio<2, 4>(QPG_AUDIO_ARGS, qt / 4, QPG_AUDIO_TAIL);
  if (qt % 2 == 0) return launch_audio<2, 2>(QPG_AUDIO_ARGS, qt / 2, QPG_AUDIO_TAIL);
  return launch_audio<2, 1>(QPG_AUDIO_ARGS, qt, QPG_AUDIO_TAIL);
#undef QPG_AUDIO_TAIL
#undef QPG_AUDIO_ARGS
}

extern "C" int qpg_audio_cosine_f64(qpg_ctx* ctx, void* stream, const float* base, int N, int T, int F,
                                    const int32_t* cand_t, int G, int n_taps, int tap_stride, const double* cn2,
                                    const float* q32, const double* qn2, int Q, double* D, int64_t ldD) {
  return audio_cosine("qpg_audio_cosine_f64", ctx, stream, base, false, N, T, F, cand_t, G, n_taps, tap_stride, cn2, q32,
                      qn2, Q, D, ldD);
}

extern "C" int qpg_audio_cosine_f64_h(qpg_ctx* ctx, void* stream, const void* base_f16, int N, int T, int F,
                                      const int32_t* cand_t, int G, int n_taps, int tap_stride, const double* cn2,
                                      const float* q32, const double* qn2, int Q, double* D, int64_t ldD) {
  QPG_REQUIRE((reinterpret_cast<uintptr_t>(base_f16) % 16) == 0, "qpg_audio_cosine_f64_h: base must be 16-byte aligned");
  return audio_cosine("qpg_audio_cosine_f64_h", ctx, stream, base_f16, true, N, T, F, cand_t, G, n_taps, tap_stride, cn2,
                      q32, qn2, Q, D, ldD);
}
