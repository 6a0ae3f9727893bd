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
                                                               const float* __restrict__ zeros, int64_t c_begin,
                                                               int64_t c_end) {
  __shared__ double red[KS][MT * NT][4][64];  // [wave][tile][acc reg][lane]

  const int64_t C = c_end;                     // candidates [c_begin, c_end) of the N*G
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int row = lane & 15, kq = lane >> 4;
  const int64_t c0 = c_begin + (int64_t)blockIdx.x * (16 * MT);
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

// ---------------------------------------------------------------------------------------------------------------
// Mixed-precision sweep (round 2): the same tiling on the f32 matrix cores (v_mfma_f32_16x16x4_f32, twice the f64
// rate), made SAFE for an exact-result pipeline by bounding its error a priori:
//   * one f32 accumulator chain covers ONE stage = 8 MFMAs = 32 products (the chain restarts from C = 0 every
//     stage), and the finished chain is added into an f64 running sum on the VALU while the next stage's MFMAs run;
//   * an f32 FMA chain of n products has |error| <= gamma_n * sum|a_i b_i| <= gamma_n |a||b| (Cauchy-Schwarz), any
//     summation order, gamma_n = n u / (1 - n u), u = 2^-24; the f64 sums add < 1e-13 relative;
//   => |D_mx - D_exact| <= gamma_32 + 1e-13 < 1.92e-6 for every (query, candidate), data independent; + 1.2e-7 when
//      the matrix is stored in f32 (distances <= 2): QPG_AUDIO_MX_ERR = 2.05e-6 covers both forms
//      (QPG_AUDIO_MX_ERR below; tests/test_gpu_matching.py measures the actual maximum, ~1e-7).
// qpg_percode_select_mixed_f64 consumes this matrix: every comparison that decides an output (per-code minimum,
// rank order of the minima) and whose operands are closer than 2*QPG_AUDIO_MX_ERR is re-evaluated with an f64 dot
// product first and, when still closer than the near-tie eps, in the reference's own arithmetic.  Operand products
// that underflow f32 would break the bound: a pair with 0 < |q||c| < 1e-16 raises stats[1] |= 2.
// ---------------------------------------------------------------------------------------------------------------
#ifndef QPG_MX_PROBE
#define QPG_MX_PROBE 0     // experiments/audio_mx ablations (results wrong): 1 = no candidate loads, 2 = no query loads,
#endif                     // 4 = no f64 flush
#ifndef QPG_MX_OCC
#define QPG_MX_OCC 1      // waves per SIMD the register allocation is held to
#endif
template <int S>
struct MxIC {
  static constexpr int value = S;
};
// AD = depth of the candidate-operand register ring: the candidate rows come from HBM (~2k cycles away under load)
// and are prefetched AD-1 stages (of 8*MT*NT MFMAs = 256*MT*NT matrix-pipe cycles) ahead; the query operand sits in
// the XCD's L2 (the whole query set is 1.2 MB) and is prefetched one stage ahead.
template <int MT, int NT, int NTAPS, int KS, int GS, int AD, int BD, bool HALF = false>
__device__ __forceinline__ void mx_ksplit_body(
    const float* __restrict__ base, int N, int T, int F, const int32_t* __restrict__ cand_t, int G, int tap_stride,
    const double* __restrict__ cn2, const float* __restrict__ q32, const double* __restrict__ qn2, int Q,
    double* __restrict__ D, int64_t ldD, const float* __restrict__ zeros, int32_t* __restrict__ stats, int64_t c_begin,
    int64_t c_end, const unsigned bx, const unsigned by, const int d_f32) {
  // candidates [c_begin, c_end) of the N*G; block = GS candidate groups x KS contraction slices (one wave each); the KS waves of a group are reduced through LDS
  __shared__ double red[GS][KS][MT * NT][4][64];  // [group][slice][tile][acc reg][lane]

  const int64_t C = c_end;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int w = wave % KS, gs = wave / KS;
  const int row = lane & 15, kq = lane >> 4;
  const int64_t c0 = c_begin + ((int64_t)bx * GS + gs) * (16 * MT);
  const int q0 = by * (NT * 16);

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
  const int KQ = NTAPS * F;
  const float* brow[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    int q = q0 + nt * 16 + row;
    if (q >= Q) q = Q - 1;
    brow[nt] = q32 + (int64_t)q * KQ + 8 * kq;
  }

  f64x4 sum[MT][NT];
  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      sum[mt][nt] = (f64x4){0.0, 0.0, 0.0, 0.0};
      acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

  struct BufA {
    f32x4 a[MT][2];
  };
  struct BufB {
    f32x4 b[NT][2];
  };
  auto loadA = [&](BufA& u, int e0, int tap) {
    if (QPG_MX_PROBE & 1) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) u.a[mt][0] = u.a[mt][1] = (f32x4){1.f, 1.f, 1.f, 1.f};
      return;
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const bool ok = at0[mt] + tap * tap_stride < T;          // padded taps read the zero page (see the f64 kernel)
      const int64_t o = aoff[mt] + (int64_t)tap * tap_stride * F + e0;
      if (HALF) {      // f16 base: one 16-byte load brings the lane's 8 features
        const f16x8 h = *reinterpret_cast<const f16x8*>(ok ? reinterpret_cast<const void*>(reinterpret_cast<const _Float16*>(base) + o)
                                                           : reinterpret_cast<const void*>(zeros));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          u.a[mt][0][i] = (float)h[i];
          u.a[mt][1][i] = (float)h[4 + i];
        }
        continue;
      }
      const float* p = ok ? base + o : zeros;
      u.a[mt][0] = *reinterpret_cast<const f32x4*>(p);
      u.a[mt][1] = *reinterpret_cast<const f32x4*>(p + 4);
    }
  };
  auto loadB = [&](BufB& u, int e0, int tap) {
    if (QPG_MX_PROBE & 2) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) u.b[nt][0] = u.b[nt][1] = (f32x4){1.f, 1.f, 1.f, 1.f};
      return;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const float* p = brow[nt] + tap * F + e0;
      u.b[nt][0] = *reinterpret_cast<const f32x4*>(p);
      u.b[nt][1] = *reinterpret_cast<const f32x4*>(p + 4);
    }
  };
  // one stage: the finished 32-product chain of every tile goes into its f64 sum, then a fresh chain starts from C = 0
  auto mma = [&](const BufA& ua, const BufB& ub) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            f32x4 cin = acc[mt][nt];
            if (h == 0 && i == 0 && !(QPG_MX_PROBE & 4)) {
#pragma unroll
              for (int r = 0; r < 4; ++r) sum[mt][nt][r] += (double)cin[r];
              cin = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ua.a[mt][h][i], ub.b[nt][h][i], cin, 0, 0, 0);
          }
  };

  static_assert(NTAPS == 6, "tap schedule below is written for 6 taps");
  static_assert(AD >= 2 && AD <= 4 && BD >= 2 && BD <= 3, "ring depths");
  constexpr int L = (AD == 4) ? 12 : 6;                 // stages per unrolled super-iteration (multiple of AD, BD, 6)
  const int eBeg = w * (F / KS);
  const int ne = (F / KS) / 32;                         // feature groups of 32 per wave; stage s = (group s/6, tap s%6)
  const int nst = NTAPS * ne;
  // blocks start their walk over the feature groups at different offsets: all blocks read the SAME query rows, and in
  // lockstep they would all hit the same few L2 channels at any moment (only the order of the f64 additions changes)
  const int rot = (int)(bx % (unsigned)ne);
  auto eof = [&](int g) {                               // g in [0, 2*ne)
    g += rot;
    g = g >= ne ? g - ne : g;
    g = g >= ne ? g - ne : g;
    return eBeg + 32 * g;
  };
  BufA ra[AD];
  BufB rb[BD];
#pragma unroll
  for (int d = 0; d < BD - 1; ++d) loadB(rb[d], eof((d / 6) % ne), d % 6);
#pragma unroll
  for (int d = 0; d < AD - 1; ++d) loadA(ra[d], eof((d / 6) % ne), d % 6);
#ifndef QPG_MX_MPL
#define QPG_MX_MPL 2
#endif
  for (int s0 = 0; s0 < nst; s0 += L) {
    const int g0 = s0 / 6;
    auto stage = [&](auto jc) {
      constexpr int j = decltype(jc)::value;
      constexpr int ja = j + AD - 1, jb = j + BD - 1;
      int ga = g0 + ja / 6, gb = g0 + jb / 6;           // prefetches past the end wrap to a valid address, unused
      ga = ga >= ne ? ga - ne : ga;
      ga = ga >= ne ? 0 : ga;
      gb = gb >= ne ? gb - ne : gb;
      gb = gb >= ne ? 0 : gb;
      // query loads first: the wait for them at the next stage must not also wait for the (younger, slower) HBM loads
      loadB(rb[jb % BD], eof(gb), jb % 6);
      loadA(ra[ja % AD], eof(ga), ja % 6);
      mma(ra[j % AD], rb[j % BD]);
#pragma unroll
      for (int sg = 0; sg < (MT + NT) * 2; ++sg) {
        __builtin_amdgcn_sched_group_barrier(0x008, QPG_MX_MPL, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 8 * MT * NT - QPG_MX_MPL * 2 * (MT + NT), 0);
      __builtin_amdgcn_sched_barrier(0);
    };
    stage(MxIC<0>{}); stage(MxIC<1>{}); stage(MxIC<2>{}); stage(MxIC<3>{}); stage(MxIC<4>{}); stage(MxIC<5>{});
    if constexpr (L == 12) {
      stage(MxIC<6>{}); stage(MxIC<7>{}); stage(MxIC<8>{}); stage(MxIC<9>{}); stage(MxIC<10>{}); stage(MxIC<11>{});
    }
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        sum[mt][nt][r] += (double)acc[mt][nt][r];
        red[gs][w][mt * NT + nt][r][lane] = sum[mt][nt][r];
      }
  __syncthreads();

  // C/D layout of v_mfma_f32_16x16x4_f32: lane l, reg r holds (cand row = 4*(l>>4) + r, query col = l&15)
  constexpr int CR = 16 * MT;
  for (int o = threadIdx.x - gs * 64 * KS; o < NT * 16 * CR; o += 64 * KS) {
    const int ql = o / CR, cr = o - ql * CR;
    const int nt = ql >> 4, qc = ql & 15;
    const int mt = cr >> 4, crr = cr & 15;
    const int r = crr & 3, l = ((crr >> 2) << 4) | qc;
    const int t = mt * NT + nt;
    double dot = red[gs][0][t][r][l];
#pragma unroll
    for (int k = 1; k < KS; ++k) dot += red[gs][k][t][r][l];
    const int q = q0 + ql;
    const int64_t cc = c0 + cr;
    if (q < Q && cc < C) {
      const double a = qn2[q], b = cn2[cc];
      const double p = a * b;
      if (p > 0.0 && p < 1e-32 && stats) atomicOr(&stats[1], 2);     // |q||c| < 1e-16: f32 products may underflow
      const double dd = cosine_from_dot(dot, a, b);
      if (d_f32) reinterpret_cast<float*>(D)[(int64_t)q * ldD + cc] = (float)dd;
      else D[(int64_t)q * ldD + cc] = dd;
    }
  }
}

template <int MT, int NT, int NTAPS, int KS, int GS, int AD, int BD, bool HALF>
__global__ __launch_bounds__(64 * KS * GS, QPG_MX_OCC) void audio_cosine_mx_kernel(
    const float* __restrict__ base, int N, int T, int F, const int32_t* __restrict__ cand_t, int G, int tap_stride,
    const double* __restrict__ cn2, const float* __restrict__ q32, const double* __restrict__ qn2, int Q,
    double* __restrict__ D, int64_t ldD, const float* __restrict__ zeros, int32_t* __restrict__ stats, int64_t c_begin,
    int64_t c_end, int d_f32) {
  mx_ksplit_body<MT, NT, NTAPS, KS, GS, AD, BD, HALF>(base, N, T, F, cand_t, G, tap_stride, cn2, q32, qn2, Q, D, ldD, zeros, stats,
                                                c_begin, c_end, blockIdx.x, blockIdx.y, d_f32);
}

// ---------------------------------------------------------------------------------------------------------------
// mx2: the mixed-precision sweep with the QUERY tile shared through LDS.  Measured on MI355X (experiments/audio_mx):
// the split-K organisation above issues 40 row-scattered 16-B/lane loads per CU per 1536 matrix-pipe cycles and is
// bound by the texture-addresser, not by the matrix cores (47 % busy; removing the loads: 262 us instead of 465).
// Here the 4 waves of a block take 16 candidates each and ALL of K; the (48 queries x 64 features) tile of a stage is
// brought in ONCE per block by LDS-DMA (12 coalesced 1-KB pieces, 3 per wave) and read back as MFMA fragments with
// ds_read_b128 (XOR-swizzled at the source so every 16-lane group covers the 16 slots of a bank row); only the
// candidate rows (4 loads per wave-stage) still go global -> VGPR.  One barrier per stage of 48 MFMAs, two LDS
// buffers.  f32 chains are still cut every 32 products (two flushes per stage), so QPG_AUDIO_MX_ERR holds.
// ---------------------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* mx_lds_ptr_t;
typedef __attribute__((address_space(1))) const void* mx_gbl_ptr_t;
#ifndef QPG_MX2_OCC
#define QPG_MX2_OCC 2
#endif
#ifndef QPG_MX2_PROBE
#define QPG_MX2_PROBE 0    // experiments/audio_mx timing ablations (results wrong): 1 no barrier, 2 no candidate loads,
#endif                     // 4 no LDS-DMA, 8 no f64 flush, 16 no fragment reads
// row -> XOR applied to the 16-byte slot index of its 256-byte LDS row.  A ds_read_b128 lane group is 16 rows, eight
// of them ({0-3,12-15} or {4-11}) at k-quarter kq and the other eight at kq^1, all reading sub-piece 4*i + kq: with
// slot = piece ^ row both eights land on disjoint slot sets ({4..11} is closed under ^1), i.e. conflict-free.
__device__ __forceinline__ int mx2_g(int r) { return r; }

// F64 = true: the SAME organisation on the f64 matrix cores (v_mfma_f64_16x16x4_f64, operands widened in registers, no
// chains to cut): the f64 sweep of qpg_audio_cosine_f64 for whole rounds of blocks.
template <int NT, int NTAPS, bool F64, bool HALF>
__global__ __launch_bounds__(256, QPG_MX2_OCC) void audio_cosine_mx2_kernel(
    const float* __restrict__ base, int N, int T, int F, const int32_t* __restrict__ cand_t, int G, int tap_stride,
    const double* __restrict__ cn2, const float* __restrict__ q32, const double* __restrict__ qn2, int Q,
    double* __restrict__ D, int64_t ldD, const float* __restrict__ zeros, int32_t* __restrict__ stats, int64_t c_begin,
    int64_t c_end, int main_blocks, int64_t c_tail_end, int d_f32) {
  // blocks past `main_blocks` are the split-K remainder (one 16-candidate tile each, candidates from c_end on): they
  // ride in the same launch so that they fill the CUs while the last round of 64-candidate blocks drains
  if (!F64 && blockIdx.x >= (unsigned)main_blocks) {
    mx_ksplit_body<1, NT, NTAPS, 4, 1, 2, 2, HALF>(base, N, T, F, cand_t, G, tap_stride, cn2, q32, qn2, Q, D, ldD, zeros,
                                                   stats, c_end, c_tail_end, blockIdx.x - (unsigned)main_blocks, blockIdx.y, d_f32);
    return;
  }
  constexpr int ROWB = 256;                       // bytes of one query row per stage (64 features)
  constexpr int STAGE_BYTES = NT * 16 * ROWB;     // 12 KB at NT = 3
  constexpr int PIECES = STAGE_BYTES / 1024;      // 1-KB DMA pieces (4 rows each)
  __shared__ __attribute__((aligned(1024))) unsigned char ring[2][STAGE_BYTES];

  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int row = lane & 15, kq = lane >> 4;
  const int64_t c0 = c_begin + ((int64_t)blockIdx.x * 4 + w) * 16;
  const int q0 = blockIdx.y * (NT * 16);
  const int KQ = NTAPS * F;

  // candidate row of this lane (clamped; stores are masked)
  int64_t c = c0 + row;
  if (c >= c_end) c = c_end - 1;
  const int j = (int)(c / G), g = (int)(c - (int64_t)j * G);
  const int at0 = cand_t[g];
  // feature of (load i, k-quarter kq, element e) within a stage = 16*i + 4*kq + e: the four lanes of a row cover 64
  // contiguous bytes per load instruction (16 half lines per instruction; 16*kq + 4*i would touch 32)
  // f16 base (HALF): the lane's 16 features of a stage are two contiguous runs of 8 (one 16-byte load each):
  // feature of (i, kq, e) = 32*(i>>1) + 8*kq + 4*(i&1) + e, and the query fragments follow that order (boff below)
  const int64_t arow_off = ((int64_t)j * T + at0) * F + (HALF ? 8 : 4) * kq;
  const float* arow = base + arow_off;
  const _Float16* arow_h = reinterpret_cast<const _Float16*>(base) + arow_off;
  // DMA source rows of this lane: piece pi covers tile rows 4*pi .. 4*pi+3, lane l -> row 4*pi + (l>>4), slot l&15
  const float* dsrc[PIECES / 4];
#pragma unroll
  for (int i = 0; i < PIECES / 4; ++i) {
    const int pi = w + 4 * i;
    const int R = 4 * pi + (lane >> 4);
    int q = q0 + R;
    if (q >= Q) q = Q - 1;
    const int p = (lane & 15) ^ mx2_g(R & 15);
    dsrc[i] = q32 + (int64_t)q * KQ + 4 * p;
  }
  // fragment read offsets inside a stage buffer: tile nt, k-quarter i -> 16 bytes
  int boff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
    boff[i] = row * ROWB + (((HALF ? 8 * (i >> 1) + 2 * kq + (i & 1) : 4 * i + kq) ^ mx2_g(row)) << 4);

  f64x4 sum[NT];
  f32x4 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    sum[nt] = (f64x4){0.0, 0.0, 0.0, 0.0};      // F64: the accumulator itself
    acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const int ng = F / 64;                           // feature groups of 64; stage s = (group s / 6, tap s % 6)
  const int nst = NTAPS * ng;
  // The DMA is issued from inline asm so that hipcc's wait-count model does not see it: with an LDS-DMA in the
  // queue the compiler waits vmcnt(0) at the first use of ANY ordinary load, which would drain the candidate
  // prefetches every stage.  Untracked entries only make its counted waits stricter (in-order completion), never
  // looser; the DMA's own completion is awaited explicitly before the stage barrier.
  const unsigned ring_lds = (unsigned)(size_t)(mx_lds_ptr_t)(&ring[0][0]);
  const unsigned w_u = (unsigned)__builtin_amdgcn_readfirstlane(w);
  auto issue_b = [&](int s, int slot) {
    const int tap = s % NTAPS, e0 = 64 * (s / NTAPS);
#pragma unroll
    for (int i = 0; i < PIECES / 4; ++i) {
      const float* gp = dsrc[i] + tap * F + e0;
      const unsigned la = ring_lds + (unsigned)slot * STAGE_BYTES + (w_u + 4 * i) * 1024;
      // m0 is the LDS base of the DMA.  Naming it as a clobber makes hipcc warn that it does not preserve reserved
      // registers across the statement - which matters only if the compiler itself keeps something in m0 around here.
      // It does not: in this translation unit m0 occurs nowhere but in these statements (on gfx9+ ordinary LDS, global
      // and MFMA instructions do not read m0), and tests/test_host_cpu.py::test_audio_object_uses_m0_only_in_the_dma_asm
      // holds the built object to that.  (The builtin form hides nothing from the wait-count model: see above.)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gp), "s"(la) : "memory", "m0");
#pragma clang diagnostic pop
    }
  };
  auto load_a = [&](f32x4 (&a)[4], int s) {
    const int tap = s % NTAPS, e0 = 64 * (s / NTAPS);
    const bool ok = at0 + tap * tap_stride < T;
    if (HALF) {
      const _Float16* ph = ok ? arow_h + (int64_t)tap * tap_stride * F + e0 : reinterpret_cast<const _Float16*>(zeros);
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const f16x8 h = *reinterpret_cast<const f16x8*>(ph + (ok ? 32 * jj : 0));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          a[2 * jj][e] = (float)h[e];
          a[2 * jj + 1][e] = (float)h[4 + e];
        }
      }
      return;
    }
    const float* p = ok ? arow + (int64_t)tap * tap_stride * F + e0 : zeros;
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const f32x4*>(p + 16 * i);
  };
  // candidate fragments: register ring of depth 3 (prefetched two stages = 96 MFMAs ahead: the rows come from HBM);
  // query tile: LDS double buffer, one stage ahead (L2-resident)
  f32x4 ra[3][4];
  issue_b(0, 0);
  load_a(ra[0], 0);
  load_a(ra[1], 1 % nst);
  for (int s0 = 0; s0 < nst; s0 += 3) {            // unrolled by the ring depth only (taps stay run-time: fewer live addresses)
   auto stage = [&](auto jc) {
    constexpr int jj = decltype(jc)::value;
    const int s = s0 + jj;
    f32x4 (&a)[4] = ra[jj % 3];
    if (!(QPG_MX2_PROBE & 1)) {
      // queue, oldest first: A(s) | DMA(s) x3 | A(s+1) x4 (x2 from an f16 base): the tile of this stage has landed
      // once no more than the A(s+1) loads are outstanding
      if (HALF) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      __builtin_amdgcn_s_barrier();                // stage s landed for every wave; every wave is done with stage s-1
    }
    __builtin_amdgcn_sched_barrier(0);
    const int sn = s + 1 < nst ? s + 1 : 0;        // (the last prefetches wrap to valid addresses, unused)
    const int sa = s + 2 < nst ? s + 2 : s + 2 - nst;
    if (!(QPG_MX2_PROBE & 4)) issue_b(sn, (s + 1) & 1);
    if (!(QPG_MX2_PROBE & 2)) load_a(ra[(jj + 2) % 3], sa);
    __builtin_amdgcn_sched_barrier(0);             // the prefetches stay in front of the stage's matrix work
    const unsigned char* S = &ring[s & 1][0];
    f32x4 b0[NT][2], b1[NT][2];
    auto read_b = [&](f32x4 (&b)[NT][2], int half) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          if (QPG_MX2_PROBE & 16) {
            b[nt][i] = (f32x4){1.f + nt, 2.f + i, 3.f, 4.f};
            continue;
          }
          b[nt][i] = *reinterpret_cast<const f32x4*>(S + nt * 16 * ROWB + boff[2 * half + i]);
        }
    };
    auto mma_half = [&](const f32x4 (&b)[NT][2], int half) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            if (F64) {
              sum[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)a[2 * half + i][e], (double)b[nt][i][e], sum[nt],
                                                             0, 0, 0);
              continue;
            }
            f32x4 cin = acc[nt];
            if (i == 0 && e == 0 && !(QPG_MX2_PROBE & 8)) {   // a 32-product chain is complete: into the f64 sum
#pragma unroll
              for (int r = 0; r < 4; ++r) sum[nt][r] += (double)cin[r];
              cin = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2 * half + i][e], b[nt][i][e], cin, 0, 0, 0);
          }
    };
    read_b(b0, 0);
    __builtin_amdgcn_sched_barrier(0);
    read_b(b1, 1);                                  // the second half's fragments come in under the first half's MFMAs
    mma_half(b0, 0);
#pragma unroll
    for (int sg = 0; sg < NT * 2; ++sg) {
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    mma_half(b1, 1);
#pragma unroll
    for (int sg = 0; sg < 8; ++sg) __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);
    __builtin_amdgcn_sched_barrier(0);
   };
   stage(MxIC<0>{}); stage(MxIC<1>{}); stage(MxIC<2>{});
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the wrapped prefetches
  // C/D layout of v_mfma_f32_16x16x4_f32: lane l, reg r holds (cand row = 4*(l>>4) + r, query col = l&15)
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int q = q0 + nt * 16 + row;
    if (q >= Q) continue;
    const double a2 = qn2[q];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      // C/D rows: f32 instruction 4*(l>>4) + r, f64 instruction (l>>4) + 4*r
      const int64_t cc = c0 + (F64 ? kq + 4 * r : 4 * kq + r);
      if (cc >= c_end) continue;
      const double dot = F64 ? sum[nt][r] : sum[nt][r] + (double)acc[nt][r];
      const double b2 = cn2[cc];
      const double p = a2 * b2;
      if (p > 0.0 && p < 1e-32 && stats) atomicOr(&stats[1], 2);
      const double dd = cosine_from_dot(dot, a2, b2);
      if (d_f32) reinterpret_cast<float*>(D)[(int64_t)q * ldD + cc] = (float)dd;
      else D[(int64_t)q * ldD + cc] = dd;
    }
  }
}

#ifndef QPG_MX_AD
#define QPG_MX_AD 2
#endif
#ifndef QPG_MX_BD
#define QPG_MX_BD 2
#endif
#ifndef QPG_MX_KS
#define QPG_MX_KS 4      // contraction slices (waves) per candidate group
#endif
#ifndef QPG_MX_GS
#define QPG_MX_GS 1      // candidate groups per block
#endif
template <int MT, int NT>
static int launch_audio_mx(qpg_ctx* ctx, void* stream, const float* base, bool half, int N, int T, int F,
                           const int32_t* cand_t, int G, int tap_stride, const double* cn2, const float* q32,
                           const double* qn2, int Q, int qtiles_y, double* D, int64_t ldD, int32_t* stats, int64_t c_begin,
                           int64_t c_end, int d_f32) {
  constexpr int KS = QPG_MX_KS, GS = QPG_MX_GS;
  dim3 grid((unsigned)((c_end - c_begin + 16 * MT * GS - 1) / (16 * MT * GS)), (unsigned)qtiles_y);
  constexpr int AD = (QPG_MX_AD == 4) ? 3 : QPG_MX_AD;     // (a depth-4 ring needs an even feature-group count)
  if (half)
    hipLaunchKernelGGL((audio_cosine_mx_kernel<MT, NT, 6, KS, GS, AD, QPG_MX_BD, true>), grid, dim3(64 * KS * GS), 0,
                       qpg_stream(stream), base, N, T, F, cand_t, G, tap_stride, cn2, q32, qn2, Q, D, ldD,
                       (const float*)ctx->zeros, stats, c_begin, c_end, d_f32);
  else
    hipLaunchKernelGGL((audio_cosine_mx_kernel<MT, NT, 6, KS, GS, AD, QPG_MX_BD, false>), grid, dim3(64 * KS * GS), 0,
                       qpg_stream(stream), base, N, T, F, cand_t, G, tap_stride, cn2, q32, qn2, Q, D, ldD,
                       (const float*)ctx->zeros, stats, c_begin, c_end, d_f32);
  QPG_LAUNCH_CHECK("audio_cosine_mx_kernel");
  return QPG_OK;
}

template <int MT>
static int launch_audio_mx_q(qpg_ctx* ctx, void* stream, const float* base, bool half, int N, int T, int F,
                             const int32_t* cand_t, int G, int tap_stride, const double* cn2, const float* q32,
                             const double* qn2, int Q, double* D, int64_t ldD, int32_t* stats, int64_t c_begin,
                             int64_t c_end, int d_f32) {
  const int qt = (Q + 15) / 16;  // widest query tile that divides the work without an empty tail (48 queries = 3)
#define QPG_MX_ARGS ctx, stream, base, half, N, T, F, cand_t, G, tap_stride, cn2, q32, qn2, Q
  if (qt % 3 == 0) return launch_audio_mx<MT, 3>(QPG_MX_ARGS, qt / 3, D, ldD, stats, c_begin, c_end, d_f32);
  if (qt % 4 == 0) return launch_audio_mx<MT, 4>(QPG_MX_ARGS, qt / 4, D, ldD, stats, c_begin, c_end, d_f32);
  if (qt % 2 == 0) return launch_audio_mx<MT, 2>(QPG_MX_ARGS, qt / 2, D, ldD, stats, c_begin, c_end, d_f32);
  return launch_audio_mx<MT, 1>(QPG_MX_ARGS, qt, D, ldD, stats, c_begin, c_end, d_f32);
#undef QPG_MX_ARGS
}

#ifndef QPG_MX_ORG
#define QPG_MX_ORG 2       // 2: query tile through LDS (mx2) + split-K remainder; 1: split-K only (experiments)
#endif
#ifndef QPG_MX_TAIL_RIDES
#define QPG_MX_TAIL_RIDES 1   // 1: the split-K remainder blocks are appended to the mx2 launch; 0: a launch of their own
#endif
static int audio_cosine_mx(const char* name, qpg_ctx* ctx, void* stream, const float* base, bool half, int N, int T, int F,
                           const int32_t* cand_t, int G, int n_taps, int tap_stride, const double* cn2, const float* q32,
                           const double* qn2, int Q, void* D_, int64_t ldD, int32_t* stats, int d_f32) {
  double* D = static_cast<double*>(D_);          // (an f32 matrix when d_f32: the kernels cast at the store)
  QPG_REQUIRE(ctx && base && cand_t && cn2 && q32 && qn2 && D, "%s: null pointer", name);
  QPG_REQUIRE(N >= 0 && T > 0 && G > 0 && Q >= 0 && tap_stride > 0 && ldD >= (int64_t)N * G, "%s: bad size", name);
  if (n_taps != 6 || F <= 0 || (F % 128) != 0) {
    qpg_set_error("%s: compiled for n_taps=6 and F %% 128 == 0 (got n_taps=%d F=%d)", name, n_taps, F);
    return QPG_EUNSUP;
  }
  if (N == 0 || Q == 0) return QPG_OK;
  const int64_t C = (int64_t)N * G;
  // Work split.  mx2 blocks take 64 candidates x 48 queries; they are all resident at once (5 fit per CU), so the
  // launch is balanced when every CU holds the same number of them: whole multiples of n_cu blocks go to mx2, the
  // remaining candidates to split-K blocks of ONE 16-candidate tile (4 waves share it), which balance to a tile per CU.
  const int ny = (Q + 47) / 48;
  const int64_t nbx = C / 64;
  int64_t main_x = 0;
  if (QPG_MX_ORG == 2) {
    if (ny >= 4) main_x = nbx;
    else main_x = (nbx * ny / ctx->n_cu) * ctx->n_cu / ny;
  }
  const int64_t c_mid = main_x * 64;
  const int64_t tail_tiles = (C - c_mid + 15) / 16;
  // a short remainder rides in the SAME launch (blocks main_x .. main_x + tail_tiles - 1: split-K, one tile each)
  const bool ride = main_x > 0 && tail_tiles > 0 && tail_tiles <= 4 * (int64_t)ctx->n_cu && ny == 1 && QPG_MX_TAIL_RIDES;
  if (main_x > 0) {
    dim3 grid((unsigned)(main_x + (ride ? tail_tiles : 0)), (unsigned)ny);
    if (half)
      hipLaunchKernelGGL((audio_cosine_mx2_kernel<3, 6, false, true>), grid, dim3(256), 0, qpg_stream(stream), base, N, T,
                         F, cand_t, G, tap_stride, cn2, q32, qn2, Q, D, ldD, (const float*)ctx->zeros, stats, (int64_t)0,
                         c_mid, (int)main_x, C, d_f32);
    else
      hipLaunchKernelGGL((audio_cosine_mx2_kernel<3, 6, false, false>), grid, dim3(256), 0, qpg_stream(stream), base, N, T,
                         F, cand_t, G, tap_stride, cn2, q32, qn2, Q, D, ldD, (const float*)ctx->zeros, stats, (int64_t)0,
                         c_mid, (int)main_x, C, d_f32);
    QPG_LAUNCH_CHECK("audio_cosine_mx2_kernel");
  }
  if (c_mid == C || ride) return QPG_OK;
  if (QPG_MX_ORG == 2 && tail_tiles * ny <= 4 * (int64_t)ctx->n_cu)
    return launch_audio_mx_q<1>(ctx, stream, base, half, N, T, F, cand_t, G, tap_stride, cn2, q32, qn2, Q, D, ldD, stats,
                                c_mid, C, d_f32);
  return launch_audio_mx_q<2>(ctx, stream, base, half, N, T, F, cand_t, G, tap_stride, cn2, q32, qn2, Q, D, ldD, stats, c_mid,
                              C, d_f32);
}

extern "C" int qpg_audio_cosine_mx(qpg_ctx* ctx, void* stream, const float* base, int N, int T, int F,
                                   const int32_t* cand_t, int G, int n_taps, int tap_stride, const double* cn2,
                                   const float* q32, const double* qn2, int Q, void* D, int d_is_f32, int64_t ldD,
                                   int32_t* stats) {
  return audio_cosine_mx("qpg_audio_cosine_mx", ctx, stream, base, false, N, T, F, cand_t, G, n_taps, tap_stride, cn2, q32,
                         qn2, Q, D, ldD, stats, d_is_f32);
}

extern "C" int qpg_audio_cosine_mx_h(qpg_ctx* ctx, void* stream, const void* base_f16, int N, int T, int F,
                                     const int32_t* cand_t, int G, int n_taps, int tap_stride, const double* cn2,
                                     const float* q32, const double* qn2, int Q, void* D, int d_is_f32, int64_t ldD,
                                     int32_t* stats) {
  QPG_REQUIRE((reinterpret_cast<uintptr_t>(base_f16) % 16) == 0, "qpg_audio_cosine_mx_h: base must be 16-byte aligned");
  return audio_cosine_mx("qpg_audio_cosine_mx_h", ctx, stream, static_cast<const float*>(base_f16), true, N, T, F, cand_t, G,
                         n_taps, tap_stride, cn2, q32, qn2, Q, D, ldD, stats, d_is_f32);
}

template <int MT, int NT>
static int launch_audio(qpg_ctx* ctx, void* stream, const void* base, bool half, int N, int T, int F,
                        const int32_t* cand_t, int G, int tap_stride, const double* cn2, const float* q32,
                        const double* qn2, int Q, int qtiles_y, double* D, int64_t ldD, int64_t c_begin, int64_t c_end) {
  dim3 grid((unsigned)((c_end - c_begin + 16 * MT - 1) / (16 * MT)), (unsigned)qtiles_y);
  // 4 waves (one per SIMD) split the feature axis.  An 8-wave split (finer work units, 6.5 instead of
  // 3.25 rounds of blocks at N_db=2048) measured slower on MI355X: 703 vs 629 us (r01 notes).
  if (half)
    hipLaunchKernelGGL((audio_cosine_f64_kernel<MT, NT, 6, 4, true>), grid, dim3(256), 0, qpg_stream(stream), base, N, T,
                       F, cand_t, G, tap_stride, cn2, q32, qn2, Q, D, ldD, (const float*)ctx->zeros, c_begin, c_end);
  else
    hipLaunchKernelGGL((audio_cosine_f64_kernel<MT, NT, 6, 4, false>), grid, dim3(256), 0, qpg_stream(stream), base, N, T,
                       F, cand_t, G, tap_stride, cn2, q32, qn2, Q, D, ldD, (const float*)ctx->zeros, c_begin, c_end);
  QPG_LAUNCH_CHECK("audio_cosine_f64_kernel");
  return QPG_OK;
}

template <int MT>
static int launch_audio_q(qpg_ctx* ctx, void* stream, const void* base, bool half, int N, int T, int F,
                          const int32_t* cand_t, int G, int tap_stride, const double* cn2, const float* q32,
                          const double* qn2, int Q, double* D, int64_t ldD, int64_t c_begin, int64_t c_end) {
  const int qt = (Q + 15) / 16;  // widest query tile that divides the work without an empty tail (48 queries = 3)
#define QPG_AUDIO_ARGS ctx, stream, base, half, N, T, F, cand_t, G, tap_stride, cn2, q32, qn2, Q
  if (qt % 3 == 0) return launch_audio<MT, 3>(QPG_AUDIO_ARGS, qt / 3, D, ldD, c_begin, c_end);
  if (qt % 4 == 0) return launch_audio<MT, 4>(QPG_AUDIO_ARGS, qt / 4, D, ldD, c_begin, c_end);
  if (qt % 2 == 0) return launch_audio<MT, 2>(QPG_AUDIO_ARGS, qt / 2, D, ldD, c_begin, c_end);
  return launch_audio<MT, 1>(QPG_AUDIO_ARGS, qt, D, ldD, c_begin, c_end);
#undef QPG_AUDIO_ARGS
}

// 1 (default): split-K only.  2: LDS-shared-query blocks (mx2<F64>) for whole rounds + split-K remainder — correct (the
// whole GPU suite passes with it) but NOT faster for f64: at 64 cycles per MFMA the split-K kernel is already at the
// matrix pipe's pace (MI355X: 594 vs 590 us at Q = 48, 0.83 vs 0.81 of the roof at Q = 768, 0.79 vs 0.76 ms per step).
#ifndef QPG_F64_ORG
#define QPG_F64_ORG 1
#endif
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
  const int64_t C = (int64_t)N * G;
  // same work split as qpg_audio_cosine_mx (f32 base only: the f16 base keeps the split-K kernel throughout)
  const int ny = (Q + 47) / 48;
  const int64_t nbx = C / 64;
  int64_t main_x = 0;
  if (QPG_F64_ORG == 2 && !half) main_x = ny >= 4 ? nbx : (nbx * ny / ctx->n_cu) * ctx->n_cu / ny;
  const int64_t c_mid = main_x * 64;
  if (main_x > 0) {
    dim3 grid((unsigned)main_x, (unsigned)ny);
    hipLaunchKernelGGL((audio_cosine_mx2_kernel<3, 6, true, false>), grid, dim3(256), 0, qpg_stream(stream),
                       static_cast<const float*>(base), N, T, F, cand_t, G, tap_stride, cn2, q32, qn2, Q, D, ldD,
                       (const float*)ctx->zeros, (int32_t*)nullptr, (int64_t)0, c_mid, (int)main_x, c_mid, 0);
    QPG_LAUNCH_CHECK("audio_cosine_mx2_kernel<f64>");
  }
  if (c_mid == C) return QPG_OK;
  const int64_t tail_tiles = (C - c_mid + 15) / 16;
  if (main_x > 0 && tail_tiles * ny <= 4 * (int64_t)ctx->n_cu)
    return launch_audio_q<1>(ctx, stream, base, half, N, T, F, cand_t, G, tap_stride, cn2, q32, qn2, Q, D, ldD, c_mid, C);
  return launch_audio_q<2>(ctx, stream, base, half, N, T, F, cand_t, G, tap_stride, cn2, q32, qn2, Q, D, ldD, c_mid, C);
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
