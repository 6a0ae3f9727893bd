// Audio (WavLM) candidate sweep, round 3: SPLIT-OPERAND f16 matrix cores, HBM-bound.
//
// Same contract as qpg_audio_cosine_mx (qpg_audio.hip): D[q][c] = cosine distance of query q and candidate c with an
// A-PRIORI error bound <= QPG_AUDIO_MX_ERR, consumed by qpg_percode_select_mixed_f64, which re-evaluates every
// comparison the bound leaves open (GestureKNN.py:666-691 is what the pair of them replaces).  What changes is where
// the time goes: the f32 matrix cores need 200 us for the 31.4 GFLOP of a 24 s clip against 2 048 windows, the HBM
// stream of the database 110 us.  This kernel moves the products to the f16 matrix cores (16x the f32 rate) without
// giving up the bound, and reads every database frame from HBM exactly ONCE.
//
// 1. Frame-major formulation.  Candidate g of a window is the six frames 6g + {0,2,..,10} (data_processing.py:264-268):
//    neighbouring candidates share three frames, so a candidate-major sweep touches every frame twice.  Group the frames
//    by threes instead: SUPER-ROW i = frames 6i, 6i+2, 6i+4 (3 x 1024 features, i = 0..26), and split every query into its
//    first / last three taps (two 3072-d columns, "lo" and "hi").  Then
//        dot(q, cand_g) = S[g][q_lo] + S[g+1][q_hi],        S = A (27 super-rows x 3072)  .  B (3072 x 2 Q)
//    — a plain GEMM over rows that do not overlap, plus one shifted add in the epilogue.
// 2. Split operands.  Every value is scaled by a power of two (one exponent for the database, one per query; exact)
//    so that the largest magnitude is in [2^14, 2^15), and stored as TWO f16 numbers: h = fl16(x), l = fl16((x - h) 2^11).
//    x - h is exact in f32 (h is x rounded to 11 bits), so x = h + 2^-11 l + delta with |delta| <= 2^-23 |x| + 2^-36.
//    A dot product becomes  sum h h'  +  2^-11 sum (h l' + l h')  (+ a dropped l l' term <= 2^-24 |x||y|): three
//    v_mfma_f32_16x16x32_f16 per 32 k-steps.  f16 x f16 products are exact in f32.
// 3. The accumulate is what limits an f16 matrix core's accuracy, so it is kept out of the bound: the h h' products run
//    in chains of TWO instructions (the first starts from C = 0, the second from the first's result: the two k-blocks of
//    an LDS stage) and every chain's 64-product sum is added to an f64 running sum on the VALU (underneath the next
//    MFMAs); only the cross terms, 2^-11 smaller, run as f32 chains over the whole contraction.
//    Error budget, relative to |q||c| (Cauchy-Schwarz, as for the f32 sweep):
//        h h' block sums      kappa 2^-24        |MFMA(A, B, 0) - exact| <= kappa 2^-24 sum |products|.  What the matrix
//                                                core does was probed on MI355X (tools/probe_mfma_f16.py): the 32
//                                                products are exact; each OCTET of k (one lane group's 8) is aligned to
//                                                its largest product and CHOPPED to 25 bits before it is summed (7 terms x
//                                                < 2^-24 of the octet's largest), the four octet sums and C meet in a wider
//                                                adder, one round-to-nearest-even at the end: kappa <= 7 + 1 + 0.5.
//                                                Worst seen in adversarial blocks 8.97.  A chain of two adds the
//                                                first result's rounding (<= 0.5 2^-24 of ITS sum; C then enters the
//                                                second instruction's final adder exactly): kappa_2 <= kappa + 0.5 in
//                                                units of 2^-24 sum |64 products|; worst seen 9.72
//                                                (tools/probe_mfma_chain.py).  ASSUMED 13 (the one measured constant
//                                                of this bound; tests/test_gpu_audio_hl.py re-measures both): 7.75e-7
//                                                (Until the end of round 3 every instruction was flushed: 12 -> 7.2e-7.)
//        cross-term chains    same model with C != 0 (C enters the final adder): error_n <= 12 2^-24 (sum_n |p| + |acc|),
//                             192 instructions per accumulator, |acc| <= P = sum |cross products| <= |q||c|:
//                             12 x 193 x 2^-24 P = 1.4e-4 P, times the 2^-11 of the cross terms: 0.7e-7
//        representation       2 x 2^-23 + 2^-24 = 3.0e-7 (needs scaled norms >= 1, else stats[1] |= 2: such operands
//                             are 2^-15 of the largest value in the database and the clip is re-matched)
//        f64 sums, scaling    < 1e-13;   f32-stored matrix 1.2e-7
//    total 1.27e-6 <= QPG_AUDIO_HL_ERR = 1.3e-6 (include/qpg.h): the select's band is 2.1 x that, 0.63 of what the
//    f32-matrix-core sweep needs - a third fewer re-evaluations.
// 4. Data layout.  The database image is written once (qpg_audio_hl_pack_db) in MFMA fragment order:
//    per window [tile 0: k-block 96][plane h|l][64 units][8 f16] then [tile 1: 96][h|l][44 units][8] — a wave's operand
//    load is ONE contiguous run (tile 1 holds its 11 live rows only).  680 MB at N = 2048: exactly the algorithmic bytes.  The
//    queries of a clip are packed the same way per chunk of 48 (1.18 MB, L2-resident), shared by a block's four waves
//    through LDS.  Two waves = one window (16-row tile x 96 columns x K = 3072 each): the shifted add of the epilogue is
//    lane shuffles plus one row handed over through LDS; no workspace, no second pass.
// Bound: HBM (0.70 GB per launch / 8 TB/s); MFMA time 46 us, VALU (f64 flush) 31 us at N = 2048.
#include "qpg_common.h"
#include <stdlib.h>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

#define HL_ROWS 27            // super-rows per window (Ga + 1)
#define HL_SUB 3              // frames per super-row
#define HL_QC 48              // queries per chunk
#define HL_CT 6               // column tiles per chunk (3 lo + 3 hi)
#define HL_GQC (16 * HL_CT)   // queries per chunk of the generic GEMM (96)
#ifndef HL_KS
#define HL_KS 2               // k-blocks per LDS stage
#endif
#define HL_PIECE 1024         // bytes of one [plane][lane][8 f16] fragment image

// value of lane (l ^ PJ), PJ = 16 / 32, through gfx950's permlane swaps (qpg_common.h: lane_xor) - the GEMM epilogues'
// reductions over a tile's four row groups went through the LDS crossbar (__shfl_xor = ds_bpermute_b32: ~65 cycles each,
// a wave's do not overlap; 96 of them per 64 x 96 item were a third of hl_gemm64h_kernel: experiments/round_scripts/r05_probe_gemm64.sh)
template <int PJ>
__device__ __forceinline__ float xor_lanes_f(float v) {
  return __int_as_float(lane_xor<PJ>(__float_as_int(v)));
}
__device__ __forceinline__ f32x4 mfma_h(h8 a, h8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// x (already scaled) -> (h, l): h = fl16(x), l = fl16((x - h) * 2^11)
__device__ __forceinline__ void split_hl(float x, _Float16& h, _Float16& l) {
  h = (_Float16)x;
  const float r = x - (float)h;                    // exact
  l = (_Float16)(r * 2048.0f);
}
// Round 4, the AUDIO images: l = fl16(x - h) at its TRUE scale.  The cross products h l' then have their real magnitude
// and can run through the SAME accumulator as h h' (audio_cosine_hl2_kernel); x - h is below half an ulp of h, so for
// the large values l stays a normal f16 number with the same 11 significant bits as before, and for scaled values below
// 2^-3 it falls into the subnormal range: an ABSOLUTE error <= 2^-25 per element, i.e. a dot-product error
// <= 2^-25 sum|y_i| <= 2^-25 sqrt(D) |y| - relative to |x||y| that is 2^-25 sqrt(6144) / |x|_scaled = 2.3e-6 / |x|_scaled.
// It is small only for operands whose SCALED norm is large, so the sweeps' validity guard demands scaled |x|^2 >=
// HL_NORM2_MIN = 2^16 (stats[1] |= 2 below it; round 4 asked for >= 1, under which a quiet row of a loud database could
// take the whole budget - ADVICE r4): <= 9.1e-9 per side, 1.8e-8 of the 1e-7 the budget leaves (1.2e-6 + 1.8e-8 <=
// QPG_AUDIO_HL_ERR).  A query always passes (its own largest element is scaled to >= 2^14); a database row fails when its
// norm is below 2^8 / 2^14..15 = 1/64 .. 1/128 of the largest magnitude in the whole track - such a clip is re-matched on
// the f64 path.  The matrix core takes f16 subnormals as they are (checked at load time: selfcheck.py).
#define HL_NORM2_MIN 65536.0
#define HL_AUDIO_LSHIFT 0
__device__ __forceinline__ void split_hl_audio(float x, _Float16& h, _Float16& l) {
  h = (_Float16)x;
  const float r = x - (float)h;                    // exact
  l = (_Float16)(HL_AUDIO_LSHIFT ? ldexpf(r, HL_AUDIO_LSHIFT) : r);
}

// ---- scale exponent of the database: max |x| -> e with max * 2^e in [2^14, 2^15) -------------------------------
__global__ __launch_bounds__(1024) void hl_absmax_kernel(const float* __restrict__ x, int64_t n, unsigned int* __restrict__ out) {
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    m = fmaxf(m, fabsf(x[i]));
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o, 64));
  __shared__ float red[16];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) m = fmaxf(m, red[i]);
    atomicMax(out, __float_as_uint(m));            // non-negative floats order like their bit patterns
  }
}

__device__ __forceinline__ int hl_exponent(float amax) {      // e: amax * 2^e in [2^14, 2^15); 0 for amax == 0 / inf / nan
  if (!(amax > 0.f) || amax > 3.0e38f) return 0;
  int e;
  frexpf(amax, &e);                                // amax = f * 2^e, f in [0.5, 1)
  e = 15 - e;
  return e > 100 ? 100 : (e < -100 ? -100 : e);    // (2^e must stay an f32 number; such data trips the norm check)
}

__global__ void hl_zero_u32_kernel(unsigned int* p) { *p = 0u; }

__global__ void hl_exponent_kernel(const unsigned int* __restrict__ amax_bits, int32_t* __restrict__ meta) {
  meta[0] = hl_exponent(__uint_as_float(amax_bits[0]));
}

// ---- database image ------------------------------------------------------------------------------------------------
// thread <-> (window j, super-row i < 27, k8 = k / 8): reads 8 consecutive features, writes one 16-byte h piece and one
// l piece.  DENSE image: a window is [tile 0: KB x 2 planes x 64 units][tile 1: KB x 2 planes x 44 units] of 16 bytes;
// tile 0 (rows 0..15): unit = lane = i + 16 * ((k % 32) / 8); tile 1 (rows 16..26, 11 live of 16): unit = 11 * ((k % 32)
// / 8) + (i - 16).  (The first layout kept 64 units for tile 1 too and never loaded rows 27..31 - but their 16-byte slots
// sat inside the 128-byte lines the live rows pulled in: PMC fetch 1.21 x the algorithmic bytes.)
#define HL_T1_UNITS 44        // 16-byte units of a tile-1 fragment: 11 live rows x 4 k-groups
#define HL_WIN_UNITS(KB) ((int64_t)(KB) * 2 * (64 + HL_T1_UNITS))
__global__ __launch_bounds__(256) void hl_pack_db_kernel(const float* __restrict__ base, int N, int T, int F, int step,
                                                         int tap_stride, const int32_t* __restrict__ meta,
                                                         _Float16* __restrict__ image) {
  const int KB = HL_SUB * F / 32, K8 = HL_SUB * F / 8;
  const int64_t n = (int64_t)N * HL_ROWS * K8;
  const float sc = ldexpf(1.0f, meta[0]);
  for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (int64_t)gridDim.x * blockDim.x) {
    const int k8 = (int)(id % K8);
    const int i = (int)((id / K8) % HL_ROWS);
    const int j = (int)(id / ((int64_t)K8 * HL_ROWS));
    const int k = k8 * 8, sub = k / F, f = k - sub * F;
    const int t = step * i + tap_stride * sub;
    f32x4 v0 = (f32x4){0.f, 0.f, 0.f, 0.f}, v1 = v0;
    if (i < HL_ROWS && t < T) {
      const f32x4* p = reinterpret_cast<const f32x4*>(base + ((int64_t)j * T + t) * F + f);
      v0 = p[0];
      v1 = p[1];
    }
    h8 hh, ll;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      _Float16 a0, b0, a1, b1;
      split_hl_audio(v0[e] * sc, a0, b0);
      split_hl_audio(v1[e] * sc, a1, b1);
      hh[e] = a0; ll[e] = b0; hh[4 + e] = a1; ll[4 + e] = b1;
    }
    const int kb = k / 32, kg = (k & 31) >> 3;
    const int64_t win = (int64_t)j * HL_WIN_UNITS(KB);
    int64_t u0;
    int pl;
    if (i < 16) {
      pl = 64;
      u0 = win + (int64_t)kb * 128 + i + 16 * kg;
    } else {
      pl = HL_T1_UNITS;
      u0 = win + (int64_t)KB * 128 + (int64_t)kb * 2 * HL_T1_UNITS + 11 * kg + (i - 16);
    }
    reinterpret_cast<h8*>(image)[u0] = hh;
    reinterpret_cast<h8*>(image)[u0 + pl] = ll;
  }
}

// ---- query image -----------------------------------------------------------------------------------------------------
// one block per query slot (slots >= Q are written as zeros: chunks are always 48 wide).
// image[((((chunk*KB + kb)*6 + ct)*2 + plane)*64 + lane][8], ct = half*3 + (q%48)/16, lane = q%16 + 16*((k%32)/8),
// element k of half `half` = q32[q][half*3F + k].  qexp[q] = the query's scale exponent.
__global__ __launch_bounds__(768) void hl_pack_queries_kernel(const float* __restrict__ q32, int Q, int F,
                                                              _Float16* __restrict__ image, int32_t* __restrict__ qexp) {
  const int q = blockIdx.x, tid = threadIdx.x;
  const int D = 2 * HL_SUB * F, KB = HL_SUB * F / 32, K8h = HL_SUB * F / 8;      // K8h 8-element groups per half
  __shared__ float red[12];
  __shared__ int e_s;
  float m = 0.f;
  const bool live = q < Q;
  const float* row = q32 + (int64_t)q * D;
  if (live)
    for (int i = tid; i < D / 4; i += blockDim.x) {
      const f32x4 v = reinterpret_cast<const f32x4*>(row)[i];
      m = fmaxf(fmaxf(fmaxf(m, fabsf(v.x)), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o, 64));
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  if (tid == 0) {
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) m = fmaxf(m, red[i]);
    e_s = hl_exponent(m);
    if (live) qexp[q] = e_s;
  }
  __syncthreads();
  const float sc = ldexpf(1.0f, e_s);
  const int chunk = q / HL_QC, qq = q % HL_QC;
  for (int id = tid; id < 2 * K8h; id += blockDim.x) {
    const int half = id / K8h, k8 = id - half * K8h, k = k8 * 8;
    f32x4 v0 = (f32x4){0.f, 0.f, 0.f, 0.f}, v1 = v0;
    if (live) {
      const f32x4* p = reinterpret_cast<const f32x4*>(row + (int64_t)half * HL_SUB * F + k);
      v0 = p[0];
      v1 = p[1];
    }
    h8 hh, ll;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      _Float16 a0, b0, a1, b1;
      split_hl_audio(v0[e] * sc, a0, b0);
      split_hl_audio(v1[e] * sc, a1, b1);
      hh[e] = a0; ll[e] = b0; hh[4 + e] = a1; ll[4 + e] = b1;
    }
    const int kb = k / 32, ct = half * 3 + qq / 16, lane = (qq & 15) + 16 * ((k & 31) >> 3);
    const int64_t piece = ((((int64_t)chunk * KB + kb) * HL_CT + ct) * 2);
    reinterpret_cast<h8*>(image)[(piece + 0) * 64 + lane] = hh;
    reinterpret_cast<h8*>(image)[(piece + 1) * 64 + lane] = ll;
  }
}

// ---- fused query pack: gather + squared norms + scale exponent + image, ONE launch in front of the sweep -------------
// (qpg_audio_pack_queries followed by qpg_audio_hl_pack_queries: two launches and the gap between them, 18 us on the
// critical path of a clip; the select's re-evaluations still need q32 / qn2, so both are written)
__device__ __forceinline__ void hl_pack_query_block(int q, int tid, const float* __restrict__ qbase, int M, int T, int F,
                                                    const int32_t* __restrict__ q_win, const int32_t* __restrict__ q_t,
                                                    int Q, int tap_stride, float* __restrict__ q32,
                                                    double* __restrict__ qn2, _Float16* __restrict__ image,
                                                    int32_t* __restrict__ qexp) {
  const int D = 2 * HL_SUB * F, KB = HL_SUB * F / 32, K8h = HL_SUB * F / 8, n8 = 2 * K8h;
  __shared__ float redm[12];
  __shared__ double reds[12];
  __shared__ int e_s;
  const bool live = q < Q;
  const int wq = live ? q_win[q] : 0, t0 = live ? q_t[q] : 0;
  constexpr int MAXU = 2;                                   // 8-element groups per thread (n8 <= 768 * MAXU)
  f32x4 v[MAXU][2];
  float m = 0.f;
  double s = 0.0;
#pragma unroll
  for (int u = 0; u < MAXU; ++u) {
    const int id = tid + u * 768;
    v[u][0] = v[u][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (live && id < n8) {
      const int e = id * 8, tap = e / F, f = e - tap * F;
      const int t = t0 + tap * tap_stride;
      if (t < T) {
        const f32x4* p = reinterpret_cast<const f32x4*>(qbase + ((int64_t)wq * T + t) * F + f);
        v[u][0] = p[0];
        v[u][1] = p[1];
      }
      reinterpret_cast<f32x4*>(q32 + (int64_t)q * D + e)[0] = v[u][0];
      reinterpret_cast<f32x4*>(q32 + (int64_t)q * D + e)[1] = v[u][1];
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          m = fmaxf(m, fabsf(v[u][h][c]));
          s += (double)v[u][h][c] * (double)v[u][h][c];
        }
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    m = fmaxf(m, __shfl_down(m, o, 64));
    s += __shfl_down(s, o, 64);
  }
  if ((tid & 63) == 0) {
    redm[tid >> 6] = m;
    reds[tid >> 6] = s;
  }
  __syncthreads();
  if (tid == 0) {
    for (int i = 1; i < 12; ++i) {
      m = fmaxf(m, redm[i]);
      s += reds[i];
    }
    e_s = hl_exponent(m);
    if (live) {
      qexp[q] = e_s;
      qn2[q] = s;
    }
  }
  __syncthreads();
  const float sc = ldexpf(1.0f, e_s);
  const int chunk = q / HL_QC, qq = q % HL_QC;
#pragma unroll
  for (int u = 0; u < MAXU; ++u) {
    const int id = tid + u * 768;
    if (id >= n8) continue;
    const int half = id / K8h, k8 = id - half * K8h, k = k8 * 8;
    h8 hh, ll;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      _Float16 a0, b0, a1, b1;
      split_hl_audio(v[u][0][e] * sc, a0, b0);
      split_hl_audio(v[u][1][e] * sc, a1, b1);
      hh[e] = a0; ll[e] = b0; hh[4 + e] = a1; ll[4 + e] = b1;
    }
    const int kb = k / 32, ct = half * 3 + qq / 16, lane = (qq & 15) + 16 * ((k & 31) >> 3);
    const int64_t piece = ((((int64_t)chunk * KB + kb) * HL_CT + ct) * 2);
    reinterpret_cast<h8*>(image)[(piece + 0) * 64 + lane] = hh;
    reinterpret_cast<h8*>(image)[(piece + 1) * 64 + lane] = ll;
  }
}


__global__ __launch_bounds__(768) void hl_pack_queries_fused_kernel(const float* __restrict__ qbase, int M, int T, int F,
                                                                    const int32_t* __restrict__ q_win,
                                                                    const int32_t* __restrict__ q_t, int Q, int tap_stride,
                                                                    float* __restrict__ q32, double* __restrict__ qn2,
                                                                    _Float16* __restrict__ image,
                                                                    int32_t* __restrict__ qexp) {
  hl_pack_query_block(blockIdx.x, threadIdx.x, qbase, M, T, F, q_win, q_t, Q, tap_stride, q32, qn2, image, qexp);
}

// ---- one launch for a clip's whole query side (round 4): the audio pack above AND the text side's query pack ------------------
// Blocks [0, n_aud) = hl_pack_query_block; blocks [n_aud, n_aud + text slots) = one text query each: gather
// clip_context[win][row] (GestureKNN.py:549-551), sklearn's f32 normalisation (bit-exact: the four einsum lane chains of
// qpg_core.hip's l2_normalize_rows_kernel, run by threads 0..3), the normalised row to qn (the select's exact evaluation
// reads it) and its split-f16 column image (hl_pack_cols_kernel's).  The text side used to start with two tiny launches
// of its own; behind a sweep that holds every register of every CU they did not get a wave slot before the sweep was over
// (the text pack sat 100 us in the queue: profiles/r04_step_timeline_graph*.md), and the whole text chain moved behind it.
struct ClipPackText {
  const float* ctx;        // [Mt][R][Dt]
  const int32_t* q_win;    // [Qt]
  const int32_t* q_row;    // [Qt]
  int R, Dt, Qt;
  float* qn;               // [Qt][Dt]
  _Float16* image;         // column image (qpg_hl_cols_bytes)
  int32_t* qexp;
};

__global__ __launch_bounds__(768) void hl_pack_clip_kernel(const float* __restrict__ qbase, int M, int T, int F,
                                                           const int32_t* __restrict__ q_win,
                                                           const int32_t* __restrict__ q_t, int Q, int tap_stride,
                                                           float* __restrict__ q32, double* __restrict__ qn2,
                                                           _Float16* __restrict__ image, int32_t* __restrict__ qexp,
                                                           int n_aud, ClipPackText tx) {
  const int tid = threadIdx.x;
  if ((int)blockIdx.x < n_aud) {
    hl_pack_query_block(blockIdx.x, tid, qbase, M, T, F, q_win, q_t, Q, tap_stride, q32, qn2, image, qexp);
    return;
  }
  const int qi = (int)blockIdx.x - n_aud;
  const int D = tx.Dt;
  extern __shared__ __attribute__((aligned(16))) float rowbuf[];         // [D] the normalised row
  __shared__ float n_s, red[12];
  __shared__ int e_s;
  const bool live = qi < tx.Qt;
  const float* p = live ? tx.ctx + ((int64_t)tx.q_win[qi] * tx.R + tx.q_row[qi]) * D : tx.ctx;
  if (tid < 4) {                                       // the norm, in NumPy einsum's order (lane chains l = 0..3)
    const int l = tid;
    float a = 0.f;
    const int nfull = D >> 4;
    int g = 0;
    for (; g + 8 <= nfull; g += 8) {
      float v[32];
#pragma unroll
      for (int jx = 0; jx < 32; ++jx) v[jx] = p[(g + (jx >> 2)) * 16 + (jx & 3) * 4 + l];
#pragma unroll
      for (int jx = 0; jx < 8; ++jx) {
#pragma unroll
        for (int u = 3; u >= 0; --u) a = f_add(f_mul(v[jx * 4 + u], v[jx * 4 + u]), a);
      }
    }
    for (; g < nfull; ++g) {
#pragma unroll
      for (int u = 3; u >= 0; --u) {
        const float v = p[g * 16 + u * 4 + l];
        a = f_add(f_mul(v, v), a);
      }
    }
    for (int i = nfull * 16; i < D; i += 4) {
      const float v = (i + l < D) ? p[i + l] : 0.f;
      a = f_add(f_mul(v, v), a);
    }
    const float o1 = __shfl_xor(a, 1, 64);
    const float pair = f_add(a, o1);
    const float o2 = __shfl_xor(pair, 2, 64);
    float n = f_sqrt(f_add(pair, o2));
    if (n < 10.f * 1.1920928955078125e-07f) n = 1.f;   // sklearn _handle_zeros_in_scale
    if (l == 0) n_s = n;
  }
  __syncthreads();
  const float n = n_s;
  float m = 0.f;
  for (int e = tid; e < D; e += 768) {
    const float v = live ? f_div(p[e], n) : 0.f;
    rowbuf[e] = v;
    if (live) tx.qn[(int64_t)qi * D + e] = v;
    m = fmaxf(m, fabsf(v));
  }
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o, 64));
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  if (tid == 0) {
    for (int i = 1; i < 12; ++i) m = fmaxf(m, red[i]);
    e_s = hl_exponent(m);
    if (live) tx.qexp[qi] = e_s;
  }
  __syncthreads();
  const float sc = ldexpf(1.0f, e_s);
  const int KB = D / 32, K8 = D / 8;
  const int chunk = qi / (16 * HL_CT), qq = qi % (16 * HL_CT);
  for (int k8 = tid; k8 < K8; k8 += 768) {
    const int k = k8 * 8;
    h8 hh, ll;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      _Float16 a0, b0;
      split_hl(rowbuf[k + e] * sc, a0, b0);
      hh[e] = a0;
      ll[e] = b0;
    }
    const int kb = k / 32, ct = qq / 16, lane = (qq & 15) + 16 * ((k & 31) >> 3);
    const int64_t piece = ((((int64_t)chunk * KB + kb) * HL_CT + ct) * 2);
    reinterpret_cast<h8*>(tx.image)[(piece + 0) * 64 + lane] = hh;
    reinterpret_cast<h8*>(tx.image)[(piece + 1) * 64 + lane] = ll;
  }
}

// ---- the sweep ---------------------------------------------------------------------------------------------------------
struct HlArgs {
  const _Float16* db;      // database image
  const _Float16* qi;      // query image (n_chunks x KB x 6 x 2 x 1 KB)
  const int32_t* meta;     // [0] database scale exponent
  const int32_t* qexp;     // [Q]
  const double* cn2;       // [N][G] squared norms of the candidates (unscaled)
  const double* qn2;       // [Q]
  void* D;                 // [Q][ldD] f32 (d_f32) or f64
  const float* zeros;      // the context's zero page (>= 16 bytes): what padding rows read
  int64_t ldD;
  int32_t* stats;
  int N, G, Q, KB, d_f32;   // N: database windows (MODE 0) / 32-row groups (MODE 1)
  int j0;                  // first window / 32-row group of this launch (N: one past the last)
  int chunks;              // query chunks of this launch
  float* tmin;             // MODE 1, optional: [Q][ldT] minimum of every 16-row tile (rows 16 i .. 16 i + 15)
  int64_t ldT;
  uint16_t* tmask;         // MODE 1, optional (with tmin): [Q][ldT] bit r = row 16 i + r lies within `band` of the tile's minimum
  float band;              // ... the matrix itself is then not needed (D may be NULL): the select opens tiles by their masks
};

#define HL_RING (2 * HL_KS)  // k-blocks of database fragments in flight per wave (2 KB each): two stages
#define HL_PS (3 * HL_KS)    // pair-steps per stage
// ablation hooks (experiments/audio_hl): -DQPG_HL_PROBE=<bits> compiles parts of the k loop out; the product build
// defines nothing.  1: no f64 flush; 2: query fragments read from LDS once; 4: database fragments not reloaded;
// 8: no cross-term MFMAs; 16: no query staging (loads, LDS stores, barriers) after the first stage
#ifndef QPG_HL_PROBE
#define QPG_HL_PROBE 0
#endif

// Block = HL_WPB database windows x 2 row tiles = 2 HL_WPB waves; wave w: window HL_WPB*blockIdx.x + (w >> 1), rows
// 16*(w & 1) .. +15, all 96 columns of the chunk (a 16 x 96 tile of S, K = 3072).  Two waves per SIMD, <= 256 registers
// each, the f64 sums in architectural VGPRs.  Organisations measured and dropped (experiments/audio_hl): 4 waves with
// 32 x 96 tiles (the f64 sums land in AGPRs and the flush drowns in v_accvgpr moves: 275 us); 8 waves with 32 x 48 tiles
// (half the LDS reads, twice the database-fragment loads: 234 us against 206).
//
// The k loop is ONE basic block, software-pipelined by hand (left to itself hipcc waits lgkmcnt(0) right behind every
// pair of fragment reads, drains the HBM ring with vmcnt(0) at each __syncthreads and puts dependent MFMAs back to back:
// 257 us):
//   pair-step = two column tiles of one k-block: 6 MFMAs (2 h h' block sums from C = 0, 2 x 2 cross terms, the two
//   MFMAs of one accumulator two instructions apart), the 4 fragment reads of the NEXT pair-step, and the f64 flush of
//   the PREVIOUS pair-step's block sums (16 VALU operations), interleaved by sched_group_barrier.
//   stage = HL_KS k-blocks; the query fragments of stage s+1 are stored to the other LDS buffer at the start of stage s,
//   and the ONE barrier of a stage (LDS-only: s_waitcnt lgkmcnt(0) + s_barrier, the HBM ring stays in flight) sits in
//   the last pair-step, behind the arrival of that pair-step's own fragments: behind it nobody reads this stage's
//   buffer any more and everybody's stores of the next stage are done, so the next stage's first fragments are
//   prefetched under the last pair-step's MFMAs - no bubble.
#ifndef HL_NT
#define HL_NT 0
#endif
#ifndef HL_PD
#define HL_PD 1            // pair-steps between a fragment read and its use (2 needs HL_PS %% 3 == 0 and 2 * HL_PS %% 3 == 0)
#endif
#ifndef HL_WPB
#define HL_WPB 4           // database windows per block (2 waves each)
#endif
#ifndef HL_MINW
#define HL_MINW 2          // waves per SIMD the register allocation must leave room for
#endif
#define HL_THREADS (128 * HL_WPB)
#define HL_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
// The 16-row organisation of round 3's audio sweep, kept as the plain distance GEMM over generic rows of a clip's <= 48
// text queries (qpg_hl_gemm_distance / _tilemin: groups of 32 rows; D[q][row] = 1 - <row, q> for unit-norm operands; the
// chunk's 96 columns are 96 queries).  The audio sweep itself runs on audio_cosine_hl2_kernel (32-row wave tiles) since
// round 4; its 16-row form left the library in round 5 (every grid it accepted, the 32-row kernel accepts).
__global__ __launch_bounds__(HL_THREADS, HL_MINW) void hl_gemm16_kernel(HlArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // 2 * HL_KS * HL_CT * 2 * HL_PIECE bytes
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wl = w >> 1, t = w & 1;                           // window of the block, row tile
  // Block -> (window group, query chunk), XCD-aware.  With several chunks (16 clips per sweep, W clips on a row shard,
  // cfg-3's 1 000 queries) every chunk needs the whole database image; launched chunk-major each chunk streamed it from
  // HBM again.  Workgroup b runs on XCD b % 8 (observed dispatch order), so linear block L takes window group
  // (L / 8 / chunks) * 8 + L % 8 and chunk (L / 8) % chunks: an XCD works through ALL chunks of a window group back to
  // back, and after the first of them the group's 1.3 MB of fragments come out of that XCD's L2.  One chunk: identity.
  const int xl = (int)blockIdx.x, slot = xl >> 3;
  const int wgrp = (slot / a.chunks) * 8 + (xl & 7);
  const int chunk = slot % a.chunks;
  if (a.j0 + wgrp * HL_WPB >= a.N) return;                    // (the grid is padded to whole groups of 8: nothing to do)
  const int j = a.j0 + wgrp * HL_WPB + wl;                     // (j0: first window of a partial launch)
  const int KB = a.KB, n_stage = KB / HL_KS;
  const bool win_ok = j < a.N;
  // row fragments: plane p of k-block kb: one 16-byte load per lane, 1 KB per wave, contiguous.  Row groups past N read
  // the context's zero page instead (stride 0): no branch in the loop.
  const bool row_ok = win_ok;
  const int64_t unit0 = (((int64_t)j * 2 + t) * KB * 2) * 64 + lane;
  const h8* dbp = row_ok ? reinterpret_cast<const h8*>(a.db) + unit0 : reinterpret_cast<const h8*>(a.zeros);
  const int pl_step = row_ok ? 64 : 0;                                   // h8 units per plane / k-block
  const int kb_step = 2 * pl_step;
  // (HL_NT: the database image is read ONCE - a non-temporal load keeps it from evicting the query image, which every
  // block re-reads, out of the XCD's L2)
  auto load_a = [&](int kb, h8 (&dst)[2]) {
#if HL_NT
    dst[0] = __builtin_nontemporal_load(dbp + (int64_t)kb * kb_step);
    dst[1] = __builtin_nontemporal_load(dbp + (int64_t)kb * kb_step + pl_step);
#else
    dst[0] = dbp[(int64_t)kb * kb_step];
    dst[1] = dbp[(int64_t)kb * kb_step + pl_step];
#endif
  };
  // query image of this chunk: stage s = HL_KS*6*2 pieces of 1 KB; 16-byte units, stage_units / HL_THREADS per thread
  constexpr int stage_units = HL_KS * HL_CT * 2 * 64;
  const h8* qsrc = reinterpret_cast<const h8*>(a.qi) + (int64_t)chunk * KB * HL_CT * 2 * 64;
  constexpr int QLD = stage_units / HL_THREADS;
  h8 qreg[QLD];
  auto load_q = [&](int s) {
    s = s < n_stage ? s : n_stage - 1;                        // (past the end: a harmless re-read)
#pragma unroll
    for (int u = 0; u < QLD; ++u) qreg[u] = qsrc[(int64_t)s * stage_units + u * HL_THREADS + tid];
  };
  auto store_q = [&](int buf) {
    h8* dst = reinterpret_cast<h8*>(lds) + buf * stage_units;
#pragma unroll
    for (int u = 0; u < QLD; ++u) dst[u * HL_THREADS + tid] = qreg[u];
  };
  auto lds_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };

  double acc[HL_CT][4];
  f32x4 xacc[HL_CT];
#pragma unroll
  for (int c = 0; c < HL_CT; ++c) {
    xacc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[c][r] = 0.0;
  }
  h8 ring[HL_RING][2];
#pragma unroll
  for (int i = 0; i < HL_RING; ++i) load_a(i, ring[i]);
  load_q(0);
  store_q(0);
  load_q(1);
  lds_barrier();
  store_q(1);                                                   // stage 1's fragments: read after stage 0's barrier
  load_q(2);
  const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
  // fragments of a pair-step: [column tile of the pair][plane]; pair-step ps of a stage: k2 = ps / 3, c0 = 2 * (ps % 3)
  h8 B[HL_PD + 1][2][2];                                        // ring: fragments are read HL_PD pair-steps ahead
  auto ld_b = [&](int buf, int ps, h8 (&d)[2][2]) {
    const h8* qb = reinterpret_cast<const h8*>(lds) + buf * stage_units + lane;
    const int k2 = ps / 3, c0 = 2 * (ps % 3);
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) d[jj][pl] = qb[((k2 * HL_CT + c0 + jj) * 2 + pl) * 64];
  };
  f32x4 hp[2] = {zero4, zero4};                                 // the previous pair-step's chain sums, not yet flushed
  f32x4 hold[HL_CT];                                            // first halves of the current stage's chains (k-block 0)
  static_assert(HL_KS == 2, "the h h' chains pair the two k-blocks of a stage");
#pragma unroll
  for (int i = 0; i < HL_PD; ++i) ld_b(0, i, B[i]);

  for (int s2 = 0; s2 < n_stage; s2 += 2) {                     // two stages per trip: ring / buffer indices are static
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) {
      const int s = s2 + ss;
#pragma unroll
      for (int ps = 0; ps < HL_PS; ++ps) {
        const int k2 = ps / 3, c0 = 2 * (ps % 3);
        const int pc0 = 2 * ((ps + 2) % 3);                     // the previous pair-step's column tiles
        const int kb = s * HL_KS + k2;
        h8 (&af)[2] = ring[(ss * HL_KS + k2) % HL_RING];
        // (HL_PS is a multiple of HL_PD + 1 for the shipped shapes: the ring index is static)
        h8 (&Bc)[2][2] = B[(ss * HL_PS + ps) % (HL_PD + 1)];
        h8 (&Bn)[2][2] = B[(ss * HL_PS + ps + HL_PD) % (HL_PD + 1)];
        if (ps == HL_PS - HL_PD) {
          // the fragments of this stage's remaining pair-steps have arrived (lgkmcnt(0)): from here on nobody reads
          // this stage's buffer, and every wave's stores of the next stage (issued at the start of this one) are done
          lds_barrier();
        }
        if (ps >= HL_PS - HL_PD) ld_b((ss + 1) & 1, ps + HL_PD - HL_PS, Bn);      // next stage's first pair-steps
        else ld_b(ss, ps + HL_PD, Bn);
        // h h': chains of two instructions - the stage's first k-block starts them (C = 0), its second finishes them
        const f32x4 h0 = mfma_h(af[0], Bc[0][0], k2 == 0 ? zero4 : hold[c0]);
        const f32x4 h1 = mfma_h(af[0], Bc[1][0], k2 == 0 ? zero4 : hold[c0 + 1]);
        if (!(QPG_HL_PROBE & 8)) {
          xacc[c0] = mfma_h(af[0], Bc[0][1], xacc[c0]);         // cross terms: f32 chains (2^-11 smaller)
          xacc[c0 + 1] = mfma_h(af[0], Bc[1][1], xacc[c0 + 1]);
          xacc[c0] = mfma_h(af[1], Bc[0][0], xacc[c0]);
          xacc[c0 + 1] = mfma_h(af[1], Bc[1][0], xacc[c0 + 1]);
        }
        // f64 running sums: the previous pair-step's chains, if it finished any (it belonged to a stage's second k-block)
        if (!(QPG_HL_PROBE & 1) && ((ps + HL_PS - 1) % HL_PS) / 3 == HL_KS - 1) {
#pragma unroll
          for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[pc0 + jj][r] += (double)hp[jj][r];
        }
        if (k2 == 0) {
          hold[c0] = h0;
          hold[c0 + 1] = h1;
        } else {
          hp[0] = h0;
          hp[1] = h1;
        }
        if (ps % 3 == 2 && !(QPG_HL_PROBE & 4)) {                                  // this k-block is done: refill its slot
          const int kn = kb + HL_RING < KB ? kb + HL_RING : KB - 1;
          load_a(kn, af);
        }
        if (ps == HL_PS - 1) {                                  // (behind the barrier) next-but-one stage -> this buffer
          store_q(ss);
          load_q(s + 3);
        }
        // issue order: an MFMA, then a fragment read and a share of the flush underneath it
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          HL_SGB(0x008, 1);
          if (i < 4) HL_SGB(0x100, 1);
          HL_SGB(0x002, 3);
        }
      }
    }
  }
#pragma unroll
  for (int jj = 0; jj < 2; ++jj)                                // the last pair-step's block sums (column tiles 4, 5)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[4 + jj][r] += (double)hp[jj][r];
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // surplus prefetches must not outlive their registers
  __builtin_amdgcn_s_barrier();

  const int cg = lane & 15, rg = lane >> 4;
  // ---- generic epilogue: D[q][row] = 1 - S 2^-(e_c + e_q), four consecutive rows per lane: one 16-byte store
  if (!win_ok) return;
  const int e_c1 = a.meta[0];
#pragma unroll
  for (int ct = 0; ct < HL_CT; ++ct) {
    const int q = chunk * (16 * HL_CT) + ct * 16 + cg;
    if (q >= a.Q) continue;
    const int e_q1 = a.qexp[q];
    f32x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      o[r] = (float)(1.0 - ldexp(acc[ct][r] + (double)xacc[ct][r] * (1.0 / 2048.0), -(e_c1 + e_q1)));
    if (a.D)
      *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.D) + (int64_t)q * a.ldD + (int64_t)j * 32 + 16 * t + 4 * rg) = o;
    if (a.tmin) {           // the tile's minimum over its 16 rows: lanes cg, cg + 16, cg + 32, cg + 48 hold 4 rows each
      float m = fminf(fminf(o[0], o[1]), fminf(o[2], o[3]));
      m = fminf(m, xor_lanes_f<16>(m));
      m = fminf(m, xor_lanes_f<32>(m));
      if (rg == 0) a.tmin[(int64_t)q * a.ldT + (int64_t)j * 2 + t] = m;
      if (a.tmask) {
        // round 4: which of the 16 rows lie within the band of the TILE's minimum - a superset of the rows within the
        // band of the code's minimum whenever the tile is opened at all (code minimum <= tile minimum, and the f32
        // addition is monotone).  With it the select never reads the matrix: 6 bytes per (query, tile) instead of 64.
        const float lim = m + a.band;
        unsigned int bits = ((o[0] <= lim) ? 1u : 0u) | ((o[1] <= lim) ? 2u : 0u) | ((o[2] <= lim) ? 4u : 0u) |
                            ((o[3] <= lim) ? 8u : 0u);
        bits <<= 4 * rg;
        bits |= (unsigned int)lane_xor<16>((int)bits);
        bits |= (unsigned int)lane_xor<32>((int)bits);
        if (rg == 0) a.tmask[(int64_t)q * a.ldT + (int64_t)j * 2 + t] = (uint16_t)bits;
      }
    }
  }
}


// ---- the sweep on 32-ROW wave tiles (round 4): audio_cosine_hl2_kernel ----------------------------------------------------
// Why: in round 3's kernel (hl_gemm16_kernel's organisation) a wave owns 16 rows x 96 columns and reads the whole 12 KB of a k-block's query
// fragments from LDS for its 18 MFMAs.  Per CU and k-block step that is 8 waves x 12 ds_read_b128 x 8 cycles = 768 LDS
// cycles against 2 waves x 18 x 16 = 576 matrix cycles per SIMD: the LDS pipe is busier than the matrix pipes (168 k
// LDS cycles per CU and launch = 99 us at the 1.69 GHz the kernel runs at), which is why the kernel still took 140 us
// with the HBM stream compiled out (profiles/r03_pmc_audio_hl.md).  LDS bytes per MFMA depend on ONE thing, the rows a
// wave owns; what kept a wave at 16 was the register file (f64 sums + chain halves + cross accumulators).
// Here a wave owns a whole window (two row tiles, 27 live rows) x 96 columns, and the registers come from the numerics:
//   * the l planes are stored at their true scale (split_hl_audio), so the cross products h l' + l h' carry their real
//     magnitude and run through the SAME f32 accumulator as h h': one chain = the 6 instructions of a stage's two
//     k-blocks for one (row tile, column tile), started from C = 0 and added to the f64 running sum when it is done -
//     no cross accumulators, no held chain halves;
//   * bound of such a chain: the four cross instructions come FIRST, so their block errors and the roundings of the
//     running result are relative to a sum that is <= 2^-10 of the h h' products' (<= 17 x 2^-24 x 2^-10 in units of
//     sum|h h' products|: 1e-9); the two h h' instructions that follow are round 3's chain of two with a tiny C in front:
//     kappa_6 <= kappa_2 + 0.02, kappa_2 <= 13 as assumed in §4.1 (the load-time self-check measures chains of six in
//     this order).  The budget is round 3's - chain sums 7.75e-7, representation 3.0e-7, f32 matrix 1.2e-7 - minus the
//     cross chains' own 0.7e-7: 1.2e-6 <= QPG_AUDIO_HL_ERR = 1.3e-6, unchanged.
// Consequences: 4 fragment reads per 12 MFMAs (half the LDS traffic), 8 windows per block (256 blocks at N = 2048: one
// round), the shifted add of the epilogue inside one wave (no LDS exchange), the f64 flush unchanged per row.
// Block = 8 waves = 8 windows; stage = 2 k-blocks of the chunk's query image (24 KB), double-buffered; database
// fragments HBM -> VGPR, ring of 4 k-blocks (a stage's two slots are refilled when the stage is done: one stage = 144
// MFMAs of lead per wave).
// 1 / sqrt(x) in f64 from the f32 estimate and two Newton steps (relative error < 1e-15 for x in f32's normal range): the
// epilogue's cosine = 1 - dot * rsqrt(|q|^2) * rsqrt(|c|^2) then costs ~12 f64 operations per candidate instead of the ~90
// of two IEEE square roots and a division - a third of what was left of the kernel behind its k loop.  (The sweep's
// values carry the a-priori bound of 1.3e-6; this adds 3 ulp of f64.)
__device__ __forceinline__ double fast_rsqrt_f64(double x) {
  const float xf = (float)x;
  double y = (double)__builtin_amdgcn_rsqf(xf);
  const double hx = 0.5 * x;
  y = y * (1.5 - hx * y * y);
  y = y * (1.5 - hx * y * y);
  return y;
}
#define H2_W 8
#ifndef H2_QFIRST
#define H2_QFIRST 1      // the query loads go out in front of the ring's loads for every ring depth (see the k loop)
#endif
#ifndef H2_PIN
#define H2_PIN 1         // two-plane kernel: where the ring's refill is issued - 0: hipcc's choice; 1: behind the stage's last
#endif                   // MFMA; 2: two loads behind the last use of each register pair
#ifndef H2_PRO
#define H2_PRO 1         // prologue in the k loop's request order
#endif
// ablation hooks (experiments/audio_hl): -DH2_PROBE=<bits>; the product build defines nothing.  1: database fragments read
// from one address (no HBM stream); 2: no f64 flush; 8: no stage barrier; 16: no query-fragment reads from LDS (the
// prologue's fragments for every step); 32: no query staging (global -> register -> LDS) inside the k loop; 64: no MFMAs (the fragment loads stay)
#ifndef H2_PROBE
#define H2_PROBE 0
#endif
#ifndef H2_NT
#define H2_NT 1          // one query chunk: the image is read ONCE - non-temporal fragment loads (round 5: the kernel with its
#endif                   // matrix work compiled out takes 130 us with plain loads, 113 with these; 0: plain loads always)
#ifndef H2_RS2
#define H2_RS2 2          // stages of database fragments in flight, two-plane kernel (3 / 4: probes only - no registers)
#endif
// ONE-PLANE database image (round 5: the track stored in IEEE f16, GestureDB feature_dtype "f16").  An f16 value IS its
// own h plane: no scaling (exponent 0), no l plane, no representation error on the database side - half the bytes (a
// window is [tile 0: KB x 64 units][tile 1: KB x 44 units] of 16 bytes: 340 MB at N = 2048) and TWO products per element
// instead of three (h h' and the query's l h).  A chain is the four instructions of a stage's two k-blocks, the two cross
// blocks first; its bound is the six-instruction chain's (a sub-chain of it: selfcheck.py measures both orders).  Budget
// relative to |q||c|: chain sums 13.05 x 2^-24 = 7.8e-7, query representation 2^-23 = 1.2e-7 (+ 1e-8 for its subnormal
// l), f32-stored matrix 1.2e-7: 1.03e-6 <= QPG_AUDIO_HL_ERR (the select keeps ONE band for both images).
// The 32 registers the l plane's ring held go into the ring's depth: RS = 4 stages (8 k-blocks) of fragments in flight.
#define HL1_WIN_UNITS(KB) ((int64_t)(KB) * (64 + HL_T1_UNITS))
// PL: planes of the database image (2: h | l, f32 track; 1: the f16 track); RS: stages of database fragments in flight
template <int PL, int RS, bool NT>
__global__ __launch_bounds__(64 * H2_W, 2) void audio_cosine_hl2_kernel(HlArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // 2 x 2 x 6 x 2 x 1 KB
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int xl = (int)blockIdx.x, slot = xl >> 3;
  const int wgrp = (slot / a.chunks) * 8 + (xl & 7);                   // XCD-aware, as hl_gemm16_kernel
  const int chunk = slot % a.chunks;
  if (a.j0 + wgrp * H2_W >= a.N) return;
  const int j = a.j0 + wgrp * H2_W + w;
  const bool win_ok = j < a.N;
  const int KB = a.KB, n_stage = KB / 2;
  const int cg = lane & 15, rg = lane >> 4;
  // database fragments of this window: tile 0 = 64 units per plane and k-block; tile 1 = its 11 live rows, 44 units.
  // Addresses = a block-uniform base (scalar registers) + a 32-bit byte offset per lane (a block's 8 windows span
  // < 3 MB): one address register per load and a 32-bit add per k-block (64-bit pointers per lane - which the zero-page
  // redirect of dead lanes needed - cost two registers and a 64-bit multiply-add per load, and the one-plane kernel
  // spilled an address INSIDE the k loop: a scratch reload's vmcnt(0) drains the whole fragment ring).  Dead lanes need no
  // zeros: row i of an MFMA's result depends on row i of A only, rows 27..31 are never read by the epilogue, and a window
  // past N is dropped there too - such lanes read their nearest live neighbour's bytes (same cache line, no traffic).
  const int jb = a.j0 + wgrp * H2_W;                                   // first window of the block (uniform)
  const uint32_t wj = (uint32_t)(win_ok ? w : a.N - 1 - jb);
  const uint32_t win_units = (uint32_t)(PL == 2 ? HL_WIN_UNITS(KB) : HL1_WIN_UNITS(KB));
  const unsigned char* sbase = reinterpret_cast<const unsigned char*>(a.db) + (int64_t)jb * win_units * 16;
  const uint32_t o0 = (wj * win_units + (uint32_t)lane) * 16u;
  const uint32_t o1 = (wj * win_units + (uint32_t)KB * (64 * PL) + 11u * rg + (cg < 11 ? cg : 10)) * 16u;
  constexpr uint32_t st0 = (H2_PROBE & 1) ? 0u : 64u * PL * 16u, st1 = (H2_PROBE & 1) ? 0u : (uint32_t)PL * HL_T1_UNITS * 16u;
  auto ld_frag = [](const unsigned char* p) -> h8 {
    if (NT) return __builtin_nontemporal_load(reinterpret_cast<const h8*>(p));
    return *reinterpret_cast<const h8*>(p);
  };
  auto load_a = [&](int kb, h8 (&d)[2 * PL]) {                         // [tile 0 h (, l), tile 1 h (, l)]
    kb = kb < KB ? kb : KB - 1;
    const uint32_t a0 = o0 + (uint32_t)kb * st0, a1 = o1 + (uint32_t)kb * st1;
    d[0] = ld_frag(sbase + a0);
    if (PL == 2) d[1] = ld_frag(sbase + a0 + 64 * 16);
    d[PL] = ld_frag(sbase + a1);
    if (PL == 2) d[PL + 1] = ld_frag(sbase + a1 + HL_T1_UNITS * 16);
  };
  constexpr int stage_units = 2 * HL_CT * 2 * 64;
  constexpr int QLD = stage_units / (64 * H2_W);
  const unsigned char* qsrc = reinterpret_cast<const unsigned char*>(a.qi) + (int64_t)chunk * KB * HL_CT * 2 * 1024;
  const uint32_t qoff = (uint32_t)tid * 16u;
  h8 qreg[QLD];
  auto load_q = [&](int s) {
    s = s < n_stage ? s : n_stage - 1;
#pragma unroll
    for (int u = 0; u < QLD; ++u)
      qreg[u] = *reinterpret_cast<const h8*>(qsrc + (int64_t)s * (stage_units * 16) + u * (64 * H2_W * 16) + qoff);
  };
  auto store_q = [&](int buf) {
    h8* dst = reinterpret_cast<h8*>(lds) + buf * stage_units;
#pragma unroll
    for (int u = 0; u < QLD; ++u) dst[u * (64 * H2_W) + tid] = qreg[u];
  };
  auto lds_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
  double acc[2][HL_CT][4];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int c = 0; c < HL_CT; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[t][c][r] = 0.0;
  h8 ring[2 * RS][2 * PL];
  // Prologue in the k loop's OWN request order - the third stage's query loads in front of the ring's last stage: hipcc's
  // wait for a register is the tightest over every path into the loop, and with the query loads youngest here the loop
  // head waited vmcnt(3) for them - which drains the whole ring once per trip (round 5: 137 -> see DESIGN 4.1a).
#pragma unroll
  for (int i = 0; i < (H2_PRO ? 2 * (RS - 1) : 2 * RS); ++i) load_a(i, ring[i]);
  load_q(0);
  store_q(0);
  load_q(1);
  lds_barrier();
  store_q(1);
  load_q(2);
  if (H2_PRO) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 2 * (RS - 1); i < 2 * RS; ++i) load_a(i, ring[i]);
    __builtin_amdgcn_sched_barrier(0);
  }
  const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
  h8 B[2][2][2];                                                       // [step parity][k-block of the stage][plane]
  auto ld_b = [&](int buf, int c, h8 (&d)[2][2]) {
    const h8* qb = reinterpret_cast<const h8*>(lds) + buf * stage_units + lane;
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) d[k2][pl] = qb[((k2 * HL_CT + c) * 2 + pl) * 64];
  };
  ld_b(0, 0, B[0]);
  if (H2_PROBE & 16) ld_b(0, 1, B[1]);
  f32x4 dp[2] = {zero4, zero4};                                        // the previous step's two chains, not yet flushed
  // a trip = an even number of stages that is a multiple of RS: ring slots, LDS buffers and fragment parities are static
  constexpr int TRIP = (RS % 2 == 0) ? RS : 2 * RS;
  for (int s0 = 0; s0 < n_stage; s0 += TRIP) {
#pragma unroll
    for (int ss = 0; ss < TRIP; ++ss) {
      const int s = s0 + ss;
      h8 (&A0)[2 * PL] = ring[(ss % RS) * 2 + 0];
      h8 (&A1)[2 * PL] = ring[(ss % RS) * 2 + 1];
#pragma unroll
      for (int c = 0; c < HL_CT; ++c) {
        const int st = ss * HL_CT + c;
        h8 (&Bc)[2][2] = B[st & 1];
        h8 (&Bn)[2][2] = B[(st + 1) & 1];
        if (c == HL_CT - 1) {
          if (!(H2_PROBE & 8)) lds_barrier();  // every fragment of this stage has arrived; the next stage's are stored
          if (!(H2_PROBE & 16)) ld_b((ss + 1) & 1, 0, Bn);
          if (!(H2_PROBE & 32) && (H2_QFIRST || RS > 2)) {
            // vmcnt counts IN ORDER: the wait for a stage's query fragments (one stage after their request) also waits
            // for every OLDER request.  With the query loads behind the ring's (below: the order of the two-stage ring,
            // where it does not matter) a deeper ring buys nothing - the data of stage s + RS - 1 must be there at the
            // end of stage s.  So here they go out FIRST (the data of stage s + RS - 2: two stages of lead for RS = 4),
            // and a scheduling fence keeps hipcc from sinking them behind the ring's loads again (it did).
            store_q(ss & 1);
            load_q(s + 3);
            __builtin_amdgcn_sched_barrier(0);
          }
        } else if (!(H2_PROBE & 16)) {
          ld_b(ss & 1, c + 1, Bn);
        }
        // two chains (row tile 0 / 1), interleaved.  ORDER INSIDE A CHAIN: the cross-term instructions first (h l', l h' of
        // both k-blocks: the running sum stays 2^-10 of the h h' scale, and so do their roundings), the two h h'
        // instructions last - the chain then behaves like round 3's chain of two (measured: every instruction that
        // re-rounds a FULL-SIZE running sum costs ~1.2-1.5 units of 2^-24 sum|products|, so h h' first would mean
        // kappa_6 = 14.4 against 9.9 this way; selfcheck.py measures this order)
        f32x4 d0, d1;
        if (H2_PROBE & 64) {                                       // probe: the loads and everything else, no matrix work
#pragma unroll
          for (int i = 0; i < 2 * PL; ++i) asm volatile("" ::"v"(A0[i]), "v"(A1[i]));
          d0 = zero4;
          d1 = zero4;
        } else {
          d0 = mfma_h(A0[0], Bc[0][1], (H2_PROBE & 2) ? dp[0] : zero4);     // (probe 2: one endless chain, no flush)
          d1 = mfma_h(A0[PL], Bc[0][1], (H2_PROBE & 2) ? dp[1] : zero4);
          if (PL == 2) {
            d0 = mfma_h(A0[PL - 1], Bc[0][0], d0);
            d1 = mfma_h(A0[2 * PL - 1], Bc[0][0], d1);
          }
          d0 = mfma_h(A1[0], Bc[1][1], d0);
          d1 = mfma_h(A1[PL], Bc[1][1], d1);
          if (PL == 2) {
            d0 = mfma_h(A1[PL - 1], Bc[1][0], d0);
            d1 = mfma_h(A1[2 * PL - 1], Bc[1][0], d1);
          }
          d0 = mfma_h(A0[0], Bc[0][0], d0);
          d1 = mfma_h(A0[PL], Bc[0][0], d1);
          d0 = mfma_h(A1[0], Bc[1][0], d0);
          d1 = mfma_h(A1[PL], Bc[1][0], d1);
        }
        // f64 running sums: the PREVIOUS step's chains (zeros in front of the first step)
        if (!(H2_PROBE & 2)) {
          const int pc = (c + HL_CT - 1) % HL_CT;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            acc[0][pc][r] += (double)dp[0][r];
            acc[1][pc][r] += (double)dp[1][r];
          }
        }
        dp[0] = d0;
        dp[1] = d1;
        if (c == HL_CT - 1) {
          // the stage is done: its two ring slots take the k-blocks RS stages ahead; the freed LDS buffer takes the
          // stage after next (behind the barrier above: nobody reads this stage's buffer any more)
          load_a(2 * (s + RS), A0);
          load_a(2 * (s + RS) + 1, A1);
          if (!(H2_PROBE & 32) && !H2_QFIRST && RS <= 2) {
            store_q(ss & 1);
            load_q(s + 3);
          }
        }
#pragma unroll
        for (int i = 0; i < 4 + 4 * PL; ++i) { // issue order: an MFMA, then a fragment read / a share of the flush under it
          HL_SGB(0x008, 1);
          if (i < 4) HL_SGB(0x100, 1);
          HL_SGB(0x002, PL == 2 ? 2 : 3);
          if (H2_PIN == 2 && PL == 2 && c == HL_CT - 1 && (i == 3 || i == 7 || i == 9 || i == 11)) HL_SGB(0x020, 2);
        }
        // the ring's refill goes out HERE, behind the stage's last MFMAs (their registers are free then) - named, or hipcc
        // issues it somewhere in the middle of the next stage and the ring loses a third of its lead
        if (c == HL_CT - 1 && H2_PIN == 1 && PL == 2) HL_SGB(0x020, 4 * PL);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {                                        // the last step's chains (column tile 5)
    acc[0][HL_CT - 1][r] += (double)dp[0][r];
    acc[1][HL_CT - 1][r] += (double)dp[1][r];
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");          // surplus prefetches must not outlive their registers
  if (!win_ok) return;
  // ---- epilogue: dot(q, cand g) = S[g][lo column] + S[g + 1][hi column]; the wave holds all 27 super-rows of its window:
  // lane (cg, rg) has rows 4 rg .. 4 rg + 3 of tile t for column cg of every column tile
  const int e_c = PL == 2 ? a.meta[0] : 0;
  const int src = (lane + 16) & 63;
#pragma unroll
  for (int ct = 0; ct < 3; ++ct) {
    const int q = chunk * HL_QC + ct * 16 + cg;
    const bool q_ok = q < a.Q;
    const double qq = q_ok ? a.qn2[q] : 1.0;
    const int e_q = q_ok ? a.qexp[q] : 0;
    // row + 1 of the hi column: register r + 1 of the same lane, register 0 of the lane 16 further (the next row group;
    // for the last row group of tile 0 that lane is row group 0 of TILE 1), nothing behind tile 1's last group
    const double nx0 = __shfl(acc[0][3 + ct][0], src, 64);
    const double nx1 = __shfl(acc[1][3 + ct][0], src, 64);
    // the ordinary case - both norms in f32's normal range - takes the fast reciprocal square roots; zero / tiny / huge
    // norms take cosine_from_dot (sklearn's degenerate-row semantics)
    const bool q_fast = qq > 1e-30 && qq < 1e30;
    const double iq = q_fast ? fast_rsqrt_f64(qq) : 0.0;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      double hs[4];
      hs[0] = acc[t][3 + ct][1];
      hs[1] = acc[t][3 + ct][2];
      hs[2] = acc[t][3 + ct][3];
      hs[3] = t == 0 ? (rg < 3 ? nx0 : nx1) : (rg < 3 ? nx1 : 0.0);
      // the four candidates of this lane are consecutive: their squared norms are one 32-byte run
      const int g0 = 16 * t + 4 * rg;
      double cc4[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) cc4[r] = (g0 + r < a.G) ? a.cn2[(int64_t)j * a.G + g0 + r] : 1.0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int g = g0 + r;
        if (!q_ok || g >= a.G) continue;
        const double dot = ldexp(acc[t][ct][r] + hs[r], -(e_c + e_q));
        const int64_t c = (int64_t)j * a.G + g;
        const double cc = cc4[r];
        // validity range of the representation bound (HL_NORM2_MIN above); the one-plane image holds the database exactly
        if (a.stats && ((PL == 2 && cc > 0.0 && ldexp(cc, 2 * e_c) < HL_NORM2_MIN) ||
                        (qq > 0.0 && ldexp(qq, 2 * e_q) < HL_NORM2_MIN)))
          atomicOr(&a.stats[1], 2);
        double dd;
        if (q_fast && cc > 1e-30 && cc < 1e30) dd = 1.0 - dot * (iq * fast_rsqrt_f64(cc));
        else dd = cosine_from_dot(dot, qq, cc);
        if (a.d_f32) reinterpret_cast<float*>(a.D)[(int64_t)q * a.ldD + c] = (float)dd;
        else reinterpret_cast<double*>(a.D)[(int64_t)q * a.ldD + c] = dd;
      }
    }
  }
}

// ---- C ABI ---------------------------------------------------------------------------------------------------------------
#define HL_F_MAX (1 << 20)    // feature widths beyond this are refused by the size helpers (their products stay inside int64)
extern "C" int64_t qpg_audio_hl_db_bytes(int N, int F) {            // database image + 64 bytes of metadata
  return (N <= 0 || F <= 0 || F > HL_F_MAX) ? 0 : (int64_t)N * HL_WIN_UNITS((int64_t)HL_SUB * F / 32) * 16 + 64;
}
extern "C" int64_t qpg_audio_hl_query_bytes(int Q, int F) {
  if (Q <= 0 || F <= 0 || F > HL_F_MAX) return 0;
  const int64_t chunks = ((int64_t)Q + HL_QC - 1) / HL_QC;
  return chunks * ((int64_t)HL_SUB * F / 32) * HL_CT * 2 * HL_PIECE + chunks * HL_QC * 4;
}

static bool hl_grid_ok(int T, int F, int G, int n_taps, int tap_stride, int step) {
  if (tap_stride <= 0 || tap_stride > (1 << 20) || F > HL_F_MAX) return false;
  return n_taps == 2 * HL_SUB && G == HL_ROWS - 1 && step == HL_SUB * tap_stride && (F % 32) == 0 && F >= 32 &&
         ((HL_SUB * F / 32) % (2 * HL_KS)) == 0 && T > 0;
}

extern "C" int qpg_audio_hl_supported(int T, int F, int G, int n_taps, int tap_stride, int cand_step) {
  return hl_grid_ok(T, F, G, n_taps, tap_stride, cand_step) ? 1 : 0;
}

extern "C" int qpg_audio_hl_pack_db(qpg_ctx* ctx, void* stream, const float* base, int N, int T, int F, int G, int n_taps,
                                    int tap_stride, int cand_step, void* image, int64_t image_bytes) {
  const char* name = "qpg_audio_hl_pack_db";
  QPG_REQUIRE(ctx && base && image && N > 0, "%s: bad argument", name);
  QPG_REQUIRE(hl_grid_ok(T, F, G, n_taps, tap_stride, cand_step),
              "%s: needs 6 taps, 26 grid positions %d frames apart (= 3 x tap_stride) and F %% 128 == 0", name,
              HL_SUB * tap_stride);
  QPG_REQUIRE(image_bytes >= qpg_audio_hl_db_bytes(N, F) && (reinterpret_cast<uintptr_t>(image) % 16) == 0 &&
                  (reinterpret_cast<uintptr_t>(base) % 16) == 0,
              "%s: image too small or misaligned (qpg_audio_hl_db_bytes)", name);
  hipStream_t st = qpg_stream(stream);
  unsigned char* img = static_cast<unsigned char*>(image);
  const int64_t body = qpg_audio_hl_db_bytes(N, F) - 64;
  int32_t* meta = reinterpret_cast<int32_t*>(img + body);
  unsigned int* amax = reinterpret_cast<unsigned int*>(meta + 4);
  hipLaunchKernelGGL(hl_zero_u32_kernel, dim3(1), dim3(1), 0, st, amax);
  hipLaunchKernelGGL(hl_absmax_kernel, dim3(1024), dim3(1024), 0, st, base, (int64_t)N * T * F, amax);
  hipLaunchKernelGGL(hl_exponent_kernel, dim3(1), dim3(1), 0, st, (const unsigned int*)amax, meta);
  hipLaunchKernelGGL(hl_pack_db_kernel, dim3(4096), dim3(256), 0, st, base, N, T, F, cand_step, tap_stride,
                     (const int32_t*)meta, reinterpret_cast<_Float16*>(img));
  QPG_LAUNCH_CHECK("hl_pack_db_kernel");
  return QPG_OK;
}

extern "C" int qpg_audio_hl_pack_queries(qpg_ctx* ctx, void* stream, const float* q32, int Q, int F, void* image,
                                         int64_t image_bytes) {
  const char* name = "qpg_audio_hl_pack_queries";
  QPG_REQUIRE(ctx && q32 && image && Q > 0 && F > 0 && (F % 32) == 0, "%s: bad argument", name);
  QPG_REQUIRE(image_bytes >= qpg_audio_hl_query_bytes(Q, F) && (reinterpret_cast<uintptr_t>(image) % 16) == 0 &&
                  (reinterpret_cast<uintptr_t>(q32) % 16) == 0,
              "%s: image too small or misaligned (qpg_audio_hl_query_bytes)", name);
  const int chunks = (Q + HL_QC - 1) / HL_QC;
  unsigned char* img = static_cast<unsigned char*>(image);
  int32_t* qexp = reinterpret_cast<int32_t*>(img + (int64_t)chunks * (HL_SUB * F / 32) * HL_CT * 2 * HL_PIECE);
  hipLaunchKernelGGL(hl_pack_queries_kernel, dim3(chunks * HL_QC), dim3(768), 0, qpg_stream(stream), q32, Q, F,
                     reinterpret_cast<_Float16*>(img), qexp);
  QPG_LAUNCH_CHECK("hl_pack_queries_kernel");
  return QPG_OK;
}

extern "C" int qpg_audio_pack_queries_hl(qpg_ctx* ctx, void* stream, const float* qbase, int M, int T, int F,
                                         const int32_t* q_win, const int32_t* q_t, int Q, int n_taps, int tap_stride,
                                         float* q32, double* qn2, void* image, int64_t image_bytes) {
  const char* name = "qpg_audio_pack_queries_hl";
  QPG_REQUIRE(ctx && qbase && q_win && q_t && q32 && qn2 && image && M > 0 && T > 0 && Q > 0 && tap_stride > 0,
              "%s: bad argument", name);
  QPG_REQUIRE(n_taps == 2 * HL_SUB && F > 0 && (F % 32) == 0 && 2 * HL_SUB * F / 8 <= 2 * 768,
              "%s: needs 6 taps, F %% 32 == 0, F <= 2048", name);
  QPG_REQUIRE(image_bytes >= qpg_audio_hl_query_bytes(Q, F) && (reinterpret_cast<uintptr_t>(image) % 16) == 0 &&
                  (reinterpret_cast<uintptr_t>(q32) % 16) == 0 && (reinterpret_cast<uintptr_t>(qbase) % 16) == 0,
              "%s: image too small or misaligned (qpg_audio_hl_query_bytes)", name);
  const int chunks = (Q + HL_QC - 1) / HL_QC;
  unsigned char* img = static_cast<unsigned char*>(image);
  int32_t* qexp = reinterpret_cast<int32_t*>(img + (int64_t)chunks * (HL_SUB * F / 32) * HL_CT * 2 * HL_PIECE);
  hipLaunchKernelGGL(hl_pack_queries_fused_kernel, dim3(chunks * HL_QC), dim3(768), 0, qpg_stream(stream), qbase, M, T, F,
                     q_win, q_t, Q, tap_stride, q32, qn2, reinterpret_cast<_Float16*>(img), qexp);
  QPG_LAUNCH_CHECK("hl_pack_queries_fused_kernel");
  return QPG_OK;
}

// The audio pack above and the text side's query pack in ONE launch (hl_pack_clip_kernel): text_ctx [dev] f32 [Mt][R][Dt],
// tq_win / tq_row [dev] i32 [Qt] (window and context row of every text query), qn_out [dev] f32 [Qt][Dt] (the normalised
// queries, bit-identical to qpg_text_pack_queries_f32's), cols_image = qpg_hl_cols_bytes(Qt, Dt) bytes (identical to
// qpg_hl_pack_cols' image of qn_out).  Dt % 128 == 0, Dt <= 8192.
extern "C" int qpg_clip_pack_hl(qpg_ctx* ctx, void* stream, const float* qbase, int M, int T, int F, const int32_t* q_win,
                                const int32_t* q_t, int Q, int n_taps, int tap_stride, float* q32, double* qn2, void* image,
                                int64_t image_bytes, const float* text_ctx, int Mt, int R, int Dt, const int32_t* tq_win,
                                const int32_t* tq_row, int Qt, float* qn_out, void* cols_image, int64_t cols_bytes) {
  const char* name = "qpg_clip_pack_hl";
  QPG_REQUIRE(ctx && qbase && q_win && q_t && q32 && qn2 && image && M > 0 && T > 0 && Q > 0 && tap_stride > 0,
              "%s: bad argument", name);
  QPG_REQUIRE(n_taps == 2 * HL_SUB && F > 0 && (F % 32) == 0 && 2 * HL_SUB * F / 8 <= 2 * 768,
              "%s: needs 6 taps, F %% 32 == 0, F <= 2048", name);
  QPG_REQUIRE(image_bytes >= qpg_audio_hl_query_bytes(Q, F) && (reinterpret_cast<uintptr_t>(image) % 16) == 0 &&
                  (reinterpret_cast<uintptr_t>(q32) % 16) == 0 && (reinterpret_cast<uintptr_t>(qbase) % 16) == 0,
              "%s: image too small or misaligned (qpg_audio_hl_query_bytes)", name);
  QPG_REQUIRE(text_ctx && tq_win && tq_row && qn_out && cols_image && Mt > 0 && R > 0 && Qt > 0 && Dt > 0 &&
                  (Dt % 128) == 0 && Dt <= 8192 && cols_bytes >= qpg_hl_cols_bytes(Qt, Dt) &&
                  (reinterpret_cast<uintptr_t>(cols_image) % 16) == 0,
              "%s: bad text-side argument (Dt %% 128 == 0, qpg_hl_cols_bytes)", name);
  const int chunks = (Q + HL_QC - 1) / HL_QC;
  unsigned char* img = static_cast<unsigned char*>(image);
  int32_t* qexp = reinterpret_cast<int32_t*>(img + (int64_t)chunks * (HL_SUB * F / 32) * HL_CT * 2 * HL_PIECE);
  const int tchunks = (Qt + HL_GQC - 1) / HL_GQC;
  unsigned char* cimg = static_cast<unsigned char*>(cols_image);
  ClipPackText tx;
  tx.ctx = text_ctx; tx.q_win = tq_win; tx.q_row = tq_row; tx.R = R; tx.Dt = Dt; tx.Qt = Qt; tx.qn = qn_out;
  tx.image = reinterpret_cast<_Float16*>(cimg);
  tx.qexp = reinterpret_cast<int32_t*>(cimg + (int64_t)tchunks * (Dt / 32) * HL_CT * 2 * HL_PIECE);
  const int n_aud = chunks * HL_QC;
  hipLaunchKernelGGL(hl_pack_clip_kernel, dim3(n_aud + tchunks * HL_GQC), dim3(768), (size_t)Dt * 4, qpg_stream(stream),
                     qbase, M, T, F, q_win, q_t, Q, tap_stride, q32, qn2, reinterpret_cast<_Float16*>(img), qexp, n_aud, tx);
  QPG_LAUNCH_CHECK("hl_pack_clip_kernel");
  return QPG_OK;
}

extern "C" int qpg_audio_cosine_hl(qpg_ctx* ctx, void* stream, const void* db_image, int N, int F, int G,
                                   const double* cn2, const void* q_image, const double* qn2, int Q, void* D,
                                   int d_is_f32, int64_t ldD, int32_t* stats) {
  const char* name = "qpg_audio_cosine_hl";
  QPG_REQUIRE(ctx && db_image && cn2 && q_image && qn2 && D, "%s: null pointer", name);
  QPG_REQUIRE(N > 0 && Q > 0 && G == HL_ROWS - 1 && (F % 32) == 0 && ((HL_SUB * F / 32) % (2 * HL_KS)) == 0 &&
                  ldD >= (int64_t)N * G,
              "%s: bad size", name);
  const int chunks = (Q + HL_QC - 1) / HL_QC, KB = HL_SUB * F / 32;
  HlArgs a;
  const unsigned char* dbi = static_cast<const unsigned char*>(db_image);
  const unsigned char* qi = static_cast<const unsigned char*>(q_image);
  a.db = reinterpret_cast<const _Float16*>(dbi);
  a.meta = reinterpret_cast<const int32_t*>(dbi + (qpg_audio_hl_db_bytes(N, F) - 64));
  a.qi = reinterpret_cast<const _Float16*>(qi);
  a.qexp = reinterpret_cast<const int32_t*>(qi + (int64_t)chunks * KB * HL_CT * 2 * HL_PIECE);
  a.cn2 = cn2; a.qn2 = qn2; a.D = D; a.zeros = ctx->zeros; a.ldD = ldD; a.stats = stats; a.N = N; a.j0 = 0; a.G = G; a.Q = Q;
  a.KB = KB; a.d_f32 = d_is_f32; a.tmin = nullptr; a.ldT = 0; a.tmask = nullptr; a.band = 0.f; a.chunks = chunks;
  const int64_t g8 = ((int64_t)N + H2_W - 1) / H2_W;
  QPG_REQUIRE(((g8 + 7) / 8) * 8 * chunks < 0x7fffffffll, "%s: too many blocks", name);
  const bool nt = H2_NT == 2 || (H2_NT == 1 && chunks == 1);      // several chunks re-read the image out of the XCD's L2
  void (*kern)(HlArgs) = nt ? audio_cosine_hl2_kernel<2, H2_RS2, true> : audio_cosine_hl2_kernel<2, H2_RS2, false>;
  hipLaunchKernelGGL(kern,
                     dim3((unsigned)(((g8 + 7) / 8) * 8 * chunks)), dim3(64 * H2_W), 2 * 2 * HL_CT * 2 * HL_PIECE,
                     qpg_stream(stream), a);
  QPG_LAUNCH_CHECK("audio_cosine_hl2_kernel");
  return QPG_OK;
}

// ---- the one-plane image of an f16-stored track (round 5; BASELINE.json configs[4] "fp16 features") -------------------------
// thread <-> (window j, super-row i < 27, k8): one 16-byte piece of the f16 base, copied to its fragment position (the
// layout of hl_pack_db_kernel without the l planes).  No scaling: the values are the database.
#define H1_RS 3               // stages of database fragments in flight (audio_cosine_hl2_kernel<1, H1_RS>)
#define H1_TRIP 6             // stages per trip of its k loop: KB must be a multiple of 2 * H1_TRIP
__global__ __launch_bounds__(256) void hl1_pack_db_kernel(const _Float16* __restrict__ base, int N, int T, int F, int step,
                                                          int tap_stride, _Float16* __restrict__ image) {
  const int KB = HL_SUB * F / 32, K8 = HL_SUB * F / 8;
  const int64_t n = (int64_t)N * HL_ROWS * K8;
  for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (int64_t)gridDim.x * blockDim.x) {
    const int k8 = (int)(id % K8);
    const int i = (int)((id / K8) % HL_ROWS);
    const int j = (int)(id / ((int64_t)K8 * HL_ROWS));
    const int k = k8 * 8, sub = k / F, f = k - sub * F;
    const int t = step * i + tap_stride * sub;
    h8 v = (h8){0, 0, 0, 0, 0, 0, 0, 0};
    if (t < T) v = *reinterpret_cast<const h8*>(base + ((int64_t)j * T + t) * F + f);
    const int kb = k / 32, kg = (k & 31) >> 3;
    const int64_t win = (int64_t)j * HL1_WIN_UNITS(KB);
    const int64_t u0 = i < 16 ? win + (int64_t)kb * 64 + i + 16 * kg
                              : win + (int64_t)KB * 64 + (int64_t)kb * HL_T1_UNITS + 11 * kg + (i - 16);
    reinterpret_cast<h8*>(image)[u0] = v;
  }
}

extern "C" int64_t qpg_audio_hl1_db_bytes(int N, int F) {
  return (N <= 0 || F <= 0 || F > HL_F_MAX) ? 0 : (int64_t)N * HL1_WIN_UNITS((int64_t)HL_SUB * F / 32) * 16;
}

static bool hl1_grid_ok(int T, int F, int G, int n_taps, int tap_stride, int step) {
  return hl_grid_ok(T, F, G, n_taps, tap_stride, step) && ((HL_SUB * F / 32) % (2 * H1_TRIP)) == 0;
}

extern "C" int qpg_audio_hl1_supported(int T, int F, int G, int n_taps, int tap_stride, int cand_step) {
  return hl1_grid_ok(T, F, G, n_taps, tap_stride, cand_step) ? 1 : 0;
}

extern "C" int qpg_audio_hl1_pack_db(qpg_ctx* ctx, void* stream, const void* base_f16, int N, int T, int F, int G, int n_taps,
                                     int tap_stride, int cand_step, void* image, int64_t image_bytes) {
  const char* name = "qpg_audio_hl1_pack_db";
  QPG_REQUIRE(ctx && base_f16 && image && N > 0, "%s: bad argument", name);
  QPG_REQUIRE(hl1_grid_ok(T, F, G, n_taps, tap_stride, cand_step),
              "%s: needs 6 taps, 26 grid positions %d frames apart (= 3 x tap_stride) and F %% 128 == 0", name,
              HL_SUB * tap_stride);
  QPG_REQUIRE(image_bytes >= qpg_audio_hl1_db_bytes(N, F) && (reinterpret_cast<uintptr_t>(image) % 16) == 0 &&
                  (reinterpret_cast<uintptr_t>(base_f16) % 16) == 0,
              "%s: image too small or misaligned (qpg_audio_hl1_db_bytes)", name);
  hipLaunchKernelGGL(hl1_pack_db_kernel, dim3(4096), dim3(256), 0, qpg_stream(stream),
                     static_cast<const _Float16*>(base_f16), N, T, F, cand_step, tap_stride, static_cast<_Float16*>(image));
  QPG_LAUNCH_CHECK("hl1_pack_db_kernel");
  return QPG_OK;
}

// The sweep over the one-plane image: qpg_audio_cosine_hl's contract (same query image, same D, same bound), cn2 = the
// squared norms of the ROUNDED track's candidates.
extern "C" int qpg_audio_cosine_hl1(qpg_ctx* ctx, void* stream, const void* db_image, int N, int F, int G, const double* cn2,
                                    const void* q_image, const double* qn2, int Q, void* D, int d_is_f32, int64_t ldD,
                                    int32_t* stats) {
  const char* name = "qpg_audio_cosine_hl1";
  QPG_REQUIRE(ctx && db_image && cn2 && q_image && qn2 && D, "%s: null pointer", name);
  QPG_REQUIRE(N > 0 && Q > 0 && G == HL_ROWS - 1 && (F % 32) == 0 && ((HL_SUB * F / 32) % (2 * H1_TRIP)) == 0 &&
                  ldD >= (int64_t)N * G,
              "%s: bad size", name);
  const int chunks = (Q + HL_QC - 1) / HL_QC, KB = HL_SUB * F / 32;
  HlArgs a;
  const unsigned char* qi = static_cast<const unsigned char*>(q_image);
  a.db = static_cast<const _Float16*>(db_image);
  a.meta = nullptr;
  a.qi = reinterpret_cast<const _Float16*>(qi);
  a.qexp = reinterpret_cast<const int32_t*>(qi + (int64_t)chunks * KB * HL_CT * 2 * HL_PIECE);
  a.cn2 = cn2; a.qn2 = qn2; a.D = D; a.zeros = ctx->zeros; a.ldD = ldD; a.stats = stats; a.N = N; a.j0 = 0; a.G = G; a.Q = Q;
  a.KB = KB; a.d_f32 = d_is_f32; a.tmin = nullptr; a.ldT = 0; a.tmask = nullptr; a.band = 0.f; a.chunks = chunks;
  const int64_t g8 = ((int64_t)N + H2_W - 1) / H2_W;
  QPG_REQUIRE(((g8 + 7) / 8) * 8 * chunks < 0x7fffffffll, "%s: too many blocks", name);
  const bool nt = H2_NT == 2 || (H2_NT == 1 && chunks == 1);
  void (*kern)(HlArgs) = nt ? audio_cosine_hl2_kernel<1, H1_RS, true> : audio_cosine_hl2_kernel<1, H1_RS, false>;
  hipLaunchKernelGGL(kern,
                     dim3((unsigned)(((g8 + 7) / 8) * 8 * chunks)), dim3(64 * H2_W), 2 * 2 * HL_CT * 2 * HL_PIECE,
                     qpg_stream(stream), a);
  QPG_LAUNCH_CHECK("audio_cosine_hl2_kernel<1>");
  return QPG_OK;
}

// ---- generic split-f16 distance GEMM (BASELINE.json configs[2]: the prefilter of the exact-f32 cosine family) ----------
// rows: x [R][D] f32 (R % 32 == 0, unit-norm rows) -> image [R/32][2 row tiles][KB][plane][64 lanes][8 f16] + exponent;
// columns: q [Q][D] f32 -> image [chunk of 96][KB][6 column tiles][plane][64][8] + one exponent per query.
__global__ __launch_bounds__(256) void hl_pack_rows_kernel(const float* __restrict__ x, int64_t R, int D,
                                                           const int32_t* __restrict__ meta, _Float16* __restrict__ image) {
  const int KB = D / 32, K8 = D / 8;
  const int64_t n = R * K8;
  const float sc = ldexpf(1.0f, meta[0]);
  for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (int64_t)gridDim.x * blockDim.x) {
    const int k8 = (int)(id % K8);
    const int64_t row = id / K8;
    const int i = (int)(row & 31);
    const int64_t j = row >> 5;
    const int k = k8 * 8;
    const f32x4* p = reinterpret_cast<const f32x4*>(x + row * D + k);
    const f32x4 v0 = p[0], v1 = p[1];
    h8 hh, ll;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      _Float16 a0, b0, a1, b1;
      split_hl(v0[e] * sc, a0, b0);
      split_hl(v1[e] * sc, a1, b1);
      hh[e] = a0; ll[e] = b0; hh[4 + e] = a1; ll[4 + e] = b1;
    }
    const int kb = k / 32, lane = (i & 15) + 16 * ((k & 31) >> 3);
    const int64_t piece = ((j * 2 + (i >> 4)) * KB + kb) * 2;
    reinterpret_cast<h8*>(image)[(piece + 0) * 64 + lane] = hh;
    reinterpret_cast<h8*>(image)[(piece + 1) * 64 + lane] = ll;
  }
}

__global__ __launch_bounds__(256) void hl_pack_cols_kernel(const float* __restrict__ q, int Q, int D,
                                                           _Float16* __restrict__ image, int32_t* __restrict__ qexp) {
  const int qi = blockIdx.x, tid = threadIdx.x;
  const int KB = D / 32, K8 = D / 8;
  __shared__ float red[4];
  __shared__ int e_s;
  const bool live = qi < Q;
  const float* row = q + (int64_t)qi * D;
  float m = 0.f;
  if (live)
    for (int i = tid; i < D / 4; i += blockDim.x) {
      const f32x4 v = reinterpret_cast<const f32x4*>(row)[i];
      m = fmaxf(fmaxf(fmaxf(m, fabsf(v.x)), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o, 64));
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  if (tid == 0) {
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) m = fmaxf(m, red[i]);
    e_s = hl_exponent(m);
    if (live) qexp[qi] = e_s;
  }
  __syncthreads();
  const float sc = ldexpf(1.0f, e_s);
  const int chunk = qi / HL_GQC, qq = qi % HL_GQC;
  for (int k8 = tid; k8 < K8; k8 += blockDim.x) {
    const int k = k8 * 8;
    f32x4 v0 = (f32x4){0.f, 0.f, 0.f, 0.f}, v1 = v0;
    if (live) {
      v0 = reinterpret_cast<const f32x4*>(row + k)[0];
      v1 = reinterpret_cast<const f32x4*>(row + k)[1];
    }
    h8 hh, ll;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      _Float16 a0, b0, a1, b1;
      split_hl(v0[e] * sc, a0, b0);
      split_hl(v1[e] * sc, a1, b1);
      hh[e] = a0; ll[e] = b0; hh[4 + e] = a1; ll[4 + e] = b1;
    }
    const int kb = k / 32, ct = qq / 16, lane = (qq & 15) + 16 * ((k & 31) >> 3);
    const int64_t piece = ((((int64_t)chunk * KB + kb) * HL_CT + ct) * 2);
    reinterpret_cast<h8*>(image)[(piece + 0) * 64 + lane] = hh;
    reinterpret_cast<h8*>(image)[(piece + 1) * 64 + lane] = ll;
  }
}

extern "C" int64_t qpg_hl_rows_bytes(int64_t R, int D) {
  return (R <= 0 || D <= 0 || D > HL_F_MAX || R > (int64_t(1) << 40)) ? 0 : R * D * 4 + 64;
}
extern "C" int64_t qpg_hl_cols_bytes(int Q, int D) {
  if (Q <= 0 || D <= 0 || D > HL_F_MAX) return 0;
  const int64_t chunks = ((int64_t)Q + HL_GQC - 1) / HL_GQC;
  return chunks * (D / 32) * HL_CT * 2 * HL_PIECE + chunks * HL_GQC * 4;
}

extern "C" int qpg_hl_pack_rows(qpg_ctx* ctx, void* stream, const float* x, int64_t R, int D, void* image,
                                int64_t image_bytes) {
  const char* name = "qpg_hl_pack_rows";
  QPG_REQUIRE(ctx && x && image && R > 0 && (R % 32) == 0 && D > 0 && (D % 128) == 0, "%s: needs R %% 32 == 0, D %% 128 == 0",
              name);
  QPG_REQUIRE(image_bytes >= qpg_hl_rows_bytes(R, D) && (reinterpret_cast<uintptr_t>(image) % 16) == 0 &&
                  (reinterpret_cast<uintptr_t>(x) % 16) == 0,
              "%s: image too small or misaligned (qpg_hl_rows_bytes)", name);
  hipStream_t st = qpg_stream(stream);
  unsigned char* img = static_cast<unsigned char*>(image);
  int32_t* meta = reinterpret_cast<int32_t*>(img + (qpg_hl_rows_bytes(R, D) - 64));
  unsigned int* amax = reinterpret_cast<unsigned int*>(meta + 4);
  hipLaunchKernelGGL(hl_zero_u32_kernel, dim3(1), dim3(1), 0, st, amax);
  hipLaunchKernelGGL(hl_absmax_kernel, dim3(1024), dim3(1024), 0, st, x, R * D, amax);
  hipLaunchKernelGGL(hl_exponent_kernel, dim3(1), dim3(1), 0, st, (const unsigned int*)amax, meta);
  hipLaunchKernelGGL(hl_pack_rows_kernel, dim3(4096), dim3(256), 0, st, x, R, D, (const int32_t*)meta,
                     reinterpret_cast<_Float16*>(img));
  QPG_LAUNCH_CHECK("hl_pack_rows_kernel");
  return QPG_OK;
}

extern "C" int qpg_hl_pack_cols(qpg_ctx* ctx, void* stream, const float* q, int Q, int D, void* image, int64_t image_bytes) {
  const char* name = "qpg_hl_pack_cols";
  QPG_REQUIRE(ctx && q && image && Q > 0 && D > 0 && (D % 128) == 0, "%s: bad argument (D %% 128 == 0)", name);
  QPG_REQUIRE(image_bytes >= qpg_hl_cols_bytes(Q, D) && (reinterpret_cast<uintptr_t>(image) % 16) == 0 &&
                  (reinterpret_cast<uintptr_t>(q) % 16) == 0,
              "%s: image too small or misaligned (qpg_hl_cols_bytes)", name);
  const int chunks = (Q + HL_GQC - 1) / HL_GQC;
  unsigned char* img = static_cast<unsigned char*>(image);
  int32_t* qexp = reinterpret_cast<int32_t*>(img + (int64_t)chunks * (D / 32) * HL_CT * 2 * HL_PIECE);
  hipLaunchKernelGGL(hl_pack_cols_kernel, dim3(chunks * HL_GQC), dim3(256), 0, qpg_stream(stream), q, Q, D,
                     reinterpret_cast<_Float16*>(img), qexp);
  QPG_LAUNCH_CHECK("hl_pack_cols_kernel");
  return QPG_OK;
}

// ---- round 6: the query side of a prefilter + by-code batch in ONE launch ------------------------------------------------
// cfg-3's step spent ~40 of its 300 us in three tiny launches in front of the GEMM - sklearn's normalisation (16 blocks: four
// lanes per row walking NumPy-einsum's chains), the split-f16 column image, the chain-permuted copy the by-code select reads
// - and the gaps between them.  One block per query does all three: the norm by four lanes in the reference's order
// (l2_normalize_rows_kernel's code, bit for bit), the normalised row through LDS, then qn (optional), the column image
// (hl_pack_cols_kernel's layout and exponent) and the permuted row (perm32_kernel's layout).  Padding queries of the last
// chunk of 96 write zero fragments, as hl_pack_cols_kernel does.
__global__ __launch_bounds__(128) void hl_prepare_queries_kernel(const float* __restrict__ q, int Q, int D,
                                                                 float* __restrict__ qn, _Float16* __restrict__ image,
                                                                 int32_t* __restrict__ qexp, float* __restrict__ qperm) {
  extern __shared__ __attribute__((aligned(16))) float rowbuf[];         // [D] the raw row, then the normalised row
  __shared__ float n_s, red[2];
  __shared__ int e_s;
  const int qi = blockIdx.x, tid = threadIdx.x;
  const bool live = qi < Q;
  const float* p = q + (int64_t)(live ? qi : 0) * D;
  // the raw row through LDS first (one coalesced round trip for the whole block): the four chain lanes below then walk LDS,
  // not four batches of strided global loads - the same additions in the same order
  for (int e = tid * 4; e < D; e += 128 * 4) *reinterpret_cast<f32x4*>(rowbuf + e) = *reinterpret_cast<const f32x4*>(p + e);
  __syncthreads();
  if (tid < 4) {                                       // the norm, in NumPy einsum's order (lane chains l = 0..3)
    const int l = tid;
    const float* pr = rowbuf;
    float a = 0.f;
    const int nfull = D >> 4;
    for (int g = 0; g < nfull; ++g) {
#pragma unroll
      for (int u = 3; u >= 0; --u) {
        const float v = pr[g * 16 + u * 4 + l];
        a = f_add(f_mul(v, v), a);
      }
    }
    for (int i = nfull * 16; i < D; i += 4) {
      const float v = (i + l < D) ? pr[i + l] : 0.f;
      a = f_add(f_mul(v, v), a);
    }
    const float o1 = __shfl_xor(a, 1, 64);
    const float pair = f_add(a, o1);
    const float o2 = __shfl_xor(pair, 2, 64);
    float n = f_sqrt(f_add(pair, o2));
    if (n < 10.f * 1.1920928955078125e-07f) n = 1.f;   // sklearn _handle_zeros_in_scale
    if (l == 0) n_s = n;
  }
  __syncthreads();
  const float n = n_s;
  float m = 0.f;
  for (int e = tid; e < D; e += 128) {
    const float v = live ? f_div(rowbuf[e], n) : 0.f;
    rowbuf[e] = v;                                     // (element e is read and written by this thread only)
    if (live && qn) qn[(int64_t)qi * D + e] = v;
    m = fmaxf(m, fabsf(v));
  }
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o, 64));
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  if (tid == 0) {
    m = fmaxf(red[0], red[1]);
    e_s = hl_exponent(m);
    if (live) qexp[qi] = e_s;
  }
  __syncthreads();
  const float sc = ldexpf(1.0f, e_s);
  const int KB = D / 32, K8 = D / 8;
  const int chunk = qi / HL_GQC, qq = qi % HL_GQC;
  for (int k8 = tid; k8 < K8; k8 += 128) {
    const int k = k8 * 8;
    h8 hh, ll;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      _Float16 a0, b0;
      split_hl(rowbuf[k + e] * sc, a0, b0);
      hh[e] = a0;
      ll[e] = b0;
    }
    const int kb = k / 32, ct = qq / 16, lane = (qq & 15) + 16 * ((k & 31) >> 3);
    const int64_t piece = ((((int64_t)chunk * KB + kb) * HL_CT + ct) * 2);
    reinterpret_cast<h8*>(image)[(piece + 0) * 64 + lane] = hh;
    reinterpret_cast<h8*>(image)[(piece + 1) * 64 + lane] = ll;
  }
  if (live && qperm) {
    // y[32 G + 8 k + j] = x[16 (2 G + (j >> 2)) + 4 (3 - (j & 3)) + k]   (perm32_kernel, csrc/qpg_sorted.hip)
    float* y = qperm + (int64_t)qi * D;
    for (int e = tid; e < D; e += 128) {
      const int G = e >> 5, k = (e & 31) >> 3, j = e & 7;
      y[e] = rowbuf[16 * (2 * G + (j >> 2)) + 4 * (3 - (j & 3)) + k];
    }
  }
}

extern "C" int qpg_hl_prepare_queries(qpg_ctx* ctx, void* stream, const float* q, int Q, int D, float* qn, void* image,
                                      int64_t image_bytes, float* qperm) {
  const char* name = "qpg_hl_prepare_queries";
  QPG_REQUIRE(ctx && q && image && Q > 0 && D > 0 && (D % 128) == 0 && D <= 8192, "%s: bad argument (D %% 128 == 0, D <= 8192)",
              name);
  QPG_REQUIRE(image_bytes >= qpg_hl_cols_bytes(Q, D) && (reinterpret_cast<uintptr_t>(image) % 16) == 0 &&
                  (reinterpret_cast<uintptr_t>(q) % 16) == 0,
              "%s: image too small or misaligned (qpg_hl_cols_bytes; q and the image 16-byte aligned)", name);
  QPG_REQUIRE(q != qn && q != qperm && (qn == nullptr || qn != qperm), "%s: outputs must not alias the input", name);
  const int chunks = (Q + HL_GQC - 1) / HL_GQC;
  unsigned char* img = static_cast<unsigned char*>(image);
  int32_t* qexp = reinterpret_cast<int32_t*>(img + (int64_t)chunks * (D / 32) * HL_CT * 2 * HL_PIECE);
  hipLaunchKernelGGL(hl_prepare_queries_kernel, dim3(chunks * HL_GQC), dim3(128), (size_t)D * 4, qpg_stream(stream), q, Q, D,
                     qn, reinterpret_cast<_Float16*>(img), qexp, qperm);
  QPG_LAUNCH_CHECK("hl_prepare_queries_kernel");
  return QPG_OK;
}

// ---- the prefilter GEMM on 32-ROW wave tiles (round 4): hl_gemm32_kernel --------------------------------------------------
// hl_gemm16_kernel inherits round 3's sweep organisation (16 rows x 96 columns per wave, f64 block sums).  For the
// prefilter of the exact-f32 cosine family the band is dominated by sklearn's own rounding (8.6e-5 at D = 512), so the
// h h' products may stay in the MFMA's f32 accumulator for the whole (short) K: a chain of KB instructions is within
// (kappa_1 + KB) 2^-24 sum|products| of the exact sum (every instruction: its own block error + one rounding of the running
// sum), QPG_HL_GEMM32_ERR(D) = (12 + D/32) 2^-24 + 5.2e-7 (cross terms, representation, f32 store: as §4.1) = 2.2e-6 at
// D = 512 (sorted_rows.gemm32_err) - and without the f64 sums a wave holds TWO row tiles: 32 rows x 96 columns, 36 MFMAs
// per 12 KB of fragment reads (half the LDS traffic per MFMA).  Block = 8 waves = 256 rows; query stages of 2 k-blocks
// (24 KB), double-buffered.
// What made the first versions of this kernel no faster than the 16-row one (experiments/gemm32/README.md): with K of
// only 12-16 k-blocks a (row block, chunk) item is ~9 us of MFMAs, and the rows' fragments were requested ONE k-block
// (0.3 us) ahead - every k-block waited for L2 / the Infinity Cache.  Now: persistent blocks over contiguous ranges of
// items, and a ring of four k-blocks whose slots are refilled, as soon as a k-block is done, with the same k-block of the
// NEXT stage (or of the next item's first stage): three k-blocks = 108 MFMAs per wave of lead.
#define G32_KS 2           // k-blocks per LDS stage of the query image (24 KB)
#define G32_RING 4         // k-blocks of row fragments in registers: two stages
#ifndef G32_PD
#define G32_PD 2           // steps between a column tile's fragment read and its use
#endif
template <int CT>
__global__ __launch_bounds__(512, 2) void hl_gemm32_kernel(HlArgs a, int n_items) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // 2 x G32_KS x 6 x 2 x 1 KB = 48 KB
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  // Work items (row block rb of 256 rows, query chunk c), id = rb * chunks + c.  XCD-AWARE order: workgroup b runs on XCD
  // b % 8 (observed dispatch order), XCD x owns the row blocks rb % 8 == x, and the k-th item of its l-th block is entry
  // k * (blocks per XCD) + l of the XCD's list (row block major, chunk minor) - at any time the CUs of an XCD work on
  // ~3 row blocks x all chunks, so a row block's 256 KB of fragments are fetched from HBM once and hit that XCD's L2 for
  // the other chunks.  (Contiguous item ranges per block - every CU on its own row block - made the 32 panels of an XCD
  // twice its L2: 2.1 GB per cfg-3 step came out of the Infinity Cache and the GEMM sat at 0.40-0.44 ms.)
  const int nb = (int)gridDim.x, nx = (nb % 8 == 0) ? 8 : 1;
  const int xcd = (int)blockIdx.x % nx, lx = (int)blockIdx.x / nx, lpx = nb / nx;
  const int n_rb = n_items / a.chunks;
  const int nit = ((n_rb - xcd + nx - 1) / nx) * a.chunks;             // items of this XCD
  auto vid = [&](int k) {
    const int i = k * lpx + lx;
    return i < nit ? ((i / a.chunks) * nx + xcd) * a.chunks + i % a.chunks : -1;
  };
  const int it0 = vid(0);
  if (it0 < 0) return;
  const int KB = a.KB, n_stage = KB / G32_KS;                          // (KB % 4 == 0: an even number of stages)
  const int cg = lane & 15, rg = lane >> 4;
  const int e_c1 = a.meta[0];
  constexpr int stage_units = G32_KS * HL_CT * 2 * 64;                 // h8 units per stage
  constexpr int QLD = stage_units / 512;
  h8 qreg[QLD];
  auto load_q = [&](int item, int s) {
    const h8* qsrc = reinterpret_cast<const h8*>(a.qi) + (int64_t)(item % a.chunks) * KB * HL_CT * 2 * 64;
#pragma unroll
    for (int u = 0; u < QLD; ++u) qreg[u] = qsrc[(int64_t)s * stage_units + u * 512 + tid];
  };
  auto store_q = [&](int buf) {
    h8* dst = reinterpret_cast<h8*>(lds) + buf * stage_units;
#pragma unroll
    for (int u = 0; u < QLD; ++u) dst[u * 512 + tid] = qreg[u];
  };
  auto lds_barrier = [&]() {                 // LDS-only: the rows' fragment loads stay in flight across it
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
  // rows of this wave in item `item`: 32-row group j; fragments [tile][kb][plane][64 lanes]; groups past N: the zero page
  const h8* dbp;
  int64_t t_step;
  int kb_step, pl_step, j;
  bool ok;
  auto set_rows = [&](int item) {
    j = (item / a.chunks) * 8 + w;
    ok = j < a.N;
    dbp = ok ? reinterpret_cast<const h8*>(a.db) + (int64_t)j * 2 * KB * 2 * 64 + lane : reinterpret_cast<const h8*>(a.zeros);
    t_step = ok ? (int64_t)KB * 2 * 64 : 0;
    kb_step = ok ? 128 : 0;
    pl_step = ok ? 64 : 0;
  };
  auto load_a = [&](int kb, h8 (&d)[2][2]) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) d[t][pl] = dbp[t * t_step + (int64_t)kb * kb_step + pl * pl_step];
  };
  f32x4 hh[2][CT], xx[2][CT];
  h8 A[G32_RING][2][2];                      // [ring slot][row tile][plane]
  h8 Bq[G32_PD + 1][2];                      // ring: a column tile's fragments are read G32_PD steps ahead
  auto ld_b = [&](int buf, int k2, int c, h8 (&d)[2]) {
    const h8* qb = reinterpret_cast<const h8*>(lds) + buf * stage_units + lane;
    d[0] = qb[((k2 * HL_CT + c) * 2 + 0) * 64];
    d[1] = qb[((k2 * HL_CT + c) * 2 + 1) * 64];
  };
  set_rows(it0);
  load_q(it0, 0);
#pragma unroll
  for (int i = 0; i < G32_RING; ++i) load_a(i, A[i]);
  store_q(0);
  __syncthreads();
  constexpr int NS = G32_KS * CT;            // steps of a stage: (k-block, column tile); 6 MFMAs each
  static_assert((2 * NS) % (G32_PD + 1) == 0, "the fragment ring index must be static over a two-stage trip");
#pragma unroll
  for (int i = 0; i < G32_PD; ++i) ld_b(0, i / CT, i % CT, Bq[i]);
  for (int kk = 0;; ++kk) {
    const int item = vid(kk);
    if (item < 0) break;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        hh[t][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        xx[t][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    const int j_cur = j;
    const bool ok_cur = ok;
    const int nitem = vid(kk + 1) >= 0 ? vid(kk + 1) : item;          // (behind the last item: harmless re-reads)
    for (int s2 = 0; s2 < n_stage; s2 += 2) {                          // two stages per trip: ring / buffer indices static
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) {
        const int s = s2 + ss;
        const bool last_s = s + 1 == n_stage;
        load_q(last_s ? nitem : item, last_s ? 0 : s + 1);            // in flight underneath this stage's MFMAs
#pragma unroll
        for (int st = 0; st < NS; ++st) {
          const int k2 = st / CT, c = st % CT;
          h8 (&Ac)[2][2] = A[ss * G32_KS + k2];
          h8 (&Bc)[2] = Bq[(ss * NS + st) % (G32_PD + 1)];
          h8 (&Bn)[2] = Bq[(ss * NS + st + G32_PD) % (G32_PD + 1)];
          if (st == NS - G32_PD) {
            // the next stage's fragments go to the other buffer (last read one stage ago, whose barrier everybody
            // passed), one LDS-only barrier; from here on the reads go to the next stage's buffer
            store_q((ss + 1) & 1);
            lds_barrier();
          }
          if (st >= NS - G32_PD) ld_b((ss + 1) & 1, (st + G32_PD - NS) / CT, (st + G32_PD - NS) % CT, Bn);
          else ld_b(ss, (st + G32_PD) / CT, (st + G32_PD) % CT, Bn);
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            hh[t][c] = mfma_h(Ac[t][0], Bc[0], hh[t][c]);
            xx[t][c] = mfma_h(Ac[t][0], Bc[1], xx[t][c]);
            xx[t][c] = mfma_h(Ac[t][1], Bc[0], xx[t][c]);
          }
          if (c == CT - 1) {                 // this k-block is done: its slot takes the k-block FOUR further on - of this
            const int nk = s * G32_KS + k2 + G32_RING;          // item, or (behind its last k-block) of the next item
            if (nk < KB) {
              load_a(nk, Ac);
            } else {
              if (nk == KB) set_rows(nitem);
              load_a(nk - KB, Ac);
            }
          }
#pragma unroll
          for (int i = 0; i < 6; ++i) {       // issue order: an MFMA, a fragment read underneath it
            HL_SGB(0x008, 1);
            if (i < 2) HL_SGB(0x100, 1);
          }
        }
      }
    }
    // epilogue of the item: d = 1 - (hh + 2^-11 xx) 2^-(e_c + e_q); lane (cg, rg) holds rows 4 rg .. 4 rg + 3 of column cg
    if (!ok_cur) continue;
    const int chunk = item % a.chunks;
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      const int q = chunk * (16 * HL_CT) + c * 16 + cg;
      if (q >= a.Q) continue;
      const int e_q1 = a.qexp[q];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          o[r] = (float)(1.0 - ldexp((double)hh[t][c][r] + (double)xx[t][c][r] * (1.0 / 2048.0), -(e_c1 + e_q1)));
        if (a.D)
          *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.D) + (int64_t)q * a.ldD + (int64_t)j_cur * 32 + 16 * t + 4 * rg) = o;
        if (a.tmin) {
          float m = fminf(fminf(o[0], o[1]), fminf(o[2], o[3]));
          m = fminf(m, xor_lanes_f<16>(m));
          m = fminf(m, xor_lanes_f<32>(m));
          if (rg == 0) a.tmin[(int64_t)q * a.ldT + (int64_t)j_cur * 2 + t] = m;
          if (a.tmask) {
            const float lim = m + a.band;
            unsigned int bits = ((o[0] <= lim) ? 1u : 0u) | ((o[1] <= lim) ? 2u : 0u) | ((o[2] <= lim) ? 4u : 0u) |
                                ((o[3] <= lim) ? 8u : 0u);
            bits <<= 4 * rg;
            bits |= (unsigned int)lane_xor<16>((int)bits);
            bits |= (unsigned int)lane_xor<32>((int)bits);
            if (rg == 0) a.tmask[(int64_t)q * a.ldT + (int64_t)j_cur * 2 + t] = (uint16_t)bits;
          }
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");       // surplus prefetches must not outlive their registers
}

// ---- the prefilter GEMM on the h PLANES ALONE, 64-row wave tiles (round 5): hl_gemm64h_kernel ------------------------------
// For the exact-f32 cosine family the prefilter only has to be BOUNDED: dropping both cross terms (h l' + l h') and l l'
// costs at most (2 x 2^-11 + 2^-22) |x||q| (h = fl16(x): relative error <= 2^-11 per element in the scaled normal range,
// Cauchy-Schwarz over the row) = 9.8e-4 for unit vectors - the band widens from 8.8e-5 to 2.1e-3, which on BASELINE
// configs[2]'s rows (nearest-neighbour gaps ~1.6e-2 per code) lists ~12 % more pairs for the exact-order evaluation, and
// the GEMM issues ONE v_mfma_f32_16x16x32_f16 per 16 x 16 x 32 block instead of three and reads half the row image.
// With the cross-term accumulators gone a wave holds FOUR row tiles x six column tiles (64 x 96, 96 accumulator
// registers): one 1 KB query fragment from LDS feeds four MFMAs (gemm32: 2 KB per six), every accumulator is revisited
// six steps later (no dependent MFMAs back to back).  Block = 8 waves = 512 rows; query stages of FOUR k-blocks of the h
// plane (24 KB), double-buffered; the ring of row fragments is one stage deep (four k-blocks x four tiles), a k-block's
// slot refilled with the same k-block of the next stage as soon as its last column tile is done.  Items, XCD-aware order
// and the persistent blocks are hl_gemm32_kernel's.  Output: tile minima and row masks only, TILE-MAJOR
// ([tile][a.ldT], a.ldT >= Q: the 16 queries of a column tile are one 64-byte store; the by-code select reads a tile's
// queries contiguously).  Needs KB % 4 == 0 (D % 128 == 0) and an even number of 32-row groups.
#define G64_KS 4
// ablation hooks (experiments/gemm32, experiments/round_scripts/r05_probe_gemm64.sh): -DG64_PROBE=<bits>; the product build defines nothing.
// 1: no epilogue (the accumulators are only kept alive); 2: row fragments from one address (no row stream); 4: no query
// staging inside the k loop; 8: no MFMAs (loads, staging and epilogue stay); 16: no query-fragment reads from LDS
#ifndef G64_PROBE
#define G64_PROBE 0
#endif
#ifndef G64_PD
#define G64_PD 2          // steps between a column tile's fragment read and its use (3 and 5 measured: see DESIGN 4.4)
#endif
// PAIR: two stages per trip of the k loop, LDS buffer indices static (an even number of stages: D % 256 == 0, configs[2]);
// !PAIR: one stage per trip, the buffer index carried in a register (any D % 128 == 0; hipcc's over-tight wait at the loop
// head - see DESIGN.md 4.4 - is then paid per stage instead of per two)
// NW: waves per block (= 64-row groups per item).  8: one block per CU; 4: TWO blocks per CU, so that the two waves of a SIMD
// belong to different blocks and one's epilogue (~2 800 issue cycles per item against 6 144 cycles of MFMAs) runs under the
// other's MFMAs - with eight waves per block both waves of a SIMD reach their epilogues together.
template <int CT, bool PAIR, int NW>
__global__ __launch_bounds__(64 * NW, 2) void hl_gemm64h_kernel(HlArgs a, int n_items) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // 2 x G64_KS x CT x 1 KB = 48 KB
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);              // (scalar: the rows' base addresses stay in SGPRs)
  const int nb = (int)gridDim.x, nx = (nb % 8 == 0) ? 8 : 1;
  const int xcd = (int)blockIdx.x % nx, lx = (int)blockIdx.x / nx, lpx = nb / nx;
  const int n_rb = n_items / a.chunks;
  const int nit = ((n_rb - xcd + nx - 1) / nx) * a.chunks;             // items of this XCD
  // This block's items: i_k = k lpx + lx < nit, item = (row block (i / chunks) nx + xcd, chunk i % chunks).  The quotient
  // and remainder are carried from item to item (one division pair per BLOCK; with `item` as a number the loop held ~12
  // run-time integer divisions per item.  Measured equal - the 34 us of the kernel's "skeleton" in experiments/round_scripts/r05_probe_gemm64.sh
  // are its 64 fragment loads per wave and item going through the CU's 64 B/clk vector L1, not index arithmetic).
  struct Item {
    int i, q, rb, ch;                        // sequence number on this XCD, its quotient by chunks, row block (global), chunk
  };
  const int dq = lpx / a.chunks, dr = lpx % a.chunks;
  auto next_item = [&](Item t) {
    t.i += lpx;
    t.q += dq;
    t.ch += dr;
    if (t.ch >= a.chunks) {
      t.ch -= a.chunks;
      ++t.q;
    }
    t.rb = t.q * nx + xcd;
    return t;
  };
  if (lx >= nit) return;
  Item it_cur{lx, lx / a.chunks, (lx / a.chunks) * nx + xcd, lx % a.chunks};
  const int KB = a.KB, n_stage = KB / G64_KS;                          // (KB % 4 == 0; the LDS buffer of a stage: s & 1, dynamic)
  const int cg = lane & 15, rg = lane >> 4;
  const int e_c1 = a.meta[0];
  constexpr int stage_units = G64_KS * CT * 64;                        // h8 units per stage (h plane only)
  constexpr int NT = 64 * NW;
  constexpr int QLD = stage_units / NT;
  static_assert(stage_units % NT == 0, "a stage is a whole number of 16-byte units per thread");
  h8 qreg[QLD];
  uint32_t qoff[QLD];                                                  // (v / 64) * 2 KB + (v % 64) * 16 B: the h piece's unit
#pragma unroll
  for (int u = 0; u < QLD; ++u) {
    const uint32_t v = (uint32_t)(u * NT + tid);
    qoff[u] = (v >> 6) * 2048u + (v & 63u) * 16u;
  }
  auto load_q = [&](int chunk, int s) {                                // the h pieces of the stage's (k-block, column tile)s
    const unsigned char* qsrc = reinterpret_cast<const unsigned char*>(a.qi) +
                                ((int64_t)chunk * KB + (int64_t)s * G64_KS) * CT * 2048;
#pragma unroll
    for (int u = 0; u < QLD; ++u) qreg[u] = *reinterpret_cast<const h8*>(qsrc + qoff[u]);
  };
  auto store_q = [&](int buf) {
    h8* dst = reinterpret_cast<h8*>(lds) + buf * stage_units;
#pragma unroll
    for (int u = 0; u < QLD; ++u) dst[u * NT + tid] = qreg[u];
  };
  auto lds_barrier = [&]() {                 // LDS-only: the rows' fragment loads stay in flight across it
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
  // rows of this wave in an item: 64-row group jj = two 32-row groups = four 16-row tiles, [tile][kb][plane][64 lanes];
  // a group past the image reads the zero page (all strides 0).  Everything here is wave-uniform: addresses = a byte
  // base in scalar registers + one 32-bit byte offset per lane.
  typedef const __attribute__((address_space(1))) unsigned char* gbytes_t;   // (global: survives the register pins below)
  typedef const __attribute__((address_space(1))) h8* gh8_t;
  struct Rows {
    gbytes_t base;
    uint32_t t_step, kb_step;                 // bytes
    int jj;
    bool ok;
  };
  auto rows_of = [&](int rb) {
    Rows r;
    r.jj = rb * NW + w;
    r.ok = 2 * r.jj < a.N;
    r.base = r.ok ? (gbytes_t)(reinterpret_cast<const unsigned char*>(a.db) + (int64_t)r.jj * 4 * KB * 2048)
                  : (gbytes_t)reinterpret_cast<const unsigned char*>(a.zeros);
    r.t_step = (r.ok && !(G64_PROBE & 2)) ? (uint32_t)KB * 2048u : 0u;
    r.kb_step = (r.ok && !(G64_PROBE & 2)) ? 2048u : 0u;
    if (G64_PROBE & 2) r.base = (gbytes_t)reinterpret_cast<const unsigned char*>(a.db);
    // (pinned in scalar registers HERE: left alone, hipcc sinks this arithmetic - a division and selects, i.e. branches -
    // into the stage's basic block, next to the refill loads)
    asm volatile("" : "+s"(r.base), "+s"(r.t_step), "+s"(r.kb_step));
    return r;
  };
  const uint32_t loff = (uint32_t)lane * 16u;
  auto load_a = [&](gbytes_t base, uint32_t t_step, uint32_t off, h8 (&d)[4]) {
#pragma unroll
    for (int t = 0; t < 4; ++t) d[t] = *(gh8_t)(base + ((uint32_t)t * t_step + off + loff));
  };
  f32x4 hh[4][CT];
  h8 A[G64_KS][4];                           // [ring slot = k-block of the stage][row tile]
  h8 Bq[G64_PD + 1];                         // ring: a column tile's fragment is read G64_PD steps ahead
  auto ld_b = [&](int buf, int k2, int c, h8& d) {
    d = (reinterpret_cast<const h8*>(lds) + buf * stage_units + lane)[(k2 * CT + c) * 64];
  };
  Rows cur = rows_of(it_cur.rb);
  load_q(it_cur.ch, 0);
#pragma unroll
  for (int i = 0; i < G64_KS; ++i) load_a(cur.base, cur.t_step, (uint32_t)i * cur.kb_step, A[i]);
  store_q(0);
  __syncthreads();
  constexpr int NS = G64_KS * CT;            // steps of a stage: (k-block, column tile); 4 MFMAs each
  static_assert((2 * NS) % (G64_PD + 1) == 0 && NS % (G64_PD + 1) == 0, "the fragment ring index must be static");
#pragma unroll
  for (int i = 0; i < G64_PD; ++i) ld_b(0, i / CT, i % CT, Bq[i]);
  if (G64_PROBE & 16) ld_b(0, 0, 0, Bq[G64_PD]);
  auto epilogue = [&](const Rows& done, int chunk) {
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      const int q = chunk * (16 * CT) + c * 16 + cg;
      if (q >= a.Q) continue;
      const float sc = ldexpf(1.0f, -(e_c1 + a.qexp[q]));            // (a power of two: the product below is exact)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = 1.0f - hh[t][c][r] * sc;
        float m = fminf(fminf(o[0], o[1]), fminf(o[2], o[3]));
        m = fminf(m, xor_lanes_f<16>(m));
        m = fminf(m, xor_lanes_f<32>(m));
        const float lim = m + a.band;
        unsigned int bits = ((o[0] <= lim) ? 1u : 0u) | ((o[1] <= lim) ? 2u : 0u) | ((o[2] <= lim) ? 4u : 0u) |
                            ((o[3] <= lim) ? 8u : 0u);
        bits <<= 4 * rg;
        bits |= (unsigned int)lane_xor<16>((int)bits);
        bits |= (unsigned int)lane_xor<32>((int)bits);
        if (rg == 0) {
          const int64_t o_i = ((int64_t)done.jj * 4 + t) * a.ldT + q;
          a.tmin[o_i] = m;
          a.tmask[o_i] = (uint16_t)bits;
        }
      }
    }
  };
  int sbuf = 0;                                                        // LDS buffer of the next stage to run
  for (;;) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int c = 0; c < CT; ++c) hh[t][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    Item it_nxt = next_item(it_cur);
    const bool more = it_nxt.i < nit;
    if (!more) it_nxt = it_cur;                                         // (behind the last item: harmless re-reads)
    const Rows nxt = rows_of(it_nxt.rb);
    for (int s2 = 0; s2 < n_stage; s2 += PAIR ? 2 : 1) {               // ring slots and the fragment ring's indices are static
#pragma unroll
      for (int sp = 0; sp < (PAIR ? 2 : 1); ++sp) {
        const int s = s2 + sp;
        const int ss = PAIR ? sp : sbuf;                               // this stage's LDS buffer
        if (!PAIR) sbuf ^= 1;
        const bool last_s = s + 1 == n_stage;
        if (!(G64_PROBE & 4)) load_q(last_s ? it_nxt.ch : it_cur.ch, last_s ? 0 : s + 1);   // in flight underneath this stage's MFMAs
        // the stage's k-blocks are refilled with the same k-blocks of the next stage - of this item, or (behind its last
        // stage) of the next item's first stage: scalars chosen HERE, so that the stage stays ONE basic block
        gbytes_t rf_base = cur.base + (uint32_t)(s + 1) * G64_KS * cur.kb_step;
        uint32_t rf_t = cur.t_step, rf_k = cur.kb_step;
        if (last_s) {
          rf_base = nxt.base;
          rf_t = nxt.t_step;
          rf_k = nxt.kb_step;
        }
        asm volatile("" : "+s"(rf_base), "+s"(rf_t), "+s"(rf_k));
#pragma unroll
        for (int st = 0; st < NS; ++st) {
          const int k2 = st / CT, c = st % CT;
          h8 (&Ac)[4] = A[k2];
          h8& Bc = Bq[st % (G64_PD + 1)];                              // (NS % (G64_PD + 1) == 0)
          h8& Bn = Bq[(st + G64_PD) % (G64_PD + 1)];
          if (st == NS - G64_PD) {
            if (!(G64_PROBE & 4)) store_q(ss ^ 1);
            lds_barrier();
          }
          if (!(G64_PROBE & 16)) {
            if (st >= NS - G64_PD) ld_b(ss ^ 1, (st + G64_PD - NS) / CT, (st + G64_PD - NS) % CT, Bn);
            else ld_b(ss, (st + G64_PD) / CT, (st + G64_PD) % CT, Bn);
          }
          if (G64_PROBE & 8) {
#pragma unroll
            for (int t = 0; t < 4; ++t) asm volatile("" ::"v"(Ac[t]), "v"(Bc));
          } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) hh[t][c] = mfma_h(Ac[t], Bc, hh[t][c]);
          }
          if (c == CT - 1) load_a(rf_base, rf_t, (uint32_t)k2 * rf_k, Ac);      // this k-block is done: its slot is refilled
          // issue order (every class named, or hipcc sinks the loads to their uses and waits vmcnt(0) there): the
          // stage's query loads first; per step an MFMA, the fragment read underneath it, three MFMAs; the refill's
          // four loads behind the step that frees their registers; the LDS stores of the next stage at its barrier
          if (st == 0) HL_SGB(0x020, QLD);
          if (st == NS - G64_PD) HL_SGB(0x200, QLD);
          HL_SGB(0x008, 1);
          HL_SGB(0x100, 1);
          HL_SGB(0x008, 3);
          if (c == CT - 1) HL_SGB(0x020, 4);
        }
      }
    }
    // epilogue of the item: d = 1 - hh 2^-(e_c + e_q); lane (cg, rg) holds rows 4 rg .. 4 rg + 3 of column cg
    const Rows done = cur;
    const int chunk = it_cur.ch;
    cur = nxt;
    it_cur = it_nxt;
    if (done.ok) {
      if (G64_PROBE & 1) {
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
          for (int t = 0; t < 4; ++t) asm volatile("" ::"v"(hh[t][c]));
      } else {
        epilogue(done, chunk);
      }
    }
    if (!more) break;                        // (that was the last item)
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");       // surplus prefetches must not outlive their registers
}

// The h-plane prefilter (hl_gemm64h_kernel): tile minima + row masks, TILE-MAJOR: tile_min / tile_mask [R / 16][ldQ],
// ldQ >= Q.  `band` must cover QPG_HL_GEMM_H_ERR (sorted_rows.gemm_h_err).  Needs D % 256 == 0 and R % 64 == 0.
// Measurement hook (-DQPG_DEBUG_HOOKS builds only): waves per block of hl_gemm64h_kernel (8: one block per CU; 4: two).
QPG_HOOK_VAR(int, g_gemm64_nw, 4);
#ifdef QPG_DEBUG_HOOKS
extern "C" int qpg_debug_gemm64_waves(int nw) {
  QPG_REQUIRE(nw == 4 || nw == 8, "qpg_debug_gemm64_waves: 4 or 8");
  g_gemm64_nw = nw;
  return QPG_OK;
}
#endif

extern "C" int qpg_hl_gemm_tilemin_h(qpg_ctx* ctx, void* stream, const void* rows_image, int64_t R, int D,
                                     const void* cols_image, int Q, float band, float* tile_min, uint16_t* tile_mask,
                                     int64_t ldQ) {
  const char* name = "qpg_hl_gemm_tilemin_h";
  QPG_REQUIRE(ctx && rows_image && cols_image && tile_min && tile_mask, "%s: null pointer", name);
  QPG_REQUIRE(R > 0 && (R % 64) == 0 && R / 32 < 0x7fffffff && Q > 0 && D > 0 && (D % 128) == 0 && ldQ >= Q && band >= 0.f,
              "%s: bad size (R %% 64 == 0, D %% 128 == 0, ldQ >= Q, band >= 0)", name);
  const int chunks = (Q + HL_GQC - 1) / HL_GQC, KB = D / 32;
  HlArgs a;
  const unsigned char* ri = static_cast<const unsigned char*>(rows_image);
  const unsigned char* ci = static_cast<const unsigned char*>(cols_image);
  a.db = reinterpret_cast<const _Float16*>(ri);
  a.meta = reinterpret_cast<const int32_t*>(ri + (qpg_hl_rows_bytes(R, D) - 64));
  a.qi = reinterpret_cast<const _Float16*>(ci);
  a.qexp = reinterpret_cast<const int32_t*>(ci + (int64_t)chunks * KB * HL_CT * 2 * HL_PIECE);
  a.cn2 = nullptr; a.qn2 = nullptr; a.D = nullptr; a.zeros = ctx->zeros; a.ldD = 0; a.stats = nullptr;
  a.N = (int)(R / 32); a.j0 = 0; a.chunks = chunks; a.G = 0; a.Q = Q; a.KB = KB; a.d_f32 = 1; a.tmin = tile_min; a.ldT = ldQ;
  a.tmask = tile_mask; a.band = band;
  const size_t lds64 = 2 * (size_t)G64_KS * HL_CT * HL_PIECE;         // 48 KB
  static bool raised64 = false;
  if (!raised64) {
    bool ok = true;
#define G64_RAISE(P_, W_) ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(hl_gemm64h_kernel<HL_CT, P_, W_>), \
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds64) == hipSuccess
    G64_RAISE(true, 8); G64_RAISE(false, 8); G64_RAISE(true, 4); G64_RAISE(false, 4);
#undef G64_RAISE
    if (!ok) {
      qpg_set_error("%s: cannot raise the dynamic LDS limit", name);
      return QPG_EHIP;
    }
    raised64 = true;
  }
  const int nw = g_gemm64_nw;                                          // waves per block: 4 (two blocks per CU) or 8
  const int64_t g8 = (R / 64 + nw - 1) / nw;                           // row blocks of nw x 64 rows
  QPG_REQUIRE(g8 * chunks < 0x7fffffffll, "%s: too many work items", name);
  const int n_items = (int)(g8 * chunks);
  const int slots = ctx->n_cu * (8 / nw);
  const int n_blocks = n_items < slots ? n_items : slots;
  const bool pair = (KB / G64_KS) % 2 == 0;
#define G64_GO(P_, W_) hipLaunchKernelGGL((hl_gemm64h_kernel<HL_CT, P_, W_>), dim3(n_blocks), dim3(64 * W_), lds64, qpg_stream(stream), a, n_items)
  if (nw == 8) { if (pair) G64_GO(true, 8); else G64_GO(false, 8); }
  else { if (pair) G64_GO(true, 4); else G64_GO(false, 4); }
#undef G64_GO
  QPG_LAUNCH_CHECK("hl_gemm64h_kernel");
  return QPG_OK;
}

static int hl_gemm_impl(const char* name, qpg_ctx* ctx, void* stream, const void* rows_image, int64_t R, int D,
                        const void* cols_image, int Q, float* Dm, int64_t ldD, float* tile_min, int64_t ldT,
                        uint16_t* tile_mask, float band) {
  QPG_REQUIRE(ctx && rows_image && cols_image && (Dm || (tile_min && tile_mask)), "%s: null pointer", name);
  QPG_REQUIRE(R > 0 && (R % 32) == 0 && R / 32 < 0x7fffffff && Q > 0 && D > 0 && (D % 128) == 0 &&
                  (!Dm || (ldD >= R && (ldD % 4) == 0 && (reinterpret_cast<uintptr_t>(Dm) % 16) == 0)),
              "%s: bad size (R %% 32 == 0, D %% 128 == 0, ldD %% 4 == 0, 16-byte aligned output)", name);
  QPG_REQUIRE(!tile_min || ldT >= R / 16, "%s: tile_min needs ldT >= R / 16", name);
  QPG_REQUIRE(!tile_mask || (tile_min && band >= 0.f), "%s: tile masks need the tile minima and a band >= 0", name);
  const int chunks = (Q + HL_GQC - 1) / HL_GQC, KB = D / 32;
  HlArgs a;
  const unsigned char* ri = static_cast<const unsigned char*>(rows_image);
  const unsigned char* ci = static_cast<const unsigned char*>(cols_image);
  a.db = reinterpret_cast<const _Float16*>(ri);
  a.meta = reinterpret_cast<const int32_t*>(ri + (qpg_hl_rows_bytes(R, D) - 64));
  a.qi = reinterpret_cast<const _Float16*>(ci);
  a.qexp = reinterpret_cast<const int32_t*>(ci + (int64_t)chunks * KB * HL_CT * 2 * HL_PIECE);
  a.cn2 = nullptr; a.qn2 = nullptr; a.D = Dm; a.zeros = ctx->zeros; a.ldD = ldD; a.stats = nullptr;
  a.N = (int)(R / 32); a.j0 = 0; a.chunks = chunks; a.G = 0; a.Q = Q; a.KB = KB; a.d_f32 = 1; a.tmin = tile_min; a.ldT = ldT; a.tmask = tile_mask; a.band = band;
  // (a clip's 48 text queries stay on the 16-row kernel: it runs UNDER the audio sweep, where the slimmer kernel gets more
  // of the slots the sweep leaves - measured inside the step: 0.273-0.283 ms against 0.281-0.282 with the 32-row kernel,
  // also with three column tiles: experiments/gemm32/README.md)
  if ((KB % G32_RING) == 0 && Q > 48) {
    const size_t lds32 = 2 * (size_t)G32_KS * HL_CT * 2 * HL_PIECE;   // 48 KB
    static bool raised32 = false;
    if (!raised32) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(hl_gemm32_kernel<6>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds32) != hipSuccess) {
        qpg_set_error("%s: cannot raise the dynamic LDS limit", name);
        return QPG_EHIP;
      }
      raised32 = true;
    }
    const int64_t g8 = (a.N + 7) / 8;                                 // row blocks of 8 x 32 rows
    QPG_REQUIRE(g8 * chunks < 0x7fffffffll, "%s: too many work items", name);
    const int n_items = (int)(g8 * chunks);
    const int n_blocks = n_items < ctx->n_cu ? n_items : ctx->n_cu;   // one resident block per CU (96 KB of LDS)
    hipLaunchKernelGGL(hl_gemm32_kernel<6>, dim3(n_blocks), dim3(512), lds32, qpg_stream(stream), a, n_items);
    QPG_LAUNCH_CHECK("hl_gemm32_kernel");
    return QPG_OK;
  }
  const size_t lds_bytes = 2 * HL_KS * HL_CT * 2 * HL_PIECE;
  const int64_t rgroups = (a.N + HL_WPB - 1) / HL_WPB;
  QPG_REQUIRE(((rgroups + 7) / 8) * 8 * chunks < 0x7fffffffll, "%s: too many blocks", name);
  hipLaunchKernelGGL(hl_gemm16_kernel, dim3((unsigned)(((rgroups + 7) / 8) * 8 * chunks)), dim3(HL_THREADS),
                     lds_bytes, qpg_stream(stream), a);
  QPG_LAUNCH_CHECK("hl_gemm16_kernel");
  return QPG_OK;
}

extern "C" int qpg_hl_gemm_distance(qpg_ctx* ctx, void* stream, const void* rows_image, int64_t R, int D,
                                    const void* cols_image, int Q, float* Dm, int64_t ldD, float* tile_min, int64_t ldT) {
  return hl_gemm_impl("qpg_hl_gemm_distance", ctx, stream, rows_image, R, D, cols_image, Q, Dm, ldD, tile_min, ldT, nullptr,
                      0.f);
}

// The same GEMM WITHOUT its matrix (round 4): per (query, 16-row tile) the minimum and a 16-bit mask of the rows within
// `band` of it - all qpg_percode_select_sorted_f32 needs (tile_mask argument).  6 bytes per (query, tile) leave the
// kernel instead of 64: cfg-3's 375 MB prefilter matrix is never written or read.
extern "C" int qpg_hl_gemm_tilemin(qpg_ctx* ctx, void* stream, const void* rows_image, int64_t R, int D,
                                   const void* cols_image, int Q, float band, float* tile_min, uint16_t* tile_mask,
                                   int64_t ldT) {
  return hl_gemm_impl("qpg_hl_gemm_tilemin", ctx, stream, rows_image, R, D, cols_image, Q, nullptr, 0, tile_min, ldT,
                      tile_mask, band);
}

// ---- hardware probe: one v_mfma_f32_16x16x32_f16 per 16 x 16 tile of (A rows, B rows), C given -------------------------------
// a, b: [tiles][16][32] f16; c: [tiles][16][16] f32 (may be NULL = 0); out: [tiles][16][16] f32 = A . B^T + C as the
// matrix core computes it.  tests/test_gpu_audio_hl.py measures its error against exact sums (the kappa of the bound).
__global__ __launch_bounds__(64) void hl_probe_kernel(const _Float16* __restrict__ A, const _Float16* __restrict__ B,
                                                      const float* __restrict__ C, float* __restrict__ out) {
  const int tile = blockIdx.x, lane = threadIdx.x;
  const h8 a = reinterpret_cast<const h8*>(A + (int64_t)tile * 512 + (lane & 15) * 32)[lane >> 4];
  const h8 b = reinterpret_cast<const h8*>(B + (int64_t)tile * 512 + (lane & 15) * 32)[lane >> 4];
  f32x4 c = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (C)
#pragma unroll
    for (int r = 0; r < 4; ++r) c[r] = C[(int64_t)tile * 256 + (4 * (lane >> 4) + r) * 16 + (lane & 15)];
  const f32x4 d = mfma_h(a, b, c);
#pragma unroll
  for (int r = 0; r < 4; ++r) out[(int64_t)tile * 256 + (4 * (lane >> 4) + r) * 16 + (lane & 15)] = d[r];
}

extern "C" int qpg_probe_mfma_f16_tile(qpg_ctx* ctx, void* stream, const void* a, const void* b, const float* c, int tiles,
                                       float* out) {
  QPG_REQUIRE(ctx && a && b && out && tiles > 0, "qpg_probe_mfma_f16_tile: bad argument");
  hipLaunchKernelGGL(hl_probe_kernel, dim3(tiles), dim3(64), 0, qpg_stream(stream), static_cast<const _Float16*>(a),
                     static_cast<const _Float16*>(b), c, out);
  QPG_LAUNCH_CHECK("hl_probe_kernel");
  return QPG_OK;
}
