// Audio (WavLM) candidate sweep on the INT8 matrix cores, exact integer arithmetic.
//
// The f64 sweep (qpg_audio.hip) is bound by the f64 matrix pipe: 31.4 GFLOP per clip at 78.6 TFLOP/s is
// >= 0.4 ms whatever the tiling.  But the operands are f32, and a dot product of fixed-point numbers is an
// INTEGER computation that the int8 matrix cores (v_mfma_i32_16x16x64_i8, ~16x the f64 rate) do exactly:
//
//   * every 1024-element frame row gets one block exponent E (2^(E-1) <= max|x| < 2^E) and is rounded to
//     31-bit fixed point  X = rint(x * 2^(30-E)),  |X| <= 2^30   (error <= 2^(E-31) per element);
//   * X is split into four balanced base-256 digit planes  X = d0*256^3 + d1*256^2 + d2*256 + d3,
//     d in [-128,127], stored as int8 planes (4 B per element: the same HBM bytes as the f32 base);
//   * X.Y = sum_{s,t} 256^(6-s-t) <d_s, e_t>: the 13 digit-pair products with s+t <= 4 are int8 GEMMs
//     accumulated in i32 by diagonal u = s+t (|sum| < 2^27 per 1024-element tap, no overflow), the dropped
//     pairs (u >= 5) are below 2^-38 of full scale;
//   * per tap the five diagonals are recombined in f64 (exact powers of two) and scaled by the two block
//     exponents; the six taps of a candidate are summed in f64.
//
// The result D'[q][c] differs from the exact distance only by the fixed-point rounding (measured <= 2e-11,
// rigorous bound computed per query on the device), so the per-code minimum is found among D' and only
// the candidates within that bound of each minimum are re-evaluated with exact f64 arithmetic
// (refine kernels below): final distances and winners are those of the f64 path.
//
// Tiling: block = 4 waves = 64 consecutive candidates x 48 queries, every wave owns 16 candidates and walks
// the whole contraction (6 taps x 16 chunks of 64 features); the query digit tile of a chunk (4 planes x 48
// rows x 64 B) is shared by the 4 waves through LDS (double-buffered, rows padded to 80 B: conflict-free
// ds_read_b128), candidate digits stream straight from HBM (16 B per lane per plane).
#include <stdlib.h>

#include "qpg_common.h"   // from qpgesture_amd/csrc (see check_i8.py for the build line)

extern "C" int qpg_i8_slice_rows(qpg_ctx*, void*, const float*, int64_t, int, int8_t*, double*, double*);
extern "C" int qpg_audio_cosine_i8(qpg_ctx*, void*, const int8_t*, const double*, int, int, int, const int32_t*, int, int,
                                   int, const double*, const int8_t*, const double*, const double*, int, double*, int64_t);

typedef int v4i __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------
// digit planes of f32 rows:  planes[s][row][e] (int8), scale[row] = 2^(E-30) (0 for an all-zero row)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void i8_slice_rows_kernel(const float* __restrict__ x, int64_t rows, int F,
                                                            int8_t* __restrict__ planes, double* __restrict__ scale,
                                                            double* __restrict__ norm2) {
  __shared__ float red[4];
  __shared__ double redd[4];
  const int64_t r = blockIdx.x;
  const float* p = x + r * F;
  float m = 0.f;
  double s2 = 0.0;
  for (int e = threadIdx.x; e < F; e += 256) {
    const float v = p[e];
    m = fmaxf(m, fabsf(v));
    s2 += (double)v * (double)v;
  }
  for (int o = 32; o > 0; o >>= 1) {
    m = fmaxf(m, __shfl_xor(m, o, 64));
    s2 += __shfl_xor(s2, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    red[threadIdx.x >> 6] = m;
    redd[threadIdx.x >> 6] = s2;
  }
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  int E = 0;
  if (m > 0.f) E = ilogbf(m) + 1;            // 2^(E-1) <= m < 2^E
  if (threadIdx.x == 0) {
    scale[r] = (m > 0.f) ? ldexp(1.0, E - 30) : 0.0;
    if (norm2) norm2[r] = (redd[0] + redd[1]) + (redd[2] + redd[3]);
  }
  const int64_t plane = rows * (int64_t)F;
  for (int e = threadIdx.x; e < F; e += 256) {
    int X = (m > 0.f) ? (int)rintf(ldexpf(p[e], 30 - E)) : 0;   // exact scaling, one rounding to integer
    int d3 = ((X + 128) & 255) - 128; X = (X - d3) >> 8;
    int d2 = ((X + 128) & 255) - 128; X = (X - d2) >> 8;
    int d1 = ((X + 128) & 255) - 128; X = (X - d1) >> 8;
    const int64_t o = r * F + e;
    planes[o] = (int8_t)X;
    planes[plane + o] = (int8_t)d1;
    planes[2 * plane + o] = (int8_t)d2;
    planes[3 * plane + o] = (int8_t)d3;
  }
}

extern "C" int qpg_i8_slice_rows(qpg_ctx* ctx, void* stream, const float* x, int64_t rows, int F, int8_t* planes,
                                 double* scale, double* norm2) {
  QPG_REQUIRE(ctx && x && planes && scale && rows >= 0 && F > 0, "qpg_i8_slice_rows: bad argument");
  QPG_REQUIRE(rows < 0x7fffffffll, "qpg_i8_slice_rows: too many rows for one launch");
  if (rows == 0) return QPG_OK;
  hipLaunchKernelGGL(i8_slice_rows_kernel, dim3((unsigned)rows), dim3(256), 0, qpg_stream(stream), x, rows, F, planes,
                     scale, norm2);
  QPG_LAUNCH_CHECK("i8_slice_rows_kernel");
  return QPG_OK;
}

// ---------------------------------------------------------------------------------------------
#define I8_QT 48            // queries per block (3 MFMA column tiles)
#define I8_NT 3
#define I8_ROWB 80          // padded LDS row: 64 digit bytes + 16
#define I8_CH 64            // features per chunk (one MFMA K)

__device__ __forceinline__ double cosine_from_dot_i8(double dot, double qn2, double cn2) {
  const double tiny = 10.0 * 2.220446049250313e-16;
  const double nq = sqrt(qn2), nc = sqrt(cn2);
  const bool zq = nq < tiny, zc = nc < tiny;
  if (zq || zc) {
    const double a = zq ? qn2 : 1.0, b = zc ? cn2 : 1.0;
    return 0.5 * (a + b - 2.0 * (dot / ((zq ? 1.0 : nq) * (zc ? 1.0 : nc))));
  }
  return 1.0 - dot / (nq * nc);
}

template <int NTAPS>
__global__ __launch_bounds__(256) void audio_dot_i8_kernel(const int8_t* __restrict__ A, const double* __restrict__ sA,
                                                           int N, int T, int F, const int32_t* __restrict__ cand_t,
                                                           int G, int tap_stride, const double* __restrict__ cn2,
                                                           const int8_t* __restrict__ Bq,
                                                           const double* __restrict__ sQ,
                                                           const double* __restrict__ qn2, int Q,
                                                           double* __restrict__ D, int64_t ldD) {
  __shared__ __attribute__((aligned(16))) unsigned char bs[2][4][I8_QT][I8_ROWB];   // 30 KB
  const int64_t C = (int64_t)N * G;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int row = lane & 15, grp = lane >> 4;
  const int64_t c0 = (int64_t)blockIdx.x * 64 + w * 16;
  const int q0 = blockIdx.y * I8_QT;
  const int64_t planeA = (int64_t)N * T * F;
  const int64_t planeQ = (int64_t)Q * NTAPS * F;

  int64_t c = c0 + row;
  if (c >= C) c = C - 1;
  const int j = (int)(c / G), g = (int)(c - (int64_t)j * G);
  const int t0 = cand_t[g];
  const int8_t* arow = A + ((int64_t)j * T + t0) * F + 16 * grp;

  // B tile staging: 4 planes x 48 rows x 64 B = 768 x 16 B; thread -> 3 pieces
  const unsigned char* bsrc[3];
  int bdst[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int piece = threadIdx.x + 256 * k;        // 0..767
    const int s = piece / 192, rem = piece - s * 192, qr = rem >> 2, seg = rem & 3;
    int q = q0 + qr;
    if (q >= Q) q = Q - 1;
    bsrc[k] = reinterpret_cast<const unsigned char*>(Bq) + s * planeQ + (int64_t)q * NTAPS * F + 16 * seg;
    bdst[k] = (s * I8_QT + qr) * I8_ROWB + 16 * seg;
  }

  v4i acc[I8_NT][5];
  double dsum[I8_NT][4];
#pragma unroll
  for (int nt = 0; nt < I8_NT; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) dsum[nt][r] = 0.0;

  const int nch = F / I8_CH;
  const int total = NTAPS * nch;
  // prologue: stage chunk 0
  {
    unsigned char* dst = &bs[0][0][0][0];
#pragma unroll
    for (int k = 0; k < 3; ++k) *reinterpret_cast<v4i*>(dst + bdst[k]) = *reinterpret_cast<const v4i*>(bsrc[k]);
  }
  __syncthreads();

  // the 4 candidate rows this lane holds in the i32 C/D layout (row = 4*(lane>>4) + reg): frame-scale row
  // offsets, resolved once; the per-tap scales are fetched at the START of a tap and used 16 chunks later
  int64_t srow[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    int64_t cc = c0 + 4 * grp + r;
    if (cc >= C) cc = C - 1;
    const int jj = (int)(cc / G), gg = (int)(cc - (int64_t)jj * G);
    srow[r] = (int64_t)jj * T + cand_t[gg];
  }
  int st0[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) st0[r] = (int)(srow[r] % T);
  int qcol[I8_NT];
#pragma unroll
  for (int nt = 0; nt < I8_NT; ++nt) {
    int q = q0 + nt * 16 + row;
    qcol[nt] = q >= Q ? Q - 1 : q;
  }
  double sa_t[4], sq_t[I8_NT];

  for (int it = 0; it < total; ++it) {
    const int tap = it / nch, e0 = (it - tap * nch) * I8_CH;
    if (e0 == 0) {
#pragma unroll
      for (int nt = 0; nt < I8_NT; ++nt)
#pragma unroll
        for (int u = 0; u < 5; ++u) acc[nt][u] = (v4i){0, 0, 0, 0};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool okr = st0[r] + tap * tap_stride < T;
        const double v = sA[okr ? srow[r] + tap * tap_stride : srow[r]];
        sa_t[r] = okr ? v : 0.0;
      }
#pragma unroll
      for (int nt = 0; nt < I8_NT; ++nt) sq_t[nt] = sQ[(int64_t)qcol[nt] * NTAPS + tap];
    }
    // next chunk's query digits -> registers (written to the other LDS buffer after the MFMAs)
    v4i bnext[3];
    const bool more = it + 1 < total;
    if (more) {
      const int tap2 = (it + 1) / nch, e2 = ((it + 1) - tap2 * nch) * I8_CH;
#pragma unroll
      for (int k = 0; k < 3; ++k) bnext[k] = *reinterpret_cast<const v4i*>(bsrc[k] + (int64_t)tap2 * F + e2);
    }
    // candidate digits of this chunk: 4 planes x 16 B (the other resident wave of the SIMD covers the latency)
    const bool ok = t0 + tap * tap_stride < T;
    const int8_t* ap = arow + (ok ? (int64_t)tap * tap_stride * F : 0) + e0;
    v4i a[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const v4i v = *reinterpret_cast<const v4i*>(ap + s * planeA);
      a[s] = ok ? v : (v4i){0, 0, 0, 0};
    }
    const unsigned char* bb = &bs[it & 1][0][0][0];
#pragma unroll
    for (int nt = 0; nt < I8_NT; ++nt) {
      v4i b[4];
#pragma unroll
      for (int t = 0; t < 4; ++t)
        b[t] = *reinterpret_cast<const v4i*>(bb + (t * I8_QT + nt * 16 + row) * I8_ROWB + 16 * grp);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int t = 0; t < 4; ++t)
          if (s + t <= 4)
            acc[nt][s + t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[s], b[t], acc[nt][s + t], 0, 0, 0);
    }
    if (more) {
      unsigned char* dst = &bs[(it + 1) & 1][0][0][0];
#pragma unroll
      for (int k = 0; k < 3; ++k) *reinterpret_cast<v4i*>(dst + bdst[k]) = bnext[k];
    }
    if (e0 + I8_CH == F) {
      // end of a tap: recombine the diagonals (exact powers of two) and apply the two block exponents.
      // i32 C/D layout: col (query) = lane & 15, row (candidate) = 4*(lane>>4) + reg
#pragma unroll
      for (int nt = 0; nt < I8_NT; ++nt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          double P = (double)acc[nt][4][r];                       // u = 4: weight 256^2 (applied below)
          P = (double)acc[nt][3][r] * 256.0 + P;                  // every term is an integer < 2^27 times a power
          P = (double)acc[nt][2][r] * 65536.0 + P;                // of two: the sum is exact up to 2^-53 relative
          P = (double)acc[nt][1][r] * 16777216.0 + P;
          P = (double)acc[nt][0][r] * 4294967296.0 + P;
          dsum[nt][r] += (P * 65536.0) * (sa_t[r] * sq_t[nt]);
        }
      }
    }
    __syncthreads();
  }

  // epilogue: distances, staged through LDS so that every query row gets a 128-B run along the candidates
  double* stage = reinterpret_cast<double*>(&bs[0][0][0][0]);        // [4 waves][48 q][16 cands] = 24 KB
#pragma unroll
  for (int nt = 0; nt < I8_NT; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) stage[(w * I8_QT + nt * 16 + row) * 16 + 4 * grp + r] = dsum[nt][r];
  __syncthreads();
  for (int o = threadIdx.x; o < 4 * I8_QT * 16; o += 256) {
    const int ww = o / (I8_QT * 16), rem = o - ww * (I8_QT * 16), ql = rem >> 4, cr = rem & 15;
    const int q = q0 + ql;
    const int64_t cc = (int64_t)blockIdx.x * 64 + ww * 16 + cr;
    if (q < Q && cc < C) D[(int64_t)q * ldD + cc] = cosine_from_dot_i8(stage[o], qn2[q], cn2[cc]);
  }
}

// ---------------------------------------------------------------------------------------------
// LDS-DMA variant: both operands arrive by global_load_lds (no staging registers) into a DP-deep ring,
// so the contraction loop is { counted vmcnt wait, one barrier, issue the DMA of chunk i+DP-1, ds_read_b128,
// 39 MFMAs }.  Candidate digits: each wave DMAs its own 16 rows (lane-linear 1 KiB per plane, read back by
// the same lane).  Query digits: one shared tile per chunk, 16-B pieces placed lane-linear with the feature
// segment XOR-swizzled by (row>>2)&3 on the SOURCE address, so the fragment reads are conflict-free.
// (r01: the register-staged version above is latency-bound at 495 us; one chunk = 0.3 us of matrix work
// cannot cover HBM latency from registers without dropping to one wave per SIMD.)
// ---------------------------------------------------------------------------------------------
#define I8_BSLOT (4 * I8_QT * 64)          // 12288 B: 4 planes x 48 rows x 64 B, unpadded (swizzled)
#define I8_ASLOT (4 * 4 * 1024)            // 16384 B: 4 waves x 4 planes x 1 KiB

template <int N_>
__device__ __forceinline__ void wait_vmcnt() {
  if (N_ == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if (N_ == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  else if (N_ == 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
  else if (N_ == 21) asm volatile("s_waitcnt vmcnt(21)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

template <int NTAPS, int DP>
__global__ __launch_bounds__(256) void audio_dot_i8_dma_kernel(
    const int8_t* __restrict__ A, const double* __restrict__ sA, int N, int T, int F,
    const int32_t* __restrict__ cand_t, int G, int tap_stride, const double* __restrict__ cn2,
    const int8_t* __restrict__ Bq, const double* __restrict__ sQ, const double* __restrict__ qn2, int Q,
    double* __restrict__ D, int64_t ldD) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  unsigned char* aring = lds;                                  // [DP][4 waves][4 planes][1024]
  unsigned char* bring = lds + DP * I8_ASLOT;                   // [DP][4 planes][48][64] swizzled
  double* sA_l = reinterpret_cast<double*>(bring + DP * I8_BSLOT);   // [4 waves][NTAPS][16]
  double* sQ_l = sA_l + 4 * NTAPS * 16;                               // [NTAPS][48]

  const int64_t C = (int64_t)N * G;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int row = lane & 15, grp = lane >> 4;
  const int64_t c0 = (int64_t)blockIdx.x * 64 + w * 16;
  const int q0 = blockIdx.y * I8_QT;
  const int64_t planeA = (int64_t)N * T * F;
  const int64_t planeQ = (int64_t)Q * NTAPS * F;

  // ---- per-lane DMA sources
  int64_t c = c0 + row;
  if (c >= C) c = C - 1;
  const int j = (int)(c / G), g = (int)(c - (int64_t)j * G);
  const int t0 = cand_t[g];
  const int8_t* arow = A + ((int64_t)j * T + t0) * F + 16 * grp;   // + s*planeA + tap*stride*F + e0
  const unsigned char* bsrc[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int piece = (w * 3 + k) * 64 + lane;                   // 0..767, lane-linear within the instruction
    const int s = piece / 192, u = piece - s * 192, qr = u >> 2, segpos = u & 3;
    const int seg = segpos ^ ((qr >> 2) & 3);
    int q = q0 + qr;
    if (q >= Q) q = Q - 1;
    bsrc[k] = reinterpret_cast<const unsigned char*>(Bq) + s * planeQ + (int64_t)q * NTAPS * F + 16 * seg;
  }
  // ---- block exponents of this block's rows / queries -> LDS (ordinary loads, all done before the ring starts)
  for (int o = tid; o < 4 * NTAPS * 16; o += 256) {
    const int ww = o / (NTAPS * 16), rem = o - ww * (NTAPS * 16), tap = rem >> 4, r = rem & 15;
    int64_t cc = (int64_t)blockIdx.x * 64 + ww * 16 + r;
    if (cc >= C) cc = C - 1;
    const int jj = (int)(cc / G), gg = (int)(cc - (int64_t)jj * G);
    const int tt = cand_t[gg] + tap * tap_stride;
    sA_l[o] = tt < T ? sA[(int64_t)jj * T + tt] : 0.0;           // zero scale == zero-padded tap
  }
  for (int o = tid; o < NTAPS * I8_QT; o += 256) {
    const int tap = o / I8_QT, ql = o - tap * I8_QT;
    int q = q0 + ql;
    if (q >= Q) q = Q - 1;
    sQ_l[o] = sQ[(int64_t)q * NTAPS + tap];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int nch = F / I8_CH;
  const int total = NTAPS * nch;
  auto issue = [&](int it) {
    const int tap = it / nch, e0 = (it - tap * nch) * I8_CH;
    const int slot = it % DP;
    // taps past the end of the window read a valid frame instead; their block scale is 0
    const int64_t toff = (t0 + tap * tap_stride < T) ? (int64_t)tap * tap_stride * F : 0;
    unsigned char* adst = aring + (slot * 4 + w) * 4096;
#pragma unroll
    for (int s = 0; s < 4; ++s)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(arow + s * planeA + toff + e0), (lds_ptr_t)(adst + s * 1024), 16,
                                       0, 0);
    unsigned char* bdst = bring + slot * I8_BSLOT + (w * 3) * 1024;
#pragma unroll
    for (int k = 0; k < 3; ++k)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(bsrc[k] + (int64_t)tap * F + e0), (lds_ptr_t)(bdst + k * 1024), 16,
                                       0, 0);
  };

  v4i acc[I8_NT][5];
  double dsum[I8_NT][4];
#pragma unroll
  for (int nt = 0; nt < I8_NT; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) dsum[nt][r] = 0.0;

#pragma unroll
  for (int p = 0; p < DP - 1; ++p) issue(p);

  for (int it = 0; it < total; ++it) {
    const int tap = it / nch, e0 = (it - tap * nch) * I8_CH;
    if (e0 == 0) {
#pragma unroll
      for (int nt = 0; nt < I8_NT; ++nt)
#pragma unroll
        for (int u = 0; u < 5; ++u) acc[nt][u] = (v4i){0, 0, 0, 0};
    }
    // chunk `it` was issued DP-1 iterations ago; DP-2 younger chunks (7 DMAs each) may stay in flight
    if (it + DP - 2 < total) wait_vmcnt<7 * (DP - 2)>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();     // everyone's pieces of chunk `it` landed; everyone is done with chunk it-1
    if (it + DP - 1 < total) issue(it + DP - 1);       // overwrites the slot of chunk it-1

    const int slot = it % DP;
    const unsigned char* ab = aring + (slot * 4 + w) * 4096 + lane * 16;
    v4i a[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) a[s] = *reinterpret_cast<const v4i*>(ab + s * 1024);
    const unsigned char* bb = bring + slot * I8_BSLOT;
#pragma unroll
    for (int nt = 0; nt < I8_NT; ++nt) {
      const int qr = nt * 16 + row;
      const int segpos = grp ^ ((qr >> 2) & 3);
      v4i b[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) b[t] = *reinterpret_cast<const v4i*>(bb + t * 3072 + (qr * 4 + segpos) * 16);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int t = 0; t < 4; ++t)
          if (s + t <= 4)
            acc[nt][s + t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[s], b[t], acc[nt][s + t], 0, 0, 0);
    }
    if (e0 + I8_CH == F) {
      // end of a tap: recombine the diagonals (exact powers of two), apply the two block exponents.
      // i32 C/D layout: col (query) = lane & 15, row (candidate) = 4*(lane>>4) + reg
#pragma unroll
      for (int nt = 0; nt < I8_NT; ++nt) {
        const double sq = sQ_l[tap * I8_QT + nt * 16 + row];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double sa = sA_l[(w * NTAPS + tap) * 16 + 4 * grp + r];
          double P = (double)acc[nt][4][r];
          P = (double)acc[nt][3][r] * 256.0 + P;
          P = (double)acc[nt][2][r] * 65536.0 + P;
          P = (double)acc[nt][1][r] * 16777216.0 + P;
          P = (double)acc[nt][0][r] * 4294967296.0 + P;
          dsum[nt][r] += (P * 65536.0) * (sa * sq);
        }
      }
    }
  }

  __syncthreads();
  double* stage = reinterpret_cast<double*>(aring);                 // [4 waves][48 q][16 cands] = 24 KB
#pragma unroll
  for (int nt = 0; nt < I8_NT; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) stage[(w * I8_QT + nt * 16 + row) * 16 + 4 * grp + r] = dsum[nt][r];
  __syncthreads();
  for (int o = tid; o < 4 * I8_QT * 16; o += 256) {
    const int ww = o / (I8_QT * 16), rem = o - ww * (I8_QT * 16), ql = rem >> 4, cr = rem & 15;
    const int q = q0 + ql;
    const int64_t cc = (int64_t)blockIdx.x * 64 + ww * 16 + cr;
    if (q < Q && cc < C) D[(int64_t)q * ldD + cc] = cosine_from_dot_i8(stage[o], qn2[q], cn2[cc]);
  }
}

extern "C" int qpg_audio_cosine_i8(qpg_ctx* ctx, void* stream, const int8_t* A, const double* sA, int N, int T, int F,
                                   const int32_t* cand_t, int G, int n_taps, int tap_stride, const double* cn2,
                                   const int8_t* Bq, const double* sQ, const double* qn2, int Q, double* D,
                                   int64_t ldD) {
  QPG_REQUIRE(ctx && A && sA && cand_t && cn2 && Bq && sQ && qn2 && D, "qpg_audio_cosine_i8: null pointer");
  QPG_REQUIRE(N >= 0 && T > 0 && G > 0 && Q >= 0 && tap_stride > 0 && ldD >= (int64_t)N * G,
              "qpg_audio_cosine_i8: bad size");
  if (n_taps != 6 || F <= 0 || (F % I8_CH) != 0) {
    qpg_set_error("qpg_audio_cosine_i8: compiled for n_taps=6 and F %% 64 == 0 (got n_taps=%d F=%d)", n_taps, F);
    return QPG_EUNSUP;
  }
  if (N == 0 || Q == 0) return QPG_OK;
  const int64_t C = (int64_t)N * G;
  dim3 grid((unsigned)((C + 63) / 64), (unsigned)((Q + I8_QT - 1) / I8_QT));
  static int variant = -1;
  if (variant < 0) {
    const char* e = getenv("QPG_I8_VARIANT");     // tuning aid: 0 = register-staged, 2/3/4 = LDS-DMA ring depth
    variant = e ? atoi(e) : 3;
  }
  hipStream_t st = qpg_stream(stream);
#define QPG_I8_ARGS A, sA, N, T, F, cand_t, G, tap_stride, cn2, Bq, sQ, qn2, Q, D, ldD
#define QPG_I8_LDS(DP) ((size_t)(DP) * (I8_ASLOT + I8_BSLOT) + sizeof(double) * (4 * 6 * 16 + 6 * I8_QT))
  if (variant == 0) {
    hipLaunchKernelGGL((audio_dot_i8_kernel<6>), grid, dim3(256), 0, st, QPG_I8_ARGS);
  } else if (variant == 2) {
    hipLaunchKernelGGL((audio_dot_i8_dma_kernel<6, 2>), grid, dim3(256), QPG_I8_LDS(2), st, QPG_I8_ARGS);
  } else if (variant == 4) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&audio_dot_i8_dma_kernel<6, 4>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)QPG_I8_LDS(4));
    hipLaunchKernelGGL((audio_dot_i8_dma_kernel<6, 4>), grid, dim3(256), QPG_I8_LDS(4), st, QPG_I8_ARGS);
  } else {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&audio_dot_i8_dma_kernel<6, 3>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)QPG_I8_LDS(3));
    hipLaunchKernelGGL((audio_dot_i8_dma_kernel<6, 3>), grid, dim3(256), QPG_I8_LDS(3), st, QPG_I8_ARGS);
  }
#undef QPG_I8_ARGS
  QPG_LAUNCH_CHECK("audio_dot_i8_kernel");
  return QPG_OK;
}
