// Experiment: approximate audio sweep on the f32 matrix cores with the QUERY tile shared through LDS.
// Block = 4 waves x 32 candidates (MT=2 tiles of 16) x 48 queries (NT=3 tiles of 16); every wave walks the whole
// contraction (6 taps x F features), so there is no cross-wave reduction; per 64-feature chunk the 48x64 query tile
// is staged once per block (global -> registers one chunk ahead -> LDS double buffer, one barrier per chunk) instead
// of being streamed from L2 by every wave.  f32 MFMA partial sums (64 products) are flushed into f64 per chunk.
#include "qpg_common.h"

namespace {

constexpr int LQ = 48;          // queries per block (3 tiles)
constexpr int CH = 64;          // features per chunk
constexpr int PITCH = CH + 4;   // LDS row pitch in floats (16-B aligned, conflict-free ds_read_b128 per quarter wave)

__device__ __forceinline__ double cosine_from_dot(double dot, double qn2, double cn2) {
  const double tiny = 10.0 * 2.220446049250313e-16;
  double nq = sqrt(qn2), nc = sqrt(cn2);
  bool zq = nq < tiny, zc = nc < tiny;
  if (zq || zc) {
    double a = zq ? qn2 : 1.0, b = zc ? cn2 : 1.0;
    double cross = dot / ((zq ? 1.0 : nq) * (zc ? 1.0 : nc));
    return 0.5 * (a + b - 2.0 * cross);
  }
  return 1.0 - dot / (nq * nc);
}

__global__ __launch_bounds__(256) void audio_cosine_f32lds_kernel(const float* __restrict__ base, int N, int T, int F,
                                                                  const int32_t* __restrict__ cand_t, int G,
                                                                  int tap_stride, const double* __restrict__ cn2,
                                                                  const float* __restrict__ q32,
                                                                  const double* __restrict__ qn2, int Q,
                                                                  float* __restrict__ D, int64_t ldD) {
  __shared__ __attribute__((aligned(16))) float Bs[2][LQ][PITCH];
  const int64_t C = (int64_t)N * G;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int row = lane & 15, kq = lane >> 4;
  const int64_t c0 = (int64_t)blockIdx.x * 128 + w * 32;
  const int q0 = blockIdx.y * LQ;
  const int KQ = 6 * F;

  const float* arow[2];
  int at0[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    int64_t c = c0 + mt * 16 + row;
    if (c >= C) c = C - 1;
    const int j = (int)(c / G), g = (int)(c - (int64_t)j * G);
    at0[mt] = cand_t[g];
    arow[mt] = base + ((int64_t)j * T + at0[mt]) * F + 8 * kq;
  }
  // B staging: thread -> 3 float4 of the 48 x 64 chunk: element index i = tid + 256 s  ->  q = i / 16, f4 = i % 16
  f32x4 bst[3];
  auto b_fetch = [&](int tap, int e0) {
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const int i = tid + 256 * s;
      int q = q0 + (i >> 4);
      if (q >= Q) q = Q - 1;
      bst[s] = *reinterpret_cast<const f32x4*>(q32 + (int64_t)q * KQ + tap * F + e0 + (i & 15) * 4);
    }
  };
  auto b_commit = [&](int buf) {
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const int i = tid + 256 * s;
      *reinterpret_cast<f32x4*>(&Bs[buf][i >> 4][(i & 15) * 4]) = bst[s];
    }
  };

  f64x4 acc[2][3];
  f32x4 facc[2][3];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) {
      acc[mt][nt] = (f64x4){0.0, 0.0, 0.0, 0.0};
      facc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

  struct ABuf {
    f32x4 a[2][2];
  };
  auto a_load = [&](ABuf& u, int tap, int e) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const bool ok = at0[mt] + tap * tap_stride < T;
      const float* p = arow[mt] + (ok ? (int64_t)tap * tap_stride * F : 0) + e;
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(p), v1 = *reinterpret_cast<const f32x4*>(p + 4);
      const f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
      u.a[mt][0] = ok ? v0 : z;
      u.a[mt][1] = ok ? v1 : z;
    }
  };
  auto mma = [&](const ABuf& u, int buf, int g) {
    f32x4 b[3][2];
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) {
      const float* p = &Bs[buf][nt * 16 + row][g * 32 + 8 * kq];
      b[nt][0] = *reinterpret_cast<const f32x4*>(p);
      b[nt][1] = *reinterpret_cast<const f32x4*>(p + 4);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < 3; ++nt)
            facc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(u.a[mt][h][i], b[nt][h][i], facc[mt][nt], 0, 0, 0);
  };
  auto flush = [&]() {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 3; ++nt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[mt][nt][r] += (double)facc[mt][nt][r];
        facc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
  };

  const int nch = F / CH;            // chunks per tap
  const int total = 6 * nch;
  b_fetch(0, 0);
  b_commit(0);
  // candidate operands are requested a whole chunk (2 groups = 96 MFMAs = 3072 matrix cycles) ahead: with only
  // C/32 waves in the launch (1.6 per SIMD at N_db = 2048) nothing else hides the HBM latency
  ABuf u[2][2];
  a_load(u[0][0], 0, 0);
  a_load(u[0][1], 0, 32);
  __syncthreads();
  for (int it = 0; it < total; it += 2) {
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      const int i2 = it + par;
      const int tap = i2 / nch, e0 = (i2 - tap * nch) * CH;
      (void)e0;
      const int itn = i2 + 1 < total ? i2 + 1 : i2;
      const int tapn = itn / nch, e0n = (itn - tapn * nch) * CH;
      b_fetch(tapn, e0n);                     // next chunk's query tile: in flight during this chunk's MFMAs
      a_load(u[par ^ 1][0], tapn, e0n);
      a_load(u[par ^ 1][1], tapn, e0n + 32);
      __builtin_amdgcn_sched_barrier(0);
      mma(u[par][0], par, 0);
      mma(u[par][1], par, 1);
      flush();
      b_commit(par ^ 1);
      __syncthreads();
    }
  }

  // f32 16x16x4 C/D: lane l, reg r -> candidate row 4*(l>>4) + r, query col l&15: 4 consecutive candidates per lane
#pragma unroll
  for (int nt = 0; nt < 3; ++nt) {
    const int q = q0 + nt * 16 + row;
    if (q >= Q) continue;
    const double qn = qn2[q];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int64_t cc = c0 + mt * 16 + 4 * kq;
      f32x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t c = cc + r < C ? cc + r : C - 1;
        o[r] = (float)cosine_from_dot(acc[mt][nt][r], qn, cn2[c]);
      }
      if (cc + 3 < C) {
        *reinterpret_cast<f32x4*>(D + (int64_t)q * ldD + cc) = o;
      } else {
        for (int r = 0; r < 4; ++r)
          if (cc + r < C) D[(int64_t)q * ldD + cc + r] = o[r];
      }
    }
  }
}

}  // namespace

extern "C" int qpg_audio_cosine_approx_lds(qpg_ctx* ctx, void* stream, const float* base, int N, int T, int F,
                                           const int32_t* cand_t, int G, int n_taps, int tap_stride,
                                           const double* cn2, const float* q32, const double* qn2, int Q, float* D32,
                                           int64_t ldD) {
  QPG_REQUIRE(ctx && base && cand_t && cn2 && q32 && qn2 && D32, "null pointer");
  QPG_REQUIRE(n_taps == 6 && F % CH == 0 && (ldD % 4) == 0, "unsupported shape");
  if (N == 0 || Q == 0) return QPG_OK;
  const int64_t C = (int64_t)N * G;
  dim3 grid((unsigned)((C + 127) / 128), (unsigned)((Q + LQ - 1) / LQ));
  hipLaunchKernelGGL(audio_cosine_f32lds_kernel, grid, dim3(256), 0, qpg_stream(stream), base, N, T, F, cand_t, G,
                     tap_stride, cn2, q32, qn2, Q, D32, ldD);
  QPG_LAUNCH_CHECK("audio_cosine_f32lds_kernel");
  return QPG_OK;
}
