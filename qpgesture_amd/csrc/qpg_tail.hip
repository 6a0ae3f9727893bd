// The state-dependent tail of CodeKNN.search_code_knn (GestureKNN.py:501-664) for a whole clip:
// pose-signature term, rank fusion, phase gate, window chaining.  Everything that depends only
// on the query position (the candidate sweeps and their ranks) has already been computed for all
// Q steps; what is left is sequential in the running (last code, last phase block) state and is
// O(K) per step, so it runs as ONE workgroup that walks the M*steps chain without returning to
// the host.  Float paths reproduce the reference's arithmetic: combined scores in float64 in the
// reference's operation order, the 128-d phase-gate cosine in scikit-learn's float32 order.
#include "qpg_common.h"

// ---------------------------------------------------------------------------------------------
// pose-signature distance table: out[p][c] = |sig[p] - sig[c]|_2 (f32), +inf on the diagonal
// (GestureKNN.py:531-536).  The difference is taken in f32 like the reference; the sum of
// squares is accumulated in f64 and rounded once (the reference's np.linalg.norm sums in f32 in
// a BLAS-dependent order; see DESIGN.md "Tie contract").
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void l2_table_kernel(const float* __restrict__ sig, int K, int Dm,
                                                       float* __restrict__ out) {
  const int p = blockIdx.x;
  for (int c = threadIdx.x; c < K; c += blockDim.x) {
    double s = 0.0;
    for (int e = 0; e < Dm; ++e) {
      const float d = f_sub(sig[(int64_t)p * Dm + e], sig[(int64_t)c * Dm + e]);
      s += (double)d * (double)d;
    }
    out[(int64_t)p * K + c] = (c == p) ? __builtin_inff() : (float)sqrt(s);
  }
}

extern "C" int qpg_l2_table_f32(qpg_ctx* ctx, void* stream, const float* sig, int K, int Dm, float* out) {
  QPG_REQUIRE(ctx && sig && out && K > 0 && Dm > 0, "qpg_l2_table_f32: bad argument");
  hipLaunchKernelGGL(l2_table_kernel, dim3(K), dim3(256), 0, qpg_stream(stream), sig, K, Dm, out);
  QPG_LAUNCH_CHECK("l2_table_kernel");
  return QPG_OK;
}

// ---------------------------------------------------------------------------------------------
// sklearn-exact f32 cosine of two 128-d vectors, cooperatively by 4 lanes (lane l = einsum lane).
// `a`/`b` are LDS arrays of 128 floats.  All 4 lanes return the same value.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float lane4_sum(float v, int l) {
  // (l0 + l1) + (l2 + l3) in exactly that order, via shuffles within the aligned group of 4
  const float o1 = __shfl_xor(v, 1, 64);
  const float pair = (l & 1) ? f_add(o1, v) : f_add(v, o1);  // lanes 0,1 -> l0+l1 ; lanes 2,3 -> l2+l3
  const float o2 = __shfl_xor(pair, 2, 64);
  return (l & 2) ? f_add(o2, pair) : f_add(pair, o2);         // (l0+l1)+(l2+l3)
}

__device__ __forceinline__ float einsum_sq_128(const float* v, int l) {
  float a = 0.f;
#pragma unroll
  for (int g = 0; g < 8; ++g) {
#pragma unroll
    for (int u = 3; u >= 0; --u) {
      const float x = v[g * 16 + u * 4 + l];
      a = f_add(f_mul(x, x), a);
    }
  }
  return lane4_sum(a, l);
}

__device__ float gate_cosine_128(float* a, float* b, int l) {
  const float eps10 = 10.f * 1.1920928955078125e-07f;
  float na = f_sqrt(einsum_sq_128(a, l)), nb = f_sqrt(einsum_sq_128(b, l));
  if (na < eps10) na = 1.f;
  if (nb < eps10) nb = 1.f;
  // diff into `a` (each lane touches only its own elements e with (e & 3) == l)
  for (int e = l; e < 128; e += 4) a[e] = f_sub(f_div(a[e], na), f_div(b[e], nb));
  return f_mul(0.5f, einsum_sq_128(a, l));
}

struct ArgMin {
  double v;
  int i;
};
__device__ __forceinline__ ArgMin amin(ArgMin a, ArgMin b) {
  return (b.v < a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}

// block-wide stable argmin of val over threads (tid = code index); excluded index gets +inf.
__device__ ArgMin block_argmin(double val, int idx, ArgMin* scratch) {
  ArgMin m{val, idx};
  for (int o = 32; o > 0; o >>= 1) {
    ArgMin t{__shfl_down(m.v, o, 64), __shfl_down(m.i, o, 64)};
    m = amin(m, t);
  }
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) scratch[w] = m;
  __syncthreads();
  ArgMin r = scratch[0];
  for (int k = 1; k < nw; ++k) r = amin(r, scratch[k]);
  return r;
}


struct TailArgs {
  const int16_t* aud_rank;   // [Q][K]
  const int32_t* aud_idx;    // [Q][K] local candidate index j*Ga+g (idx_base already removed), -1 = absent
  const int16_t* txt_rank;
  const int32_t* txt_idx;
  const int16_t* pos_rank;   // [K][K]
  const int16_t* freq_rank;  // [K]
  const int32_t* code;       // [N][code_ld]
  int code_ld;
  const int32_t* aud_cidx;   // [Ga] code column of grid position
  const int32_t* aud_pslot;  // [Ga] phase start frame int(k/398*240)
  int Ga;
  const int32_t* txt_cidx;
  const int32_t* txt_pslot;
  int Gt;
  const float* phase;        // [N][Tp][2][8]  (p, a)
  int Tp;
  int mode;
  int M, steps, step_codes, codes_per_window;
  int K;
  int seed_code;
  const float* seed_phase;   // [8][16]
  int32_t* out_codes;        // [M][codes_per_window]
  float* out_phase;          // [M][steps][8][16]
  int32_t* out_vote;         // [M][steps]
  int32_t* out_status;       // [1] 0 ok, 1 = an absent code won a rank fusion (reference would raise IndexError)
};

__global__ __launch_bounds__(1024) void match_steps_kernel(TailArgs A) {
  __shared__ ArgMin scratch[16];
  __shared__ float prev[8 * 16];        // running phase block (8 frames x [8 phase | 8 amp])
  __shared__ float head[2][8 * 16];     // candidate first-8-frame blocks
  __shared__ float tail[2][8 * 16];     // candidate last-8-frame blocks
  __shared__ float va[2][128], vb[2][128];
  __shared__ float score[2];
  __shared__ int cand_code[2], cand_j[2], cand_g[2], cand_src[2];
  __shared__ int s_prev_code, s_final, s_bad;
  __shared__ int wincodes[64];

  const int tid = threadIdx.x, K = A.K;
  if (tid < 128) prev[tid] = A.seed_phase[tid];
  if (tid == 0) {
    s_prev_code = A.seed_code;
    s_bad = 0;
  }
  __syncthreads();

  for (int w = 0; w < A.M; ++w) {
    for (int s = 0; s < A.steps; ++s) {
      const int q = w * A.steps + s;
      const int pc = s_prev_code;
      // ---- rank fusion (GestureKNN.py:540-545, 553-555, 574-576), f64 in the reference's op order
      double ca = __builtin_inf(), ct = __builtin_inf();
      if (tid < K) {
        const double pos_score = (double)A.pos_rank[(int64_t)pc * K + tid] + (double)A.freq_rank[tid] * 0.05;
        if (A.mode != QPG_MODE_TXT) ca = pos_score + (double)A.aud_rank[(int64_t)q * K + tid];
        if (A.mode != QPG_MODE_AUD) ct = pos_score + (double)A.txt_rank[(int64_t)q * K + tid];
      }
      int c0, c1, src0, src1;
      if (A.mode == QPG_MODE_AUD_TXT) {
        c0 = block_argmin(ca, tid, scratch).i;
        c1 = block_argmin(ct, tid, scratch).i;
        src0 = 0;
        src1 = 1;
      } else {
        const double v = (A.mode == QPG_MODE_AUD) ? ca : ct;
        c0 = block_argmin(v, tid, scratch).i;
        c1 = block_argmin(tid == c0 ? __builtin_inf() : v, tid, scratch).i;
        src0 = src1 = (A.mode == QPG_MODE_AUD) ? 0 : 1;
      }
      if (tid < 2) {
        const int c = tid ? c1 : c0, src = tid ? src1 : src0;
        const int32_t ci = src ? A.txt_idx[(int64_t)q * K + c] : A.aud_idx[(int64_t)q * K + c];
        const int G = src ? A.Gt : A.Ga;
        cand_code[tid] = c;
        cand_src[tid] = src;
        if (ci < 0) {
          s_bad = 1;
          cand_j[tid] = 0;
          cand_g[tid] = 0;
        } else {
          cand_j[tid] = ci / G;
          cand_g[tid] = ci - (ci / G) * G;
        }
      }
      __syncthreads();
      // ---- fetch both candidates' phase blocks: rows [ps, ps+8) and [ps+24, ps+32) (GestureKNN.py:632-637)
      if (tid < 512) {
        const int k = tid >> 8, part = (tid >> 7) & 1, e = tid & 127;   // part 0 = head, 1 = tail
        const int r = e >> 4, col = e & 15;                              // row, [phase 0..7 | amp 0..7]
        const int ps = (cand_src[k] ? A.txt_pslot : A.aud_pslot)[cand_g[k]];
        const int t = ps + (part ? 24 : 0) + r;
        const float v = A.phase[(((int64_t)cand_j[k] * A.Tp + t) * 2 + (col >> 3)) * 8 + (col & 7)];
        (part ? tail : head)[k][e] = v;
      }
      __syncthreads();
      // ---- gate vectors: a = [prev[-5:], head[:3]], b = [prev[-3:], head[:5]]  (128 floats each)
      if (tid < 256) {
        const int k = tid >> 7, e = tid & 127;
        va[k][e] = (e < 80) ? prev[48 + e] : head[k][e - 80];
        vb[k][e] = (e < 48) ? prev[80 + e] : head[k][e - 48];
      }
      __syncthreads();
      if (tid < 8) {
        const int k = tid >> 2, l = tid & 3;
        const float sc = gate_cosine_128(va[k], vb[k], l);
        if (l == 0) score[k] = sc;
      }
      __syncthreads();
      if (tid == 0) s_final = (score[1] < score[0]) ? 1 : 0;   // list.index(min): first on ties
      __syncthreads();
      const int fi = s_final;
      // ---- append the winner's 4 codes, carry its last-8-frame block (GestureKNN.py:648-657)
      if (tid < 128) {
        prev[tid] = tail[fi][tid];
        A.out_phase[(((int64_t)w * A.steps + s) * 128) + tid] = tail[fi][tid];
      }
      if (tid < A.step_codes) {
        const int col = (cand_src[fi] ? A.txt_cidx : A.aud_cidx)[cand_g[fi]] + tid;
        const int cv = A.code[(int64_t)cand_j[fi] * A.code_ld + col];
        wincodes[s * A.step_codes + tid] = cv;
      }
      if (tid == 0) A.out_vote[w * A.steps + s] = fi;
      __syncthreads();
      if (tid == 0) s_prev_code = wincodes[s * A.step_codes + A.step_codes - 1];
      __syncthreads();
    }
    // window result = first codes_per_window codes; the next window is seeded by the LAST KEPT code
    // (motion_output[-1][-1], GestureKNN.py:800) and the last phase block.
    if (tid < A.codes_per_window) A.out_codes[(int64_t)w * A.codes_per_window + tid] = wincodes[tid];
    if (tid == 0) s_prev_code = wincodes[A.codes_per_window - 1];
    __syncthreads();
  }
  if (tid == 0) A.out_status[0] = s_bad;
}

extern "C" int qpg_match_steps(qpg_ctx* ctx, void* stream, const int16_t* aud_rank, const int32_t* aud_idx,
                               const int16_t* txt_rank, const int32_t* txt_idx, const int16_t* pos_rank,
                               const int16_t* freq_rank, const int32_t* code, int code_ld, const int32_t* aud_cidx,
                               const int32_t* aud_pslot, int Ga, const int32_t* txt_cidx, const int32_t* txt_pslot,
                               int Gt, const float* phase, int Tp, int mode, int M, int steps, int K, int seed_code,
                               const float* seed_phase, int32_t* out_codes, float* out_phase, int32_t* out_vote,
                               int32_t* out_status) {
  QPG_REQUIRE(ctx && pos_rank && freq_rank && code && phase && seed_phase && out_codes && out_phase && out_vote &&
                  out_status,
              "qpg_match_steps: null pointer");
  QPG_REQUIRE(mode >= 0 && mode <= 2, "qpg_match_steps: bad mode %d", mode);
  QPG_REQUIRE(mode == QPG_MODE_TXT || (aud_rank && aud_idx && aud_cidx && aud_pslot && Ga > 0),
              "qpg_match_steps: audio tables missing");
  QPG_REQUIRE(mode == QPG_MODE_AUD || (txt_rank && txt_idx && txt_cidx && txt_pslot && Gt > 0),
              "qpg_match_steps: text tables missing");
  QPG_REQUIRE(M >= 0 && steps > 0 && steps * 4 <= 64 && K > 0 && K <= 1024 && seed_code >= 0 && seed_code < K,
              "qpg_match_steps: bad size");
  if (M == 0) return QPG_OK;
  TailArgs A;
  A.aud_rank = aud_rank; A.aud_idx = aud_idx; A.txt_rank = txt_rank; A.txt_idx = txt_idx;
  A.pos_rank = pos_rank; A.freq_rank = freq_rank; A.code = code; A.code_ld = code_ld;
  A.aud_cidx = aud_cidx; A.aud_pslot = aud_pslot; A.Ga = Ga; A.txt_cidx = txt_cidx; A.txt_pslot = txt_pslot; A.Gt = Gt;
  A.phase = phase; A.Tp = Tp; A.mode = mode; A.M = M; A.steps = steps; A.step_codes = 4;
  A.codes_per_window = (steps * 4 < 30) ? steps * 4 : 30;
  A.K = K; A.seed_code = seed_code; A.seed_phase = seed_phase;
  A.out_codes = out_codes; A.out_phase = out_phase; A.out_vote = out_vote; A.out_status = out_status;
  int threads = K < 512 ? 512 : ((K + 63) / 64) * 64;
  hipLaunchKernelGGL(match_steps_kernel, dim3(1), dim3(threads), 0, qpg_stream(stream), A);
  QPG_LAUNCH_CHECK("match_steps_kernel");
  return QPG_OK;
}
