// VQ-VAE training-step pieces that are not convolutions (gfx950): the loss terms of VQVAE.forward
// (codebook/models/vqvae.py:240-302), the bottleneck's commit / fit / prenorm statistics (bottleneck.py:96-118,
// 156-186) and the EMA codebook update with random restart (bottleneck.py:63-94).
//
// All reductions are two-stage and ordered (per-block partial -> one block sums the partials in index order), with
// f64 accumulators: results are run-to-run deterministic and independent of the launch's scheduling.
#include "qpg_common.h"

namespace {

constexpr int RED_BLOCKS = 1024;   // partial slots of one reduction
constexpr int RED_THREADS = 256;

template <int NV>
__device__ __forceinline__ void block_reduce_store(double (&v)[NV], double* __restrict__ partial) {
  __shared__ double sm[RED_THREADS / 64][NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    double x = v[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6][i] = x;
  }
  __syncthreads();
  if (threadIdx.x < NV) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < RED_THREADS / 64; ++w) s += sm[w][threadIdx.x];
    partial[(size_t)blockIdx.x * NV + threadIdx.x] = s;
  }
}

// one block: out[i] = sum_b partial[b][i] in index order (pairwise tree over a fixed layout -> deterministic)
template <int NV>
__device__ __forceinline__ void final_reduce(const double* __restrict__ partial, int nblocks, double (&out)[NV]) {
  __shared__ double sm[RED_THREADS][NV];
  double acc[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) acc[i] = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += RED_THREADS)
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] += partial[(size_t)b * NV + i];
#pragma unroll
  for (int i = 0; i < NV; ++i) sm[threadIdx.x][i] = acc[i];
  __syncthreads();
  for (int s = RED_THREADS / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s)
#pragma unroll
      for (int i = 0; i < NV; ++i) sm[threadIdx.x][i] += sm[threadIdx.x + s][i];
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) out[i] = sm[0][i];
}

// ---------------------------------------------------------------------------------------------
// Reconstruction / velocity / acceleration / regularisation sums (vqvae.py:244-258).
// x_out, x_tgt: [B][T][C].  partial: [RED_BLOCKS][4] = sum|xo-xt|, sum|d1(xo)-d1(xt)|, sum|d2(xo)-d2(xt)|, sum d2(xo)^2
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(RED_THREADS) void vq_loss_partial_kernel(const float* __restrict__ xo,
                                                                      const float* __restrict__ xt, int B, int T, int C,
                                                                      double* __restrict__ partial) {
  const int64_t n = (int64_t)B * T * C;
  double v[4] = {0.0, 0.0, 0.0, 0.0};
  for (int64_t i = (int64_t)blockIdx.x * RED_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * RED_THREADS) {
    const int t = (int)((i / C) % T);
    const float o0 = xo[i], g0 = xt[i];
    v[0] += (double)fabsf(g0 - o0);
    if (t >= 1) {
      const float o1 = xo[i - C], g1 = xt[i - C];
      v[1] += (double)fabsf((g0 - g1) - (o0 - o1));
      if (t + 1 < T) {
        const float o2 = xo[i + C], g2 = xt[i + C];
        const float ao = o2 + o1 - 2.0f * o0;
        const float ag = g2 + g1 - 2.0f * g0;
        v[2] += (double)fabsf(ag - ao);
        v[3] += (double)(ao * ao);
      }
    }
  }
  block_reduce_store<4>(v, partial);
}

// out[6] = {loss, recons, regularization, velocity, acceleration, commit}
__global__ __launch_bounds__(RED_THREADS) void vq_loss_final_kernel(const double* __restrict__ partial, int nblocks,
                                                                    int B, int T, int C,
                                                                    const float* __restrict__ commit_loss,
                                                                    float w_commit, float w_reg, float w_vel,
                                                                    float w_acc, float* __restrict__ out) {
  double s[4];
  final_reduce<4>(partial, nblocks, s);
  if (threadIdx.x == 0) {
    const double n0 = (double)B * T * C, n1 = (double)B * (T - 1) * C, n2 = (double)B * (T - 2) * C;
    const float recons = (float)(s[0] / n0);
    const float vel = T > 1 ? (float)(s[1] / n1) : 0.0f;
    const float acc = T > 2 ? (float)(s[2] / n2) : 0.0f;
    const float reg = T > 2 ? (float)(s[3] / n2) : 0.0f;
    const float commit = commit_loss ? *commit_loss : 0.0f;
    // vqvae.py:267: recons + commit*w + reg*w + vel*w + acc*w, summed left to right in f32
    float loss = recons + commit * w_commit;
    loss = loss + w_reg * reg;
    loss = loss + w_vel * vel;
    loss = loss + w_acc * acc;
    out[0] = loss;
    out[1] = recons;
    out[2] = reg;
    out[3] = vel;
    out[4] = acc;
    out[5] = commit;
  }
}

// d loss / d x_out (the L1 / second-difference terms are piecewise linear: sign() sub-gradients, 0 at 0 like torch).
// One thread per element gathers the (up to) 1 + 2 + 3 + 3 terms that touch x_out[b][t][c].
__device__ __forceinline__ float sgnf(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }

__global__ __launch_bounds__(256) void vq_loss_grad_kernel(const float* __restrict__ xo, const float* __restrict__ xt,
                                                           int B, int T, int C, float g_rec, float g_vel, float g_acc,
                                                           float g_reg, float* __restrict__ dxo) {
  const int64_t n = (int64_t)B * T * C;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int t = (int)((i / C) % T);
  auto O = [&](int dt) { return xo[i + (int64_t)dt * C]; };
  auto G = [&](int dt) { return xt[i + (int64_t)dt * C]; };
  // recons: mean|xt - xo|  ->  -sign(xt - xo) / n0
  float g = -g_rec * sgnf(G(0) - O(0));
  // velocity: v[t] = (xo[t]-xo[t-1]) vs target, t = 1..T-1; loss term |tv - v|; d/dxo[t] = -s[t] + s[t+1]
  if (t >= 1) g -= g_vel * sgnf((G(0) - G(-1)) - (O(0) - O(-1)));
  if (t + 1 < T) g += g_vel * sgnf((G(1) - G(0)) - (O(1) - O(0)));
  // acceleration a[m] = xo[m+1] + xo[m-1] - 2 xo[m], m = 1..T-2: |ta - a| ; regularisation a^2
  auto acc_term = [&](int m_rel, float coef) {   // contribution of a[t+m_rel] to d/dxo[t] with da/dxo = coef
    const int m = t + m_rel;
    if (m < 1 || m > T - 2) return;
    const float ao = O(m_rel + 1) + O(m_rel - 1) - 2.0f * O(m_rel);
    const float ag = G(m_rel + 1) + G(m_rel - 1) - 2.0f * G(m_rel);
    g += coef * (-g_acc * sgnf(ag - ao) + g_reg * 2.0f * ao);
  };
  acc_term(-1, 1.0f);
  acc_term(0, -2.0f);
  acc_term(1, 1.0f);
  dxo[i] = g;
}

// ---------------------------------------------------------------------------------------------
// Bottleneck statistics: commit = |k[ids]-z|^2 / (R E) (bottleneck.py:176), fit = mean(min_d) (:125),
// prenorm = |z - mean(z)| / sqrt(R E) (:105).   partial [RED_BLOCKS][4] = sum z, sum z^2, sum (k[id]-z)^2, sum dmin
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(RED_THREADS) void vq_latent_partial_kernel(const float* __restrict__ z,
                                                                        const float* __restrict__ zq,
                                                                        const float* __restrict__ dmin, int64_t R,
                                                                        int E, double* __restrict__ partial) {
  const int64_t n = R * E;
  double v[4] = {0.0, 0.0, 0.0, 0.0};
  for (int64_t i = (int64_t)blockIdx.x * RED_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * RED_THREADS) {
    const int64_t r = i / E;
    const int e = (int)(i - r * E);
    const float zv = z[i];
    const float d = zq[i] - zv;
    v[0] += (double)zv;
    v[1] += (double)zv * (double)zv;
    v[2] += (double)(d * d);
    if (e == 0 && dmin) v[3] += (double)dmin[r];
  }
  block_reduce_store<4>(v, partial);
}

// out[3] = {commit, fit, prenorm}
__global__ __launch_bounds__(RED_THREADS) void vq_latent_final_kernel(const double* __restrict__ partial, int nblocks,
                                                                      int64_t R, int E, float* __restrict__ out) {
  double s[4];
  final_reduce<4>(partial, nblocks, s);
  if (threadIdx.x == 0) {
    const double n = (double)R * E;
    const double mean = s[0] / n;
    double dev = s[1] - n * mean * mean;
    if (dev < 0.0) dev = 0.0;
    out[0] = (float)(s[2] / n);
    out[1] = (float)(s[3] / (double)R);
    out[2] = (float)(sqrt(dev) / sqrt(n));
  }
}

// d commit / d z = 2 (z - k[id]) * scale     (x_d is detached: bottleneck.py:176)
__global__ __launch_bounds__(256) void vq_commit_grad_kernel(const float* __restrict__ z, const float* __restrict__ zq,
                                                             int64_t R, int E, float scale,
                                                             const float* __restrict__ dzq, float* __restrict__ dz) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= R * E) return;
  float g = 2.0f * scale * (z[i] - zq[i]);
  if (dzq) g += dzq[i];      // straight-through estimator: d x_d / d x = I  (bottleneck.py:179)
  dz[i] = g;
}

// ---------------------------------------------------------------------------------------------
// EMA codebook update (bottleneck.py:63-94)
// ---------------------------------------------------------------------------------------------
// _k_sum[c][:] = sum_{r: ids[r]==c} z[r][:], _k_elem[c] = count.  Block (c, s) handles rows [1024 s, 1024 s + 1024)
// of code c: every thread tests 8 consecutive ids, a block-wide exclusive scan of the hit counts gives each thread
// its slot in the ordered row list, then all threads add the listed rows in ascending order.  The chunk partials
// are added in chunk order by vq_code_sums_reduce_kernel: a fixed summation order (deterministic), and a popular
// code's rows are spread over R/1024 blocks instead of serialising in one.
__global__ __launch_bounds__(128) void vq_code_sums_kernel(const float* __restrict__ z, const int64_t* __restrict__ ids,
                                                           int64_t R, int E, int K, float* __restrict__ part,
                                                           int* __restrict__ part_n) {
  const int c = blockIdx.x, sidx = blockIdx.y;
  __shared__ int rows[1024];
  __shared__ int wave_tot[2];
  const int e4 = E >> 2;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)sidx * 1024;
  const int64_t rb = r0 + (int64_t)threadIdx.x * 8;
  unsigned hits = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j)
    if (rb + j < R && ids[rb + j] == c) hits |= 1u << j;
  const int n = __popc(hits);
  int incl = n;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o, 64);
    if (lane >= o) incl += v;
  }
  if (lane == 63) wave_tot[wv] = incl;
  __syncthreads();
  int pos = incl - n + (wv ? wave_tot[0] : 0);
  const int nr = wave_tot[0] + wave_tot[1];
#pragma unroll
  for (int j = 0; j < 8; ++j)
    if (hits & (1u << j)) rows[pos++] = threadIdx.x * 8 + j;
  __syncthreads();
  f32x4 acc[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};   // E <= 1024
  int j = 0;
  for (; j + 4 <= nr; j += 4) {                            // 4 rows in flight, added in row order
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int q = threadIdx.x + u * 128;
      if (q < e4) {
        const f32x4 a0 = reinterpret_cast<const f32x4*>(z + (r0 + rows[j]) * E)[q];
        const f32x4 a1 = reinterpret_cast<const f32x4*>(z + (r0 + rows[j + 1]) * E)[q];
        const f32x4 a2 = reinterpret_cast<const f32x4*>(z + (r0 + rows[j + 2]) * E)[q];
        const f32x4 a3 = reinterpret_cast<const f32x4*>(z + (r0 + rows[j + 3]) * E)[q];
        acc[u] = (((acc[u] + a0) + a1) + a2) + a3;
      }
    }
  }
  for (; j < nr; ++j) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int q = threadIdx.x + u * 128;
      if (q < e4) acc[u] += reinterpret_cast<const f32x4*>(z + (r0 + rows[j]) * E)[q];
    }
  }
  float* dst = part + ((size_t)sidx * K + c) * E;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int q = threadIdx.x + u * 128;
    if (q < e4) reinterpret_cast<f32x4*>(dst)[q] = acc[u];
  }
  if (threadIdx.x == 0) part_n[(size_t)sidx * K + c] = nr;
}

__global__ __launch_bounds__(256) void vq_code_sums_reduce_kernel(const float* __restrict__ part,
                                                                  const int* __restrict__ part_n, int S, int K, int E,
                                                                  float* __restrict__ ksum, float* __restrict__ kelem) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t n = (int64_t)K * E;
  if (i < n) {
    // chunks that hold no row of the code contribute an exact +0: skipping them keeps the sum identical to the
    // plain ascending-row sum
    float v = 0.f;
    const int c = (int)(i / E);
    for (int s = 0; s < S; ++s)
      if (part_n[(size_t)s * K + c]) v += part[(size_t)s * n + i];
    ksum[i] = v;
  } else if (i < n + K) {
    const int c = (int)(i - n);
    int t = 0;
    for (int s = 0; s < S; ++s) t += part_n[(size_t)s * K + c];
    kelem[c] = (float)t;
  }
}

// per code: EMA of k_sum / k_elem, new k (or the random-restart row), refresh of the transposed copy kT and |k|^2
// the quantiser GEMM uses; dk2[c] = |k_new - k_old|^2.
__global__ __launch_bounds__(128) void vq_ema_apply_kernel(float* __restrict__ k, float* __restrict__ k_sum,
                                                           float* __restrict__ k_elem, const float* __restrict__ bsum,
                                                           const float* __restrict__ belem,
                                                           const float* __restrict__ k_rand, float mu, float threshold,
                                                           int K, int E, float* __restrict__ kT, int ldkT,
                                                           float* __restrict__ kk, double* __restrict__ dk2) {
  const int c = blockIdx.x;
  const float ne = mu * k_elem[c] + (1.0f - mu) * belem[c];
  const float usage = ne >= threshold ? 1.0f : 0.0f;
  double d2 = 0.0, n2 = 0.0;
  for (int e = threadIdx.x; e < E; e += 128) {
    const size_t i = (size_t)c * E + e;
    const float ns = mu * k_sum[i] + (1.0f - mu) * bsum[i];
    k_sum[i] = ns;
    // bottleneck.py:78-79: usage * (k_sum / k_elem) + (1 - usage) * k_rand   (0 * inf = nan is the reference's too)
    const float kn = usage * (ns / ne) + (1.0f - usage) * k_rand[i];
    const float ko = k[i];
    k[i] = kn;
    if (kT) kT[(size_t)e * ldkT + c] = kn;
    d2 += (double)(kn - ko) * (double)(kn - ko);
    n2 += (double)kn * (double)kn;
  }
  __syncthreads();
  if (threadIdx.x == 0) k_elem[c] = ne;
  __shared__ double sm[2][2];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    d2 += __shfl_down(d2, o, 64);
    n2 += __shfl_down(n2, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    sm[threadIdx.x >> 6][0] = d2;
    sm[threadIdx.x >> 6][1] = n2;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    dk2[c] = sm[0][0] + sm[1][0];
    if (kk) kk[c] = (float)(sm[0][1] + sm[1][1]);
  }
}

// one block: out[4] = {entropy, used_curr, usage, dk}   (bottleneck.py:80-86)
__global__ __launch_bounds__(RED_THREADS) void vq_ema_stats_kernel(const float* __restrict__ belem,
                                                                   const float* __restrict__ k_elem,
                                                                   const double* __restrict__ dk2, float threshold,
                                                                   int K, int E, float* __restrict__ out) {
  __shared__ double sm[RED_THREADS];
  __shared__ double total_s;
  auto reduce = [&](double v) -> double {
    __syncthreads();
    sm[threadIdx.x] = v;
    __syncthreads();
    for (int s = RED_THREADS / 2; s > 0; s >>= 1) {
      if ((int)threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
      __syncthreads();
    }
    return sm[0];
  };
  double t = 0.0;
  for (int c = threadIdx.x; c < K; c += RED_THREADS) t += (double)belem[c];
  const double total = reduce(t);
  if (threadIdx.x == 0) total_s = total;
  __syncthreads();
  const float totf = (float)total_s;
  double ent = 0.0, used = 0.0, usage = 0.0, d2 = 0.0;
  for (int c = threadIdx.x; c < K; c += RED_THREADS) {
    const float p = belem[c] / totf;
    ent -= (double)(p * logf(p + 1e-8f));
    used += belem[c] >= threshold ? 1.0 : 0.0;
    usage += k_elem[c] >= threshold ? 1.0 : 0.0;
    d2 += dk2[c];
  }
  ent = reduce(ent);
  used = reduce(used);
  usage = reduce(usage);
  d2 = reduce(d2);
  if (threadIdx.x == 0) {
    out[0] = (float)ent;
    out[1] = (float)used;
    out[2] = (float)usage;
    out[3] = (float)(sqrt(d2) / sqrt((double)K * E));
  }
}

// Adam (torch.optim.Adam defaults the reference uses, train.py:71: betas, eps, no weight decay, no amsgrad):
// m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
__global__ __launch_bounds__(256) void adam_step_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                        float lr, float b1, float b2, float eps, float bc1,
                                                        float bc2_sqrt) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float gi = g[i];
  const float mi = b1 * m[i] + (1.0f - b1) * gi;
  const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
  m[i] = mi;
  v[i] = vi;
  const float denom = sqrtf(vi) / bc2_sqrt + eps;
  p[i] = p[i] - (lr / bc1) * (mi / denom);
}


// ---------------------------------------------------------------------------------------------
// Backward-weights of the convolutions (autograd of nn.Conv1d / ConvTranspose1d):
//   dW[tap][ci][co] = sum_{b,t} A(x[b][t*in_stride + in_offset + tap*dil][ci]) * dy[b][t*out_stride + out_offset][co]
//   db[co]          = sum_{b,t} dy[b][t*out_stride + out_offset][co]
// A GEMM whose contraction is the B*T_out position axis: both operands are read exactly as they lie in memory
// (position-major, channel contiguous), which is the layout v_mfma_f32_32x32x2_f32 wants when the contraction
// index is the slow one.  Tile = 128 (ci) x 128 (co) per block per tap, 4 waves of 64x64; the position axis is
// split over blockIdx.z (the output has only taps*Cin*Cout/16K tiles, far fewer than CUs) and the partial
// sums are added in split order by wgrad_reduce_kernel (deterministic).
// ---------------------------------------------------------------------------------------------
struct WgradArgs {
  const float* x;    // [B][T_in][Cin]
  const float* dy;   // [B][T_y][Cout]
  float* ws;         // [S][taps][Cin_pad][Cout_pad] then [S][Cout_pad]
  const float* zeros;  // >= 64 B of zeros (ctx): where out-of-range loads are pointed, so that no select follows them
  int B, T_in, Cin, Cin_pad, Cout, Cout_pad, taps;
  int in_stride, in_offset, dil, T_out, out_stride, out_offset, T_y;
  int relu_in, S;
  int64_t rows_per_split;
};

#define WG_BK 16     // positions per chunk (a multiple of 16; 32 - half the barriers, twice the LDS - measured equal: 0.435 vs 0.433 ms)
template <bool VECX, bool VECY>
__global__ __launch_bounds__(256, 4) void conv_wgrad_mfma_kernel(WgradArgs a) {
  __shared__ __attribute__((aligned(16))) float Xs[WG_BK][128];
  __shared__ __attribute__((aligned(16))) float Ys[WG_BK][128];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = w & 1, wn = w >> 1;
  const int mblocks = (a.Cin_pad + 127) / 128;
  const int tap = blockIdx.x / mblocks;
  const int ci0 = (blockIdx.x - tap * mblocks) * 128;
  const int co0 = blockIdx.y * 128;
  const int split = blockIdx.z;
  const int64_t M = (int64_t)a.B * a.T_out;
  const int64_t r_begin = (int64_t)split * a.rows_per_split;
  int64_t r_end = r_begin + a.rows_per_split;
  if (r_end > M) r_end = M;
  const bool do_bias = blockIdx.x == 0;

  // staging: thread -> position rows kr, kr + 16, ... of the chunk and two 4-channel groups 64 channels apart, so that
  // eight consecutive lanes write 128 consecutive bytes of a row (ds_write_b128 without bank conflicts)
  constexpr int RH = WG_BK / 16;
  const int kr = tid >> 4, c4 = (tid & 15) * 4;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float bsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  // raw prefetch registers: out-of-range elements are read from the context's zero page, so the loaded values need no
  // select and the loads of the next chunk stay in flight across the MFMA block; ReLU happens at the LDS commit
  float xv[RH][8], yv[RH][8];
  // this thread's first position row, advanced by WG_BK per chunk (no division inside the loop)
  int64_t fr = r_begin + kr;
  int fb = (int)(fr / a.T_out);
  int ft = (int)(fr - (int64_t)fb * a.T_out);

  auto fetch = [&]() {
    int hb = fb, ht = ft;
#pragma unroll
    for (int h = 0; h < RH; ++h) {
      const bool live = fr + 16 * h < r_end;
      const int t_in = ht * a.in_stride + a.in_offset + tap * a.dil;
      const bool x_ok = live && t_in >= 0 && t_in < a.T_in;
      const float* xrow = a.x + ((int64_t)hb * a.T_in + (x_ok ? t_in : 0)) * a.Cin + ci0 + c4;
      const float* yrow = a.dy + ((int64_t)hb * a.T_y + (int64_t)ht * a.out_stride + a.out_offset) * a.Cout + co0 + c4;
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        if (VECX) {
          const bool ok = x_ok && (ci0 + c4 + 64 * v) < a.Cin;
          const f32x4 q = *reinterpret_cast<const f32x4*>(ok ? xrow + 64 * v : a.zeros);
#pragma unroll
          for (int i = 0; i < 4; ++i) xv[h][4 * v + i] = q[i];
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const bool ok = x_ok && (ci0 + c4 + 64 * v + i) < a.Cin;
            xv[h][4 * v + i] = *(ok ? xrow + 64 * v + i : a.zeros);
          }
        }
        if (VECY) {
          const bool ok = live && (co0 + c4 + 64 * v) < a.Cout;
          const f32x4 q = *reinterpret_cast<const f32x4*>(ok ? yrow + 64 * v : a.zeros);
#pragma unroll
          for (int i = 0; i < 4; ++i) yv[h][4 * v + i] = q[i];
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const bool ok = live && (co0 + c4 + 64 * v + i) < a.Cout;
            yv[h][4 * v + i] = *(ok ? yrow + 64 * v + i : a.zeros);
          }
        }
      }
      ht += 16;
      while (ht >= a.T_out) {
        ht -= a.T_out;
        ++hb;
      }
    }
    fr += WG_BK;
    fb = hb;
    ft = ht;
  };

  // ablation hooks (experiments/conv_probe): -DQPG_WGRAD_PROBE=n compiles parts of the chunk loop out; the product build
  // defines nothing.  1: no global prefetch after the first chunk; 2: and no commit / barriers; 3: MFMAs on constant
  // registers; 4: as 3, and the partial tile is not stored.
#ifndef QPG_WGRAD_PROBE
#define QPG_WGRAD_PROBE 0
#endif
  if (r_begin < r_end) fetch();
  for (int64_t r0 = r_begin; r0 < r_end; r0 += WG_BK) {
#if QPG_WGRAD_PROBE < 2
    __syncthreads();
#pragma unroll
    for (int h = 0; h < RH; ++h) {
      if (a.relu_in) {
#pragma unroll
        for (int i = 0; i < 8; ++i) xv[h][i] = fmaxf(xv[h][i], 0.f);
      }
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        *reinterpret_cast<f32x4*>(&Xs[kr + 16 * h][c4 + 64 * v]) =
            f32x4{xv[h][4 * v], xv[h][4 * v + 1], xv[h][4 * v + 2], xv[h][4 * v + 3]};
        *reinterpret_cast<f32x4*>(&Ys[kr + 16 * h][c4 + 64 * v]) =
            f32x4{yv[h][4 * v], yv[h][4 * v + 1], yv[h][4 * v + 2], yv[h][4 * v + 3]};
      }
      if (do_bias) {
#pragma unroll
        for (int i = 0; i < 8; ++i) bsum[i] += yv[h][i];
      }
    }
    __syncthreads();
#endif
#if QPG_WGRAD_PROBE < 1
    if (r0 + WG_BK < r_end) fetch();
#endif
    float aq[2][2], bq[2][2];
    auto lds_read = [&](int ks, int slot) {
#if QPG_WGRAD_PROBE >= 3
      aq[slot][0] = aq[slot][1] = bq[slot][0] = bq[slot][1] = 1.0f;
      return;
#endif
      const int k = ks * 2 + (lane >> 5);
      aq[slot][0] = Xs[k][wm * 64 + (lane & 31)];
      aq[slot][1] = Xs[k][wm * 64 + 32 + (lane & 31)];
      bq[slot][0] = Ys[k][wn * 64 + (lane & 31)];
      bq[slot][1] = Ys[k][wn * 64 + 32 + (lane & 31)];
    };
    lds_read(0, 0);
#pragma unroll
    for (int ks = 0; ks < WG_BK / 2; ++ks) {
      const int cur = ks & 1;
      if (ks + 1 < WG_BK / 2) lds_read(ks + 1, cur ^ 1);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[cur][0], bq[cur][0], acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[cur][0], bq[cur][1], acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[cur][1], bq[cur][0], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[cur][1], bq[cur][1], acc[1][1], 0, 0, 0);
    }
  }

  // partial dW tile: C layout col = lane&31 (co), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (ci)
  float* wsz = a.ws + ((int64_t)split * a.taps + tap) * a.Cin_pad * a.Cout_pad;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int co = co0 + wn * 64 + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = ci0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
#if QPG_WGRAD_PROBE >= 4
        if (acc[i][j][r] != 12345.678f) continue;
#endif
        if (ci < a.Cin_pad) wsz[(int64_t)ci * a.Cout_pad + co] = acc[i][j][r];
      }
    }
  if (do_bias) {
    // column sums of this split's dy rows: add the 16 staging threads of a column in row order
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) Xs[kr][c4 + 64 * (i >> 2) + (i & 3)] = bsum[i];
    __syncthreads();
    if (tid < 128) {
      float sacc = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) sacc += Xs[k][tid];
      float* bws = a.ws + (int64_t)a.S * a.taps * a.Cin_pad * a.Cout_pad + (int64_t)split * a.Cout_pad;
      bws[co0 + tid] = sacc;
    }
  }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, int S, int64_t n_w,
                                                           int Cout_pad, float* __restrict__ dw,
                                                           float* __restrict__ db, int accumulate_bias) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n_w) {
    float v = 0.f;
    for (int s = 0; s < S; ++s) v += ws[(int64_t)s * n_w + i];
    dw[i] = v;
  } else if (db && i < n_w + Cout_pad) {
    const int c = (int)(i - n_w);
    const float* bws = ws + (int64_t)S * n_w;
    float v = 0.f;
    for (int s = 0; s < S; ++s) v += bws[(int64_t)s * Cout_pad + c];
    db[c] = accumulate_bias ? db[c] + v : v;
  }
}

}  // namespace

extern "C" int64_t qpg_vq_reduce_ws_bytes(void) { return (int64_t)RED_BLOCKS * 4 * sizeof(double); }

extern "C" int qpg_vq_loss_f32(qpg_ctx* ctx, void* stream, const float* x_out, const float* x_target, int B, int T,
                               int C, const float* commit_loss, float w_commit, float w_reg, float w_vel, float w_acc,
                               void* ws, int64_t ws_bytes, float* out6) {
  QPG_REQUIRE(ctx && x_out && x_target && ws && out6 && B > 0 && T > 0 && C > 0, "qpg_vq_loss_f32: bad argument");
  QPG_REQUIRE(ws_bytes >= qpg_vq_reduce_ws_bytes(), "qpg_vq_loss_f32: workspace too small");
  const int64_t n = (int64_t)B * T * C;
  int nb = (int)((n + RED_THREADS - 1) / RED_THREADS);
  if (nb > RED_BLOCKS) nb = RED_BLOCKS;
  hipLaunchKernelGGL(vq_loss_partial_kernel, dim3(nb), dim3(RED_THREADS), 0, qpg_stream(stream), x_out, x_target, B, T,
                     C, (double*)ws);
  QPG_LAUNCH_CHECK("vq_loss_partial_kernel");
  hipLaunchKernelGGL(vq_loss_final_kernel, dim3(1), dim3(RED_THREADS), 0, qpg_stream(stream), (const double*)ws, nb, B,
                     T, C, commit_loss, w_commit, w_reg, w_vel, w_acc, out6);
  QPG_LAUNCH_CHECK("vq_loss_final_kernel");
  return QPG_OK;
}

extern "C" int qpg_vq_loss_grad_f32(qpg_ctx* ctx, void* stream, const float* x_out, const float* x_target, int B, int T,
                                    int C, float w_reg, float w_vel, float w_acc, float upstream, float* d_x_out) {
  QPG_REQUIRE(ctx && x_out && x_target && d_x_out && B > 0 && T > 0 && C > 0, "qpg_vq_loss_grad_f32: bad argument");
  const int64_t n = (int64_t)B * T * C;
  const float g_rec = upstream / (float)n;
  const float g_vel = T > 1 ? upstream * w_vel / (float)((int64_t)B * (T - 1) * C) : 0.0f;
  const float g_acc = T > 2 ? upstream * w_acc / (float)((int64_t)B * (T - 2) * C) : 0.0f;
  const float g_reg = T > 2 ? upstream * w_reg / (float)((int64_t)B * (T - 2) * C) : 0.0f;
  hipLaunchKernelGGL(vq_loss_grad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, qpg_stream(stream), x_out,
                     x_target, B, T, C, g_rec, g_vel, g_acc, g_reg, d_x_out);
  QPG_LAUNCH_CHECK("vq_loss_grad_kernel");
  return QPG_OK;
}

extern "C" int qpg_vq_latent_stats_f32(qpg_ctx* ctx, void* stream, const float* z, const float* zq, const float* dmin,
                                       int64_t R, int E, void* ws, int64_t ws_bytes, float* out3) {
  QPG_REQUIRE(ctx && z && zq && ws && out3 && R > 0 && E > 0, "qpg_vq_latent_stats_f32: bad argument");
  QPG_REQUIRE(ws_bytes >= qpg_vq_reduce_ws_bytes(), "qpg_vq_latent_stats_f32: workspace too small");
  const int64_t n = R * E;
  int nb = (int)((n + RED_THREADS - 1) / RED_THREADS);
  if (nb > RED_BLOCKS) nb = RED_BLOCKS;
  hipLaunchKernelGGL(vq_latent_partial_kernel, dim3(nb), dim3(RED_THREADS), 0, qpg_stream(stream), z, zq, dmin, R, E,
                     (double*)ws);
  QPG_LAUNCH_CHECK("vq_latent_partial_kernel");
  hipLaunchKernelGGL(vq_latent_final_kernel, dim3(1), dim3(RED_THREADS), 0, qpg_stream(stream), (const double*)ws, nb,
                     R, E, out3);
  QPG_LAUNCH_CHECK("vq_latent_final_kernel");
  return QPG_OK;
}

extern "C" int qpg_vq_commit_grad_f32(qpg_ctx* ctx, void* stream, const float* z, const float* zq, int64_t R, int E,
                                      float scale, const float* d_zq, float* d_z) {
  QPG_REQUIRE(ctx && z && zq && d_z && R > 0 && E > 0, "qpg_vq_commit_grad_f32: bad argument");
  const int64_t n = R * E;
  hipLaunchKernelGGL(vq_commit_grad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, qpg_stream(stream), z, zq,
                     R, E, scale / (float)n, d_zq, d_z);
  QPG_LAUNCH_CHECK("vq_commit_grad_kernel");
  return QPG_OK;
}

extern "C" int64_t qpg_vq_code_sums_ws_bytes(int64_t R, int E, int K) {
  const int64_t S = (R + 1023) / 1024;
  return S * K * ((int64_t)E * sizeof(float) + sizeof(int));
}

extern "C" int qpg_vq_code_sums_f32(qpg_ctx* ctx, void* stream, const float* z, const int64_t* ids, int64_t R, int E,
                                    int K, float* batch_sum, float* batch_elem, void* ws, int64_t ws_bytes) {
  QPG_REQUIRE(ctx && z && ids && batch_sum && batch_elem && ws && R >= 0 && K > 0, "qpg_vq_code_sums_f32: bad argument");
  QPG_REQUIRE(E > 0 && (E % 4) == 0 && E <= 1024, "qpg_vq_code_sums_f32: emb_width must be a multiple of 4, <= 1024");
  QPG_REQUIRE(ws_bytes >= qpg_vq_code_sums_ws_bytes(R, E, K), "qpg_vq_code_sums_f32: workspace too small");
  const int S = (int)((R + 1023) / 1024);
  float* part = (float*)ws;
  int* part_n = (int*)(part + (size_t)S * K * E);
  if (S > 0) {
    hipLaunchKernelGGL(vq_code_sums_kernel, dim3(K, S), dim3(128), 0, qpg_stream(stream), z, ids, R, E, K, part,
                       part_n);
    QPG_LAUNCH_CHECK("vq_code_sums_kernel");
  }
  const int64_t n = (int64_t)K * E + K;
  hipLaunchKernelGGL(vq_code_sums_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, qpg_stream(stream),
                     (const float*)part, (const int*)part_n, S, K, E, batch_sum, batch_elem);
  QPG_LAUNCH_CHECK("vq_code_sums_reduce_kernel");
  return QPG_OK;
}

extern "C" int qpg_vq_ema_update_f32(qpg_ctx* ctx, void* stream, float* k, float* k_sum, float* k_elem,
                                     const float* batch_sum, const float* batch_elem, const float* k_rand, float mu,
                                     float threshold, int K, int E, float* kT, int ldkT, float* kk, void* ws,
                                     int64_t ws_bytes, float* out4) {
  QPG_REQUIRE(ctx && k && k_sum && k_elem && batch_sum && batch_elem && k_rand && ws && out4 && K > 0 && E > 0,
              "qpg_vq_ema_update_f32: bad argument");
  QPG_REQUIRE(ws_bytes >= (int64_t)K * (int64_t)sizeof(double), "qpg_vq_ema_update_f32: workspace too small");
  QPG_REQUIRE(!kT || ldkT >= K, "qpg_vq_ema_update_f32: ldkT < K");
  hipLaunchKernelGGL(vq_ema_apply_kernel, dim3(K), dim3(128), 0, qpg_stream(stream), k, k_sum, k_elem, batch_sum,
                     batch_elem, k_rand, mu, threshold, K, E, kT, ldkT, kk, (double*)ws);
  QPG_LAUNCH_CHECK("vq_ema_apply_kernel");
  hipLaunchKernelGGL(vq_ema_stats_kernel, dim3(1), dim3(RED_THREADS), 0, qpg_stream(stream), batch_elem, k_elem,
                     (const double*)ws, threshold, K, E, out4);
  QPG_LAUNCH_CHECK("vq_ema_stats_kernel");
  return QPG_OK;
}

extern "C" int qpg_adam_step_f32(qpg_ctx* ctx, void* stream, float* param, const float* grad, float* exp_avg,
                                 float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2, float eps,
                                 int64_t step) {
  QPG_REQUIRE(ctx && param && grad && exp_avg && exp_avg_sq && n >= 0 && step >= 1, "qpg_adam_step_f32: bad argument");
  if (n == 0) return QPG_OK;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  hipLaunchKernelGGL(adam_step_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, qpg_stream(stream), param, grad,
                     exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, (float)bc1, (float)sqrt(bc2));
  QPG_LAUNCH_CHECK("adam_step_kernel");
  return QPG_OK;
}

extern "C" int64_t qpg_conv1d_wgrad_ws_floats(int taps, int Cin_pad, int Cout_pad, int splits) {
  return (int64_t)splits * ((int64_t)taps * Cin_pad * Cout_pad + Cout_pad);
}

extern "C" int qpg_conv1d_bwd_weight_f32(qpg_ctx* ctx, void* stream, const float* x, int B, int T_in, int Cin,
                                         const float* dy, int taps, int Cin_pad, int Cout, int Cout_pad, int in_stride,
                                         int in_offset, int dil, int T_out, int out_stride, int out_offset, int T_y,
                                         int relu_in, float* dw, float* db, int accumulate_bias, float* ws,
                                         int64_t ws_floats) {
  QPG_REQUIRE(ctx && x && dy && dw && ws, "qpg_conv1d_bwd_weight_f32: null pointer");
  QPG_REQUIRE(B > 0 && T_in > 0 && Cin > 0 && taps > 0 && Cout > 0 && T_out > 0 && T_y > 0 && out_stride > 0 &&
                  in_stride > 0 && dil > 0,
              "qpg_conv1d_bwd_weight_f32: bad size");
  QPG_REQUIRE(Cin_pad >= Cin && Cin_pad % 16 == 0 && Cout_pad >= Cout && Cout_pad % 128 == 0,
              "qpg_conv1d_bwd_weight_f32: packed gradient must be padded like the weights");
  const int64_t M = (int64_t)B * T_out;
  const int mblocks = (Cin_pad + 127) / 128;
  const int64_t tiles = (int64_t)taps * mblocks * (Cout_pad / 128);
  const int64_t n_w = (int64_t)taps * Cin_pad * Cout_pad;
  // ONE round of blocks: a CU holds four blocks of this kernel (__launch_bounds__(256, 4): <= 128 VGPRs; 16 KB of LDS),
  // so tiles * S is kept at or below 4 * n_cu.  (Rounded UP, as it was until round 3, 48 tiles x 22 splits = 1056 blocks
  // needed a second round for the last 32 of them and the launch took two rounds' time.)
  int S = (int)(4 * (int64_t)ctx->n_cu / tiles);
  if (S < 1) S = 1;
  const int64_t max_by_rows = (M + 4 * WG_BK - 1) / (4 * WG_BK);     // at least four chunks of positions per split
  if (S > max_by_rows) S = (int)max_by_rows;
  const int64_t max_by_ws = ws_floats / (n_w + Cout_pad);
  if (S > max_by_ws) S = (int)max_by_ws;
  QPG_REQUIRE(S >= 1, "qpg_conv1d_bwd_weight_f32: workspace smaller than one partial (%lld floats needed)",
              (long long)(n_w + Cout_pad));
  WgradArgs a;
  a.x = x; a.dy = dy; a.ws = ws; a.zeros = ctx->zeros;
  a.B = B; a.T_in = T_in; a.Cin = Cin; a.Cin_pad = Cin_pad; a.Cout = Cout; a.Cout_pad = Cout_pad; a.taps = taps;
  a.in_stride = in_stride; a.in_offset = in_offset; a.dil = dil; a.T_out = T_out;
  a.out_stride = out_stride; a.out_offset = out_offset; a.T_y = T_y; a.relu_in = relu_in; a.S = S;
  a.rows_per_split = ((M + S - 1) / S + WG_BK - 1) / WG_BK * WG_BK;
  const bool vx = (Cin % 4) == 0 && (reinterpret_cast<uintptr_t>(x) % 16) == 0;
  const bool vy = (Cout % 4) == 0 && (reinterpret_cast<uintptr_t>(dy) % 16) == 0;
  dim3 grid((unsigned)(taps * mblocks), (unsigned)(Cout_pad / 128), (unsigned)S);
  hipStream_t st = qpg_stream(stream);
  if (vx && vy) hipLaunchKernelGGL((conv_wgrad_mfma_kernel<true, true>), grid, dim3(256), 0, st, a);
  else if (vx) hipLaunchKernelGGL((conv_wgrad_mfma_kernel<true, false>), grid, dim3(256), 0, st, a);
  else if (vy) hipLaunchKernelGGL((conv_wgrad_mfma_kernel<false, true>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((conv_wgrad_mfma_kernel<false, false>), grid, dim3(256), 0, st, a);
  QPG_LAUNCH_CHECK("conv_wgrad_mfma_kernel");
  const int64_t n = n_w + (db ? Cout_pad : 0);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float*)ws, S,
                     n_w, Cout_pad, dw, db, accumulate_bias);
  QPG_LAUNCH_CHECK("wgrad_reduce_kernel");
  return QPG_OK;
}
