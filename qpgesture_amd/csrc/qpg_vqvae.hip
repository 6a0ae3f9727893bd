// Gesture VQ-VAE encode / quantise / decode (codebook/models/{vqvae,encdec,resnet,bottleneck}.py).
//
// Every layer of the reference's Encoder / Decoder is a 1-D convolution (strided k4, dilated k3,
// 1x1, k3, and ConvTranspose1d k4 s2 p1 which splits into two 2-tap convolutions, one per output
// parity), i.e. a GEMM with M = batch x time positions, N = output channels, K = taps x input
// channels.  They all run through ONE implicit-GEMM kernel on the f32 matrix cores
// (v_mfma_f32_32x32x2_f32: exact f32 FMA chains, no reduced-precision path, so decoded poses stay
// within the 1e-4 parity bar of the torch-CPU reference).
//
// Layout: activations are channels-last [B][T][C] f32 (the pose tensors are (B,T,135) on disk and at
// the API, so the reference's NTC<->NCT permutes disappear; the K axis of every tap is contiguous).
// Weights are repacked once at load to [tap][Cin_pad][Cout_pad] (K-major rows, output channel
// contiguous, zero padded to the tile) so the B operand needs no guards.
//
// Tile: 64 positions x 128 channels per 256-thread block, 4 waves = 2(M) x 2(N), each wave two
// 32x32 MFMA tiles; K advances in 16-wide slices staged through LDS (A rows padded to 17 floats:
// conflict-free column reads; B rows read along the channel axis: conflict-free).  Fused in the
// epilogue: bias, optional ReLU, optional residual add; fused in the A load: optional input ReLU
// (ResConv1DBlock = x + conv1x1(relu(conv3_dil(relu(x)))), resnet.py:31-46).
#include "qpg_common.h"

#define CV_BM 64
#define CV_BN 128
#define CV_BK 16

struct ConvArgs {
  const float* x;      // [B][T_in][Cin]
  const float* w;      // [taps][Cin_pad][Cout_pad]
  const float* bias;   // [Cout_pad]
  const float* res;    // residual, same indexing as y, or null
  float* y;            // [B][T_y][Cout]
  int B, T_in, Cin, Cin_pad, Cout, Cout_pad, taps;
  int in_stride, in_offset, dil;   // t_in = t*in_stride + in_offset + tap*dil
  int T_out;                       // output positions computed per batch item in this launch
  int out_stride, out_offset, T_y; // y row = t*out_stride + out_offset, T_y rows per batch item
  int relu_in, relu_out;
};

__global__ __launch_bounds__(256) void conv1d_mfma_f32_kernel(ConvArgs a) {
  __shared__ float As[CV_BM][CV_BK + 1];
  __shared__ __attribute__((aligned(16))) float Bs[CV_BK][CV_BN];

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = w & 1, wn = w >> 1;
  const int64_t M = (int64_t)a.B * a.T_out;
  const int64_t m0 = (int64_t)blockIdx.x * CV_BM;
  const int n0 = blockIdx.y * CV_BN;

  // A staging: thread -> (row, 4 consecutive k)
  const int ar = tid >> 2, ak = (tid & 3) * 4;
  const int64_t am = m0 + ar;
  const bool a_live = am < M;
  const int ab = a_live ? (int)(am / a.T_out) : 0;
  const int at = a_live ? (int)(am - (int64_t)ab * a.T_out) : 0;
  // B staging: thread -> (k row, 8 consecutive n)
  const int bk = tid >> 4, bn = (tid & 15) * 8;

  f32x16 acc0, acc1;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc0[i] = acc1[i] = 0.f;

  for (int tap = 0; tap < a.taps; ++tap) {
    const int t_in = at * a.in_stride + a.in_offset + tap * a.dil;
    const bool t_ok = a_live && t_in >= 0 && t_in < a.T_in;
    const float* xrow = a.x + ((int64_t)ab * a.T_in + (t_ok ? t_in : 0)) * a.Cin;
    const float* wtap = a.w + (int64_t)tap * a.Cin_pad * a.Cout_pad;
    for (int c0 = 0; c0 < a.Cin_pad; c0 += CV_BK) {
      float av[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ci = c0 + ak + i;
        float v = (t_ok && ci < a.Cin) ? xrow[ci] : 0.f;
        if (a.relu_in) v = fmaxf(v, 0.f);
        av[i] = v;
      }
      const float* wp = wtap + (int64_t)(c0 + bk) * a.Cout_pad + n0 + bn;
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(wp), b1 = *reinterpret_cast<const f32x4*>(wp + 4);
      __syncthreads();   // previous slice fully consumed
#pragma unroll
      for (int i = 0; i < 4; ++i) As[ar][ak + i] = av[i];
      *reinterpret_cast<f32x4*>(&Bs[bk][bn]) = b0;
      *reinterpret_cast<f32x4*>(&Bs[bk][bn + 4]) = b1;
      __syncthreads();
#pragma unroll
      for (int ks = 0; ks < CV_BK / 2; ++ks) {
        const int k = ks * 2 + (lane >> 5);
        const float av_ = As[wm * 32 + (lane & 31)][k];
        const float bv0 = Bs[k][wn * 64 + (lane & 31)], bv1 = Bs[k][wn * 64 + 32 + (lane & 31)];
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av_, bv0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av_, bv1, acc1, 0, 0, 0);
      }
    }
  }

  // C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int n = n0 + wn * 64 + half * 32 + (lane & 31);
    if (n >= a.Cout) continue;
    const float bias = a.bias ? a.bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const int64_t m = m0 + wm * 32 + row;
      if (m >= M) continue;
      const int b = (int)(m / a.T_out);
      const int t = (int)(m - (int64_t)b * a.T_out);
      const int64_t o = ((int64_t)b * a.T_y + (int64_t)t * a.out_stride + a.out_offset) * a.Cout + n;
      float v = (half ? acc1[r] : acc0[r]) + bias;
      if (a.relu_out) v = fmaxf(v, 0.f);
      if (a.res) v = a.res[o] + v;
      a.y[o] = v;
    }
  }
}

extern "C" int qpg_conv1d_f32(qpg_ctx* ctx, void* stream, const float* x, int B, int T_in, int Cin, const float* w,
                              const float* bias, int taps, int Cin_pad, int Cout, int Cout_pad, int in_stride,
                              int in_offset, int dil, int T_out, int out_stride, int out_offset, int T_y,
                              const float* residual, int relu_in, int relu_out, float* y) {
  QPG_REQUIRE(ctx && x && w && y, "qpg_conv1d_f32: null pointer");
  QPG_REQUIRE(B >= 0 && T_in > 0 && Cin > 0 && taps > 0 && Cout > 0 && T_out >= 0 && T_y > 0 && out_stride > 0 &&
                  in_stride > 0 && dil > 0,
              "qpg_conv1d_f32: bad size");
  QPG_REQUIRE(Cin_pad >= Cin && Cin_pad % CV_BK == 0 && Cout_pad >= Cout && Cout_pad % CV_BN == 0,
              "qpg_conv1d_f32: packed weights must be padded to Cin %% %d == 0, Cout %% %d == 0", CV_BK, CV_BN);
  if (B == 0 || T_out == 0) return QPG_OK;
  ConvArgs a;
  a.x = x; a.w = w; a.bias = bias; a.res = residual; a.y = y;
  a.B = B; a.T_in = T_in; a.Cin = Cin; a.Cin_pad = Cin_pad; a.Cout = Cout; a.Cout_pad = Cout_pad; a.taps = taps;
  a.in_stride = in_stride; a.in_offset = in_offset; a.dil = dil; a.T_out = T_out;
  a.out_stride = out_stride; a.out_offset = out_offset; a.T_y = T_y; a.relu_in = relu_in; a.relu_out = relu_out;
  const int64_t M = (int64_t)B * T_out;
  dim3 grid((unsigned)((M + CV_BM - 1) / CV_BM), (unsigned)(Cout_pad / CV_BN));
  hipLaunchKernelGGL(conv1d_mfma_f32_kernel, grid, dim3(256), 0, qpg_stream(stream), a);
  QPG_LAUNCH_CHECK("conv1d_mfma_f32_kernel");
  return QPG_OK;
}

// ---------------------------------------------------------------------------------------------
// BottleneckBlock.quantise (bottleneck.py:120-126): distance = sum(x^2) - 2 x.k^T + sum(k^2), argmin.
// The x.k^T GEMM runs through qpg_conv1d_f32 (taps = 1, weights = k^T); this kernel finishes one row per
// wave: d[c] = (xx - 2*dot[c]) + kk[c] in f32 in that order, min with lowest-index ties.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vq_argmin_kernel(const float* __restrict__ z, const float* __restrict__ dot,
                                                        const float* __restrict__ kk, int64_t R, int E, int K,
                                                        int64_t* __restrict__ ids, float* __restrict__ dmin,
                                                        float* __restrict__ dsecond) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  float xx = 0.f;
  for (int e = lane; e < E; e += 64) {
    const float v = z[r * E + e];
    xx = fmaf(v, v, xx);
  }
  for (int o = 32; o > 0; o >>= 1) xx += __shfl_xor(xx, o, 64);
  float best = __builtin_inff(), second = __builtin_inff();
  int bi = 0x7fffffff;
  for (int c = lane; c < K; c += 64) {
    const float d = (xx - 2.f * dot[r * K + c]) + kk[c];
    if (d < best || (d == best && c < bi)) {
      second = best;
      best = d;
      bi = c;
    } else if (d < second) {
      second = d;
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64), os = __shfl_xor(second, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ob < best || (ob == best && oi < bi)) {
      second = fminf(best, os);
      best = ob;
      bi = oi;
    } else {
      second = fminf(second, ob);
    }
  }
  if (lane == 0) {
    ids[r] = bi;
    if (dmin) dmin[r] = best;
    if (dsecond) dsecond[r] = second;
  }
}

extern "C" int qpg_vq_argmin_f32(qpg_ctx* ctx, void* stream, const float* z, const float* dot, const float* kk,
                                 int64_t R, int E, int K, int64_t* ids, float* dmin, float* dsecond) {
  QPG_REQUIRE(ctx && z && dot && kk && ids && R >= 0 && E > 0 && K > 0, "qpg_vq_argmin_f32: bad argument");
  if (R == 0) return QPG_OK;
  hipLaunchKernelGGL(vq_argmin_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, qpg_stream(stream), z, dot, kk, R,
                     E, K, ids, dmin, dsecond);
  QPG_LAUNCH_CHECK("vq_argmin_kernel");
  return QPG_OK;
}

// BottleneckBlock.dequantise (bottleneck.py:128-130): F.embedding gather, written channels-last.
__global__ __launch_bounds__(256) void vq_gather_kernel(const float* __restrict__ k, const int64_t* __restrict__ ids,
                                                        int64_t R, int E, int K, float* __restrict__ out,
                                                        int* __restrict__ status) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int e4 = E >> 2;
  if (i >= R * e4) return;
  const int64_t r = i / e4;
  const int c = (int)(i - r * e4);
  int64_t id = ids[r];
  if (id < 0 || id >= K) {
    if (status) *status = 1;
    id = 0;
  }
  reinterpret_cast<f32x4*>(out)[i] = reinterpret_cast<const f32x4*>(k)[id * e4 + c];
}

extern "C" int qpg_vq_gather_f32(qpg_ctx* ctx, void* stream, const float* k, const int64_t* ids, int64_t R, int E,
                                 int K, float* out, int32_t* status) {
  QPG_REQUIRE(ctx && k && ids && out && R >= 0 && E > 0 && (E % 4) == 0 && K > 0, "qpg_vq_gather_f32: bad argument");
  if (R == 0) return QPG_OK;
  const int64_t n = R * (E / 4);
  hipLaunchKernelGGL(vq_gather_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, qpg_stream(stream), k, ids, R,
                     E, K, out, status);
  QPG_LAUNCH_CHECK("vq_gather_kernel");
  return QPG_OK;
}
