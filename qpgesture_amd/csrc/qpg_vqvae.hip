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
// Tile: 64 or 128 positions x 128 channels per 256-thread block, 4 waves = 2(M) x 2(N), each wave 1x2 or
// 2x2 32x32 MFMA tiles; K advances in 16-wide slices staged through LDS (A rows padded to 17 floats:
// conflict-free column reads; B rows read along the channel axis: conflict-free).  Fused in the
// epilogue: bias, optional ReLU, optional residual add; fused in the A load: optional input ReLU
// (ResConv1DBlock = x + conv1x1(relu(conv3_dil(relu(x)))), resnet.py:31-46).
#include "qpg_common.h"

void qpg_launch_sub_inplace(void* stream, float* a, const float* b, int64_t n);

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
  const float* zeros;  // >= 64 B of zeros (ctx): where out-of-range 16-B tile loads are pointed
  const float* gate;   // backward-data only: same indexing as y, result is zeroed where gate <= 0 (ReLU backward)
  int wt_rows, wt_pitch;   // backward-data only: the forward layer's Cin_pad (rows per tap) and Cout_pad (row pitch)
  int tap_base, tap_step;  // weight tap used for input tap j = tap_base + j*tap_step (flips / parity subsets)
  int ksplit;          // > 1: blockIdx.z owns a contiguous range of K slices and writes raw partial sums to ws
  float* ws;           // [ksplit][M][Cout_pad]
  int vec_out;         // Cout % 4 == 0, y / gate / res 16-byte aligned, B * T_out < 2^31: the epilogue moves 16-byte rows
};

// MT = 32-row MFMA tiles per wave along M (block tile = 64*MT positions x 128 channels); VEC = 16-B loads of
// the activation rows (needs Cin % 4 == 0; the 135-channel input layer takes the scalar path).
// The next K slice's global loads are issued into registers before the current slice's MFMAs (register
// double-buffer), so HBM/L2 latency overlaps the matrix pipe with a single LDS buffer.
// WT = backward-data: the contraction runs over the packed weights' OUTPUT-channel axis and the result is indexed
// by their input-channel axis (B tile = W[tap]^T read in place, no transposed copy of the weights is kept).
template <int MT, bool VEC, bool WT = false>
__global__ __launch_bounds__(256, MT == 2 ? 4 : 2) void conv1d_mfma_f32_kernel(ConvArgs a) {
  constexpr int BM = 64 * MT;
  constexpr int AR = BM * CV_BK / 256;                 // A floats per thread per slice: 4 (MT=1) or 8 (MT=2)
  __shared__ __attribute__((aligned(16))) float As[1][BM][CV_BK + 1];
  __shared__ __attribute__((aligned(16))) float Bs[1][CV_BK][CV_BN];

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = w & 1, wn = w >> 1;
  const int64_t M = (int64_t)a.B * a.T_out;
  const int64_t m0 = (int64_t)blockIdx.x * BM;
  const int n0 = blockIdx.y * CV_BN;

  // A staging: thread -> (row, AR consecutive k)
  const int ar = tid / (CV_BK / AR), ak = (tid % (CV_BK / AR)) * AR;
  const int64_t am = m0 + ar;
  const bool a_live = am < M;
  const int ab = a_live ? (int)(am / a.T_out) : 0;
  const int at = a_live ? (int)(am - (int64_t)ab * a.T_out) : 0;
  // B staging: thread -> (k row, 8 consecutive n); transposed: thread -> (n, 8 consecutive k)
  const int bk = WT ? (tid & 1) * 8 : tid >> 4, bn = WT ? tid >> 1 : (tid & 15) * 8;

  f32x16 acc[MT][2];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mt][h][i] = 0.f;

  const int nslice = a.Cin_pad / CV_BK;
  const int total = a.taps * nslice;
  // Global loads are issued RAW (clamped addresses, no select on the result) so that nothing waits on them until
  // `commit` turns them into LDS stores: the loads of slice it+1 stay in flight across the whole MFMA block of
  // slice it.  `okm` remembers which elements are real (bit per 4-group / per element).
  float av[AR];
  f32x4 b0, b1;
  unsigned okm = 0;
  // slice iterator (tap, c0) advanced incrementally; everything that depends only on the tap (row pointer, range
  // check, weight tap pointer) is recomputed at tap boundaries, not per slice
  int f_tap = 0, f_c0 = 0;
  const float* xrow_tap = a.x;
  const float* w_tap = a.w;
  bool t_ok = false;
  const float* xbase = a.x + (int64_t)ab * a.T_in * a.Cin + ak;
  const bool w_ok = !WT || n0 + bn < a.wt_rows;
  auto set_tap = [&](int tap) {
    const int t_in = at * a.in_stride + a.in_offset + tap * a.dil;
    t_ok = a_live && t_in >= 0 && t_in < a.T_in;
    xrow_tap = xbase + (int64_t)(t_ok ? t_in : 0) * a.Cin;
    const int wtap = a.tap_base + tap * a.tap_step;
    // WT: rows of the packed weights are the forward layer's input channels (this launch's n axis), the
    // contraction index k runs along a row; a.wt_rows rows per tap, pitch a.wt_pitch
    w_tap = WT ? a.w + ((int64_t)wtap * a.wt_rows + (w_ok ? n0 + bn : 0)) * a.wt_pitch + bk
               : a.w + ((int64_t)wtap * a.Cin_pad + bk) * a.Cout_pad + n0 + bn;
  };
  auto seek = [&](int it) {
    f_tap = it / nslice;
    f_c0 = (it - f_tap * nslice) * CV_BK;
    set_tap(f_tap);
  };
  auto fetch = [&]() {
    const int c0 = f_c0;
    const float* xrow = xrow_tap + c0;
    okm = 0;
    if (VEC) {
      // Cin % 4 == 0 and Cin_pad == round_up(Cin,16): a 4-group is either fully inside or fully outside; an
      // outside group is read from the context's zero page, so the loaded value needs no select at all
#pragma unroll
      for (int v = 0; v < AR / 4; ++v) {
        const bool ok = t_ok && (c0 + ak + 4 * v) < a.Cin;
        const f32x4 x4 = *reinterpret_cast<const f32x4*>(ok ? xrow + 4 * v : a.zeros);
#pragma unroll
        for (int i = 0; i < 4; ++i) av[4 * v + i] = x4[i];
      }
    } else {
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        const bool ok = t_ok && (c0 + ak + i) < a.Cin;
        av[i] = *(ok ? xrow + i : a.zeros);
      }
    }
    const float* wp = WT ? w_tap + c0 : w_tap + (int64_t)c0 * a.Cout_pad;
#if defined(QPG_CONV_PROBE) && QPG_CONV_PROBE == 5      // probe: the weight tile comes from the zero page (one hot line)
    wp = a.zeros;
#endif
    b0 = *reinterpret_cast<const f32x4*>(wp);
    b1 = *reinterpret_cast<const f32x4*>(wp + 4);
    if (WT && !w_ok) okm = 0x80000000u;
    f_c0 += CV_BK;
    if (f_c0 == a.Cin_pad) {
      f_c0 = 0;
      ++f_tap;
      set_tap(f_tap);
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      float v = av[i];
      if (a.relu_in) v = __builtin_amdgcn_fmed3f(v, 0.f, __builtin_inff());   // one v_med3_f32 = max(v, 0)
      As[buf][ar][ak + i] = v;
    }
    if (WT) {
      const bool z = (okm & 0x80000000u) != 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        Bs[buf][bk + i][bn] = z ? 0.f : b0[i];
        Bs[buf][bk + 4 + i][bn] = z ? 0.f : b1[i];
      }
    } else {
      *reinterpret_cast<f32x4*>(&Bs[buf][bk][bn]) = b0;
      *reinterpret_cast<f32x4*>(&Bs[buf][bk][bn + 4]) = b1;
    }
  };

  // split-K (short sequences: a handful of blocks would otherwise walk the whole contraction alone)
  int it0 = 0, it1 = total;
  if (a.ksplit > 1) {
    const int per = (total + a.ksplit - 1) / a.ksplit;
    it0 = blockIdx.z * per;
    it1 = it0 + per < total ? it0 + per : total;
  }
  // Pipeline: slice it+1's global loads are in flight (raw, in registers) across slice it's whole MFMA block; they
  // are committed to the single LDS tile at the top of the next iteration.  (An LDS double buffer with one barrier
  // per slice was measured slower: 4.57 vs 4.42 ms for the B=256 encode.)
  if (it0 < it1) {
    seek(it0);
    fetch();
  }
  for (int it = it0; it < it1; ++it) {
    constexpr int buf = 0;
#if !defined(QPG_CONV_PROBE) || QPG_CONV_PROBE == 1
    __syncthreads();   // previous slice fully consumed
    commit(buf);
    __syncthreads();
#endif
#if !defined(QPG_CONV_PROBE)
    if (it + 1 < it1) fetch();           // in flight during the MFMAs below
#endif
    // LDS operand reads run one k-pair ahead of the MFMAs that consume them
    float bq[2][2], aq[2][MT];
    auto lds_read = [&](int ks, int slot) {
#if defined(QPG_CONV_PROBE) && QPG_CONV_PROBE >= 3
      bq[slot][0] = bq[slot][1] = 1.0f;
      for (int mt = 0; mt < MT; ++mt) aq[slot][mt] = 1.0f;
      return;
#endif
      const int k = ks * 2 + (lane >> 5);
      bq[slot][0] = Bs[buf][k][wn * 64 + (lane & 31)];
      bq[slot][1] = Bs[buf][k][wn * 64 + 32 + (lane & 31)];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) aq[slot][mt] = As[buf][(wm * MT + mt) * 32 + (lane & 31)][k];
    };
    lds_read(0, 0);
#pragma unroll
    for (int ks = 0; ks < CV_BK / 2; ++ks) {
      const int cur = ks & 1;
      if (ks + 1 < CV_BK / 2) lds_read(ks + 1, cur ^ 1);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[cur][mt], bq[cur][0], acc[mt][0], 0, 0, 0);
        acc[mt][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[cur][mt], bq[cur][1], acc[mt][1], 0, 0, 0);
      }
    }
  }

  // C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  if (a.ksplit > 1) {
    float* wsz = a.ws + (int64_t)blockIdx.z * M * a.Cout_pad;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int n = n0 + wn * 64 + half * 32 + (lane & 31);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t m = m0 + (wm * MT + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (m < M) wsz[m * a.Cout_pad + n] = acc[mt][half][r];
        }
    }
    return;
  }
#if !defined(QPG_CONV_PROBE) || QPG_CONV_PROBE < 4
  if (a.vec_out) {
    // A lane of the 32x32 accumulator tile holds ONE channel of 16 positions: written out as it lies, every store /
    // gate / residual access is 64 scattered dwords.  Each wave turns its tiles through 2 KB of the (now idle) operand
    // tiles instead - 16 positions x 32 channels per pass, written as the accumulators lie, read back as 16-byte rows -
    // so that a wave instruction moves whole 128-byte lines, a quarter of the memory instructions.  (Probe 4 of
    // experiments/conv_probe: the scalar epilogue was 15 % of a k3 512 -> 512 layer at T = 120.)
    __syncthreads();                                   // the last slice's operand reads are done
    float* stage = w < 2 ? &As[0][0][0] + w * 512 : &Bs[0][0][0] + (w - 2) * 512;
    const unsigned T_out = (unsigned)a.T_out;
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
#pragma unroll
          for (int gg = 0; gg < 2; ++gg)
#pragma unroll
            for (int i = 0; i < 4; ++i)
              stage[(8 * gg + 4 * (lane >> 5) + i) * 32 + (lane & 31)] = acc[mt][half][4 * (2 * p + gg) + i];
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            f32x4 v = *reinterpret_cast<const f32x4*>(stage + (lane + 64 * j) * 4);
            const int n = n0 + wn * 64 + half * 32 + (lane & 7) * 4;
            const int64_t m = m0 + (wm * MT + mt) * 32 + 16 * p + (lane >> 3) + 8 * j;
            if (m < M && n < a.Cout) {
              const unsigned b = (unsigned)m / T_out, t = (unsigned)m - b * T_out;
              const int64_t o = ((int64_t)b * a.T_y + (int64_t)t * a.out_stride + a.out_offset) * a.Cout + n;
              if (a.bias) v += *reinterpret_cast<const f32x4*>(a.bias + n);
              if (a.relu_out) v = f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
              if (a.gate) {
                const f32x4 g = *reinterpret_cast<const f32x4*>(a.gate + o);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = g[e] > 0.f ? v[e] : 0.f;
              }
              if (a.res) {
                const f32x4 rr = *reinterpret_cast<const f32x4*>(a.res + o);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = rr[e] + v[e];
              }
              *reinterpret_cast<f32x4*>(a.y + o) = v;
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    return;
  }
#endif
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int n = n0 + wn * 64 + half * 32 + (lane & 31);
    if (n >= a.Cout) continue;
    const float bias = a.bias ? a.bias[n] : 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int64_t m = m0 + (wm * MT + mt) * 32 + row;
        if (m >= M) continue;
        const int b = (int)(m / a.T_out);
        const int t = (int)(m - (int64_t)b * a.T_out);
        const int64_t o = ((int64_t)b * a.T_y + (int64_t)t * a.out_stride + a.out_offset) * a.Cout + n;
        float v = acc[mt][half][r] + bias;
#if defined(QPG_CONV_PROBE) && QPG_CONV_PROBE == 4
        if (v != 12345.678f) continue;        // probe: keep the MFMAs alive, skip the epilogue's memory traffic
#endif
        if (a.relu_out) v = fmaxf(v, 0.f);
        if (a.gate) v = a.gate[o] > 0.f ? v : 0.f;
        if (a.res) v = a.res[o] + v;
        a.y[o] = v;
      }
    }
  }
}

// split-K epilogue: partial sums added in slice order (fixed), then bias / ReLU / residual as in the fused path
__global__ __launch_bounds__(256) void conv_splitk_reduce_kernel(ConvArgs a) {
  const int64_t M = (int64_t)a.B * a.T_out;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * a.Cout) return;
  const int64_t m = i / a.Cout;
  const int n = (int)(i - m * a.Cout);
  float v = 0.f;
  for (int z = 0; z < a.ksplit; ++z) v += a.ws[((int64_t)z * M + m) * a.Cout_pad + n];
  v += a.bias ? a.bias[n] : 0.f;
  const int b = (int)(m / a.T_out);
  const int t = (int)(m - (int64_t)b * a.T_out);
  const int64_t o = ((int64_t)b * a.T_y + (int64_t)t * a.out_stride + a.out_offset) * a.Cout + n;
  if (a.relu_out) v = fmaxf(v, 0.f);
  if (a.gate) v = a.gate[o] > 0.f ? v : 0.f;
  if (a.res) v = a.res[o] + v;
  a.y[o] = v;
}

static int conv_launch(qpg_ctx* ctx, void* stream, ConvArgs& a, bool wt, float* ws, int64_t ws_floats) {
  const int64_t M = (int64_t)a.B * a.T_out;
  const bool vec = (a.Cin % 4) == 0 && (reinterpret_cast<uintptr_t>(a.x) % 16) == 0;
  // 128-row tiles once there are enough rows to fill the chip with them, 64-row tiles for short sequences
  const bool big = M * (a.Cout_pad / CV_BN) >= (int64_t)128 * 2 * ctx->n_cu;
  const int BM = big ? 128 : 64;
  const int64_t blocks = ((M + BM - 1) / BM) * (a.Cout_pad / CV_BN);
  const int total = a.taps * (a.Cin_pad / CV_BK);
  // short sequences (one clip's decode: 180..1440 rows): split the contraction over blockIdx.z so that the
  // launch fills the chip, partial sums go through the caller's scratch and are added in a fixed order
  int ks = 1;
  if (ws && blocks < ctx->n_cu && total >= 16) {
    ks = (int)((ctx->n_cu + blocks - 1) / blocks);
    if (ks > 8) ks = 8;
    if (ks > total / 8) ks = total / 8;
    if ((int64_t)ks * M * a.Cout_pad > ws_floats) ks = (int)(ws_floats / (M * a.Cout_pad));
    if (ks < 2) ks = 1;
  }
  a.ksplit = ks;
  a.ws = ws;
  const auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) % 16) == 0; };
  a.vec_out = (a.Cout % 4) == 0 && M < ((int64_t)1 << 31) && al16(a.y) && al16(a.gate) && al16(a.res) && al16(a.bias);
  dim3 grid((unsigned)((M + BM - 1) / BM), (unsigned)(a.Cout_pad / CV_BN), (unsigned)ks);
  hipStream_t st = qpg_stream(stream);
  if (wt) {
    if (big && vec) hipLaunchKernelGGL((conv1d_mfma_f32_kernel<2, true, true>), grid, dim3(256), 0, st, a);
    else if (big) hipLaunchKernelGGL((conv1d_mfma_f32_kernel<2, false, true>), grid, dim3(256), 0, st, a);
    else if (vec) hipLaunchKernelGGL((conv1d_mfma_f32_kernel<1, true, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv1d_mfma_f32_kernel<1, false, true>), grid, dim3(256), 0, st, a);
  } else {
    if (big && vec) hipLaunchKernelGGL((conv1d_mfma_f32_kernel<2, true>), grid, dim3(256), 0, st, a);
    else if (big) hipLaunchKernelGGL((conv1d_mfma_f32_kernel<2, false>), grid, dim3(256), 0, st, a);
    else if (vec) hipLaunchKernelGGL((conv1d_mfma_f32_kernel<1, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv1d_mfma_f32_kernel<1, false>), grid, dim3(256), 0, st, a);
  }
  QPG_LAUNCH_CHECK("conv1d_mfma_f32_kernel");
  if (ks > 1) {
    const int64_t n = M * a.Cout;
    hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a);
    QPG_LAUNCH_CHECK("conv_splitk_reduce_kernel");
  }
  return QPG_OK;
}

extern "C" int qpg_conv1d_f32(qpg_ctx* ctx, void* stream, const float* x, int B, int T_in, int Cin, const float* w,
                              const float* bias, int taps, int Cin_pad, int Cout, int Cout_pad, int in_stride,
                              int in_offset, int dil, int T_out, int out_stride, int out_offset, int T_y,
                              const float* residual, int relu_in, int relu_out, float* y, float* ws,
                              int64_t ws_floats) {
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
  a.gate = nullptr; a.tap_base = 0; a.tap_step = 1; a.wt_rows = 0; a.wt_pitch = 0; a.zeros = ctx->zeros;
  return conv_launch(ctx, stream, a, false, ws, ws_floats);
}

/* Backward-data of the same convolutions (autograd of nn.Conv1d / ConvTranspose1d, encdec.py / resnet.py): a
 * convolution of dy with the transposed weights, read in place from the forward layer's packed tensor. */
extern "C" int qpg_conv1d_bwd_data_f32(qpg_ctx* ctx, void* stream, const float* dy, int B, int T_in, int C_dy,
                                       const float* w_fwd, int taps, int fwd_Cin, int fwd_Cin_pad, int fwd_Cout_pad,
                                       int tap_base, int tap_step, int in_stride, int in_offset, int dil, int T_out,
                                       int out_stride, int out_offset, int T_y, const float* gate,
                                       const float* residual, float* dx, float* ws, int64_t ws_floats) {
  QPG_REQUIRE(ctx && dy && w_fwd && dx, "qpg_conv1d_bwd_data_f32: null pointer");
  QPG_REQUIRE(B >= 0 && T_in > 0 && C_dy > 0 && taps > 0 && fwd_Cin > 0 && T_out >= 0 && T_y > 0 && out_stride > 0 &&
                  in_stride > 0 && dil > 0,
              "qpg_conv1d_bwd_data_f32: bad size");
  QPG_REQUIRE(fwd_Cin_pad % CV_BK == 0 && fwd_Cin_pad >= fwd_Cin && fwd_Cout_pad % CV_BN == 0 && fwd_Cout_pad >= C_dy,
              "qpg_conv1d_bwd_data_f32: the forward layer's packing (Cin %% %d, Cout %% %d) is required", CV_BK, CV_BN);
  if (B == 0 || T_out == 0) return QPG_OK;
  ConvArgs a;
  a.x = dy; a.w = w_fwd; a.bias = nullptr; a.res = residual; a.y = dx;
  a.B = B; a.T_in = T_in; a.Cin = C_dy; a.Cin_pad = (C_dy + CV_BK - 1) / CV_BK * CV_BK;
  a.Cout = fwd_Cin; a.Cout_pad = (fwd_Cin_pad + CV_BN - 1) / CV_BN * CV_BN; a.taps = taps;
  a.in_stride = in_stride; a.in_offset = in_offset; a.dil = dil; a.T_out = T_out;
  a.out_stride = out_stride; a.out_offset = out_offset; a.T_y = T_y; a.relu_in = 0; a.relu_out = 0;
  a.zeros = ctx->zeros; a.gate = gate; a.tap_base = tap_base; a.tap_step = tap_step; a.wt_rows = fwd_Cin_pad; a.wt_pitch = fwd_Cout_pad;
  return conv_launch(ctx, stream, a, true, ws, ws_floats);
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

// ---------------------------------------------------------------------------------------------
// Whole-network orchestration (host C++): the layer sequences of Encoder / Decoder (encdec.py:53-136)
// issued back to back on the caller's stream.  Scratch = three [B][T_max][width] buffers.
// ---------------------------------------------------------------------------------------------
// split-K scratch of the whole-network calls: the 4th region of the workspace
#define VQ_SPLITK_ROWS 2048
struct SplitWs {
  float* p;
  int64_t n;
};
static int conv_call(qpg_ctx* ctx, void* stream, const qpg_conv_desc& c, const float* x, int B, int T_in, int in_stride,
                     int in_offset, int dil, int T_out, int out_stride, int out_offset, int T_y, const float* res,
                     int relu_in, int relu_out, float* y, SplitWs sw = SplitWs{nullptr, 0}) {
  return qpg_conv1d_f32(ctx, stream, x, B, T_in, c.cin, c.w, c.b, c.taps, c.cin_pad, c.cout, c.cout_pad, in_stride,
                        in_offset, dil, T_out, out_stride, out_offset, T_y, res, relu_in, relu_out, y, sw.p, sw.n);
}

static int ipow(int b, int e) {
  int r = 1;
  while (e-- > 0) r *= b;
  return r;
}

static bool model_ok(const qpg_vq_model* m) {
  return m && m->down_t > 0 && m->down_t <= QPG_VQ_MAX_DOWN && m->depth > 0 && m->depth <= QPG_VQ_MAX_DEPTH &&
         m->width > 0 && m->emb > 0 && m->bins > 0 && m->in_dim > 0 && m->growth > 0 && m->k && m->kk;
}

extern "C" int64_t qpg_vq_workspace_floats(const qpg_vq_model* m, int B, int T) {
  if (!model_ok(m) || B < 0 || T < 0) return -1;
  int64_t cmax = m->width > m->emb ? m->width : m->emb;
  if (m->bins > cmax) cmax = m->bins;
  const int64_t cpad = ((cmax + CV_BN - 1) / CV_BN) * CV_BN;
  return 3 * (((int64_t)B * T * cmax + 3) / 4 * 4) + 8 * (int64_t)VQ_SPLITK_ROWS * cpad + 64;
}

// x + conv1x1(relu(conv3_dil(relu(x)))) for each block; ping-pongs between cur and alt, h is the hidden buffer
static int resnet_run(qpg_ctx* ctx, void* stream, const qpg_conv_desc (*blocks)[2], int depth, int growth, bool reverse,
                      int B, int T, float*& cur, float*& alt, float* h, SplitWs sw) {
  for (int d = 0; d < depth; ++d) {
    const int dil = ipow(growth, reverse ? depth - 1 - d : d);                       // resnet.py:57-62
    int rc = conv_call(ctx, stream, blocks[d][0], cur, B, T, 1, -dil, dil, T, 1, 0, T, nullptr, 1, 1, h, sw);
    if (rc) return rc;
    rc = conv_call(ctx, stream, blocks[d][1], h, B, T, 1, 0, 1, T, 1, 0, T, cur, 0, 0, alt, sw);
    if (rc) return rc;
    float* t = cur; cur = alt; alt = t;
  }
  return QPG_OK;
}

// ---- transposed-formulation path (csrc/qpg_convt.hip): taken when the descriptor carries the T-packed images ----
extern "C" int qpg_convt_f32(qpg_ctx*, void*, const float*, int, int, int, const float*, const float*, int, int, int, int,
                             int, int, int, int, int, int, int, const float*, int, int, float*);
extern "C" int qpg_pad_channels_f32(qpg_ctx*, void*, const float*, int64_t, int, int, float*);
extern "C" int qpg_resblock_f32(qpg_ctx*, void*, const float*, int, int, int, const float*, const float*, const float*,
                                float*, float*);

static bool tpath_ok(const qpg_vq_model* m, bool enc) {
  if (m->width != 512 || m->emb != 512) return false;
  for (int i = 0; i < m->down_t; ++i) {
    if (enc ? !m->enc_down[i].wt : (!m->dec_up_even[i].wt || !m->dec_up_odd[i].wt)) return false;
    for (int d = 0; d < m->depth; ++d) {
      if (!(enc ? m->enc_res_pack[i][d] : m->dec_res_pack[i][d])) return false;
      const qpg_conv_desc* r = enc ? m->enc_res[i][d] : m->dec_res[i][d];
      if (!r[0].wt || !r[1].wt) return false;
    }
  }
  return enc ? (m->enc_out.wt && m->kT.wt) : (m->dec_in.wt && m->dec_out.wt);
}

static int convt_call(qpg_ctx* ctx, void* stream, const qpg_conv_desc& c, const float* x, int Cx, int B, int T_in,
                      int in_stride, int in_offset, int dil, int T_out, int out_stride, int out_offset, int T_y,
                      const float* res, int relu_in, int relu_out, float* y) {
  return qpg_convt_f32(ctx, stream, x, B, T_in, Cx, c.wt, c.b, c.taps, c.cin_pad, c.cout, c.cout_pad, in_stride,
                       in_offset, dil, T_out, out_stride, out_offset, T_y, res, relu_in, relu_out, y);
}

// Resnet1D with the fused block kernel where a launch of 64-row tiles fills the chip, the two-launch form below that
static int resnet_tpath(qpg_ctx* ctx, void* stream, const qpg_conv_desc (*blocks)[2], const float* const* packs, int depth,
                        int growth, bool reverse, int B, int T, float*& cur, float*& alt, float* h) {
  const int64_t tiles = ((int64_t)B * T + 63) / 64;
  const bool fused = tiles * 4 >= (int64_t)ctx->n_cu * 3;
  for (int d = 0; d < depth; ++d) {
    const int dil = ipow(growth, reverse ? depth - 1 - d : d);
    int rc;
    if (fused) {
      rc = qpg_resblock_f32(ctx, stream, cur, B, T, dil, packs[d], blocks[d][0].b, blocks[d][1].b, alt, nullptr);
    } else {
      rc = convt_call(ctx, stream, blocks[d][0], cur, 512, B, T, 1, -dil, dil, T, 1, 0, T, nullptr, 1, 1, h);
      if (rc) return rc;
      rc = convt_call(ctx, stream, blocks[d][1], h, 512, B, T, 1, 0, 1, T, 1, 0, T, cur, 0, 0, alt);
    }
    if (rc) return rc;
    float* t = cur; cur = alt; alt = t;
  }
  return QPG_OK;
}

static int encode_tpath(qpg_ctx* ctx, void* stream, const qpg_vq_model* m, const float* x, int B, int T, float* cur,
                        float* alt, float* h, int64_t* ids, float* latent, float* margin) {
  // pose rows (135 floats = 540 B) -> 16-byte aligned rows of cin_pad floats
  const int cp = m->enc_down[0].cin_pad;
  int rc = qpg_pad_channels_f32(ctx, stream, x, (int64_t)B * T, m->in_dim, cp, h);
  if (rc) return rc;
  const float* in = h;
  int Cx = cp, Tc = T;
  for (int i = 0; i < m->down_t; ++i) {
    const int To = Tc / 2;
    rc = convt_call(ctx, stream, m->enc_down[i], in, Cx, B, Tc, 2, -1, 1, To, 1, 0, To, nullptr, 0, 0, alt);
    if (rc) return rc;
    { float* t = cur; cur = alt; alt = t; }
    Tc = To;
    rc = resnet_tpath(ctx, stream, m->enc_res[i], m->enc_res_pack[i], m->depth, m->growth, false, B, Tc, cur, alt, h);
    if (rc) return rc;
    in = cur;
    Cx = 512;
  }
  float* z = latent ? latent : alt;
  rc = convt_call(ctx, stream, m->enc_out, cur, 512, B, Tc, 1, -1, 1, Tc, 1, 0, Tc, nullptr, 0, 0, z);
  if (rc) return rc;
  const int64_t R = (int64_t)B * Tc;
  QPG_REQUIRE(R < 0x7fffffffll, "qpg_vq_encode_f32: too many latent rows");
  float* dot = h;
  rc = convt_call(ctx, stream, m->kT, z, 512, 1, (int)R, 1, 0, 1, (int)R, 1, 0, (int)R, nullptr, 0, 0, dot);
  if (rc) return rc;
  float* dmin = margin ? cur : nullptr;
  rc = qpg_vq_argmin_f32(ctx, stream, z, dot, m->kk, R, m->emb, m->bins, ids, dmin, margin);
  if (rc) return rc;
  if (margin) {
    qpg_launch_sub_inplace(stream, margin, dmin, R);
    QPG_LAUNCH_CHECK("sub_inplace_kernel");
  }
  return QPG_OK;
}

extern "C" int qpg_vq_encode_f32(qpg_ctx* ctx, void* stream, const qpg_vq_model* m, const float* x, int B, int T,
                                 float* ws, int64_t ws_floats, int64_t* ids, float* latent, float* margin) {
  QPG_REQUIRE(ctx && model_ok(m) && x && ws && ids, "qpg_vq_encode_f32: bad argument");
  const int hop = ipow(2, m->down_t);
  QPG_REQUIRE(B >= 0 && T > 0 && T % hop == 0, "qpg_vq_encode_f32: T must be a multiple of %d", hop);
  QPG_REQUIRE(ws_floats >= qpg_vq_workspace_floats(m, B, T), "qpg_vq_encode_f32: workspace too small");
  if (B == 0) return QPG_OK;
  int64_t cmax = m->width > m->emb ? m->width : m->emb;
  if (m->bins > cmax) cmax = m->bins;
  const int64_t slab = (((int64_t)B * T * cmax + 3) / 4) * 4;                          // T/2 rows suffice; keep simple
  float *cur = ws, *alt = ws + slab, *h = ws + 2 * slab;
  const SplitWs sw{ws + 3 * slab, ws_floats - 3 * slab};
  if (tpath_ok(m, true)) return encode_tpath(ctx, stream, m, x, B, T, cur, alt, h, ids, latent, margin);
  const float* in = x;
  int Tc = T;
  for (int i = 0; i < m->down_t; ++i) {
    const int To = Tc / 2;
    int rc = conv_call(ctx, stream, m->enc_down[i], in, B, Tc, 2, -1, 1, To, 1, 0, To, nullptr, 0, 0, alt, sw);
    if (rc) return rc;
    { float* t = cur; cur = alt; alt = t; }
    Tc = To;
    rc = resnet_run(ctx, stream, m->enc_res[i], m->depth, m->growth, false, B, Tc, cur, alt, h, sw);
    if (rc) return rc;
    in = cur;
  }
  float* z = latent ? latent : alt;
  int rc = conv_call(ctx, stream, m->enc_out, cur, B, Tc, 1, -1, 1, Tc, 1, 0, Tc, nullptr, 0, 0, z, sw);
  if (rc) return rc;
  // quantise: dot = z . k^T (1-tap conv over the flattened rows), then the argmin kernel
  const int64_t R = (int64_t)B * Tc;
  QPG_REQUIRE(R < 0x7fffffffll, "qpg_vq_encode_f32: too many latent rows");
  float* dot = h;
  rc = conv_call(ctx, stream, m->kT, z, 1, (int)R, 1, 0, 1, (int)R, 1, 0, (int)R, nullptr, 0, 0, dot, sw);
  if (rc) return rc;
  float* dmin = margin ? cur : nullptr;                                               // cur is free now
  rc = qpg_vq_argmin_f32(ctx, stream, z, dot, m->kk, R, m->emb, m->bins, ids, dmin, margin);
  if (rc) return rc;
  if (margin) {   // margin currently holds the runner-up distance: subtract the minimum in place
    qpg_launch_sub_inplace(stream, margin, dmin, R);
    QPG_LAUNCH_CHECK("sub_inplace_kernel");
  }
  return QPG_OK;
}

__global__ void sub_inplace_kernel(float* a, const float* b, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = a[i] - b[i];
}
void qpg_launch_sub_inplace(void* stream, float* a, const float* b, int64_t n) {
  hipLaunchKernelGGL(sub_inplace_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, qpg_stream(stream), a, b, n);
}

static int decode_tpath(qpg_ctx* ctx, void* stream, const qpg_vq_model* m, const int64_t* ids, int B, int L, float* cur,
                        float* alt, float* h, float* out, int32_t* status) {
  int rc = qpg_vq_gather_f32(ctx, stream, m->k, ids, (int64_t)B * L, m->emb, m->bins, alt, status);   // dequantise
  if (rc) return rc;
  int Tc = L;
  rc = convt_call(ctx, stream, m->dec_in, alt, 512, B, Tc, 1, -1, 1, Tc, 1, 0, Tc, nullptr, 0, 0, cur);
  if (rc) return rc;
  for (int i = 0; i < m->down_t; ++i) {
    rc = resnet_tpath(ctx, stream, m->dec_res[i], m->dec_res_pack[i], m->depth, m->growth, m->reverse_dec != 0, B, Tc,
                      cur, alt, h);
    if (rc) return rc;
    // ConvTranspose1d(k4,s2,p1): y[2m] = x[m-1].W3 + x[m].W1 ; y[2m+1] = x[m].W2 + x[m+1].W0
    const qpg_conv_desc &ce = m->dec_up_even[i], &co = m->dec_up_odd[i];
    rc = qpg_convt_pair_f32(ctx, stream, cur, B, Tc, 512, ce.wt, ce.b, -1, 0, co.wt, co.b, 0, 1, ce.taps, ce.cin_pad, ce.cout,
                            ce.cout_pad, 1, 1, Tc, 2, 2 * Tc, alt);
    if (rc) return rc;
    { float* t = cur; cur = alt; alt = t; }
    Tc *= 2;
  }
  return convt_call(ctx, stream, m->dec_out, cur, 512, B, Tc, 1, -1, 1, Tc, 1, 0, Tc, nullptr, 0, 0, out);
}

extern "C" int qpg_vq_decode_f32(qpg_ctx* ctx, void* stream, const qpg_vq_model* m, const int64_t* ids, int B, int L,
                                 float* ws, int64_t ws_floats, float* out, int32_t* status) {
  QPG_REQUIRE(ctx && model_ok(m) && ids && ws && out, "qpg_vq_decode_f32: bad argument");
  QPG_REQUIRE(B >= 0 && L > 0 && (m->emb % 4) == 0, "qpg_vq_decode_f32: bad size");
  const int hop = ipow(2, m->down_t);
  const int T = L * hop;
  QPG_REQUIRE(ws_floats >= qpg_vq_workspace_floats(m, B, T), "qpg_vq_decode_f32: workspace too small");
  if (B == 0) return QPG_OK;
  int64_t cmax = m->width > m->emb ? m->width : m->emb;
  if (m->bins > cmax) cmax = m->bins;
  const int64_t slab = (((int64_t)B * T * cmax + 3) / 4) * 4;
  float *cur = ws, *alt = ws + slab, *h = ws + 2 * slab;
  const SplitWs sw{ws + 3 * slab, ws_floats - 3 * slab};
  if (tpath_ok(m, false)) return decode_tpath(ctx, stream, m, ids, B, L, cur, alt, h, out, status);
  int rc = qpg_vq_gather_f32(ctx, stream, m->k, ids, (int64_t)B * L, m->emb, m->bins, alt, status);   // dequantise
  if (rc) return rc;
  int Tc = L;
  rc = conv_call(ctx, stream, m->dec_in, alt, B, Tc, 1, -1, 1, Tc, 1, 0, Tc, nullptr, 0, 0, cur, sw);
  if (rc) return rc;
  for (int i = 0; i < m->down_t; ++i) {
    rc = resnet_run(ctx, stream, m->dec_res[i], m->depth, m->growth, m->reverse_dec != 0, B, Tc, cur, alt, h, sw);
    if (rc) return rc;
    // ConvTranspose1d(k4,s2,p1): y[2m] = x[m-1].W3 + x[m].W1 ; y[2m+1] = x[m].W2 + x[m+1].W0
    rc = conv_call(ctx, stream, m->dec_up_even[i], cur, B, Tc, 1, -1, 1, Tc, 2, 0, 2 * Tc, nullptr, 0, 0, alt, sw);
    if (rc) return rc;
    rc = conv_call(ctx, stream, m->dec_up_odd[i], cur, B, Tc, 1, 0, 1, Tc, 2, 1, 2 * Tc, nullptr, 0, 0, alt, sw);
    if (rc) return rc;
    { float* t = cur; cur = alt; alt = t; }
    Tc *= 2;
  }
  return conv_call(ctx, stream, m->dec_out, cur, B, Tc, 1, -1, 1, Tc, 1, 0, Tc, nullptr, 0, 0, out, sw);
}
