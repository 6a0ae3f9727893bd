// Split-operand f16 convolutions of the gesture VQ-VAE (round 5; VERDICT r4 #5: "split-f16 MFMA for the 512 -> 512 GEMMs
// with the same bound-and-recheck discipline, reported beside the f32 figure, never instead").
//
// The f32 matrix cores run at 1/16 of the f16 rate, and every convolution of encdec.py:8-51 / resnet.py:31-46 is a GEMM
// over K = taps x Cin (512 .. 2048).  As in the audio sweep (qpg_audio_hl.hip) every operand is written as x = h + l,
// h = fl16(x), l = fl16(x - h) (x - h is exact in f32), and a product becomes THREE v_mfma_f32_16x16x32_f16 - h l', l h',
// h h' (f16 x f16 products are exact in f32; the dropped l l' term is <= 2^-22 |x||y|) - accumulated in the MFMA's f32
// accumulator over the whole contraction: 3/16 of the f32 matrix time per flop.  What it is NOT: bit-identical to the
// f32 FMA chains of qpg_convt.hip / qpg_vqvae.hip.  Per GEMM the result is within ~(K / 32) 1.3 x 2^-24 of the exact
// sum relative to sum |products| (every chained instruction re-rounds the running sum; measured constants:
// selfcheck.py) + 2^-21 for the representation - the same order as an f32 FMA chain's own K 2^-24 worst case - so the
// latents agree with the f32 path to ~1e-5 and the poses to << 1e-4, but a code id whose two best codebook distances are
// closer than that may flip.  The host therefore keeps the f32 path as THE result and uses this one under a margin
// check: ids whose runner-up margin is inside the measured latent difference bound are re-encoded on the f32 kernels
// (qpgesture_amd/vqvae.py: encode(precision="f16x3")).
//
// Formulation (as qpg_convt.hip): y^T[co][m] = sum_k W^T[co][k] x^T[k][m], k = tap x Cin_pad + ci, m = (batch, time).
//   A operand = weights, 16 channels x 32 k per tile, pre-split and pre-packed in fragment order at load time, scaled by a
//     power of two per layer (max |w| 2^e in [2^9, 2^10): the l plane stays a normal f16 number; exact);
//   B operand = activations, 32 k x 16 positions per tile, read from the channels-last f32 rows (8 consecutive channels of
//     one position = one lane's fragment), ReLU'd if asked, split h | l on the fly and staged through LDS so that the
//     block's four waves share them; unscaled (|x| < 65504 is checked: status |= 1 otherwise, and the host takes the f32
//     path); small values' l planes are f16 subnormals: an absolute error <= 2^-25 per element, < 1e-6 of any output;
//   block = 128 channels x 128 positions, 4 waves = 2 x 2 of 64 x 64 (16 accumulator tiles = 64 VGPRs), K in slices of
//     32 through a double-buffered 2 x 32 KB LDS stage, ONE barrier per slice, 48 MFMAs per wave per 16 fragment reads.
// Epilogue: accumulators x 2^-e + bias, optional ReLU, optional residual, 16-byte stores (a lane holds 4 consecutive
// channels of one position).
#include "qpg_common.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

#define C16_BM 128          // positions per block
#define C16_BN 128          // output channels per block
#define C16_BK 32           // contraction slice
#define C16_SLICE_BYTES 16384   // one operand's slice: 8 tiles x 2 planes x 1 KB
// ablation hooks (experiments/round_scripts/r05_probe_conv16.sh): -DC16_PROBE=<bits>; the product build defines nothing.  1: no MFMAs; 2: no
// split + LDS stores in the loop; 4: no global requests in the loop; 8: no fragment reads from LDS; 16: no slice barrier;
// 32: the activation requests of a wave coalesced into one 2 KB run (timing only: wrong values)
#ifndef C16_PROBE
#define C16_PROBE 0
#endif
#ifndef C16_SGB
#define C16_SGB 5          // VALU operations named behind every MFMA of a slice (0: hipcc's own order)
#endif

struct Conv16Args {
  const float* x;          // [B][T_in][Cx] channels-last, Cx % 8 == 0
  const _Float16* wimg;    // [Cout_pad / 128][n_slice][8 tiles][2 planes][64 lanes][8]
  const float* bias;       // [Cout_pad] or null
  const float* res;        // residual indexed like y, or null
  float* y;                // [B][T_y][Cout]
  int B, T_in, Cx, Cin, Cin_pad, Cout, taps;
  int in_stride, in_offset, dil, T_out, out_stride, out_offset, T_y;
  int relu_in, relu_out, w_exp;
  const int32_t* w_exp_dev;   // or: the exponent is read from the image's metadata on the device (no host read-back)
  const float* zeros;
  int32_t* status;
};

__device__ __forceinline__ f32x4 mfma16h(h8 a, h8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

__global__ __launch_bounds__(256, 2) void conv1d_hl_kernel(Conv16Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // [2 buffers][weights 16 KB | activations 16 KB]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = w & 1, wn = w >> 1;                         // position half / channel half of this wave's 64 x 64 tile
  const int64_t M = (int64_t)a.B * a.T_out;
  const int64_t m0 = (int64_t)blockIdx.x * C16_BM;
  const int nb = blockIdx.y;
  const int spt = a.Cin_pad / C16_BK;                        // slices per tap
  const int n_slice = a.taps * spt;
  // ---- staging roles
  // activations: wave w stages k-group w (8 channels) of positions lane, lane + 64
  int sb[2], st[2];
  bool slive[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int64_t m = m0 + lane + 64 * i;
    slive[i] = m < M;
    sb[i] = slive[i] ? (int)(m / a.T_out) : 0;
    st[i] = slive[i] ? (int)(m - (int64_t)sb[i] * a.T_out) : 0;
  }
  const _Float16* wsrc = a.wimg + (int64_t)nb * n_slice * (C16_SLICE_BYTES / 2);
  f32x4 xr[2][2];
  bool xok[2];
  h8 wr[4];
  unsigned bad = 0;
  auto fetch = [&](int s) {
    const int tap = s / spt, c0 = (s - tap * spt) * C16_BK + 8 * w;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      // (no branch: a position outside the sequence / a channel group past Cin reads a CLAMPED in-range address and is
      // zeroed when it is split - a pointer select became control flow and cut the slice into basic blocks)
      const int t_in = st[i] * a.in_stride + a.in_offset + tap * a.dil;
      xok[i] = ((int)slive[i] & (int)(t_in >= 0) & (int)(t_in < a.T_in) & (int)(c0 < a.Cin)) != 0;   // (&, not &&: no branch)
      const int t_c = t_in < 0 ? 0 : (t_in < a.T_in ? t_in : a.T_in - 1);
      const int c_c = c0 + 8 <= a.Cx ? c0 : a.Cx - 8;
      const f32x4* p = reinterpret_cast<const f32x4*>(a.x + ((int64_t)sb[i] * a.T_in + t_c) * a.Cx + c_c);
      if (C16_PROBE & 32)     // TIMING ONLY (wrong values): the wave's 64 pieces of 32 bytes as ONE contiguous 2 KB run
        p = reinterpret_cast<const f32x4*>(a.x + ((int64_t)sb[0] * a.T_in) * a.Cx + (int64_t)(s & 7) * 4096 + w * 1024 + i * 512 + lane * 8);
      xr[i][0] = p[0];
      xr[i][1] = p[1];
    }
    const h8* wp = reinterpret_cast<const h8*>(wsrc + (int64_t)s * (C16_SLICE_BYTES / 2));
#pragma unroll
    for (int u = 0; u < 4; ++u) wr[u] = wp[tid + 256 * u];
  };
  typedef float f32x8_t __attribute__((ext_vector_type(8)));
  const float relu_lo1 = a.relu_in ? 0.f : -__builtin_inff();
  const f32x8_t relu_lo = {relu_lo1, relu_lo1, relu_lo1, relu_lo1, relu_lo1, relu_lo1, relu_lo1, relu_lo1};
  auto commit = [&](int buf) {
    h8* wl = reinterpret_cast<h8*>(lds + buf * 2 * C16_SLICE_BYTES);
#pragma unroll
    for (int u = 0; u < 4; ++u) wl[tid + 256 * u] = wr[u];
    h8* xl = reinterpret_cast<h8*>(lds + buf * 2 * C16_SLICE_BYTES + C16_SLICE_BYTES);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      // the split on PACKED operations (round 5, last hours: experiments/round_scripts/r05_probe_conv16.sh - this staging, one scalar
      // conversion at a time, cost the kernel as much as its MFMAs): v_pk_max_f32 for the ReLU, v_cvt_pk_f16_f32 (gfx950,
      // round to nearest even like the scalar conversion), v_pk_add_f32 for x - h; one range test per eight values
      typedef float f32x8 __attribute__((ext_vector_type(8)));
      f32x8 v = __builtin_shufflevector(xr[i][0], xr[i][1], 0, 1, 2, 3, 4, 5, 6, 7);
      const f32x8 zero8 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      v = xok[i] ? v : zero8;
      v = __builtin_elementwise_max(v, relu_lo);                       // (relu_lo = 0 or -inf: no select per element)
      const f32x8 av = __builtin_elementwise_abs(v);
      const float m01 = fmaxf(fmaxf(av[0], av[1]), fmaxf(av[2], av[3])), m23 = fmaxf(fmaxf(av[4], av[5]), fmaxf(av[6], av[7]));
      bad |= (!(fmaxf(m01, m23) <= 60000.f)) ? 1u : 0u;                 // (a NaN fails the test too)
      const h8 hh = __builtin_convertvector(v, h8);
      const f32x8 hb = __builtin_convertvector(hh, f32x8);
      const h8 ll = __builtin_convertvector(v - hb, h8);
      // fragment position of (position p = lane + 64 i, k-group w): tile p / 16, lane (p % 16) + 16 w
      const int p = lane + 64 * i, tile = p >> 4, fl = (p & 15) + 16 * w;
      xl[(tile * 2 + 0) * 64 + fl] = hh;
      xl[(tile * 2 + 1) * 64 + fl] = ll;
    }
  };
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  fetch(0);
  commit(0);
  fetch(n_slice > 1 ? 1 : 0);
  for (int s = 0; s < n_slice; ++s) {
    // slice s is in buffer s & 1; buffer (s + 1) & 1 was last read in slice s - 1.  An LDS-only barrier: the global loads
    // of the slice after next stay in flight across it (__syncthreads() is s_waitcnt vmcnt(0) first)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (!(C16_PROBE & 16)) __builtin_amdgcn_s_barrier();
    const int buf = s & 1;
    const h8* wl = reinterpret_cast<const h8*>(lds + buf * 2 * C16_SLICE_BYTES) + lane;
    const h8* xl = reinterpret_cast<const h8*>(lds + buf * 2 * C16_SLICE_BYTES + C16_SLICE_BYTES) + lane;
    h8 ah[4], al[4], bh[4], bl[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if ((C16_PROBE & 8) && s > 0) {
        asm volatile("" : "=v"(ah[i]), "=v"(al[i]), "=v"(bh[i]), "=v"(bl[i]));
        continue;
      }
      ah[i] = wl[((4 * wn + i) * 2 + 0) * 64];
      al[i] = wl[((4 * wn + i) * 2 + 1) * 64];
      bh[i] = xl[((4 * wm + i) * 2 + 0) * 64];
      bl[i] = xl[((4 * wm + i) * 2 + 1) * 64];
    }
    // The next slice's operands go to the other buffer, the slice after next is requested - UNCONDITIONALLY (behind the
    // last slices: a harmless re-stage of the last one), so that the slice is ONE basic block and the ~200 VALU
    // operations of the split (ReLU, range check, two conversions and a subtraction per element) + the LDS stores + the
    // address arithmetic of the requests can be issued UNDERNEATH the 48 MFMAs instead of in front of them (round 5, last
    // hours: hipcc had left the four phases - fragment reads, split + stores, requests, MFMAs - one after the other;
    // with two waves per SIMD the matrix pipe idled half the time).
    if (!(C16_PROBE & 2)) commit(buf ^ 1);
    if (!(C16_PROBE & 4)) fetch(s + 2 < n_slice ? s + 2 : n_slice - 1);
    if (C16_PROBE & 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(ah[i]), "v"(al[i]), "v"(bh[i]), "v"(bl[i]));
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[i][j] = mfma16h(ah[i], bl[j], acc[i][j]);
          acc[i][j] = mfma16h(al[i], bh[j], acc[i][j]);
          acc[i][j] = mfma16h(ah[i], bh[j], acc[i][j]);
        }
    }
#if C16_SGB
    // issue order: the 16 fragment reads up front (the first MFMAs need them), then per MFMA a few VALU operations of the
    // split; the LDS stores and the requests spread over the MFMAs behind the conversions that feed them
#pragma unroll
    for (int i = 0; i < 48; ++i) {
      if (i == 0) __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, C16_SGB, 0);
      if (i >= 16 && (i & 3) == 3) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      if (i >= 32 && (i & 1) == 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
#endif
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the surplus request must not outlive its registers)
  if (bad && a.status) atomicOr(a.status, 1);
  // ---- epilogue: lane (cg = lane & 15 -> position, rg = lane >> 4 -> channels 4 rg .. 4 rg + 3 of a 16-channel tile)
  const int cg = lane & 15, rg = lane >> 4;
  const float sc = __builtin_ldexpf(1.0f, -(a.w_exp_dev ? a.w_exp_dev[0] : a.w_exp));
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int64_t m = m0 + 64 * wm + 16 * j + cg;
    if (m >= M) continue;
    const int b = (int)(m / a.T_out), t = (int)(m - (int64_t)b * a.T_out);
    const int64_t row = ((int64_t)b * a.T_y + (int64_t)t * a.out_stride + a.out_offset) * a.Cout;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = nb * C16_BN + 64 * wn + 16 * i + 4 * rg;
      if (n >= a.Cout) continue;
      f32x4 v = acc[i][j] * sc;
      if (a.bias) v += *reinterpret_cast<const f32x4*>(a.bias + n);
      if (a.relu_out) v = (f32x4){fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
      if (n + 4 <= a.Cout && (a.Cout & 3) == 0) {
        if (a.res) v += *reinterpret_cast<const f32x4*>(a.res + row + n);
        *reinterpret_cast<f32x4*>(a.y + row + n) = v;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n + r < a.Cout) a.y[row + n + r] = v[r] + (a.res ? a.res[row + n + r] : 0.f);
      }
    }
  }
}

// ---- weight image ----------------------------------------------------------------------------------------------------
// w [taps][Cin_pad_w][Cout_pad_w] f32 (qpg_conv1d_f32's layout) -> split-f16 fragment image for Cin_pad = round_up(Cin, 32),
// Cout_pad = round_up(Cout, 128); meta[0] = the layer's scale exponent (written by conv16_wexp_kernel first).
__global__ __launch_bounds__(1024) void conv16_absmax_kernel(const float* __restrict__ w, int64_t n, unsigned int* __restrict__ out) {
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    m = fmaxf(m, fabsf(w[i]));
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o, 64));
  // one atomic per BLOCK (round 5: one per wave - 4 096 device-scope atomics on one address - took 48 us per layer, 2.3 ms of
  // a training step that re-packs its 48 images)
  __shared__ float red[16];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) m = fmaxf(m, red[i]);
    atomicMax(out, __float_as_uint(m));
  }
}

__global__ void conv16_wexp_kernel(unsigned int* __restrict__ amax_bits, int32_t* __restrict__ meta) {
  const float amax = __uint_as_float(amax_bits[0]);
  int e = 0;
  if (amax > 0.f && amax < 3.0e38f) {
    int ex;
    frexpf(amax, &ex);                 // amax = f 2^ex, f in [0.5, 1)
    e = 10 - ex;                       // amax 2^e in [2^9, 2^10)
    e = e > 60 ? 60 : (e < -60 ? -60 : e);
  }
  meta[0] = e;
}

__global__ __launch_bounds__(256) void conv16_pack_w_kernel(const float* __restrict__ w, int taps, int Cin, int Cin_pad_w,
                                                            int Cout, int Cout_pad_w, int Cin_pad, int n_blk,
                                                            const int32_t* __restrict__ meta, _Float16* __restrict__ img) {
  // one thread per (n block, slice, tile, lane): 8 k of one channel -> its h and l fragments
  const int spt = Cin_pad / C16_BK, n_slice = taps * spt;
  const int64_t total = (int64_t)n_blk * n_slice * 8 * 64;
  const float sc = ldexpf(1.0f, meta[0]);
  for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (int64_t)gridDim.x * blockDim.x) {
    const int ln = (int)(id & 63), tile = (int)((id >> 6) & 7);
    const int64_t rest = id >> 9;
    const int s = (int)(rest % n_slice), nbk = (int)(rest / n_slice);
    const int tap = s / spt, c0 = (s - tap * spt) * C16_BK + 8 * (ln >> 4);
    const int co = nbk * C16_BN + 16 * tile + (ln & 15);
    h8 hh, ll;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int ci = c0 + e;
      float v = 0.f;
      if (ci < Cin && co < Cout) v = w[((int64_t)tap * Cin_pad_w + ci) * Cout_pad_w + co] * sc;
      const _Float16 h = (_Float16)v;
      hh[e] = h;
      ll[e] = (_Float16)(v - (float)h);
    }
    h8* dst = reinterpret_cast<h8*>(img) + (((int64_t)nbk * n_slice + s) * 8 + tile) * 128;
    dst[ln] = hh;
    dst[64 + ln] = ll;
  }
}

static inline int64_t c16_round(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

extern "C" int64_t qpg_conv16_image_bytes(int taps, int Cin, int Cout) {
  if (taps <= 0 || Cin <= 0 || Cout <= 0 || taps > 64 || Cin > (1 << 20) || Cout > (1 << 20)) return 0;
  return taps * c16_round(Cin, C16_BK) * c16_round(Cout, C16_BN) * 4 + 64;      // 2 planes x 2 bytes per weight + meta
}

// w [dev] f32 [taps][Cin_pad_w][Cout_pad_w] -> image [dev] (qpg_conv16_image_bytes(taps, Cin, Cout) bytes, 16-byte aligned)
extern "C" int qpg_conv16_pack_weights(qpg_ctx* ctx, void* stream, const float* w, int taps, int Cin, int Cin_pad_w, int Cout,
                                       int Cout_pad_w, void* image, int64_t image_bytes) {
  const char* name = "qpg_conv16_pack_weights";
  QPG_REQUIRE(ctx && w && image && taps > 0 && Cin > 0 && Cout > 0 && Cin_pad_w >= Cin && Cout_pad_w >= Cout,
              "%s: bad argument", name);
  const int64_t need = qpg_conv16_image_bytes(taps, Cin, Cout);
  QPG_REQUIRE(need > 0 && image_bytes >= need && (reinterpret_cast<uintptr_t>(image) % 16) == 0,
              "%s: image too small or misaligned (qpg_conv16_image_bytes)", name);
  hipStream_t st = qpg_stream(stream);
  unsigned char* img = static_cast<unsigned char*>(image);
  int32_t* meta = reinterpret_cast<int32_t*>(img + need - 64);
  unsigned int* amax = reinterpret_cast<unsigned int*>(meta + 4);
  if (hipMemsetAsync(amax, 0, 4, st) != hipSuccess) {
    qpg_set_error("%s: memset failed", name);
    return QPG_EHIP;
  }
  hipLaunchKernelGGL(conv16_absmax_kernel, dim3(64), dim3(1024), 0, st, w, (int64_t)taps * Cin_pad_w * Cout_pad_w, amax);
  hipLaunchKernelGGL(conv16_wexp_kernel, dim3(1), dim3(1), 0, st, amax, meta);
  const int Cin_pad = (int)c16_round(Cin, C16_BK), n_blk = (int)(c16_round(Cout, C16_BN) / C16_BN);
  hipLaunchKernelGGL(conv16_pack_w_kernel, dim3(2048), dim3(256), 0, st, w, taps, Cin, Cin_pad_w, Cout, Cout_pad_w, Cin_pad,
                     n_blk, (const int32_t*)meta, reinterpret_cast<_Float16*>(img));
  QPG_LAUNCH_CHECK("conv16_pack_w_kernel");
  return QPG_OK;
}

// The convolution of qpg_conv1d_f32's contract (same geometry arguments) on the split-f16 path.  x rows have pitch Cx floats
// (Cx % 8 == 0, Cx >= Cin; channels >= Cin are never read); image from qpg_conv16_pack_weights for (taps, Cin, Cout);
// w_exp = the image's scale exponent (the i32 at byte qpg_conv16_image_bytes - 64 of the image, read back once by the host),
// or QPG_CONV16_WEXP_FROM_IMAGE: the kernel reads it there itself;
// status [dev] i32: |= 1 if an activation's magnitude left the f16 range (the result is then garbage: redo in f32).
extern "C" int qpg_conv16_f32(qpg_ctx* ctx, void* stream, const float* x, int B, int T_in, int Cx, int Cin, const void* image,
                              int w_exp, const float* bias, int taps, int Cout, int in_stride, int in_offset, int dil, int T_out,
                              int out_stride, int out_offset, int T_y, const float* residual, int relu_in, int relu_out,
                              float* y, int32_t* status) {
  const char* name = "qpg_conv16_f32";
  QPG_REQUIRE(ctx && x && image && y, "%s: null pointer", name);
  QPG_REQUIRE(B > 0 && T_in > 0 && taps > 0 && Cin > 0 && Cout > 0 && Cx >= Cin && (Cx % 8) == 0 && (Cin % 8) == 0 &&
                  in_stride > 0 && dil > 0 && T_out > 0 && out_stride > 0 && out_offset >= 0 && T_y > 0 &&
                  (int64_t)(T_out - 1) * out_stride + out_offset < T_y &&
                  ((w_exp >= -60 && w_exp <= 60) || w_exp == QPG_CONV16_WEXP_FROM_IMAGE),
              "%s: bad geometry (Cx %% 8 == 0, Cin %% 8 == 0)", name);
  QPG_REQUIRE((reinterpret_cast<uintptr_t>(x) % 16) == 0 && (reinterpret_cast<uintptr_t>(image) % 16) == 0 &&
                  (reinterpret_cast<uintptr_t>(y) % 16) == 0 && (!residual || (reinterpret_cast<uintptr_t>(residual) % 16) == 0) &&
                  (!bias || (reinterpret_cast<uintptr_t>(bias) % 16) == 0),
              "%s: 16-byte aligned pointers", name);
  Conv16Args a;
  a.x = x; a.wimg = static_cast<const _Float16*>(image); a.bias = bias; a.res = residual; a.y = y;
  a.B = B; a.T_in = T_in; a.Cx = Cx; a.Cin = Cin; a.Cin_pad = (int)c16_round(Cin, C16_BK); a.Cout = Cout; a.taps = taps;
  a.in_stride = in_stride; a.in_offset = in_offset; a.dil = dil; a.T_out = T_out; a.out_stride = out_stride;
  a.out_offset = out_offset; a.T_y = T_y; a.relu_in = relu_in; a.relu_out = relu_out; a.w_exp = w_exp; a.zeros = ctx->zeros;
  // (training: the weights - and with them the exponent - change every step; a host read-back per layer would be ~40
  // synchronisations per step)
  a.w_exp_dev = w_exp == QPG_CONV16_WEXP_FROM_IMAGE
                    ? reinterpret_cast<const int32_t*>(static_cast<const unsigned char*>(image) +
                                                       (qpg_conv16_image_bytes(taps, Cin, Cout) - 64))
                    : nullptr;
  a.status = status;
  const int64_t M = (int64_t)B * T_out;
  const int64_t gx = (M + C16_BM - 1) / C16_BM;
  const int gy = (int)(c16_round(Cout, C16_BN) / C16_BN);
  QPG_REQUIRE(gx < 0x7fffffffll, "%s: too many blocks", name);
  static bool raised = false;
  if (!raised) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv1d_hl_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            4 * C16_SLICE_BYTES) != hipSuccess) {
      qpg_set_error("%s: cannot raise the dynamic LDS limit", name);
      return QPG_EHIP;
    }
    raised = true;
  }
  hipLaunchKernelGGL(conv1d_hl_kernel, dim3((unsigned)gx, gy), dim3(256), 4 * C16_SLICE_BYTES, qpg_stream(stream), a);
  QPG_LAUNCH_CHECK("conv1d_hl_kernel");
  return QPG_OK;
}
