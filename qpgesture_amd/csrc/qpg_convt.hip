// Transposed-formulation convolution kernels of the gesture VQ-VAE (round 2).
//
// The layer-per-launch kernel of qpg_vqvae.hip stages BOTH operands through LDS and spends its time around the
// K loop (measured per layer, profiles/r02_encode_layers_before.md: dilated k3 layers 0.64-0.70 of the f32 matrix
// rate, 1x1 layers 0.46-0.55 - prologue / epilogue bound).  Here every convolution is computed TRANSPOSED,
//     yT[n][m] = sum_k  W^T[n][k] * xT[k][m]         (n = output channel, m = (batch, time) position),
// with v_mfma_f32_16x16x4_f32 (A = 16 channels x 4 k, B = 4 k x 16 positions, exact f32 FMA chains):
//   * a wave owns 16 positions and ALL output channels of its block: the activation operand is ONE float4 load
//     per lane per 16 contraction steps, straight from the channels-last rows into registers (no LDS, no
//     transposes); the weights - shared by the block's four waves - stream through LDS in 32 KB stages by LDS-DMA
//     (global_load_lds) from a pre-packed image, one s_barrier per 128 MFMAs per wave;
//   * the C/D layout of the MFMA (lane: column m = lane&15, rows n = 4*(lane>>4)+r) IS the B-operand layout of a
//     following GEMM over those rows (lane group g supplies k = 16*tile + 4g + r for step r), so the hidden
//     activation of a ResConv1DBlock (resnet.py:31-46:  x + conv1x1(relu(conv3_dil(relu(x)))))  never leaves the
//     accumulator registers: the 1x1 convolution consumes them directly.  One launch per block, no hidden
//     activation in memory, one epilogue instead of two.
//
// Weight image ("T-pack", built once by qpgesture_amd/vqvae.py):  Wt[nb][kb][g][nl][j] = W[k = 16 kb + 4 g + j]
// [n = nb*NB + nl]  with k = tap*Cin_pad + ci:  a lane's ds_read_b128 yields its channel's four consecutive k,
// 16 lanes of a group read 16 distinct 16-B bank slots (conflict-free), and a stage is one contiguous 32 KB run.
#include "qpg_common.h"

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

#define CT_ROWS 64              // positions per block: 4 waves x 16
#define CT_STAGE_FLOATS 8192    // 32 KB of packed weights per pipeline stage
#define CT_STAGE_BYTES 32768
#define CT_RES_LDS (2 * CT_STAGE_BYTES + 4096)   // two stage slots + the two bias vectors

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// one 32 KB stage: 32 wave-instructions of 1 KB (lane-linear in LDS), 8 per wave
__device__ __forceinline__ void ct_issue_stage(const float* src, unsigned char* slot, int w, int lane) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int p = i * 4 + w;
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + p * 256 + lane * 4), (lds_ptr_t)(slot + p * 1024), 16, 0, 0);
  }
}

__device__ __forceinline__ f32x4 relu4(f32x4 v) {
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = __builtin_amdgcn_fmed3f(v[i], 0.f, __builtin_inff());
  return v;
}

// ---------------------------------------------------------------------------------------------------------------
// Fused ResConv1DBlock, width 512:  y = x + W2 . relu(W1 (*) relu(x) + b1) + b2      (resnet.py:31-46)
// wpack: 96 stages of the k3 convolution (T-pack NB = 512: one 16-k block x 512 channels per stage) followed by
// 32 stages of the 1x1 convolution (T-pack NB = 128: four 16-k blocks x 128 channels per stage, chunk-major).
// ---------------------------------------------------------------------------------------------------------------
struct ResArgs {
  const float* x;       // [B][T][512]
  const float* wpack;   // 128 stages x 32 KB
  const float* b1;      // [512]
  const float* b2;      // [512]
  float* y;             // [B][T][512]
  float* h;             // optional [B][T][512]: relu(conv3(relu(x)) + b1), what the training step records
  int B, T, dil;
  const float* zeros;
};

// ablation hooks (experiments/resblock_probe): -DQPG_RES_PROBE=<bits> compiles parts of the stage out; the product
// build defines nothing.  1: no LDS-DMA after the prologue; 2: no wait / barrier; 4: no fragment reads after the first
// two groups of a stage; 8: no activation-fragment loads.
#ifndef QPG_RES_PROBE
#define QPG_RES_PROBE 0
#endif
#define RP_DMA (!(QPG_RES_PROBE & 1))
#define RP_BAR (!(QPG_RES_PROBE & 2))
#define RP_LDS (!(QPG_RES_PROBE & 4))
#define RP_FRAG (!(QPG_RES_PROBE & 8))

template <bool SAVE_H>
__global__ __launch_bounds__(256, 2) void resblock_fused_f32_kernel(ResArgs a) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, ml = lane & 15, g = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t M = (int64_t)a.B * a.T;
  const int64_t m = (int64_t)blockIdx.x * CT_ROWS + w * 16 + ml;
  const bool live = m < M;
  const int b = live ? (int)(m / a.T) : 0;
  const int t = live ? (int)(m - (int64_t)b * a.T) : 0;
  const float* xb = a.x + (int64_t)b * a.T * 512 + 4 * g;
  auto rowptr = [&](int tap) -> const float* {
    const int t_in = t + (tap - 1) * a.dil;
    return (live && t_in >= 0 && t_in < a.T) ? xb + (int64_t)t_in * 512 : nullptr;
  };

  f32x4 acc1[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) acc1[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // biases -> LDS behind the two stage slots (read back as one ds_read_b128 per 16-channel tile)
  float* bl = reinterpret_cast<float*>(lds + 2 * CT_STAGE_BYTES);
  bl[tid] = a.b1[tid];
  bl[256 + tid] = a.b1[256 + tid];
  bl[512 + tid] = a.b2[tid];
  bl[768 + tid] = a.b2[256 + tid];
  ct_issue_stage(a.wpack, lds, w, lane);
  // the three tap rows of this lane's position (zero page where the tap falls outside the sequence); the stage
  // loop below must stay ONE basic block (sched_group_barrier), so the tap switch is a select, not a branch
  const float* xr0 = rowptr(0);
  const float* xr1 = rowptr(1);
  const float* xr2 = rowptr(2);
  f32x4 bnext = *reinterpret_cast<const f32x4*>(xr0 ? xr0 : a.zeros);

  // ---- phase 1: hidden^T[512][16 positions per wave] over K = 3 taps x 512 channels, one 16-k block per stage
  for (int it = 0; it < 96; ++it) {
    if (RP_BAR) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();       // stage `it` landed for every wave; every wave is done reading stage it-1
    }
    // The activation fragment loaded during the previous stage is consumed HERE, before this stage's loads are
    // issued: hipcc waits vmcnt(0) at the first use of an ordinary load that has LDS-DMA behind it in the queue
    // (the DMA issued below would be drained every stage otherwise).
    f32x4 bcur = relu4(bnext);
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(bcur[j]));     // opaque use: the compiler's wait lands here
    __builtin_amdgcn_sched_barrier(0);
    const unsigned char* S = lds + (it & 1) * CT_STAGE_BYTES + (g * 512 + ml) * 16;
    f32x4 a4[2][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) a4[0][q] = *reinterpret_cast<const f32x4*>(S + q * 256);
#pragma unroll
    for (int q = 0; q < 4; ++q) a4[1][q] = *reinterpret_cast<const f32x4*>(S + (4 + q) * 256);
    if (RP_DMA)
      ct_issue_stage(a.wpack + (int64_t)(it + 1) * CT_STAGE_FLOATS, lds + ((it + 1) & 1) * CT_STAGE_BYTES, w, lane);
    if (RP_FRAG) {
      const int nx = it + 1, ntap = nx >> 5, nci = (nx & 31) * 16;
      const float* xp = ntap == 0 ? xr0 : (ntap == 1 ? xr1 : xr2);
      bnext = *reinterpret_cast<const f32x4*>((xp && nx < 96) ? xp + nci : a.zeros);
    }
#pragma unroll
    for (int grp = 0; grp < 8; ++grp) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc1[grp * 4 + q] = mfma16(a4[grp & 1][q][j], bcur[j], acc1[grp * 4 + q]);
      if (grp + 2 < 8 && RP_LDS) {
#pragma unroll
        for (int q = 0; q < 4; ++q) a4[grp & 1][q] = *reinterpret_cast<const f32x4*>(S + ((grp + 2) * 4 + q) * 256);
      }
    }
    // issue pattern: the first two groups' fragment reads, then the matrix pipe starts at once and the next stage's
    // nine loads (8 LDS-DMA pieces + the activation fragment) go out one per MFMA underneath it; every later
    // group's reads are issued a whole group (16 MFMAs = 512 cycles) ahead of their use
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 7, 0);
#pragma unroll
    for (int grp = 1; grp < 7; ++grp) {
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
  // hidden = relu(acc + b1): register r of tile nt is channel 16 nt + 4 g + r
#pragma unroll
  for (int nt = 0; nt < 32; ++nt) {
    const f32x4 bb = *reinterpret_cast<const f32x4*>(bl + 16 * nt + 4 * g);
#pragma unroll
    for (int r = 0; r < 4; ++r) acc1[nt][r] = fmaxf(acc1[nt][r] + bb[r], 0.f);
    if (SAVE_H && live) *reinterpret_cast<f32x4*>(a.h + m * 512 + 16 * nt + 4 * g) = acc1[nt];
  }

  // ---- phase 2: y^T[128-channel chunk][16 positions] = W2^T . hidden^T, the B operand is acc1 itself.  The
  // accumulators start from residual + bias, so the epilogue is the store alone.
  const float* xres = a.x + m * 512 + 4 * g;
  float* yrow = a.y + m * 512 + 4 * g;
  for (int c = 0; c < 4; ++c) {
    f32x4 acc2[8];
#pragma unroll
    for (int t2 = 0; t2 < 8; ++t2) {
      const f32x4 rr = *reinterpret_cast<const f32x4*>(live ? xres + 128 * c + 16 * t2 : a.zeros);
      const f32x4 bb = *reinterpret_cast<const f32x4*>(bl + 512 + 128 * c + 16 * t2 + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) acc2[t2][r] = rr[r] + bb[r];
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int it = 96 + c * 8 + s;
      if (RP_BAR) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
      if (s == 0) {
#pragma unroll
        for (int t2 = 0; t2 < 8; ++t2)
#pragma unroll
          for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(acc2[t2][r]));   // residual landed: wait here, not later
      }
      __builtin_amdgcn_sched_barrier(0);
      const unsigned char* S = lds + (s & 1) * CT_STAGE_BYTES + (g * 128 + ml) * 16;
      // group gi = (kbl, tg): fragment reads at S + (kbl*512 + (tg*4+q)*16)*16
      f32x4 a4[2][4];
#pragma unroll
      for (int gi = 0; gi < 2; ++gi)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          a4[gi][q] = *reinterpret_cast<const f32x4*>(S + ((gi >> 1) * 512 + ((gi & 1) * 4 + q) * 16) * 16);
      if (it + 1 < 128 && RP_DMA)
        ct_issue_stage(a.wpack + (int64_t)(it + 1) * CT_STAGE_FLOATS, lds + ((s + 1) & 1) * CT_STAGE_BYTES, w, lane);
#pragma unroll
      for (int gi = 0; gi < 8; ++gi) {
        const int kbl = gi >> 1, tg = gi & 1;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            acc2[tg * 4 + q] = mfma16(a4[gi & 1][q][j], acc1[4 * s + kbl][j], acc2[tg * 4 + q]);
        if (gi + 2 < 8 && RP_LDS) {
          const int g2 = gi + 2;
#pragma unroll
          for (int q = 0; q < 4; ++q)
            a4[gi & 1][q] = *reinterpret_cast<const f32x4*>(S + ((g2 >> 1) * 512 + ((g2 & 1) * 4 + q) * 16) * 16);
        }
      }
      __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
      if (it + 1 < 128) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
      } else {
        __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
      }
#pragma unroll
      for (int gi = 1; gi < 7; ++gi) {
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (live) {
#pragma unroll
      for (int t2 = 0; t2 < 8; ++t2) *reinterpret_cast<f32x4*>(yrow + 128 * c + 16 * t2) = acc2[t2];
    }
  }
}

extern "C" int qpg_resblock_f32(qpg_ctx* ctx, void* stream, const float* x, int B, int T, int dil, const float* wpack,
                                const float* b1, const float* b2, float* y, float* hidden) {
  QPG_REQUIRE(ctx && x && wpack && b1 && b2 && y, "qpg_resblock_f32: null pointer");
  QPG_REQUIRE(B >= 0 && T > 0 && dil > 0, "qpg_resblock_f32: bad size");
  QPG_REQUIRE(x != y, "qpg_resblock_f32: in-place operation is not supported (rows are re-read as taps)");
  if (B == 0) return QPG_OK;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(resblock_fused_f32_kernel<false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, CT_RES_LDS) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(resblock_fused_f32_kernel<true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, CT_RES_LDS) != hipSuccess) {
      qpg_set_error("qpg_resblock_f32: cannot reserve %d bytes of LDS", CT_RES_LDS);
      return QPG_EHIP;
    }
    attr_set = true;
  }
  ResArgs a;
  a.x = x; a.wpack = wpack; a.b1 = b1; a.b2 = b2; a.y = y; a.h = hidden; a.B = B; a.T = T; a.dil = dil;
  a.zeros = ctx->zeros;
  const int64_t M = (int64_t)B * T;
  const dim3 grid((unsigned)((M + CT_ROWS - 1) / CT_ROWS));
  if (hidden) hipLaunchKernelGGL(resblock_fused_f32_kernel<true>, grid, dim3(256), CT_RES_LDS, qpg_stream(stream), a);
  else hipLaunchKernelGGL(resblock_fused_f32_kernel<false>, grid, dim3(256), CT_RES_LDS, qpg_stream(stream), a);
  QPG_LAUNCH_CHECK("resblock_fused_f32_kernel");
  return QPG_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Generic convolution, transposed formulation: a block = 64 positions x 128 output channels, stage = four 16-k
// blocks x 128 channels (T-pack NB = 128).  Same argument meaning as qpg_conv1d_f32.
// ---------------------------------------------------------------------------------------------------------------
struct ConvTArgs {
  const float* x;      // [B][T_in][Cx] (Cx = row pitch in floats, multiple of 4; channels >= Cin_pad read as zero
                       //                 only through zero WEIGHTS: the rows must hold Cin_pad readable floats)
  const float* wt;     // T-pack NB=128: [Cout_pad/128][K/16][4][128][4]
  const float* bias;   // [Cout_pad] or null
  const float* res;    // indexed like y, or null
  float* y;            // [B][T_y][Cout]
  int B, T_in, Cx, Cin_pad, taps, in_stride, in_offset, dil;
  int T_out, out_stride, out_offset, T_y, Cout, relu_in, relu_out;
  int nstage;          // K / 64
  const float* zeros;
  // convt_small_f32_kernel, gridDim.z == 2 (the two output parities of a ConvTranspose1d in one launch): the second
  // half's weights / bias / offsets
  const float* wt1;
  const float* bias1;
  int in_offset1, out_offset1;
  int xcd_map;         // convt_small_f32_kernel: 1-D grid, block -> (position tile, channel group, parity) with a channel
                       // group's blocks all on ONE XCD (workgroup L runs on XCD L % 8): an XCD's L2 then holds 1/8 of a
                       // layer's weights instead of all of them
  int P, C, nz;        // (xcd_map) position tiles, channel groups, parities
};

template <bool RELU_IN>
__global__ __launch_bounds__(256, 2) void convt_f32_kernel(ConvTArgs a) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, ml = lane & 15, g = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t M = (int64_t)a.B * a.T_out;
  const int64_t m = (int64_t)blockIdx.x * CT_ROWS + w * 16 + ml;
  const int nb = blockIdx.y;
  const bool live = m < M;
  const int b = live ? (int)(m / a.T_out) : 0;
  const int t = live ? (int)(m - (int64_t)b * a.T_out) : 0;
  const float* xb = a.x + (int64_t)b * a.T_in * a.Cx + 4 * g;
  auto rowptr = [&](int tap) -> const float* {
    const int t_in = t * a.in_stride + a.in_offset + tap * a.dil;
    return (live && t_in >= 0 && t_in < a.T_in && tap < a.taps) ? xb + (int64_t)t_in * a.Cx : nullptr;
  };
  const float* wsrc = a.wt + (int64_t)nb * a.nstage * CT_STAGE_FLOATS;

  f32x4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // iterator over 16-k blocks: (tap, ci0), branch-free (the stage loop must stay ONE basic block for the issue
  // pattern below): tap rows are base + tap*step, their validity one bit per tap in a lane mask
  unsigned vmask = 0;
#pragma unroll
  for (int tp = 0; tp < 4; ++tp) vmask |= rowptr(tp) ? (1u << tp) : 0u;
  const float* xrow0 = xb + (int64_t)(t * a.in_stride + a.in_offset) * a.Cx;
  const int64_t tstep = (int64_t)a.dil * a.Cx;
  int tap = 0, ci0 = 0;
  auto next_b = [&](bool on) -> f32x4 {
    const bool ok = ((vmask >> tap) & 1u) != 0 && on;
    int64_t off = tap * tstep + ci0;
    asm volatile("" : "+s"(off));       // keep the address arithmetic unconditional (no exec-masked region)
    const f32x4 v = *reinterpret_cast<const f32x4*>(ok ? xrow0 + off : a.zeros);
    ci0 += 16;
    const int wrap = ci0 == a.Cin_pad ? 1 : 0;
    ci0 = ci0 * (1 - wrap);
    tap += wrap;
    return v;
  };
  ct_issue_stage(wsrc, lds, w, lane);
  f32x4 bnext[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) bnext[i] = next_b(true);

  for (int it = 0; it < a.nstage; ++it) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    f32x4 bcur[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      bcur[i] = RELU_IN ? relu4(bnext[i]) : bnext[i];
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(bcur[i][j]));     // an opaque use: the wait lands here
    }
    __builtin_amdgcn_sched_barrier(0);     // consume last stage's fragments before this stage's loads are issued
    const unsigned char* S = lds + (it & 1) * CT_STAGE_BYTES + (g * 128 + ml) * 16;
    f32x4 a4[2][4];
#pragma unroll
    for (int gi = 0; gi < 2; ++gi)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        a4[gi][q] = *reinterpret_cast<const f32x4*>(S + ((gi >> 1) * 512 + ((gi & 1) * 4 + q) * 16) * 16);
    const bool more = it + 1 < a.nstage;
    // past the last stage the DMA re-reads the last stage (never consumed): the issue pattern stays uniform
    ct_issue_stage(wsrc + (int64_t)(more ? it + 1 : it) * CT_STAGE_FLOATS, lds + ((it + 1) & 1) * CT_STAGE_BYTES, w, lane);
#pragma unroll
    for (int i = 0; i < 4; ++i) bnext[i] = next_b(more);
#pragma unroll
    for (int gi = 0; gi < 8; ++gi) {
      const int kbl = gi >> 1, tg = gi & 1;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[tg * 4 + q] = mfma16(a4[gi & 1][q][j], bcur[kbl][j], acc[tg * 4 + q]);
      if (gi + 2 < 8) {
        const int g2 = gi + 2;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          a4[gi & 1][q] = *reinterpret_cast<const f32x4*>(S + ((g2 >> 1) * 512 + ((g2 & 1) * 4 + q) * 16) * 16);
      }
    }
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
#pragma unroll
    for (int gi = 1; gi < 7; ++gi) {
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the surplus DMA of the last stage must not outlive the block

  if (!live) return;
  const int64_t orow = ((int64_t)b * a.T_y + (int64_t)t * a.out_stride + a.out_offset) * a.Cout;
#pragma unroll
  for (int t2 = 0; t2 < 8; ++t2) {
    const int n = nb * 128 + 16 * t2 + 4 * g;
    if (n >= a.Cout) continue;
    f32x4 o = acc[t2];
    if (a.bias) {
      const f32x4 bb = *reinterpret_cast<const f32x4*>(a.bias + n);
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] += bb[r];
    }
    if (a.relu_out) o = relu4(o);
    if (n + 4 <= a.Cout && (a.Cout & 3) == 0) {
      if (a.res) {
        const f32x4 rr = *reinterpret_cast<const f32x4*>(a.res + orow + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = rr[r] + o[r];
      }
      *reinterpret_cast<f32x4*>(a.y + orow + n) = o;
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (n + r < a.Cout) a.y[orow + n + r] = a.res ? a.res[orow + n + r] + o[r] : o[r];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Short sequences (one clip's decode: 180..1440 rows; a single window's encode): a launch of 64-row x 128-channel
// blocks would leave most of the chip idle and each block would walk the whole contraction alone (latency-bound).
// Here a block = 16 positions x 64 output channels and its four waves SPLIT THE CONTRACTION (each a quarter of the
// 16-k blocks); nothing is shared between the waves, so the weights go straight from the T-pack (L2-resident, 1 KB
// coalesced per tile and 16-k block) into registers, three 16-k blocks ahead, with no LDS staging and no barrier in
// the loop; the eight partial tiles meet in LDS once, are added in wave order (deterministic) and each wave finishes
// one 16-channel tile (bias, ReLU, residual, 16-byte stores).
// ---------------------------------------------------------------------------------------------------------------
#define CTS_NW 8     // waves per block = ways the contraction is split
// NQ = 16-channel tiles per block (4: 64 channels; 2: 32 channels — twice the blocks, each pulling half the weights,
// for launches that would otherwise leave most CUs idle)
template <bool RELU_IN, int PD, int NQ>
__global__ __launch_bounds__(64 * CTS_NW) void convt_small_f32_kernel(ConvTArgs a) {
  // (uniform per block) which of the two parity halves this block computes
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (a.xcd_map) {
    const int L = (int)blockIdx.x, xcd = L & 7, i = L >> 3, pz = a.P * a.nz;
    by = (i / pz) * 8 + xcd;
    const int rem = i % pz;
    bz = rem / a.P;
    bx = rem % a.P;
  }
  const float* const wt_sel = bz ? a.wt1 : a.wt;
  const float* const bias_sel = bz ? a.bias1 : a.bias;
  const int in_offset = bz ? a.in_offset1 : a.in_offset;
  const int out_offset = bz ? a.out_offset1 : a.out_offset;
  __shared__ __attribute__((aligned(16))) float part[CTS_NW][NQ][64][4];    // [wave][tile][lane][r]
  const int tid = threadIdx.x, lane = tid & 63, ml = lane & 15, g = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t M = (int64_t)a.B * a.T_out;
  const int64_t m = (int64_t)bx * 16 + ml;
  constexpr int NGRP = 8 / NQ;                                         // channel groups per 128-channel T-pack block
  const int nb = by / NGRP, cg0 = (by % NGRP) * (16 * NQ);
  if (nb * 128 + cg0 >= a.Cout) return;                                // padding-only chunk (135-channel output layer)
  const bool live = m < M;
  const int b = live ? (int)(m / a.T_out) : 0;
  const int t = live ? (int)(m - (int64_t)b * a.T_out) : 0;
  const float* xb = a.x + (int64_t)b * a.T_in * a.Cx + 4 * g;
  const int nkb = a.nstage * 4;                                        // 16-k blocks in the contraction
  const int per = (nkb + CTS_NW - 1) / CTS_NW;
  const int kb0 = w * per, kb1 = kb0 + per < nkb ? kb0 + per : nkb;
  const int kpt = a.Cin_pad / 16;                                      // 16-k blocks per tap
  const float* wbase = wt_sel + ((int64_t)nb * nkb * 4 + g) * 512 + (cg0 + ml) * 4;         // + kb*2048 + tile*64
  auto bfrag = [&](int kb) -> f32x4 {
    const int tap = kb / kpt, ci0 = (kb - tap * kpt) * 16;
    const int t_in = t * a.in_stride + in_offset + tap * a.dil;
    const bool ok = live && kb < kb1 && t_in >= 0 && t_in < a.T_in;
    f32x4 v = *reinterpret_cast<const f32x4*>(ok ? xb + (int64_t)t_in * a.Cx + ci0 : a.zeros);
    return RELU_IN ? relu4(v) : v;
  };
  struct Frag {
    f32x4 a[NQ], b;
  };
  auto load = [&](int kb) -> Frag {
    Frag f;
    const float* wp = wbase + (int64_t)(kb < kb1 ? kb : 0) * 2048;       // past the range: a valid address, zero B
#pragma unroll
    for (int q = 0; q < NQ; ++q) f.a[q] = *reinterpret_cast<const f32x4*>(wp + q * 64);
    f.b = bfrag(kb);
    return f;
  };
  f32x4 acc[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // A block is bound by how fast ONE CU pulls its 64-channel weight slice (393 KB for a k3 layer) out of L2, not by its
  // 2.6 us of MFMAs.  PD = 16-k blocks (5 KB per wave each) in flight; when the wave's share is a whole number of
  // PD-chunks the loop is peeled so that no chunk prefetches past the end (branch-free inside: a conditional load
  // makes hipcc drain vmcnt); PD = 0 is the general 3-deep rotation.
  if (PD > 0 && (kb1 - kb0) == per && per % (PD > 0 ? PD : 1) == 0) {
    constexpr int P = PD > 0 ? PD : 1;
    Frag f[P];
#pragma unroll
    for (int d = 0; d < P; ++d) f[d] = load(kb0 + d);
    for (int kb = kb0; kb + P < kb1; kb += P) {
#pragma unroll
      for (int d = 0; d < P; ++d) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int q = 0; q < NQ; ++q) acc[q] = mfma16(f[d].a[q][j], f[d].b[j], acc[q]);
        f[d] = load(kb + d + P);
      }
    }
#pragma unroll
    for (int d = 0; d < P; ++d)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[q] = mfma16(f[d].a[q][j], f[d].b[j], acc[q]);
  } else {
    Frag f0 = load(kb0), f1 = load(kb0 + 1), f2 = load(kb0 + 2);
    for (int kb = kb0; kb < kb1; ++kb) {
      const Frag f3 = load(kb + 3);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[q] = mfma16(f0.a[q][j], f0.b[j], acc[q]);
      f0 = f1;
      f1 = f2;
      f2 = f3;
    }
  }
#pragma unroll
  for (int q = 0; q < NQ; ++q) *reinterpret_cast<f32x4*>(&part[w][q][lane][0]) = acc[q];
  __syncthreads();
  // waves 0..NQ-1 finish tile w: channel n = nb*128 + cg0 + 16 w + 4 g + r, position m (partials added in wave order)
  if (w >= NQ) return;
  f32x4 o = *reinterpret_cast<const f32x4*>(&part[0][w][lane][0]);
#pragma unroll
  for (int s2 = 1; s2 < CTS_NW; ++s2) {
    const f32x4 p2 = *reinterpret_cast<const f32x4*>(&part[s2][w][lane][0]);
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] += p2[r];
  }
  if (!live) return;
  const int n = nb * 128 + cg0 + 16 * w + 4 * g;
  if (n >= a.Cout) return;
  const int64_t orow = ((int64_t)b * a.T_y + (int64_t)t * a.out_stride + out_offset) * a.Cout;
  if (bias_sel) {
    const f32x4 bb = *reinterpret_cast<const f32x4*>(bias_sel + n);
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] += bb[r];
  }
  if (a.relu_out) o = relu4(o);
  if (n + 4 <= a.Cout && (a.Cout & 3) == 0) {
    if (a.res) {
      const f32x4 rr = *reinterpret_cast<const f32x4*>(a.res + orow + n);
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = rr[r] + o[r];
    }
    *reinterpret_cast<f32x4*>(a.y + orow + n) = o;
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (n + r < a.Cout) a.y[orow + n + r] = a.res ? a.res[orow + n + r] + o[r] : o[r];
  }
}

// Measurement hook (tools/bench_decode.py): deep = every k-block of a wave's share requested up front when the share is 8 or 12
// k-blocks (a k3 / k2-pair layer: one round trip to the weights instead of three / two), xcd = the XCD-aware block mapping.
// Round 5, a 24 s clip's decode on one box, alternating: neither 0.281-0.289 ms; deep ring 0.325-0.332 (SLOWER: these
// layers are bound by the rate at which ~190-720 blocks pull their tiles out of L2, not by round trips - more requests
// in flight only lengthen every queue, and 176 VGPRs halve the waves per SIMD); XCD map 0.285-0.290 (no difference: a
// layer's 1-4 MB of weights fit every XCD's L2 anyway).  Both stay OFF.
// (-DQPG_DEBUG_HOOKS builds only; constants in the product)
QPG_HOOK_VAR(int, g_deep_ring, 0);
QPG_HOOK_VAR(int, g_xcd_map, 0);
#ifdef QPG_DEBUG_HOOKS
extern "C" int qpg_debug_convt_opts(int deep_ring, int xcd_map) {
  g_deep_ring = deep_ring != 0;
  g_xcd_map = xcd_map != 0;
  return QPG_OK;
}
#endif

// Measurement hook: force the short-sequence kernel's block shape (nq in {1, 2, 4} channel tiles, pd in {0, 4} ring
// depth); nq = 0 restores the launcher's own choice.  -DQPG_DEBUG_HOOKS builds only; constants in the product.
QPG_HOOK_VAR(int, g_force_nq, 0);
QPG_HOOK_VAR(int, g_force_pd, 0);
#ifdef QPG_DEBUG_HOOKS
extern "C" int qpg_debug_convt_shape(int nq, int pd) {
  QPG_REQUIRE((nq == 0 || nq == 1 || nq == 2 || nq == 4 || nq == 8) && (pd == 0 || pd == 4),
              "qpg_debug_convt_shape: nq in {0,1,2,4} (8: the 64 x 128 kernel whatever the size), pd in {0,4}");
  g_force_nq = nq;
  g_force_pd = pd;
  return QPG_OK;
}
#endif

// nz = 2: the pair form (a.wt1 / bias1 / in_offset1 / out_offset1 set); the generic kernel then takes two launches
static int convt_launch(qpg_ctx* ctx, void* stream, ConvTArgs a, int Cout_pad, int nz) {
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(convt_f32_kernel<false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 2 * CT_STAGE_BYTES) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(convt_f32_kernel<true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 2 * CT_STAGE_BYTES) != hipSuccess) {
      qpg_set_error("qpg_convt_f32: cannot reserve %d bytes of LDS", 2 * CT_STAGE_BYTES);
      return QPG_EHIP;
    }
    attr_set = true;
  }
  const bool relu_in = a.relu_in != 0;
  const int64_t M = (int64_t)a.B * a.T_out;
  // short sequences: 16-position x 64-channel blocks whose waves split the contraction (see convt_small_f32_kernel)
  if (g_force_nq != 8 && ((M + CT_ROWS - 1) / CT_ROWS) * (Cout_pad / 128) * 2 * nz < 3 * (int64_t)ctx->n_cu) {
    // Channels per block: 64, 32 or 16 (NQ = 4, 2, 1 tiles).  A block's time is the time ONE CU needs to pull its
    // operands (NQ weight tiles + 1 activation tile of 16 x K floats; two blocks on one CU take twice as long, so a
    // launch takes ceil(blocks / CUs) block times): pick the NQ with the smallest product.  tools/bench_convt_small.py
    // measures every shape on the decoder's layers (experiments/convt_small/README.md).
    int nq = 4;
    int64_t best_cost = 0;
    for (int cand = 4; cand >= 1; cand >>= 1) {
      const int64_t blocks = ((M + 15) / 16) * (Cout_pad / (16 * cand)) * nz;
      const int64_t cost = ((blocks + ctx->n_cu - 1) / ctx->n_cu) * (cand + 1);
      if (cand == 4 || cost < best_cost) {
        best_cost = cost;
        nq = cand;
      }
    }
    const int per = (a.nstage * 4 + CTS_NW - 1) / CTS_NW;          // 16-k blocks per wave
    int pd = (a.nstage * 4) % CTS_NW == 0 && per % 4 == 0 ? 4 : 0;   // ring of 4 (a ring of 6 measured ~6% slower)
    if (g_deep_ring && pd == 4 && ((per == 12 && nq <= 2) || per == 8)) pd = per;      // the whole share in flight
    if (g_force_nq && g_force_nq != 8) {                            // tools/bench_convt_small.py (qpg_debug_convt_shape)
      if (g_force_pd == 0 || (g_force_pd == 4 && (a.nstage * 4) % CTS_NW == 0 && per % 4 == 0)) {
        nq = g_force_nq;
        pd = g_force_pd;
      }
    }
    dim3 sgrid((unsigned)((M + 15) / 16), (unsigned)(Cout_pad / (16 * nq)), (unsigned)nz);
    a.P = (int)sgrid.x; a.C = (int)sgrid.y; a.nz = nz;
    a.xcd_map = (g_xcd_map && a.C % 8 == 0 && (int64_t)a.P * a.C * nz < 0x7fffffffll) ? 1 : 0;
    if (a.xcd_map) sgrid = dim3((unsigned)(a.P * a.C * nz));
#define CTS_LAUNCH(R, P, Q_) hipLaunchKernelGGL((convt_small_f32_kernel<R, P, Q_>), sgrid, dim3(64 * CTS_NW), 0, qpg_stream(stream), a)
#define CTS_LAUNCH_P(R, Q_) do { if (pd == 12 && Q_ <= 2) CTS_LAUNCH(R, (Q_ <= 2 ? 12 : 4), Q_); else if (pd == 8) CTS_LAUNCH(R, 8, Q_); \
                                 else if (pd == 4) CTS_LAUNCH(R, 4, Q_); else CTS_LAUNCH(R, 0, Q_); } while (0)
    if (relu_in) {
      if (nq == 1) CTS_LAUNCH_P(true, 1); else if (nq == 2) CTS_LAUNCH_P(true, 2); else CTS_LAUNCH_P(true, 4);
    } else {
      if (nq == 1) CTS_LAUNCH_P(false, 1); else if (nq == 2) CTS_LAUNCH_P(false, 2); else CTS_LAUNCH_P(false, 4);
    }
#undef CTS_LAUNCH_P
#undef CTS_LAUNCH
    QPG_LAUNCH_CHECK("convt_small_f32_kernel");
    return QPG_OK;
  }
  const dim3 grid((unsigned)((M + CT_ROWS - 1) / CT_ROWS), (unsigned)(Cout_pad / 128));
  for (int z = 0; z < nz; ++z) {
    if (z == 1) {
      a.wt = a.wt1; a.bias = a.bias1; a.in_offset = a.in_offset1; a.out_offset = a.out_offset1;
    }
    if (relu_in) hipLaunchKernelGGL(convt_f32_kernel<true>, grid, dim3(256), 2 * CT_STAGE_BYTES, qpg_stream(stream), a);
    else hipLaunchKernelGGL(convt_f32_kernel<false>, grid, dim3(256), 2 * CT_STAGE_BYTES, qpg_stream(stream), a);
    QPG_LAUNCH_CHECK("convt_f32_kernel");
  }
  return QPG_OK;
}

static int convt_check(qpg_ctx* ctx, const float* x, const float* wt, const float* y, int B, int T_in, int Cx, int taps,
                       int Cin_pad, int Cout, int Cout_pad, int in_stride, int dil, int T_out, int out_stride, int T_y) {
  QPG_REQUIRE(ctx && x && wt && y, "qpg_convt_f32: null pointer");
  QPG_REQUIRE(B >= 0 && T_in > 0 && Cx > 0 && taps > 0 && taps <= 4 && Cout > 0 && T_out >= 0 && T_y > 0 &&
                  out_stride > 0 && in_stride > 0 && dil > 0,
              "qpg_convt_f32: bad size (1..4 taps)");
  QPG_REQUIRE((Cx % 4) == 0 && Cx >= Cin_pad && (reinterpret_cast<uintptr_t>(x) % 16) == 0,
              "qpg_convt_f32: input rows must be 16-byte aligned and hold Cin_pad floats (pad the channels first)");
  QPG_REQUIRE(Cin_pad % 16 == 0 && (taps * Cin_pad) % 64 == 0 && Cout_pad % 128 == 0 && Cout_pad >= Cout,
              "qpg_convt_f32: T-pack needs Cin_pad %% 16 == 0, taps*Cin_pad %% 64 == 0, Cout_pad %% 128 == 0");
  return QPG_OK;
}

extern "C" int qpg_convt_f32(qpg_ctx* ctx, void* stream, const float* x, int B, int T_in, int Cx, const float* wt,
                             const float* bias, int taps, int Cin_pad, int Cout, int Cout_pad, int in_stride,
                             int in_offset, int dil, int T_out, int out_stride, int out_offset, int T_y,
                             const float* residual, int relu_in, int relu_out, float* y) {
  const int rc = convt_check(ctx, x, wt, y, B, T_in, Cx, taps, Cin_pad, Cout, Cout_pad, in_stride, dil, T_out, out_stride, T_y);
  if (rc) return rc;
  if (B == 0 || T_out == 0) return QPG_OK;
  ConvTArgs a;
  a.x = x; a.wt = wt; a.bias = bias; a.res = residual; a.y = y;
  a.B = B; a.T_in = T_in; a.Cx = Cx; a.Cin_pad = Cin_pad; a.taps = taps; a.in_stride = in_stride;
  a.in_offset = in_offset; a.dil = dil; a.T_out = T_out; a.out_stride = out_stride; a.out_offset = out_offset;
  a.T_y = T_y; a.Cout = Cout; a.relu_in = relu_in; a.relu_out = relu_out; a.nstage = taps * Cin_pad / 64;
  a.zeros = ctx->zeros;
  a.wt1 = wt; a.bias1 = bias; a.in_offset1 = in_offset; a.out_offset1 = out_offset;
  a.xcd_map = 0; a.P = a.C = 0; a.nz = 1;
  return convt_launch(ctx, stream, a, Cout_pad, 1);
}

// Two convolutions of the SAME input and shape that interleave in the output - the even / odd output frames of
// ConvTranspose1d(k4, s2, p1) (encdec.py:112-115): y[2m] = x[m-1].W3 + x[m].W1, y[2m+1] = x[m].W2 + x[m+1].W0 -
// as ONE launch of the short-sequence kernel (gridDim.z = 2), two launches of the generic one.
extern "C" int qpg_convt_pair_f32(qpg_ctx* ctx, void* stream, const float* x, int B, int T_in, int Cx, const float* wt0,
                                  const float* bias0, int in_offset0, int out_offset0, const float* wt1,
                                  const float* bias1, int in_offset1, int out_offset1, int taps, int Cin_pad, int Cout,
                                  int Cout_pad, int in_stride, int dil, int T_out, int out_stride, int T_y, float* y) {
  const int rc = convt_check(ctx, x, wt0, y, B, T_in, Cx, taps, Cin_pad, Cout, Cout_pad, in_stride, dil, T_out, out_stride, T_y);
  if (rc) return rc;
  QPG_REQUIRE(wt1, "qpg_convt_pair_f32: null pointer");
  if (B == 0 || T_out == 0) return QPG_OK;
  ConvTArgs a;
  a.x = x; a.wt = wt0; a.bias = bias0; a.res = nullptr; a.y = y;
  a.B = B; a.T_in = T_in; a.Cx = Cx; a.Cin_pad = Cin_pad; a.taps = taps; a.in_stride = in_stride;
  a.in_offset = in_offset0; a.dil = dil; a.T_out = T_out; a.out_stride = out_stride; a.out_offset = out_offset0;
  a.T_y = T_y; a.Cout = Cout; a.relu_in = 0; a.relu_out = 0; a.nstage = taps * Cin_pad / 64;
  a.zeros = ctx->zeros;
  a.wt1 = wt1; a.bias1 = bias1; a.in_offset1 = in_offset1; a.out_offset1 = out_offset1;
  a.xcd_map = 0; a.P = a.C = 0; a.nz = 2;
  return convt_launch(ctx, stream, a, Cout_pad, 2);
}

// [R][C] -> [R][Cp] with zero fill (the 135-channel pose rows are 540 B: not 16-byte aligned per row)
__global__ __launch_bounds__(256) void pad_channels_kernel(const float* __restrict__ x, int64_t R, int C, int Cp,
                                                           float* __restrict__ y) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= R * Cp) return;
  const int64_t r = i / Cp;
  const int c = (int)(i - r * Cp);
  y[i] = c < C ? x[r * C + c] : 0.f;
}

extern "C" int qpg_pad_channels_f32(qpg_ctx* ctx, void* stream, const float* x, int64_t R, int C, int Cp, float* y) {
  QPG_REQUIRE(ctx && x && y && R >= 0 && C > 0 && Cp >= C, "qpg_pad_channels_f32: bad argument");
  if (R == 0) return QPG_OK;
  const int64_t n = R * Cp;
  hipLaunchKernelGGL(pad_channels_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, qpg_stream(stream), x, R, C,
                     Cp, y);
  QPG_LAUNCH_CHECK("pad_channels_kernel");
  return QPG_OK;
}

// T-pack of one convolution's weights ON THE DEVICE (what qpgesture_amd/vqvae.py::tpack builds with torch ops):
//   out[n / nb][kb][g][n % nb][j] = w[k = 16 kb + 4 g + j][n],   w: [taps * Cin_pad][Cout_pad] (the training layout),
// n < Cout_n = Cout_pad rounded up to nb (zeros beyond Cout_pad).  A training step changes every weight, and the
// transposed-formulation kernels read T-packs: one launch per convolution and step instead of ~200 torch operations.
__global__ __launch_bounds__(256) void tpack_kernel(const float* __restrict__ w, int Kd, int cout_pad, int nb, int cout_n,
                                                    float* __restrict__ out) {
  const int64_t n_out = (int64_t)Kd * cout_n;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_out; i += (int64_t)gridDim.x * blockDim.x) {
    const int j = (int)(i & 3);
    const int nl = (int)((i >> 2) % nb);
    const int64_t r = (i >> 2) / nb;
    const int g = (int)(r & 3);
    const int64_t r2 = r >> 2;
    const int kb = (int)(r2 % (Kd / 16));
    const int nbi = (int)(r2 / (Kd / 16));
    const int k = 16 * kb + 4 * g + j, n = nbi * nb + nl;
    out[i] = n < cout_pad ? w[(int64_t)k * cout_pad + n] : 0.f;
  }
}

extern "C" int qpg_tpack_f32(qpg_ctx* ctx, void* stream, const float* w, int taps, int Cin_pad, int Cout_pad, int nb,
                             float* out) {
  QPG_REQUIRE(ctx && w && out && taps > 0 && Cin_pad > 0 && (Cin_pad % 16) == 0 && Cout_pad > 0 && nb > 0 && (nb % 4) == 0,
              "qpg_tpack_f32: bad argument (Cin_pad %% 16 == 0)");
  const int Kd = taps * Cin_pad, cout_n = (Cout_pad + nb - 1) / nb * nb;
  const int64_t n = (int64_t)Kd * cout_n;
  hipLaunchKernelGGL(tpack_kernel, dim3((unsigned)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192)), dim3(256), 0,
                     qpg_stream(stream), w, Kd, Cout_pad, nb, cout_n, out);
  QPG_LAUNCH_CHECK("tpack_kernel");
  return QPG_OK;
}
