// Round-4 experiments: prefilter GEMM organisations measured against audio_cosine_hl_kernel<1> and NOT kept.
// These kernels were compiled inside qpgesture_amd/csrc/qpg_audio_hl.hip (they use its HlArgs, h8, mfma_h, HL_* definitions);
// see README.md for the numbers.

// ---- the prefilter GEMM on 32-ROW wave tiles (round 4) ----------------------------------------------------------------------
// audio_cosine_hl_kernel<1> inherits the audio sweep's organisation: a wave owns 16 rows x 96 columns, every wave reads
// the whole 12 KB of a k-block's query fragments from LDS for its 18 MFMAs, and per CU that is 768 LDS cycles against
// 576 matrix cycles per k-block step - the LDS pipe, not the matrix pipe, sets the pace (profiles/r04_*: 39 % matrix
// utilisation in both kernels).  The audio sweep cannot give a wave more rows: its f64 running sums (which keep the f32
// accumulation out of the error bound) already fill the register file.  The prefilter of the exact-f32 cosine family can:
// its band is dominated by sklearn's own rounding (8.6e-5 at D = 512), so the h h' products may simply stay in the
// MFMA's f32 accumulator for the whole (short) K - the error of a KB-instruction chain is bounded by
// (kappa_1 + KB) 2^-24 sum|products| (every instruction: its own block error + one rounding of the running sum),
// i.e. QPG_HL_GEMM32_ERR(D) = (12 + D/32) 2^-24 + 5.2e-7 (cross terms, representation, f32 store: as §4.1) = 2.2e-6 at
// D = 512 - and without the f64 sums a wave holds TWO row tiles: 32 rows x 96 columns, 36 MFMAs per 12 KB of fragment
// reads.  Block = 8 waves = 256 rows x one chunk of 96 queries; query stages of 4 k-blocks (48 KB), double-buffered; the
// rows' fragments go HBM / L2 -> VGPR one k-block ahead.
#define G32_KS 4
#ifndef G32_SGB
#define G32_SGB 1          // 1: MFMA / fragment-read issue order pinned with sched_group_barrier (0: the compiler's order)
#endif
// PERSISTENT over work items (row block of 256 rows, query chunk): with K = D / 32 of only 12-16 k-blocks a block's
// prologue (first query stage through LDS, first row fragments) and epilogue were a third of its life, and with one
// 512-thread block per CU nothing overlapped them.  gridDim.x blocks take contiguous ranges of the items (chunk-minor,
// so a range stays on one row block as long as possible and its fragments come out of L2 after the first chunk); the
// next item's first stage and row fragments are requested underneath the current item's last stage, the epilogue's
// stores are fire-and-forget.
template <int CT>
__global__ __launch_bounds__(512, 2) void hl_gemm32_kernel(HlArgs a, int n_items) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // 2 x G32_KS x 6 x 2 x 1 KB
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int per = (n_items + (int)gridDim.x - 1) / (int)gridDim.x;
  const int it0 = (int)blockIdx.x * per, it1 = min(it0 + per, n_items);
  if (it0 >= it1) return;
  const int KB = a.KB, n_stage = KB / G32_KS;
  const int cg = lane & 15, rg = lane >> 4;
  const int e_c1 = a.meta[0];
  constexpr int stage_units = G32_KS * HL_CT * 2 * 64;                 // h8 units per stage (48 KB)
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
  h8 A[2][2][2];                             // [k-block parity][row tile][plane]
  h8 Bq[2][2];                               // [step parity][plane]: a column tile's fragments, read one step ahead
  auto ld_b = [&](int buf, int k4, int c, h8 (&d)[2]) {
    const h8* qb = reinterpret_cast<const h8*>(lds) + buf * stage_units + lane;
    d[0] = qb[((k4 * HL_CT + c) * 2 + 0) * 64];
    d[1] = qb[((k4 * HL_CT + c) * 2 + 1) * 64];
  };
  set_rows(it0);
  load_q(it0, 0);
  load_a(0, A[0]);
  store_q(0);
  __syncthreads();
  ld_b(0, 0, 0, Bq[0]);
  constexpr int NS = G32_KS * CT;            // steps of a stage: (k-block, column tile); 6 MFMAs each
  int buf = 0;
  for (int item = it0; item < it1; ++item) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        hh[t][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        xx[t][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    const int j_cur = j;
    const bool ok_cur = ok;
    for (int s = 0; s < n_stage; ++s) {
      const bool last_s = s + 1 == n_stage;
      const int nitem = last_s ? (item + 1 < it1 ? item + 1 : item) : item;   // (the very last stage: a harmless re-read)
      load_q(nitem, last_s ? 0 : s + 1);     // in flight underneath this stage's MFMAs
#pragma unroll
      for (int st = 0; st < NS; ++st) {
        const int k4 = st / CT, c = st % CT;
        h8 (&Ac)[2][2] = A[k4 & 1];
        h8 (&Bc)[2] = Bq[st & 1];
        h8 (&Bn)[2] = Bq[(st + 1) & 1];
        if (st == NS - 1) {
          // the next stage's fragments go to the other buffer (last read one stage ago, whose barrier everybody passed),
          // one LDS-only barrier, and the next stage's first fragments are read underneath this stage's last MFMAs
          store_q(buf ^ 1);
          lds_barrier();
          ld_b(buf ^ 1, 0, 0, Bn);
        } else {
          ld_b(buf, (st + 1) / CT, (st + 1) % CT, Bn);
        }
        if (c == 0) {                        // the rows' fragments one k-block ahead (the item's last: the next item's first)
          if (last_s && k4 == G32_KS - 1) {
            set_rows(nitem);
            load_a(0, A[(k4 + 1) & 1]);
          } else {
            load_a(s * G32_KS + k4 + 1, A[(k4 + 1) & 1]);
          }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          hh[t][c] = mfma_h(Ac[t][0], Bc[0], hh[t][c]);
          xx[t][c] = mfma_h(Ac[t][0], Bc[1], xx[t][c]);
          xx[t][c] = mfma_h(Ac[t][1], Bc[0], xx[t][c]);
        }
#if G32_SGB
#pragma unroll
        for (int i = 0; i < 6; ++i) {         // issue order: an MFMA, a fragment read / a row-fragment load underneath it
          HL_SGB(0x008, 1);
          if (i < 2) HL_SGB(0x100, 1);
          else if (c == 0) HL_SGB(0x020, 1);
        }
#endif
      }
      buf ^= 1;
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
          m = fminf(m, __shfl_xor(m, 16, 64));
          m = fminf(m, __shfl_xor(m, 32, 64));
          if (rg == 0) a.tmin[(int64_t)q * a.ldT + (int64_t)j_cur * 2 + t] = m;
          if (a.tmask) {
            const float lim = m + a.band;
            unsigned int bits = ((o[0] <= lim) ? 1u : 0u) | ((o[1] <= lim) ? 2u : 0u) | ((o[2] <= lim) ? 4u : 0u) |
                                ((o[3] <= lim) ? 8u : 0u);
            bits <<= 4 * rg;
            bits |= (unsigned int)__shfl_xor((int)bits, 16, 64);
            bits |= (unsigned int)__shfl_xor((int)bits, 32, 64);
            if (rg == 0) a.tmask[(int64_t)q * a.ldT + (int64_t)j_cur * 2 + t] = (uint16_t)bits;
          }
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");       // surplus prefetches must not outlive their registers
}

// ---- the prefilter GEMM with the ROW PANEL RESIDENT IN REGISTERS (round 4) ------------------------------------------------
// hl_gemm32_kernel re-reads a block's 256 KB of row fragments for every query chunk; 32 such panels per XCD are twice its
// L2, so the re-reads come out of the Infinity Cache at ~2.7 TB/s and that, not the matrix or LDS pipes, is what cfg-3's
// GEMM waits for (0.43 ms for 2.1 GB).  Here a wave keeps its 32 rows x K x (h, l) in REGISTERS - 64 KB = 256 VGPRs at
// K = 512, so ONE wave per SIMD with the 512-register budget (panel and accumulators spread over the unified file) - and
// walks through a range of query chunks: the 192 MB row image is read once (twice with the chunk range split for load
// balance), the query stages go L2 -> LDS -> fragments as before, and with 32 rows per wave the LDS pipe (4 waves x 12
// reads per k-block) stays under the matrix pipe (36 MFMAs per SIMD).
// Block = 4 waves = 128 rows; item = (row block, chunk range c0..c1); stages of 2 k-blocks (24 KB), double-buffered.
template <int KB, int CT>
__global__ __launch_bounds__(256, 1) void hl_gemmp_kernel(HlArgs a, int n_items, int csplit) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // 2 x 2 x 6 x 2 x 1 KB = 48 KB
  constexpr int KS = 2, n_stage = KB / KS;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int per = (n_items + (int)gridDim.x - 1) / (int)gridDim.x;
  const int it0 = (int)blockIdx.x * per, it1 = min(it0 + per, n_items);
  if (it0 >= it1) return;
  const int cg = lane & 15, rg = lane >> 4;
  const int e_c1 = a.meta[0];
  constexpr int stage_units = KS * HL_CT * 2 * 64;                     // h8 units per stage (24 KB)
  constexpr int QLD = stage_units / 256;
  h8 qreg[QLD];
  auto load_q = [&](int chunk, int s) {
    const h8* qsrc = reinterpret_cast<const h8*>(a.qi) + ((int64_t)chunk * KB + (int64_t)s * KS) * HL_CT * 2 * 64;
#pragma unroll
    for (int u = 0; u < QLD; ++u) qreg[u] = qsrc[u * 256 + tid];
  };
  auto store_q = [&](int buf) {
    h8* dst = reinterpret_cast<h8*>(lds) + buf * stage_units;
#pragma unroll
    for (int u = 0; u < QLD; ++u) dst[u * 256 + tid] = qreg[u];
  };
  auto lds_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
  auto ld_b = [&](int buf, int k2, int c, h8 (&d)[2]) {
    const h8* qb = reinterpret_cast<const h8*>(lds) + buf * stage_units + lane;
    d[0] = qb[((k2 * HL_CT + c) * 2 + 0) * 64];
    d[1] = qb[((k2 * HL_CT + c) * 2 + 1) * 64];
  };
  h8 P[KB][2][2];                            // the panel: [k-block][row tile][plane]
  f32x4 hh[2][CT], xx[2][CT];
  h8 Bq[2][2];
  constexpr int NS = KS * CT;
  int buf = 0;
  for (int item = it0; item < it1; ++item) {
    const int rb = item / csplit, part = item % csplit;
    const int c0 = (int)((int64_t)part * a.chunks / csplit), c1 = (int)((int64_t)(part + 1) * a.chunks / csplit);
    if (c0 >= c1) continue;
    const int j = rb * 4 + w;                                          // this wave's 32-row group
    const bool ok = j < a.N;
    __syncthreads();                                                   // (the previous item's last stage has been read)
    load_q(c0, 0);
    {
      const h8* dbp = ok ? reinterpret_cast<const h8*>(a.db) + (int64_t)j * 2 * KB * 2 * 64 + lane
                         : reinterpret_cast<const h8*>(a.zeros);
      const int64_t t_step = ok ? (int64_t)KB * 2 * 64 : 0;
      const int kb_step = ok ? 128 : 0, pl_step = ok ? 64 : 0;
#pragma unroll
      for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) P[kb][t][pl] = dbp[t * t_step + (int64_t)kb * kb_step + pl * pl_step];
    }
    buf = 0;
    store_q(0);
    __syncthreads();
    ld_b(0, 0, 0, Bq[0]);
    for (int chunk = c0; chunk < c1; ++chunk) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int c = 0; c < CT; ++c) {
          hh[t][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
          xx[t][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
      for (int s = 0; s < n_stage; ++s) {
        const bool last_s = s + 1 == n_stage;
        const int nchunk = last_s ? (chunk + 1 < c1 ? chunk + 1 : chunk) : chunk;
        load_q(nchunk, last_s ? 0 : s + 1);
        const int b0 = (buf + s) & 1;
#pragma unroll
        for (int st = 0; st < NS; ++st) {
          const int k2 = st / CT, c = st % CT;
          h8 (&Bc)[2] = Bq[(s * NS + st) & 1];
          h8 (&Bn)[2] = Bq[(s * NS + st + 1) & 1];
          if (st == NS - 1) {
            store_q(b0 ^ 1);
            lds_barrier();
            ld_b(b0 ^ 1, 0, 0, Bn);
          } else {
            ld_b(b0, (st + 1) / CT, (st + 1) % CT, Bn);
          }
          h8 (&Ac)[2][2] = P[s * KS + k2];
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            hh[t][c] = mfma_h(Ac[t][0], Bc[0], hh[t][c]);
            xx[t][c] = mfma_h(Ac[t][0], Bc[1], xx[t][c]);
            xx[t][c] = mfma_h(Ac[t][1], Bc[0], xx[t][c]);
          }
#pragma unroll
          for (int i = 0; i < 6; ++i) {
            HL_SGB(0x008, 1);
            if (i < 2) HL_SGB(0x100, 1);
          }
        }
      }
      buf = (buf + n_stage) & 1;
      if (!ok) continue;
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
            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.D) + (int64_t)q * a.ldD + (int64_t)j * 32 + 16 * t + 4 * rg) = o;
          if (a.tmin) {
            float m = fminf(fminf(o[0], o[1]), fminf(o[2], o[3]));
            m = fminf(m, __shfl_xor(m, 16, 64));
            m = fminf(m, __shfl_xor(m, 32, 64));
            if (rg == 0) a.tmin[(int64_t)q * a.ldT + (int64_t)j * 2 + t] = m;
            if (a.tmask) {
              const float lim = m + a.band;
              unsigned int bits = ((o[0] <= lim) ? 1u : 0u) | ((o[1] <= lim) ? 2u : 0u) | ((o[2] <= lim) ? 4u : 0u) |
                                  ((o[3] <= lim) ? 8u : 0u);
              bits <<= 4 * rg;
              bits |= (unsigned int)__shfl_xor((int)bits, 16, 64);
              bits |= (unsigned int)__shfl_xor((int)bits, 32, 64);
              if (rg == 0) a.tmask[(int64_t)q * a.ldT + (int64_t)j * 2 + t] = (uint16_t)bits;
            }
          }
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  }
}

