// Per-code minimum of the exact-f32 cosine family from a BOUNDED prefilter (round 3; BASELINE.json configs[2]).
//
// The reference's text distance (GestureKNN.py:716 -> sklearn paired_distances(metric='cosine') on float32) is defined
// by its arithmetic: normalise, subtract, square and sum in NumPy's einsum order, three separately rounded f32
// operations per element pair - which is why the sweeps of qpg_text.hip are VALU-bound (2.7 ms for 1 000 queries x
// 100 000 rows).  But only the per-code MINIMUM and its first-wins candidate are wanted, and both are decided by
// comparisons: a prefilter whose error against the sklearn value is bounded a priori leaves, per (query, code), a BAND of
// candidates that can be the minimum; only those are evaluated in the exact order.
//   prefilter  qpg_hl_gemm_distance (qpg_audio_hl.hip): d~ = 1 - <x^, q^> on the f16 matrix cores (split operands,
//              f64 block sums), rows SORTED BY CODE (stable: original order inside a code) and padded to 16 per code with
//              copies of the segment's first row; its epilogue also leaves the minimum of every 16-row tile;
//   bound      |d~ - d_sklearn| <= E = E_pre + E_sk:  E_pre = 1.3e-6 (the GEMM, unit-norm operands) + 2 eps1 (x^, q^ are
//              the f32-normalised rows, off the true unit vectors by eps1 = ((D/4 + 2)/2 + 2) u each);
//              E_sk = 0.5 [ 8 eps1 + 4 (D/4 + 3) u ]  (sklearn's own f32 rounding against the real-number value:
//              normalisation errors through the difference, Cauchy-Schwarz with |delta| <= 2, then the 4-lane chains of
//              D/4 squares) - 4.3e-5 at D = 512; `band` = 2.1 E is passed by the caller;
//   select     SORT_SPLIT blocks per query, a range of codes (= of tiles) each: (1) per-code minimum of d~ from the tile minima,
//              (2) tiles whose minimum is within `band` of their code's minimum are opened and their rows within the band listed, (3) the listed (query, row) pairs
//              are evaluated in sklearn's exact order from the f32 rows, (4) per code the minimum exact distance and,
//              among equals, the lowest ORIGINAL index (first-wins); tables, nearest neighbours, optionally ranks.
//              (Round 3's first version was one block per query streaming the whole row twice: 0.43 ms of cfg-3's 0.94,
//              71 us of the matcher's text side; the tile minima cut the reads from 2 R to R / 16 + the opened tiles.)
// A row sklearn's normalisation leaves at (or near) zero - an all-zero embedding: its norm is replaced by 1 - is not a unit
// vector: its distance to a unit query is 0.5 |q^|^2, not 1 - <x^, q^>.  All such rows of a code are at the SAME exact
// distance from a query, so only the one with the lowest original index can win: the builder keeps that one per code
// (zero_row[K]) OUTSIDE the GEMM's rows, and the select enters it with the prefilter value 0.5 (a zero QUERY shifts every
// value by the same -0.5: order and bands are kept, and since all rows then tie inside the band the list overflows and
// the exact sweep answers).
// The tables are bit-identical to qpg_text_percode_f32's.  A list that overflows raises stats[1] |= 16 (the host then
// runs the exact VALU sweep): real text embeddings repeat (silence), and thousands of exact ties in one code are then
// all inside the band.
#include "qpg_common.h"

__device__ __forceinline__ unsigned int okey32(float d) {
  const unsigned int b = __float_as_uint(d);
  return (b >> 31) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float okey32_value(unsigned int k) {
  return __uint_as_float((k >> 31) ? (k & 0x7fffffffu) : ~k);
}

#define SORT_SPLIT 8       // sub-blocks per query: rows are sorted by code, so a range of codes is a range of tiles and
                           // its tables need nothing from the other ranges
#define SORT_LIST 2048     // band members a sub-block can hold (x SORT_SPLIT per query)
#define SORT_TILES 1024    // opened tiles a sub-block can hold
#define SORT_THREADS 256

// Block (q, s): query q, codes [s K / S, (s + 1) K / S).  The GEMM's epilogue left the minimum of every 16-row TILE (a tile
// lies inside one code's segment: segments are padded to 16 rows with copies of their first row, so a padding row never
// lowers a minimum), so the block reads its share of the R / 16 tile minima instead of the R distances: (1) per-code
// minimum over the tiles (+ 0.5 for a code that holds an all-zero row), (2) tiles whose minimum lies inside the code's band
// are opened (16 lanes per tile) and their rows within the band listed, (3) exact sklearn-order distances of the listed
// pairs, (4) tables.
__global__ __launch_bounds__(SORT_THREADS) void percode_select_sorted_kernel(
    const float* __restrict__ Dm, int64_t ldD, const float* __restrict__ tmin, const uint16_t* __restrict__ tmask,
    int64_t ldT, int64_t R, const int16_t* __restrict__ row_code, const int32_t* __restrict__ row_index, const int32_t* __restrict__ zero_row,
    const int32_t* __restrict__ code_tile, int K, float band, const float* __restrict__ qn, const float* __restrict__ xs,
    int Dd, float absent, float* __restrict__ out_dist, int32_t* __restrict__ out_idx, int32_t* __restrict__ stats,
    int32_t idx_base, int q_block, int64_t block_stride) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int S = gridDim.y, s = blockIdx.y;
  const int k0 = (int)((int64_t)s * K / S), k1 = (int)((int64_t)(s + 1) * K / S), KL = k1 - k0;
  const int KLmax = ((K + S - 1) / S + 4) & ~3;          // (multiple of 4: keeps qrow 16-byte aligned)
  unsigned long long* ebest = reinterpret_cast<unsigned long long*>(smem);            // [KL] exact key << 32 | original index
  unsigned int* best = reinterpret_cast<unsigned int*>(ebest + KLmax);                // [KL] order key of the prefilter minimum
  int* list = reinterpret_cast<int*>(best + KLmax);                                   // [SORT_LIST] rows in a band
  int* tl = list + SORT_LIST;                                                         // [SORT_TILES] opened tiles
  float* qrow = reinterpret_cast<float*>(tl + SORT_TILES);                            // [Dd]
  __shared__ int n_list, n_tiles;
  const int q = blockIdx.x, tid = threadIdx.x;
  const float* row = Dm ? Dm + (int64_t)q * ldD : nullptr;
  const float* trow = tmin + (int64_t)q * ldT;
  const uint16_t* mrow = tmask ? tmask + (int64_t)q * ldT : nullptr;
  const int t0 = code_tile[k0], t1 = code_tile[k1];            // this range's tiles
  for (int k = tid; k < KL; k += blockDim.x) {
    ebest[k] = ~0ull;
    best[k] = (zero_row && zero_row[k0 + k] >= 0) ? okey32(0.5f) : 0xffffffffu;
  }
  for (int i = tid; i < Dd / 4; i += blockDim.x)
    reinterpret_cast<f32x4*>(qrow)[i] = reinterpret_cast<const f32x4*>(qn + (int64_t)q * Dd)[i];
  if (tid == 0) {
    n_list = 0;
    n_tiles = 0;
  }
  __syncthreads();
  // (1) per-code minimum of the prefilter values, from the tile minima
  for (int t = t0 + tid; t < t1; t += blockDim.x) {
    const int code = (row_code[(int64_t)t * 16] & 0x1fff) - k0;
    if ((unsigned)code < (unsigned)KL) atomicMin(&best[code], okey32(trow[t]));
  }
  __syncthreads();
  if (band >= 0.f) {           // (band < 0: timing diagnostics - nothing is listed, the tables come out empty)
    // (2a) tiles whose minimum is inside their code's band
    for (int t = t0 + tid; t < t1; t += blockDim.x) {
      const int code = (row_code[(int64_t)t * 16] & 0x1fff) - k0;
      if ((unsigned)code >= (unsigned)KL) continue;
      if (trow[t] <= okey32_value(best[code]) + band) {
        const int pos = atomicAdd(&n_tiles, 1);
        if (pos < SORT_TILES) tl[pos] = t;
      }
    }
    for (int k = tid; k < KL; k += blockDim.x)
      if (zero_row && zero_row[k0 + k] >= 0 && 0.5f <= okey32_value(best[k]) + band) {
        const int pos = atomicAdd(&n_list, 1);
        if (pos < SORT_LIST) list[pos] = (int)R + k0 + k;            // the code's all-zero row: xs row R is zeros
      }
    __syncthreads();
    int nt = n_tiles;
    if (nt > SORT_TILES) {
      nt = SORT_TILES;
      if (tid == 0 && stats) atomicOr(&stats[1], 16);          // (16 = this prefilter's own overflow bit)
    }
    // (2b) their rows: 16 lanes per opened tile.  With the GEMM's tile MASKS (round 4) the matrix is not read: bit r of
    // a tile's mask = row r lies within the band of the TILE's minimum, a superset of the rows within the band of the
    // code's minimum (code minimum <= tile minimum; the surplus only gets an exact evaluation it did not need).
    for (int i = tid >> 4; i < nt; i += blockDim.x >> 4) {
      const int64_t r = (int64_t)tl[i] * 16 + (tid & 15);
      const int cd = row_code[r];
      if (cd & 0x4000) continue;
      const bool in = mrow ? ((mrow[tl[i]] >> (tid & 15)) & 1) != 0
                           : row[r] <= okey32_value(best[(cd & 0x1fff) - k0]) + band;
      if (in) {
        const int pos = atomicAdd(&n_list, 1);
        if (pos < SORT_LIST) list[pos] = (int)r;
      }
    }
  }
  __syncthreads();
  int n = n_list;
  if (n > SORT_LIST) {
    n = SORT_LIST;
    if (tid == 0 && stats) atomicOr(&stats[1], 16);
  }
  // (3) exact sklearn-order distance of every listed (query, row) pair: 0.5 * einsum_sq(qn - xn), four lane chains,
  // 16-element groups visited u = 3,2,1,0, separate multiply and add, (l0 + l1) + (l2 + l3).  One thread per pair, the
  // query row in LDS, the candidate row gathered (16 loads in flight per thread: the gather is latency-bound).  Measured
  // alternative: evaluating per CODE instead (buckets of (query, row) pairs, one block per code, the rows of a code read
  // once for all queries) is SLOWER - 465 us against ~200: every lane then gathers BOTH operands in 16-byte pieces.
  // Four lanes per pair (one per chain, a lane's 96-128 floats all requested at once) is no faster either: 0.300 against
  // 0.266 ms for the cfg-3 batch, 27 against 25 us for the matcher's text side.
  for (int e = tid; e < n; e += blockDim.x) {
    const int r = list[e];
    const bool z = r >= (int)R;
    const f32x4* xp = reinterpret_cast<const f32x4*>(xs + (int64_t)(z ? (int)R : r) * Dd);
    const f32x4* qp = reinterpret_cast<const f32x4*>(qrow);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int g0 = 0; g0 < Dd / 16; g0 += 4) {
      f32x4 xv[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) xv[i] = (g0 * 4 + i) < Dd / 4 ? xp[g0 * 4 + i] : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int gg = 0; gg < 4; ++gg) {
        if (g0 + gg >= Dd / 16) break;
#pragma unroll
        for (int u = 3; u >= 0; --u) {
          const f32x4 x = xv[gg * 4 + u], qq = qp[(g0 + gg) * 4 + u];
          const float d0 = f_sub(qq.x, x.x), d1 = f_sub(qq.y, x.y), d2 = f_sub(qq.z, x.z), d3 = f_sub(qq.w, x.w);
          a0 = f_add(f_mul(d0, d0), a0);
          a1 = f_add(f_mul(d1, d1), a1);
          a2 = f_add(f_mul(d2, d2), a2);
          a3 = f_add(f_mul(d3, d3), a3);
        }
      }
    }
    const float dist = f_mul(0.5f, f_add(f_add(a0, a1), f_add(a2, a3)));
    const int cd = z ? r - (int)R : (row_code[r] & 0x1fff);
    const int oi = z ? zero_row[cd] : row_index[r];
    atomicMin(&ebest[cd - k0], ((unsigned long long)okey32(dist) << 32) | (unsigned int)oi);
  }
  __syncthreads();
  // (4) this range's part of the tables.  Exchange layout (sharded DB): row q lives in block q / q_block of a byte buffer
  // whose blocks are block_stride bytes apart (qpg_percode_select_f32's); indices leave as global candidate indices.
  if (q_block > 0) {
    const int64_t shift = (int64_t)(q / q_block) * block_stride;
    const int64_t rowoff = (int64_t)(q % q_block) * K - (int64_t)q * K;
    out_dist = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(out_dist) + shift) + rowoff;
    out_idx = reinterpret_cast<int32_t*>(reinterpret_cast<unsigned char*>(out_idx) + shift) + rowoff;
  }
  for (int k = tid; k < KL; k += blockDim.x) {
    const unsigned long long kv = ebest[k];
    const bool have = kv != ~0ull;
    out_dist[(int64_t)q * K + k0 + k] = have ? okey32_value((unsigned int)(kv >> 32)) : absent;
    out_idx[(int64_t)q * K + k0 + k] = have ? (int32_t)(kv & 0xffffffffu) + idx_base : -1;
  }
}

// (5) ranks of the finished rows (stable: value, then code - qpg_rank_rows_f32's) and the nearest neighbours: its own small
// launch.  (Letting the LAST sub-block of a query do it needs a device-scope release fence in every block; on this part
// that writes back the XCD's whole L2 - the freshly written distance matrix included: measured 1.55 ms instead of 0.43
// for the cfg-3 batch.)
__global__ __launch_bounds__(256) void sorted_finish_kernel(const float* __restrict__ dist, const int32_t* __restrict__ idx,
                                                             int K, int16_t* __restrict__ out_rank,
                                                             int32_t* __restrict__ out_nn) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned long long* skey = reinterpret_cast<unsigned long long*>(smem);     // [Kp] sort scratch
  int* scode = reinterpret_cast<int*>(skey + rank_sort_pow2(K));              // [Kp]
  float* v = reinterpret_cast<float*>(scode + rank_sort_pow2(K));             // [K]
  __shared__ unsigned long long nn_key;
  const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  if (tid == 0) nn_key = ~0ull;
  unsigned long long mine = ~0ull;
  for (int k = tid; k < K; k += blockDim.x) {
    const float dv = dist[(int64_t)q * K + k];
    const int iv = idx[(int64_t)q * K + k];
    v[k] = dv;
    if (iv >= 0) {
      const unsigned long long kv = ((unsigned long long)okey32(dv) << 32) | (unsigned int)iv;
      if (kv < mine) mine = kv;
    }
  }
  __syncthreads();
  if (out_nn) {
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long other = __shfl_xor(mine, o, 64);
      mine = other < mine ? other : mine;
    }
    if (lane == 0 && mine != ~0ull) atomicMin(&nn_key, mine);
    __syncthreads();
    if (tid == 0) out_nn[q] = nn_key != ~0ull ? (int32_t)(nn_key & 0xffffffffu) : -1;
  }
  if (!out_rank) return;
  block_sorted_ranks(v, K, skey, scode, [&](int k, int r) { out_rank[(int64_t)q * K + k] = (int16_t)r; });
}

extern "C" int qpg_percode_select_sorted_f32(qpg_ctx* ctx, void* stream, const float* Dm, int64_t ldD, const float* tile_min,
                                             const uint16_t* tile_mask, int64_t ldT, int Q, int64_t R,
                                             const int16_t* row_code,
                                             const int32_t* row_index, const int32_t* zero_row, const int32_t* code_tile,
                                             int K, float band, const float* qn, const float* xs, int Dd, float absent,
                                             float* out_dist, int32_t* out_idx, int16_t* out_rank, int32_t* out_nn,
                                             int32_t* stats, int32_t idx_base, int q_block, int64_t block_stride) {
  const char* name = "qpg_percode_select_sorted_f32";
  QPG_REQUIRE(ctx && (Dm || tile_mask) && tile_min && row_code && row_index && code_tile && qn && xs && out_dist && out_idx,
              "%s: null pointer", name);
  QPG_REQUIRE(Q >= 0 && R > 0 && (R % 16) == 0 && R < 0x7fffffffll - 0x2000 && (!Dm || ldD >= R) && ldT >= R / 16 && K > 0 &&
                  K <= 2048 && K <= 0x1fff && Dd > 0 && (Dd % 16) == 0 && (reinterpret_cast<uintptr_t>(xs) % 16) == 0 &&
                  (reinterpret_cast<uintptr_t>(qn) % 16) == 0,
              "%s: bad size / alignment (R %% 16 == 0, D %% 16 == 0, K <= 2048)", name);
  QPG_REQUIRE(q_block >= 0 && idx_base >= 0 &&
                  (q_block == 0 || (!out_rank && !out_nn && Q % q_block == 0 && block_stride % 8 == 0)),
              "%s: block layout needs Q %% q_block == 0, an 8-byte multiple stride and no rank / nn output", name);
  if (Q == 0) return QPG_OK;
  const int S = K >= 32 * SORT_SPLIT ? SORT_SPLIT : (K >= 32 ? K / 32 : 1);
  const int KLmax = ((K + S - 1) / S + 4) & ~3;          // (multiple of 4: keeps qrow 16-byte aligned)
  const size_t sh = 12 * (size_t)KLmax + 4 * (size_t)SORT_LIST + 4 * (size_t)SORT_TILES + 4 * (size_t)Dd;
  QPG_REQUIRE(sh <= 64 * 1024, "%s: K / D too large for the LDS tables", name);
  hipLaunchKernelGGL(percode_select_sorted_kernel, dim3(Q, S), dim3(SORT_THREADS), sh, qpg_stream(stream), Dm, ldD, tile_min,
                     tile_mask, ldT, R, row_code, row_index, zero_row, code_tile, K, band, qn, xs, Dd, absent, out_dist, out_idx, stats,
                     idx_base, q_block, block_stride);
  QPG_LAUNCH_CHECK("percode_select_sorted_kernel");
  if (out_rank || out_nn) {
    hipLaunchKernelGGL(sorted_finish_kernel, dim3(Q), dim3(256), 12 * (size_t)rank_sort_pow2(K) + 4 * (size_t)K, qpg_stream(stream),
                       (const float*)out_dist, (const int32_t*)out_idx, K, out_rank, out_nn);
    QPG_LAUNCH_CHECK("sorted_finish_kernel");
  }
  return QPG_OK;
}

// ---- round 5: the exact-order evaluations BY CODE (many queries per batch: BASELINE.json configs[2]) ------------------------
// percode_select_sorted_kernel gathers one 2 KB candidate row per listed (query, row) pair: 512 K pairs = 0.95 GB per
// 1 000 queries although every one of the 100 K rows is wanted by ~5 queries (profiles/r04_cfg3_pmc.md).  Here a block
// owns ONE CODE (its rows are one contiguous segment of the sorted rows) and a range of queries:
//   (1) per query the code's minimum over its tiles, from the TILE-MAJOR minima the h-plane GEMM wrote (a tile's queries
//       are one coalesced run);  (2) opened tiles' masked rows listed as (query, row) pairs;  (3) every pair evaluated in
//       sklearn's exact order by FOUR lanes, one per einsum chain: both operands are stored CHAIN-PERMUTED (perm32_kernel:
//       inside every 32-element group = 128-byte line, chain k's eight elements in visiting order at 8 k .. 8 k + 7), so that
//       lane k reads the 32 bytes it visits next and the four lanes of a pair read whole lines of the row and of the query; the rows of the
//       segment are fetched from HBM once and hit L1 / L2 for the segment's other pairs, the queries (2 MB) live in L2;
//       a_k = d^2 + a_k in the reference's order (groups ascending, u = 3, 2, 1, 0), (a_0 + a_1) + (a_2 + a_3) by two
//       lane exchanges;  (4) minimum by (exact distance, ORIGINAL index) per query in LDS, one table entry per query.
// Tables bit-identical to percode_select_sorted_kernel's and to the exact sweep's (tests/test_gpu_cfg3.py).  A pair list
// that overflows raises stats[1] |= 16 like the by-query kernel's.  (Round 3's by-code attempt - 465 us - gathered BOTH
// operands in 16-byte pieces per lane from a query-major matrix; this one reads 64-byte runs and nothing query-major.)
// y[32 G + 8 k + j] = x[16 (2 G + (j >> 2)) + 4 (3 - (j & 3)) + k]: inside every 32-element group (one 128-byte line) chain
// k's eight elements sit in visiting order - 16-element groups ascending, u = 3, 2, 1, 0 inside each (NumPy einsum's order)
__global__ __launch_bounds__(256) void perm32_kernel(const float* __restrict__ x, int64_t n32, float* __restrict__ y) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n32; i += (int64_t)gridDim.x * blockDim.x) {
    const f32x4* p = reinterpret_cast<const f32x4*>(x) + i * 8;
    f32x4 u[8];
#pragma unroll
    for (int v = 0; v < 8; ++v) u[v] = p[v];                        // u[4 h + uu]: group 2 G + h, vector uu
    f32x4* o = reinterpret_cast<f32x4*>(y) + i * 8;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      o[2 * k] = (f32x4){u[3][k], u[2][k], u[1][k], u[0][k]};
      o[2 * k + 1] = (f32x4){u[7][k], u[6][k], u[5][k], u[4][k]};
    }
  }
}

extern "C" int qpg_perm32_rows_f32(qpg_ctx* ctx, void* stream, const float* x, int64_t R, int D, float* y) {
  QPG_REQUIRE(ctx && x && y && x != y && R >= 0 && D > 0 && (D % 32) == 0 && (reinterpret_cast<uintptr_t>(x) % 16) == 0 &&
                  (reinterpret_cast<uintptr_t>(y) % 16) == 0,
              "qpg_perm32_rows_f32: bad argument (D %% 32 == 0, 16-byte aligned, out of place)");
  const int64_t n32 = R * (D / 32);
  if (n32 == 0) return QPG_OK;
  const int64_t nb = (n32 + 255) / 256;
  hipLaunchKernelGGL(perm32_kernel, dim3((unsigned)(nb < 16384 ? nb : 16384)), dim3(256), 0, qpg_stream(stream), x, n32, y);
  QPG_LAUNCH_CHECK("perm32_kernel");
  return QPG_OK;
}

// Block = ONE CODE, all queries of the launch (<= BYC_QMAX), 512 threads.  The code's rows are staged through LDS tile by
// tile (16 rows = 32 KB, coalesced loads into registers one tile ahead); per tile the queries whose band the
// tile's minimum lies in are listed from the tile-major minima / masks (~100 pairs of cfg-3's 1 000 queries) and evaluated
// by four lanes each: the row from LDS, the query (L2-resident) gathered in whole 128-byte lines, both chain-permuted.
// History of this kernel (cfg-3 batch, 1 000 queries): rows AND queries gathered per pair from memory, one block per
// (code, query range): 181 us with 64-byte runs (16-element permutation), 165 us with whole lines - four dependent
// batches of HBM-latency loads per pair whatever the block shape; rows through LDS: see DESIGN.md 4.4.
#define BYC_THREADS 512
#define BYC_QMAX 2048      // queries per launch of the kernel (the host wrapper loops over larger batches)
#define BYC_LIST 1024      // (query, row) pairs of ONE tile a block can hold

template <int PT>              // 16-byte units of a tile per thread: 16 (D / 4) / BYC_THREADS
__global__ __launch_bounds__(BYC_THREADS) void percode_select_bycode_kernel(
    const float* __restrict__ tmin_t, const uint16_t* __restrict__ tmask_t, int64_t ldQ, int Q, int64_t R,
    const int16_t* __restrict__ row_code, const int32_t* __restrict__ row_index, const int32_t* __restrict__ zero_row,
    const int32_t* __restrict__ code_tile, int K, float band, const float* __restrict__ qp, const float* __restrict__ xsp,
    int Dd, float absent, float* __restrict__ out_dist, int32_t* __restrict__ out_idx, int32_t* __restrict__ stats,
    int32_t idx_base) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int rowb = Dd * 4;                                             // bytes of a row
  float* rows0 = reinterpret_cast<float*>(smem);                       // [17][Dd]: the tile's 16 rows + a zero row
  unsigned long long* ebest = reinterpret_cast<unsigned long long*>(smem + (size_t)17 * rowb);       // [Q]
  float* cmin = reinterpret_cast<float*>(ebest + Q);                   // [Q]
  int* list_q = reinterpret_cast<int*>(cmin + Q);                      // [BYC_LIST]
  int* list_r = list_q + BYC_LIST;                                     // [BYC_LIST] row inside the tile (16: the zero row)
  __shared__ int n_list[2];                                            // tile t counts in n_list[t & 1]
  const int code = blockIdx.x, tid = threadIdx.x;
  const int t0 = code_tile[code], t1 = code_tile[code + 1], nt = t1 - t0;
  const int zr = zero_row ? zero_row[code] : -1;
  const int nv = Dd / 4;                                               // 16-byte units of a row
  // (0) the code's minimum per query (tile-major minima: coalesced; eight independent loads per trip)
  for (int q = tid; q < Q; q += BYC_THREADS) {
    const float* tp = tmin_t + (int64_t)t0 * ldQ + q;
    float m = zr >= 0 ? 0.5f : __builtin_inff();
    for (int t = 0; t < nt; t += 8) {
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = t + i < nt ? tp[(int64_t)(t + i) * ldQ] : __builtin_inff();
#pragma unroll
      for (int i = 0; i < 8; ++i) m = fminf(m, v[i]);
    }
    cmin[q] = m;
    ebest[q] = ~0ull;
  }
  for (int i = tid; i < nv; i += BYC_THREADS)                          // the zero row
    reinterpret_cast<f32x4*>(rows0 + 16 * Dd)[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (tid == 0) n_list[0] = n_list[1] = 0;
  f32x4 stage[PT];
  float nmin[BYC_QMAX / BYC_THREADS];                                  // the next tile's minima / masks of this thread's queries
  unsigned int nmask[BYC_QMAX / BYC_THREADS];
  auto fetch = [&](int t) {                                            // tile t0 + t -> registers (coalesced)
    const f32x4* src = reinterpret_cast<const f32x4*>(xsp + (int64_t)(t0 + t) * 16 * Dd);
#pragma unroll
    for (int u = 0; u < PT; ++u) {
      const int i = u * BYC_THREADS + tid;
      stage[u] = i < 16 * nv ? src[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const int64_t to = (int64_t)(t0 + t) * ldQ;
#pragma unroll
    for (int u = 0; u < BYC_QMAX / BYC_THREADS; ++u) {
      const int q = u * BYC_THREADS + tid;
      nmin[u] = q < Q ? tmin_t[to + q] : __builtin_inff();
      nmask[u] = q < Q ? tmask_t[to + q] : 0u;
    }
  };
  auto commit = [&]() {
    f32x4* dst = reinterpret_cast<f32x4*>(rows0);
#pragma unroll
    for (int u = 0; u < PT; ++u) {
      const int i = u * BYC_THREADS + tid;
      if (i < 16 * nv) dst[i] = stage[u];
    }
  };
  if (nt > 0 && band >= 0.f) {
    fetch(0);
    commit();
  }
  __syncthreads();
  const int k4 = tid & 3;
  const int G32 = Dd / 32;
  for (int t = 0; t < nt && band >= 0.f; ++t) {
    float vmin[BYC_QMAX / BYC_THREADS];
    unsigned int vmask[BYC_QMAX / BYC_THREADS];
#pragma unroll
    for (int u = 0; u < BYC_QMAX / BYC_THREADS; ++u) {
      vmin[u] = nmin[u];
      vmask[u] = nmask[u];
    }
    if (t + 1 < nt) fetch(t + 1);                                      // rows, minima, masks: in flight underneath this tile
    // (1) this tile's pairs: queries whose band the tile's minimum lies in, their masked rows (padding rows never);
    // the code's all-zero row enters with the first tile
#pragma unroll
    for (int u = 0; u < BYC_QMAX / BYC_THREADS; ++u) {
      const int q = u * BYC_THREADS + tid;
      if (q >= Q) break;
      const float lim = cmin[q] + band;
      if (vmin[u] <= lim) {
        unsigned int bits = vmask[u];
        while (bits) {
          const int r = __builtin_ctz(bits);
          bits &= bits - 1;
          if (row_code[(int64_t)(t0 + t) * 16 + r] & 0x4000) continue;
          const int pos = atomicAdd(&n_list[t & 1], 1);
          if (pos < BYC_LIST) {
            list_q[pos] = q;
            list_r[pos] = r;
          }
        }
      }
      if (t == 0 && zr >= 0 && 0.5f <= lim) {
        const int pos = atomicAdd(&n_list[t & 1], 1);
        if (pos < BYC_LIST) {
          list_q[pos] = q;
          list_r[pos] = 16;
        }
      }
    }
    __syncthreads();
    int n = n_list[t & 1];
    if (tid == 0) n_list[(t + 1) & 1] = 0;                             // (nobody touches it before the next barrier)
    if (n > BYC_LIST) {
      n = BYC_LIST;
      if (tid == 0 && stats) atomicOr(&stats[1], 16);
    }
    // (2) four lanes per pair, lane k = einsum chain k: 0.5 * ((a0 + a1) + (a2 + a3)), a_k = d * d + a_k over the chain's
    // elements in the reference's order - which the chain-permuted rows store consecutively (32 bytes per 128-byte line)
    const float* rb = rows0;
    for (int e0 = 0; e0 < n; e0 += BYC_THREADS / 4) {
      const int e = e0 + (tid >> 2);
      const bool on = e < n;
      const int q = on ? list_q[e] : 0, r = on ? list_r[e] : 16;
      const f32x4* xp = reinterpret_cast<const f32x4*>(rb + (size_t)r * Dd) + 2 * k4;
      const f32x4* qv = reinterpret_cast<const f32x4*>(qp + (int64_t)q * Dd) + 2 * k4;
      float a = 0.f;
      for (int g0 = 0; g0 < G32; g0 += 8) {
        f32x4 qq[16];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const bool in = g0 + i < G32;
          qq[2 * i] = in ? qv[(g0 + i) * 8] : (f32x4){0.f, 0.f, 0.f, 0.f};
          qq[2 * i + 1] = in ? qv[(g0 + i) * 8 + 1] : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          if (g0 + (i >> 1) >= G32) break;
          const f32x4 xv = xp[(g0 + (i >> 1)) * 8 + (i & 1)];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float d = f_sub(qq[i][j], xv[j]);
            a = f_add(f_mul(d, d), a);
          }
        }
      }
      const float s01 = f_add(a, __shfl_xor(a, 1, 64));            // lane 0: a0 + a1, lane 2: a2 + a3
      const float s = f_add(s01, __shfl_xor(s01, 2, 64));          // lane 0: (a0 + a1) + (a2 + a3)
      if (on && k4 == 0) {
        const float dist = f_mul(0.5f, s);
        const int oi = r == 16 ? zr : row_index[(int64_t)(t0 + t) * 16 + r];
        atomicMin(&ebest[q], ((unsigned long long)okey32(dist) << 32) | (unsigned int)oi);
      }
    }
    __syncthreads();                                                   // every quad is done with this tile's rows
    if (t + 1 < nt) commit();                                          // (visible behind the next tile's listing barrier)
  }
  if (nt == 0 && zr >= 0 && band >= 0.f) {
    // a code whose only rows are all-zero embeddings: every query is at 0.5 |q^|^2 from it - exact value by lane 0 alone
    for (int q = tid; q < Q; q += BYC_THREADS) {
      const float* qr = qp + (int64_t)q * Dd;
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      for (int G = 0; G < G32; ++G)
        for (int j = 0; j < 8; ++j) {
          const float d0 = qr[G * 32 + j], d1 = qr[G * 32 + 8 + j], d2 = qr[G * 32 + 16 + j], d3 = qr[G * 32 + 24 + j];
          a0 = f_add(f_mul(d0, d0), a0);
          a1 = f_add(f_mul(d1, d1), a1);
          a2 = f_add(f_mul(d2, d2), a2);
          a3 = f_add(f_mul(d3, d3), a3);
        }
      const float dist = f_mul(0.5f, f_add(f_add(a0, a1), f_add(a2, a3)));
      ebest[q] = ((unsigned long long)okey32(dist) << 32) | (unsigned int)zr;
    }
    __syncthreads();
  }
  // (3) this code's column of the tables
  for (int q = tid; q < Q; q += BYC_THREADS) {
    const unsigned long long kv = ebest[q];
    const bool have = kv != ~0ull;
    out_dist[(int64_t)q * K + code] = have ? okey32_value((unsigned int)(kv >> 32)) : absent;
    out_idx[(int64_t)q * K + code] = have ? (int32_t)(kv & 0xffffffffu) + idx_base : -1;
  }
}

extern "C" int qpg_percode_select_bycode_f32(qpg_ctx* ctx, void* stream, const float* tile_min_t, const uint16_t* tile_mask_t,
                                             int64_t ldQ, int Q, int64_t R, const int16_t* row_code,
                                             const int32_t* row_index, const int32_t* zero_row, const int32_t* code_tile,
                                             int K, float band, const float* qn_perm, const float* xs_perm, int Dd,
                                             float absent, float* out_dist, int32_t* out_idx, int16_t* out_rank,
                                             int32_t* out_nn, int32_t* stats, int32_t idx_base) {
  const char* name = "qpg_percode_select_bycode_f32";
  QPG_REQUIRE(ctx && tile_min_t && tile_mask_t && row_code && row_index && code_tile && qn_perm && xs_perm && out_dist && out_idx,
              "%s: null pointer", name);
  QPG_REQUIRE(Q >= 0 && ldQ >= Q && R > 0 && (R % 16) == 0 && R < 0x7fffffffll - 0x2000 && K > 0 && K <= 0x1fff && Dd > 0 &&
                  (Dd % 128) == 0 && Dd <= 1024 && (reinterpret_cast<uintptr_t>(xs_perm) % 16) == 0 &&
                  (reinterpret_cast<uintptr_t>(qn_perm) % 16) == 0 && idx_base >= 0,
              "%s: bad size / alignment (R %% 16 == 0, D %% 128 == 0, D <= 1024)", name);
  if (Q == 0) return QPG_OK;
  static bool raised = false;
  if (!raised) {
    bool ok = true;
#define BYC_RAISE(PT_) ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(percode_select_bycode_kernel<PT_>), \
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) == hipSuccess
    BYC_RAISE(1); BYC_RAISE(2); BYC_RAISE(3); BYC_RAISE(4); BYC_RAISE(5); BYC_RAISE(6); BYC_RAISE(7); BYC_RAISE(8);
#undef BYC_RAISE
    if (!ok) {
      qpg_set_error("%s: cannot raise the dynamic LDS limit", name);
      return QPG_EHIP;
    }
    raised = true;
  }
  for (int qa = 0; qa < Q; qa += BYC_QMAX) {                            // (tables of [Q][K]: a query range is a row range)
    const int qn = Q - qa < BYC_QMAX ? Q - qa : BYC_QMAX;
    const size_t sh = (size_t)17 * Dd * 4 + 12 * (size_t)qn + 8 * (size_t)BYC_LIST;
    QPG_REQUIRE(sh <= 96 * 1024, "%s: LDS tables too large", name);
#define BYC_GO(PT_) hipLaunchKernelGGL(percode_select_bycode_kernel<PT_>, dim3(K), dim3(BYC_THREADS), sh, qpg_stream(stream), \
                                       tile_min_t + qa, tile_mask_t + qa, ldQ, qn, R, row_code, row_index, zero_row, code_tile, K, \
                                       band, qn_perm + (int64_t)qa * Dd, xs_perm, Dd, absent, out_dist + (int64_t)qa * K,           \
                                       out_idx + (int64_t)qa * K, stats, idx_base)
    switch (Dd / 128) {                                                 // 16 (D / 4) / 512 units per thread
      case 1: BYC_GO(1); break;
      case 2: BYC_GO(2); break;
      case 3: BYC_GO(3); break;
      case 4: BYC_GO(4); break;
      case 5: BYC_GO(5); break;
      case 6: BYC_GO(6); break;
      case 7: BYC_GO(7); break;
      default: BYC_GO(8); break;
    }
#undef BYC_GO
    QPG_LAUNCH_CHECK("percode_select_bycode_kernel");
  }
  if (out_rank || out_nn) {
    hipLaunchKernelGGL(sorted_finish_kernel, dim3(Q), dim3(256), 12 * (size_t)rank_sort_pow2(K) + 4 * (size_t)K, qpg_stream(stream),
                       (const float*)out_dist, (const int32_t*)out_idx, K, out_rank, out_nn);
    QPG_LAUNCH_CHECK("sorted_finish_kernel");
  }
  return QPG_OK;
}
