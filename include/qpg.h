/*
 * qpg.h — C ABI of libqpg_hip.so: the MI355X (gfx950) implementation of QPGesture's
 * code-level motion-matching hot path (CodeKNN) and gesture VQ-VAE encode/quantise/decode.
 *
 * The reference (YoungSeng/QPGesture) is pure Python and has no FFI layer; each entry point
 * below names the reference interface it replaces (paths relative to the reference root).
 * INTEGRATION.md shows the ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer marked [dev] is a device pointer owned by the caller (the Python host keeps
 *     torch tensors alive); the library never allocates or frees caller-visible memory;
 *   - all work is enqueued on the hipStream_t passed as `void* stream` (0 = null stream) and
 *     is stream-ordered; no call synchronises the device;
 *   - every function returns QPG_OK (0) or a negative QPG_E* code and never throws;
 *     qpg_last_error() returns the text of the last failure on the calling thread;
 *   - no global mutable state except the per-device qpg_ctx.
 *   - tie rule everywhere: lowest index wins (== the reference's strict `<` first-wins scan,
 *     GestureKNN.py:686,717) and ranks are stable (documented deviation from NumPy's unstable
 *     default argsort, DESIGN.md "Tie contract").
 */
#ifndef QPG_H
#define QPG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QPG_OK 0
#define QPG_EINVAL (-1)   /* bad argument (null pointer, size out of range)          */
#define QPG_EHIP (-2)     /* a HIP runtime call or kernel launch failed               */
#define QPG_EUNSUP (-3)   /* shape not supported by the compiled kernels              */

typedef struct qpg_ctx qpg_ctx;

int qpg_version(void);
/* First 16 hex digits of the SHA-256 over the sources the library was compiled from (every .hip and .h file of csrc/
 * in name order, then include/qpg.h: qpgesture_amd/build.py:source_hash()); "unstamped" for a build that bypassed build.py.  The Python binding
 * refuses to run a library whose id differs from the tree it sits in (it rebuilds it). */
const char* qpg_build_id(void);
/* One context per device, created by the host thread that drives that device.
 * PERFORMANCE NOTE for direct callers: the matcher is a chain of a dozen short dependent launches per clip, and the HIP
 * runtime hands their arguments to the command processor through device memory only when the environment variable
 * HIP_FORCE_DEV_KERNARG=1 is set BEFORE the process's first HIP call (the Python package sets it at import; measured
 * ~5 % of a clip, 0.347 -> 0.330 ms).  qpg_dev_kernarg() reports what this process will get: 1 = set, 0 = not set. */
int qpg_ctx_create(int device, qpg_ctx** out);
int qpg_dev_kernarg(void);
int qpg_ctx_destroy(qpg_ctx* ctx);
/* Per-context knobs (the library holds no other mutable state; a knob of one context never affects another):
 *   QPG_OPT_GATE_DEDUP_FROM_CHAINS  from how many chains per qpg_match_steps(_batch) launch the phase-gate table is
 *                                   deduplicated by the previous winner (default 1 = always; 0 = never: round 4's plain
 *                                   table, bit-identical results - the tests walk on both). */
#define QPG_OPT_GATE_DEDUP_FROM_CHAINS 0
#define QPG_OPT_COUNT 4
int qpg_ctx_set_option(qpg_ctx* ctx, int option, int value);
int qpg_ctx_get_option(qpg_ctx* ctx, int option, int* value);
int qpg_last_error(char* buf, size_t n);
/* Stream-ordered signal to the host: *dst = value (system-scope store by a one-thread kernel) once everything enqueued on
 * `stream` before this call has completed.  dst: [host-pinned, device-accessible or dev] i32.  The multi-lane replay
 * pipeline (qpgesture_amd.code_knn.GraphPipeline) polls it to learn that a replay's sweep is over. */
int qpg_signal_i32(qpg_ctx*, void* stream, int32_t* dst, int32_t value);
/* The doorbell of a pre-launched graph replay: a one-thread kernel that takes its sequence number seq = ++*counter
 * (counter: [dev] i32, zero-initialised by the caller, one per captured graph) and waits until *go - seq >= 0 (go:
 * [host-pinned, device-accessible] i32 the host stores to), at most timeout_ms (1..60000; a host that never rings cannot hang
 * the device).  As the FIRST node of a captured matching step it lets the host enqueue the next replay - hipGraphLaunch is
 * ~17 us of host time - while the current one still runs, and start it by a single store once the current results are read
 * and the next seeds written (qpgesture_amd.code_knn.ClipGraph(doorbell=True)). */
int qpg_doorbell_wait(qpg_ctx*, void* stream, int32_t* counter, const int32_t* go, int32_t timeout_ms);

/* ------------------------------------------------------------------------------------------
 * Database preparation (one-off per speaker DB; replaces the host-side feature windowing of
 * codebook/Speech2GestureMatching/data_processing.py:255-274, which materialises an
 * (N,180,6144) float64 stack — here nothing is materialised, only per-candidate norms).
 * ---------------------------------------------------------------------------------------- */

/* WavLM track resampling Tin -> Tout frames: F.interpolate(mode='linear', align_corners=True) in f32, bit-exact
 * with torch's CPU kernel (data_processing.py:258-261: 199 -> 180).  x: [dev] f32 [N][Tin][F]; out: [dev] f32 [N][Tout][F]. */
int qpg_wavlm_resample_f32(qpg_ctx*, void* stream, const float* x, int64_t N, int Tin, int F, int Tout, float* out);

/* out[r] = sum_e x[r][e]^2 in float64, r < rows.  x: [dev] f32 [rows][F]. */
int qpg_frame_norm2_f64(qpg_ctx*, void* stream, const float* x, int64_t rows, int F, double* out);

/* cn2[j][g] = sum_{i<n_taps} fn2[j][cand_t[g] + i*tap_stride]  (0 past T).
 * fn2: [dev] f64 [N][T]; cand_t: [dev] i32 [G]; cn2: [dev] f64 [N][G]. */
int qpg_audio_cand_norm2(qpg_ctx*, void* stream, const double* fn2, int N, int T, const int32_t* cand_t,
                         int G, int n_taps, int tap_stride, double* cn2);

/* Row-wise L2 normalisation with scikit-learn's float32 arithmetic, bit-exact
 * (sklearn.preprocessing.normalize as used by paired_cosine_distances; zero rows stay zero).
 * x, out: [dev] f32 [rows][D]. */
int qpg_l2_normalize_rows_f32(qpg_ctx*, void* stream, const float* x, int64_t rows, int D, float* out);
/* The text queries of a clip, gathered and normalised in one launch: out[r] = normalise(x[q_win[r]][q_row[r]][:])
 * (clip_context[int(i / n_db_frm * 30)] of window q_win[r], GestureKNN.py:549-551).  x: [dev] f32 [M][R][D];
 * q_win, q_row: [dev] i32 [Q] (caller guarantees 0 <= q_win < M, 0 <= q_row < R); out: [dev] f32 [Q][D]. */
int qpg_text_pack_queries_f32(qpg_ctx*, void* stream, const float* x, int M, int R, int D, const int32_t* q_win,
                              const int32_t* q_row, int Q, float* out);

/* ------------------------------------------------------------------------------------------
 * Candidate sweeps.  Replace CodeKNN.search_audio_cands(mode='wavlm_feat') and
 * CodeKNN.search_text_cands (GestureKNN.py:666-691, 708-721) for ALL Q query steps of a clip
 * at once: the scans depend only on the query position, never on the matching state.
 * Candidate index c = j*G + g (j = DB window, g = grid position) = the reference's scan order.
 * ---------------------------------------------------------------------------------------- */

/* Gather the audio queries of a clip into contiguous rows (values stay f32: WavLM features are f32
 * in the reference too, only the arithmetic is f64) and compute their squared norms in f64.
 * qbase: [dev] f32 [M][T][F] (interpolated WavLM of the test windows);
 * q_win/q_t: [dev] i32 [Q] window and start frame of each query (GestureKNN.py:565: clip_test[i]);
 * q32: [dev] f32 [Q][n_taps*F] out; qn2: [dev] f64 [Q] out (squared norms). */
int qpg_audio_pack_queries(qpg_ctx*, void* stream, const float* qbase, int M, int T, int F,
                           const int32_t* q_win, const int32_t* q_t, int Q, int n_taps, int tap_stride,
                           float* q32, double* qn2);

/* Cosine distance of every query against every audio candidate, float64 arithmetic
 * (the reference computes this distance in float64: data_processing.py:264, sklearn keeps f64):
 *   D[q][c] = 1 - <q, cand_c> / (|q| |cand_c|),  cand_c = concat_i base[j][cand_t[g] + i*tap_stride][:]
 * (== 0.5*|q/|q| - c/|c||^2 of sklearn.metrics.pairwise.paired_cosine_distances up to f64 rounding;
 * zero-norm rows follow sklearn: they stay zero vectors).
 * base: [dev] f32 [N][T][F]; cn2: [dev] f64 [N][G]; D: [dev] f64 [Q][N*G], row stride ldD elements. */
int qpg_audio_cosine_f64(qpg_ctx*, void* stream, const float* base, int N, int T, int F,
                         const int32_t* cand_t, int G, int n_taps, int tap_stride, const double* cn2,
                         const float* q32, const double* qn2, int Q, double* D, int64_t ldD);
/* The same sweep over a base track stored in IEEE f16 (BASELINE.json configs[4] "fp16 features": half the HBM bytes of
 * the dominant array).  Values are widened f16 -> f32 -> f64 in registers and the arithmetic is the same f64 as above,
 * i.e. the result is the reference's distance on the f16-ROUNDED track (cn2 must be computed from the rounded values
 * too).  base_f16: [dev] f16 [N][T][F], 16-byte aligned. */
int qpg_audio_cosine_f64_h(qpg_ctx*, void* stream, const void* base_f16, int N, int T, int F,
                           const int32_t* cand_t, int G, int n_taps, int tap_stride, const double* cn2,
                           const float* q32, const double* qn2, int Q, double* D, int64_t ldD);

/* MIXED-PRECISION form of qpg_audio_cosine_f64 (round 2; the default path of CodeKNN.sweep_audio): the same sweep on
 * the f32 matrix cores, with every f32 accumulation chain limited to 32 products and summed in f64, which bounds the
 * result's error A PRIORI: |D[q][c] - exact| <= QPG_AUDIO_MX_ERR for every pair, whatever the data (gamma_32 = 1.91e-6
 * for the f32 chains + 1.2e-7 for an f32-stored matrix + f64 noise; derivation: qpgesture_amd/csrc/qpg_audio.hip).
 * Meant to be consumed by qpg_percode_select_mixed_f64 (and, across row shards, qpg_merge_mixed_*), which re-evaluate
 * every comparison the bound leaves undecided, so the selected candidates and ranks are those of the f64 path.  stats: [dev] i32 [4] (may be NULL): [1] |= 2 if a pair with 0 < |q||c| < 1e-16 was met (operand products
 * could underflow f32, which the bound excludes). */
#define QPG_AUDIO_MX_ERR 2.05e-6
int qpg_audio_cosine_mx(qpg_ctx*, void* stream, const float* base, int N, int T, int F, const int32_t* cand_t, int G,
                        int n_taps, int tap_stride, const double* cn2, const float* q32, const double* qn2, int Q,
                        void* D, int d_is_f32, int64_t ldD, int32_t* stats);
/* D: [dev] [Q][N*G] with row stride ldD ELEMENTS, f64 — or f32 when d_is_f32 != 0: the matrix only feeds the select's
 * two streaming passes, and rounding it to f32 (<= 1.2e-7 on a distance <= 2) is part of QPG_AUDIO_MX_ERR. */

/* qpg_audio_cosine_mx over a base track stored in IEEE f16 (as qpg_audio_cosine_f64_h): same bound, on the rounded values. */
int qpg_audio_cosine_mx_h(qpg_ctx*, void* stream, const void* base_f16, int N, int T, int F, const int32_t* cand_t, int G,
                          int n_taps, int tap_stride, const double* cn2, const float* q32, const double* qn2, int Q,
                          void* D, int d_is_f32, int64_t ldD, int32_t* stats);

/* SPLIT-OPERAND f16 form of the mixed-precision sweep (round 3; the default audio sweep where the grid allows it):
 * same contract as qpg_audio_cosine_mx with a tighter a-priori bound (|D - exact| <= QPG_AUDIO_HL_ERR for every pair,
 * consumed by qpg_percode_select_mixed_f64 with eps1 >= 2 x that), with the products on the f16 matrix cores: every value is scaled by a power
 * of two and stored as two f16 numbers (h, l) with x = h + 2^-11 l (+ <= 2^-23 |x|), the h h' block sums of every MFMA
 * are added in f64, the cross terms run as f32 chains, and the database is read in a frame-major image in which every
 * frame occurs ONCE (super-rows of three frames: dot(q, cand g) = S[g][q first 3 taps] + S[g+1][q last 3 taps]) — the
 * sweep is HBM-bound.  Derivation of the bound and layouts: qpgesture_amd/csrc/qpg_audio_hl.hip.
 *   qpg_audio_hl_supported   1 if the candidate grid has the required shape: 6 taps, G = 26 grid positions
 *                            `cand_step` = 3 x tap_stride frames apart, F %% 128 == 0 (the reference's WavLM grid,
 *                            data_processing.py:264-268 / GestureKNN.py:672-690); else use qpg_audio_cosine_mx.
 *   qpg_audio_hl_pack_db     one-off: base [dev] f32 [N][T][F] -> image [dev] (qpg_audio_hl_db_bytes(N, F) bytes, 16-byte
 *                            aligned): per window [tile 0: k-block][plane][64 units][8 f16] then [tile 1: k-block][plane]
 *                            [44 units][8 f16] (its 11 live rows only: every byte of the image is read) + scale exponent.
 *   qpg_audio_hl_pack_queries per clip: q32 [dev] f32 [Q][6 F] (qpg_audio_pack_queries) -> image [dev]
 *                            (qpg_audio_hl_query_bytes(Q, F) bytes): chunks of 48 queries + one scale exponent per query.
 *   qpg_audio_cosine_hl      D [dev] [Q][N*G] (f32 if d_is_f32, else f64; row stride ldD elements).  cn2 [dev] f64 [N][G],
 *                            qn2 [dev] f64 [Q]: the UNSCALED squared norms (qpg_audio_cand_norm2 / qpg_audio_pack_queries).
 *                            stats[1] |= 2 if a non-zero operand's SCALED squared norm is < 2^16 (a database row whose
 *                            norm is below 1/64 .. 1/128 of the largest magnitude in the whole track: outside the range
 *                            the representation bound covers - the l planes' f16 subnormals; round 5, was < 1). */
#define QPG_AUDIO_HL_ERR 1.3e-6 /* a-priori bound of qpg_audio_cosine_hl: |D - exact| <= this for every pair in range */
int qpg_audio_hl_supported(int T, int F, int G, int n_taps, int tap_stride, int cand_step);
/* qpg_audio_pack_queries + qpg_audio_hl_pack_queries in ONE launch (the pair sits on a clip's critical path): gathers the
 * queries from qbase [dev] f32 [M][T][F] like qpg_audio_pack_queries (q32, qn2 are still written: the select's
 * re-evaluations read them) and writes the split-f16 image.  6 taps, F %% 32 == 0, F <= 2048. */
int qpg_audio_pack_queries_hl(qpg_ctx*, void* stream, const float* qbase, int M, int T, int F, const int32_t* q_win,
                              const int32_t* q_t, int Q, int n_taps, int tap_stride, float* q32, double* qn2, void* image,
                              int64_t image_bytes);
int64_t qpg_audio_hl_db_bytes(int N, int F);
int64_t qpg_audio_hl_query_bytes(int Q, int F);
int qpg_audio_hl_pack_db(qpg_ctx*, void* stream, const float* base, int N, int T, int F, int G, int n_taps, int tap_stride,
                         int cand_step, void* image, int64_t image_bytes);
int qpg_audio_hl_pack_queries(qpg_ctx*, void* stream, const float* q32, int Q, int F, void* image, int64_t image_bytes);
/* Round 4: a clip's WHOLE query side in one launch - qpg_audio_pack_queries_hl's work and the text side's query pack
 * (qpg_text_pack_queries_f32 + qpg_hl_pack_cols, bit-identical outputs): text_ctx [dev] f32 [Mt][R][Dt], tq_win / tq_row
 * [dev] i32 [Qt], qn_out [dev] f32 [Qt][Dt], cols_image [dev] qpg_hl_cols_bytes(Qt, Dt) bytes.  Dt %% 128 == 0. */
int qpg_clip_pack_hl(qpg_ctx*, void* stream, const float* qbase, int M, int T, int F, const int32_t* q_win,
                     const int32_t* q_t, int Q, int n_taps, int tap_stride, float* q32, double* qn2, void* image,
                     int64_t image_bytes, const float* text_ctx, int Mt, int R, int Dt, const int32_t* tq_win,
                     const int32_t* tq_row, int Qt, float* qn_out, void* cols_image, int64_t cols_bytes);
int qpg_audio_cosine_hl(qpg_ctx*, void* stream, const void* db_image, int N, int F, int G, const double* cn2,
                        const void* q_image, const double* qn2, int Q, void* D, int d_is_f32, int64_t ldD, int32_t* stats);
/* The same sweep over a track stored in IEEE f16 (round 5; BASELINE.json configs[4] "fp16 features"): an f16 value is
 * its own h plane, so the database image has ONE plane, no scale exponent and no representation error - half the bytes
 * (N x 27 rows x 3 F f16, every byte read once) and two products per element (h h', l' h) instead of three.  Same query
 * image (qpg_audio_pack_queries_hl / qpg_clip_pack_hl), same D, same bound QPG_AUDIO_HL_ERR on the ROUNDED track's exact
 * distances (cn2 = its candidates' squared norms), same select.  Reference data path: data_processing.py:255-274 (the
 * track), GestureKNN.py:666-691 (the scan).
 *   qpg_audio_hl1_supported  qpg_audio_hl_supported's grid and 3 F / 32 a multiple of 12 (F %% 128 == 0).
 *   qpg_audio_hl1_pack_db    one-off: base_f16 [dev] f16 [N][T][F] -> image [dev] (qpg_audio_hl1_db_bytes(N, F) bytes). */
int qpg_audio_hl1_supported(int T, int F, int G, int n_taps, int tap_stride, int cand_step);
int64_t qpg_audio_hl1_db_bytes(int N, int F);
int qpg_audio_hl1_pack_db(qpg_ctx*, void* stream, const void* base_f16, int N, int T, int F, int G, int n_taps,
                          int tap_stride, int cand_step, void* image, int64_t image_bytes);
int qpg_audio_cosine_hl1(qpg_ctx*, void* stream, const void* db_image, int N, int F, int G, const double* cn2,
                         const void* q_image, const double* qn2, int Q, void* D, int d_is_f32, int64_t ldD, int32_t* stats);
/* Hardware probe behind the bound's one measured constant: out[tile] = A[tile] (16 x 32 f16) . B[tile]^T (16 x 32 f16)
 * + C[tile] (16 x 16 f32, NULL = 0) exactly as ONE v_mfma_f32_16x16x32_f16 computes it; the tests compare it with exact
 * sums (kappa: error of a 32-product block sum in units of 2^-24 sum |products|). */
int qpg_probe_mfma_f16_tile(qpg_ctx*, void* stream, const void* a, const void* b, const float* c, int tiles, float* out);

/* One-off DB preparation for the text sweep: sklearn-normalise the grid rows x[j][cand_r[g]] (bit-exact,
 * as qpg_l2_normalize_rows_f32) and store them tiled for lane-per-candidate access:
 *   xt[c/64][e/4][c%64][e%4],  c = j*G + g   (a wave's 64 lanes read 64 consecutive 16-B pieces).
 * x: [dev] f32 [N][R][Dm]; xt: [dev] f32 [ceil(N*G/64)*64 * Dm]. */
int qpg_text_pack_candidates_f32(qpg_ctx*, void* stream, const float* x, int N, int R, int Dm,
                                 const int32_t* cand_r, int G, float* xt);

/* Cosine distance with scikit-learn's float32 arithmetic, bit-exact (GestureKNN.py:716 keeps f32):
 *   D[q][c] = 0.5 * einsum_sq(qn[q] - xn_c)     (NumPy einsum summation order)
 * xt: tiled normalised candidates from qpg_text_pack_candidates_f32 (C = N*G of them);
 * qn: [dev] f32 [Q][Dm] queries normalised by qpg_l2_normalize_rows_f32; D: [dev] f32 [Q][C]. */
int qpg_text_cosine_f32(qpg_ctx*, void* stream, const float* xt, int64_t C, int Dm, const float* qn, int Q,
                        float* D, int64_t ldD);

/* The same sweep with the per-code minimum FUSED into it (large query counts, BASELINE.json configs[2]: 1 000 queries x
 * 100 000 candidates): the Q x C distance matrix never reaches HBM.  Distances are bit-identical to
 * qpg_text_cosine_f32's; winner per (query, code) = minimum distance, lowest candidate index among equals.
 * cand_code: [dev] i16 [C] code of candidate c (outside [0,K): skipped, e.g. masked rows); tiles_per_chunk: 64-candidate
 * tiles per block (work granularity); `ws`: scratch of `qpg_text_percode_ws_bytes` bytes (one packed [Q][K] table);
 * out_dist [dev] f32 [Q][K] (`absent` where a code has no candidate), out_idx [dev] i32 [Q][K] = c + idx_base or -1,
 * out_rank optional [dev] i16 [Q][K] stable ranks; out_nn optional [dev] i32 [Q] the query's global nearest
 * neighbour over all codes (candidate index, -1 if there is no valid candidate).  K <= 1024. */
int64_t qpg_text_percode_ws_bytes(int64_t C, int Q, int K, int tiles_per_chunk);
int qpg_text_percode_f32(qpg_ctx*, void* stream, const float* xt, int64_t C, int Dm, const int16_t* cand_code, int K,
                         const float* qn, int Q, int tiles_per_chunk, int32_t idx_base, float absent, void* ws,
                         int64_t ws_bytes, float* out_dist, int32_t* out_idx, int16_t* out_rank, int32_t* out_nn);
/* fp16-STORAGE variant of the pair above (BASELINE.json "fp16 features"): qpg_text_pack_candidates_f16 stores the grid
 * rows ROUNDED to IEEE f16 (xh: [dev] f16, ceil(N*G/64)*64*Dm elements, tiled [c/64][e/8][c%64][e%8], 16-byte aligned)
 * and their sklearn norms computed from the rounded values (nrm: [dev] f32 [N*G]); qpg_text_percode_f16 widens,
 * normalises with sklearn's division and continues with the same f32 arithmetic: its tables are the reference's on
 * the f16-rounded database, bit for bit.  Half the HBM bytes of the dominant array; the sweep stays VALU-bound. */
int qpg_text_pack_candidates_f16(qpg_ctx*, void* stream, const float* x, int N, int R, int Dm, const int32_t* cand_r,
                                 int G, void* xh, float* nrm);
int qpg_text_percode_f16(qpg_ctx*, void* stream, const void* xh, const float* nrm, int64_t C, int Dm,
                         const int16_t* cand_code, int K, const float* qn, int Q, int32_t idx_base, float absent,
                         void* ws, int64_t ws_bytes, float* out_dist, int32_t* out_idx, int16_t* out_rank,
                         int32_t* out_nn);

/* BOUNDED PREFILTER + EXACT REFINE for the exact-f32 cosine family (round 3; BASELINE.json configs[2] and the matcher's
 * text side, GestureKNN.py:708-721): bit-identical tables to qpg_text_percode_f32 at a fraction of its VALU work.  Build
 * side (host, once; qpgesture_amd/sorted_rows.py): sklearn-normalise the rows (qpg_l2_normalize_rows_f32), drop masked rows,
 * keep per code only the FIRST of the rows the normalisation left at zero (zero_row), sort the rest by code (stable) into
 * segments padded to 16 rows with copies of the segment's first row, R %% 32 == 0.
 *   qpg_hl_pack_rows / qpg_hl_pack_cols   split-f16 fragment images of the sorted unit rows xs [dev] f32 [R][D] and of the
 *       normalised queries qn [dev] f32 [Q][D] (chunks of 96); sizes: qpg_hl_rows_bytes / qpg_hl_cols_bytes.  D %% 128 == 0.
 *   qpg_hl_gemm_distance   Dm [dev] f32 [Q][ldD >= R]: Dm[q][r] = 1 - <xs[r], qn[q]> on the f16 matrix cores (the kernel of
 *       qpg_audio_cosine_hl with a plain epilogue; |Dm - true| <= QPG_AUDIO_HL_ERR for unit-norm operands); tile_min
 *       (optional) [dev] f32 [Q][ldT >= R / 16]: the minimum of every 16-row tile, what the select reads first.
 *   qpg_percode_select_sorted_f32   per query: per-code minimum of Dm from the tile minima, every row within `band` of it
 *       evaluated in sklearn's exact f32 order (0.5 * einsum_sq(qn - xs)), minimum exact distance and lowest ORIGINAL index
 *       per code.  row_code [dev] i16 [R]: code of sorted row r (K <= 2048), | 0x4000 for padding rows, 0x1fff for the tail
 *       past the last segment; row_index [dev] i32 [R]: original index; zero_row [dev] i32 [K] or NULL: original index of the
 *       code's first all-zero row (-1: none) - it enters with the prefilter value 0.5; xs holds R + 1 rows, row R zeros;
 *       code_tile [dev] i32 [K + 1]: first 16-row tile of every code's segment (8 blocks per query take a range of codes
 *       each; ranks / nearest neighbours: a second small launch).  A band list is 2048 rows / 1024 opened tiles per block.
 *       `band` >= 2 x (prefilter error + sklearn's own rounding against the real value): derivation in
 *       csrc/qpg_sorted.hip (8.6e-5 at D = 512).  stats[1] |= 16 if a band list overflowed (its own bit, round 4): run
 *       qpg_text_percode_f32 instead.  out_rank optional: i16 [Q][K] ranks of the table rows (qpg_rank_rows_f32's);
 *       out_nn optional: the query's global nearest neighbour (original index).  idx_base is added to every index written;
 *       q_block / block_stride: the exchange layout of qpg_percode_select_f32 (row shards; no ranks / nn then). */
int64_t qpg_hl_rows_bytes(int64_t R, int D);
int64_t qpg_hl_cols_bytes(int Q, int D);
int qpg_hl_pack_rows(qpg_ctx*, void* stream, const float* xs, int64_t R, int D, void* image, int64_t image_bytes);
int qpg_hl_pack_cols(qpg_ctx*, void* stream, const float* qn, int Q, int D, void* image, int64_t image_bytes);
/* Round 6: the query side of a prefilter batch in ONE launch - qpg_l2_normalize_rows_f32 (sklearn's normalize, bit for bit)
 * + qpg_hl_pack_cols + qpg_perm32_rows_f32 on the RAW queries q [dev] f32 [Q][D]: qn_out (optional) [dev] f32 [Q][D] the
 * normalised rows, cols_image qpg_hl_cols_bytes(Q, D) bytes, qperm_out (optional) [dev] f32 [Q][D] the chain-permuted
 * normalised rows qpg_percode_select_bycode_f32 reads.  Outputs bit-identical to the three calls.  D %% 128 == 0. */
int qpg_hl_prepare_queries(qpg_ctx*, void* stream, const float* q, int Q, int D, float* qn_out, void* cols_image,
                           int64_t image_bytes, float* qperm_out);
int qpg_hl_gemm_distance(qpg_ctx*, void* stream, const void* rows_image, int64_t R, int D, const void* cols_image, int Q,
                         float* Dm, int64_t ldD, float* tile_min, int64_t ldT);
/* Round 4: the same GEMM WITHOUT its matrix - per (query, 16-row tile) the minimum and a 16-bit mask of the rows within
 * `band` of it (tile_min f32 / tile_mask u16 [dev] [Q][ldT >= R/16]).  A tile is opened by the select only if its minimum is
 * within the band of its code's minimum, and code minimum <= tile minimum, so the masked rows are a superset of the band's
 * rows: qpg_percode_select_sorted_f32 with `tile_mask` never reads Dm (may be NULL).  6 bytes per (query, tile) instead of
 * 64: cfg-3's 375 MB prefilter matrix is neither written nor read. */
int qpg_hl_gemm_tilemin(qpg_ctx*, void* stream, const void* rows_image, int64_t R, int D, const void* cols_image, int Q,
                        float band, float* tile_min, uint16_t* tile_mask, int64_t ldT);
/* Round 5, many queries per batch (BASELINE.json configs[2]): the prefilter on the h PLANES ALONE and the exact-order
 * evaluations BY CODE.
 *   qpg_hl_gemm_tilemin_h   one f16 MFMA per 16 x 16 x 32 block (the cross terms h l' + l h' and l l' are dropped: at most
 *       (2^-10 + 2^-22) |x||q| by Cauchy-Schwarz - `band` must cover it: sorted_rows.gemm_h_err), 64-row wave tiles; writes
 *       tile minima and row masks TILE-MAJOR: tile_min f32 / tile_mask u16 [dev] [R / 16][ldQ >= Q].  R %% 64 == 0,
 *       D %% 256 == 0.  Same images as qpg_hl_gemm_tilemin (the l planes are not read).
 *   qpg_perm32_rows_f32     y[r][32 G + 8 k + j] = x[r][16 (2 G + (j >> 2)) + 4 (3 - (j & 3)) + k]: inside every 32-element
 *       group (one 128-byte line) the four einsum chains of sklearn's f32 sum of squares become four 32-byte runs in visiting
 *       order (rows once at build time, the queries per batch).  D %% 32 == 0.
 *   qpg_percode_select_bycode_f32   the tables of qpg_percode_select_sorted_f32 (bit-identical), one block per (code, query
 *       range): the code's minimum per query from the tile-major minima, the opened tiles' masked rows listed as (query,
 *       row) pairs, every pair evaluated in the exact order by four lanes (one per chain) from the chain-permuted rows
 *       xs_perm [R + 1][D] / queries qn_perm [Q][D] - a segment's rows are fetched from HBM once for all queries (the by-query
 *       kernel gathers 2 KB per pair: 0.95 GB per 1 000 queries of cfg-3).  A block = one code, its rows staged through LDS tile by tile; a tile's pair list holds 1 024 pairs;
 *       overflow: stats[1] |= 16.  No exchange layout (q_block): single-GPU batches and per-shard tables only. */
int qpg_hl_gemm_tilemin_h(qpg_ctx*, void* stream, const void* rows_image, int64_t R, int D, const void* cols_image, int Q,
                          float band, float* tile_min_t, uint16_t* tile_mask_t, int64_t ldQ);
int qpg_perm32_rows_f32(qpg_ctx*, void* stream, const float* x, int64_t R, int D, float* y);
int qpg_percode_select_bycode_f32(qpg_ctx*, void* stream, const float* tile_min_t, const uint16_t* tile_mask_t, int64_t ldQ,
                                  int Q, int64_t R, const int16_t* row_code, const int32_t* row_index,
                                  const int32_t* zero_row, const int32_t* code_tile, int K, float band,
                                  const float* qn_perm, const float* xs_perm, int D, float absent, float* out_dist,
                                  int32_t* out_idx, int16_t* out_rank, int32_t* out_nn, int32_t* stats, int32_t idx_base);
int qpg_percode_select_sorted_f32(qpg_ctx*, void* stream, const float* Dm, int64_t ldD, const float* tile_min,
                                  const uint16_t* tile_mask /* or NULL: rows are taken from Dm */, int64_t ldT,
                                  int Q, int64_t R, const int16_t* row_code, const int32_t* row_index,
                                  const int32_t* zero_row, const int32_t* code_tile, int K, float band, const float* qn,
                                  const float* xs, int D, float absent, float* out_dist, int32_t* out_idx,
                                  int16_t* out_rank, int32_t* out_nn, int32_t* stats, int32_t idx_base, int q_block,
                                  int64_t block_stride);

/* vq-wav2vec audio sweep (the mode the paper describes; flags use_wavvq/use_feature of GestureKNN.py:557-560):
 * D[q][c] = Levenshtein distance (unit costs, python-Levenshtein distance()) between the 11-symbol strings of
 * query q and candidate c, symbol = g1*320+g2 (wavvq_distances(mode='combine'), GestureKNN.py:57-67).  Strings
 * are gathered in place from symbol tracks: string[i] = track[t + tap_off[i]] (0 outside the window), which is
 * the 6-back/5-forward stack of data_processing.py:297-335 without materialising it.
 * sym_db: [dev] i32 [N][T]; cand_t: [dev] i32 [G] (int(k) of the float grid); tap_off: HOST i32 [11];
 * sym_q: [dev] i32 [Mq][Tq]; q_win/q_t: [dev] i32 [Q]; D: [dev] f32 [Q][N*G] (exact small integers). */
int qpg_wavvq_lev_f32(qpg_ctx*, void* stream, const int32_t* sym_db, int N, int T, const int32_t* cand_t, int G,
                      const int32_t* tap_off, int n_taps, const int32_t* sym_q, int Mq, int Tq,
                      const int32_t* q_win, const int32_t* q_t, int Q, float* D, int64_t ldD);

/* Segmented min + argmin by code id in one launch: per query row the per-code minimum, its
 * first-wins candidate (lowest index among equal distances == the strict `<` scan of GestureKNN.py:686-689, 717-720),
 * `absent` / -1 for codes with no candidate, and optionally the stable ranks of the 512 minima.
 * D: [dev] [Q][ldD] distances; cand_code: [dev] i16 [C] code of candidate c (values outside [0,K) are skipped);
 * out_dist [Q][K], out_idx i32 [Q][K] = c + idx_base, out_rank i16 [Q][K] or NULL.  K <= 2048.
 * q_block > 0 (sharded database): the outputs are written in EXCHANGE layout - row q goes to row q % q_block of block
 * q / q_block, blocks `block_stride` BYTES apart starting at out_dist / out_idx (one block per destination rank of the
 * all-to-all, several arrays per block); out_rank must then be NULL (ranks are taken after the merge). */
int qpg_percode_select_f64(qpg_ctx*, void* stream, const double* D, int64_t ldD, int Q, const int16_t* cand_code,
                           int64_t C, int K, double absent, int32_t idx_base, double* out_dist, int32_t* out_idx,
                           int16_t* out_rank, int q_block, int64_t block_stride);
int qpg_percode_select_f32(qpg_ctx*, void* stream, const float* D, int64_t ldD, int Q, const int16_t* cand_code,
                           int64_t C, int K, float absent, int32_t idx_base, float* out_dist, int32_t* out_idx,
                           int16_t* out_rank, int q_block, int64_t block_stride);

/* qpg_percode_select_f64 with the NEAR-TIE GUARD (SURVEY.md §7 hard part 2).  The sweep's distances (dot-product
 * form on the f64 matrix cores) and the reference's (sklearn: 0.5*|q/|q| - c/|c||^2, NumPy einsum order,
 * GestureKNN.py:685) agree to ~1e-16 but can order two distances closer than that differently.  Candidates within `eps`
 * of their code's minimum (two or more) are re-evaluated in the reference's exact arithmetic and the winner taken by
 * (reference distance, index); minima of different codes within `eps` of each other are replaced by the
 * reference-arithmetic value of their winner before ranking.  All inside the launch; nothing is flagged on ordinary data.
 * base [dev] f32 [N][T][F], cand_t [dev] i32 [G], q32 [dev] f32 [Q][n_taps*F]: the sweep's own operands (C == N*G);
 * stats [dev] i32 [4]: [0] += re-evaluated (query, candidate) pairs, [1] |= 1 if more than 256 were flagged in one row
 * (the surplus keeps its sweep value). */
int qpg_percode_select_guarded_f64(qpg_ctx*, void* stream, const double* D, int64_t ldD, int Q, const int16_t* cand_code,
                                   int64_t C, int K, double absent, int32_t idx_base, double* out_dist, int32_t* out_idx,
                                   int16_t* out_rank, int q_block, int64_t block_stride, const float* base, int T, int F,
                                   const int32_t* cand_t, int G, int n_taps, int tap_stride, const float* q32, double eps,
                                   int32_t* stats, int base_is_f16);
/* base_is_f16 != 0 (here and in qpg_percode_select_mixed_f64): `base` points at an IEEE f16 track (as swept by
 * qpg_audio_cosine_f64_h / qpg_audio_cosine_mx_h); the re-evaluation then runs on the widened, i.e. rounded, values. */

/* UNCAPPED form of the guarded select (round 3): same arguments, results and arithmetic, but every list lives in `ws`
 * ([dev] qpg_percode_select_exact_ws_bytes(Q, C, K) bytes, 16-byte aligned; sized for ALL C candidates of a row inside
 * one band), so no population of near-ties can overflow it and stats[1] is never raised: the reference's scan
 * (GestureKNN.py:685-689) visits every candidate with a strict `<` and has no cap either.  This is the path the host
 * re-matches a clip on when a capped select or the mixed-precision sweep raised stats[1].  Five launches; K <= 1024. */
int64_t qpg_percode_select_exact_ws_bytes(int Q, int64_t C, int K);
int qpg_percode_select_exact_f64(qpg_ctx*, void* stream, const double* D, int64_t ldD, int Q, const int16_t* cand_code,
                                 int64_t C, int K, double absent, int32_t idx_base, double* out_dist, int32_t* out_idx,
                                 int16_t* out_rank, int q_block, int64_t block_stride, const float* base, int T, int F,
                                 const int32_t* cand_t, int G, int n_taps, int tap_stride, const float* q32, double eps,
                                 int32_t* stats, int base_is_f16, void* ws, int64_t ws_bytes);

/* Select for the matrix of qpg_audio_cosine_mx.  Same outputs as qpg_percode_select_guarded_f64.  Two sweep values
 * further apart than eps1 (>= 2 x QPG_AUDIO_MX_ERR) are ordered like the exact distances; inside that band
 *   tier 1: all candidates within eps1 of their code's minimum (if two or more) and the winners of codes whose minima
 *           are rank neighbours within eps1 get an f64 dot product (error ~1e-15, like qpg_audio_cosine_f64);
 *   tier 2: on those values, the near-tie guard above with band eps2 (the reference's own arithmetic; 0 = off).
 * out_dist holds the sweep value (error <= QPG_AUDIO_MX_ERR) for untouched codes, the re-evaluated one otherwise.
 * Pass out_rank: without it the minima of different codes are not protected against each other.
 * qn2 [dev] f64 [Q], cn2 [dev] f64 [C]: the sweep's squared norms.  stats [dev] i32 [4]: [0] += tier-2 pairs,
 * [1] |= 1 if a list overflowed (2048 tier-1 / 256 tier-2 entries per query; results then unguarded), [2] += tier-1
 * pairs.  K <= 512. */
int qpg_percode_select_mixed_f64(qpg_ctx*, void* stream, const void* D, int d_is_f32, int64_t ldD, int Q, const int16_t* cand_code,
                                 int64_t C, int K, double absent, int32_t idx_base, double* out_dist, int32_t* out_idx,
                                 int16_t* out_rank, int q_block, int64_t block_stride, const float* base, int T, int F,
                                 const int32_t* cand_t, int G, int n_taps, int tap_stride, const float* q32,
                                 const double* qn2, const double* cn2, double eps1, double eps2, int32_t* stats,
                                 void* ws, int64_t ws_bytes, int base_is_f16);
/* ws: [dev] scratch of qpg_percode_select_mixed_ws_bytes(Q, K) bytes, 16-byte aligned, ZERO-FILLED ONCE by the caller
 * (part of it is state the launches leave all-zero for the next call; a workspace may be reused for any smaller Q) - with
 * it the call is four launches (the row streamed by 8 blocks per query | lists | tier-1 dot products, 256 waves per query
 * | merge, tier 2, ranks; an f64 matrix: three, the lists' launch streams the row itself); NULL: one launch, each query's
 * work on its own CU.  qpg_percode_select_mixed_ws_stride(K): bytes per query - query q's tier-1 list length is the i32
 * at q x stride + 24 K (diagnostics). */
int64_t qpg_percode_select_mixed_ws_bytes(int Q, int K);
int64_t qpg_percode_select_mixed_ws_stride(int K);
/* The same call with the WALK-RELEVANCE CUT of its re-evaluation lists (f32 matrix + workspace + out_rank, no block layout).
 * The walk reads, per step and previous code p, only the code(s) with the smallest fused score
 *     pos_rank[p][c] + 0.05 freq_rank[c] + rank(c)          (GestureKNN.py:540-545, :574-576; order[0], order[:2] at :593)
 * so a code whose rank is CERTAINLY above every previous code's winning score can never be read: neither its rank among
 * near-tied neighbours nor its winner among near-tied candidates is settled in f64 (unless a code that can be read lies
 * within eps1 of it).  Such a code's table entries keep the sweep's values (|value - exact| <= eps1 / 2.1), its rank is its
 * position among them; every entry the walk can read - and therefore every selected code index - is what
 * qpg_percode_select_mixed_f64 returns.  pos_rank_t [dev] i16 [K][K]: the TRANSPOSE of qpg_match_steps' pos_rank
 * (pos_rank_t[c * K + p]); freq_rank [dev] i16 [K]; top_n: 1 with the text side, 2 without; probe: best-ranked codes the
 * bound on the winning score is taken over (0: 64). */
int qpg_percode_select_mixed_f64_cut(qpg_ctx*, void* stream, const void* D, int d_is_f32, int64_t ldD, int Q,
                                     const int16_t* cand_code, int64_t C, int K, double absent, int32_t idx_base,
                                     double* out_dist, int32_t* out_idx, int16_t* out_rank, int q_block, int64_t block_stride,
                                     const float* base, int T, int F, const int32_t* cand_t, int G, int n_taps,
                                     int tap_stride, const float* q32, const double* qn2, const double* cn2, double eps1,
                                     double eps2, int32_t* stats, void* ws, int64_t ws_bytes, int base_is_f16,
                                     const int16_t* pos_rank_t, const int16_t* freq_rank, int top_n, int probe);

/* Cross-shard merge when the shards swept with qpg_audio_cosine_mx (their tables are accurate to QPG_AUDIO_MX_ERR; each
 * shard's own select has settled the near-ties inside the shard).  Three steps around two more byte exchanges:
 *   qpg_merge_mixed_phase1_f64  (owner of a query block)  approximate merge of the W received tables (same layout
 *       arguments as qpg_merge_select_f64); every shard within eps1 of a code's merged minimum — when there are two or
 *       more — and the winners of codes whose merged minima are rank neighbours within eps1 become requests to the
 *       shard holding the candidate.  req: [dev] W blocks of req_stride bytes, block w = [i64 count][R x u64
 *       (q_local << 48 | code << 32 | global candidate)]: the send buffer of an all-to-all.  ws: [dev] scratch of
 *       qpg_merge_mixed_ws_bytes(Q, K, fl_cap) bytes, kept for phase 2 (fl_cap = flagged (code, shard) entries per
 *       query).  stats[1] |= 4 on request / flag-list overflow: every written request is valid and every counted flag
 *       entry written, the surplus is dropped and the clip must be re-matched (R = Q*K and fl_cap = K*W cannot overflow).
 *   qpg_shard_refine_f64        (every shard)  req_recv = the W request blocks received (block o from owner o); the
 *       exact f64 distance of every requested (query o*q_stride + q_local, candidate - cand_base) goes to resp block o
 *       ([R x f64], resp_stride bytes apart): the send buffer of the answering all-to-all.  reference_arithmetic != 0:
 *       the distance in the reference's own arithmetic instead (sklearn normalise + einsum-order sum, as tier 2).
 *   qpg_merge_mixed_phase2_f64  (owner)  winners among the re-evaluated contenders by (exact value, candidate), stable
 *       ranks over re-evaluated and untouched minima.  stats[3] += re-evaluated entries.  eps2 > 0: contenders of one
 *       code, or minima of rank-neighbour codes, closer than eps2 raise stats[1] |= 8 (dot-product responses cannot
 *       order them like the reference; the host re-matches the clip with reference-arithmetic responses and eps2 = 0).
 * The same three calls with eps1 = the near-tie band (1e-12), reference_arithmetic = 1 and eps2 = 0 are the cross-shard
 * TIER 2: the uncapped sharded path (CodeKNN, audio_precision "exact"). */
int64_t qpg_merge_mixed_ws_bytes(int Q, int K, int fl_cap);
/* Round 4: request slots are DETERMINISTIC - (query q, code k) -> shard w sits at slot q * (R / Q) + its position among q's
 * requests to w in code order, unused slots hold ~0 (R % Q == 0) - so every rank that merges the same tables builds the
 * same request blocks and, in the all-gather form, a shard refines its own block of its own run: no request exchange.
 * Block headers (8 bytes in front of the R slots of a request / response block): [i32 count (diagnostics) | i32 trouble
 * bits].  The bits TRAVEL WITH THE EXCHANGES: phase 1 ORs the W table blocks' flag words (at `flag_off` inside a table
 * block, < 0: none) into stats[1] and seeds the request headers; the shard refine ORs the received request headers into
 * stats[1] (stats may be NULL) and seeds the response headers; phase 2 ORs those in.  resp_stride >= 8 + 8 R. */
int qpg_merge_mixed_phase1_f64(qpg_ctx*, void* stream, const void* recv, int W, int64_t src_stride, int64_t dist_off,
                               int64_t idx_off, int Q, int K, double absent, double eps1, int R, void* req,
                               int64_t req_stride, void* ws, int64_t ws_bytes, int32_t* stats, int fl_cap,
                               int64_t flag_off);
int qpg_shard_refine_f64(qpg_ctx*, void* stream, const void* req_recv, int W, int64_t req_stride, int R, int q_stride,
                         int64_t cand_base, const float* base, int base_is_f16, int T, int F, const int32_t* cand_t, int G,
                         int n_taps, int tap_stride, const float* q32, const double* qn2, const double* cn2, void* resp,
                         int64_t resp_stride, int reference_arithmetic, int32_t* stats, int Rq);
/* The trouble word riding in an exchanged buffer: stamp = every one of nblk blocks' i32 at `off` := stats[1] (sender,
 * before the exchange); gather = stats[1] |= OR of the nblk received words (receiver). */
int qpg_flags_stamp(qpg_ctx*, void* stream, void* buf, int nblk, int64_t stride, int64_t off, const int32_t* stats);
int qpg_flags_gather(qpg_ctx*, void* stream, const void* buf, int nblk, int64_t stride, int64_t off, int32_t* stats);
int qpg_merge_mixed_phase2_f64(qpg_ctx*, void* stream, const void* recv, int W, int64_t src_stride, int64_t idx_off, int Q,
                               int K, double absent, const void* ws, int64_t ws_bytes, const void* resp_recv,
                               int64_t resp_stride, double* out_dist, int32_t* out_idx, int16_t* out_rank, int32_t* stats,
                               int fl_cap, double eps2);

/* Cross-shard min + index merge after the RCCL exchange (the all-reduce(min, index) `north_star`
 * names, as all-gather / all-to-all + this kernel): source w's tables start at recv + w*src_stride (+ dist_off for the
 * [Q][K] distances, + idx_off for the [Q][K] i32 global candidate indices, -1 = absent in that shard).  Winner per
 * (query, code): minimum distance, lowest index among equals.  out_rank (optional): stable ranks of the merged row.
 * _f64 only: eps2 > 0 with stats != NULL raises stats[1] |= 8 when two shards' minima of one code, or the merged minima
 * of rank-neighbour codes, are closer than eps2 (the shards settle near-ties inside themselves only; the host then
 * re-matches the clip on the uncapped path, whose merge compares such pairs in the reference's own arithmetic). */
int qpg_merge_select_f64(qpg_ctx*, void* stream, const void* recv, int W, int64_t src_stride, int64_t dist_off,
                         int64_t idx_off, int Q, int K, double absent, double* out_dist, int32_t* out_idx,
                         int16_t* out_rank, double eps2, int32_t* stats);
int qpg_merge_select_f32(qpg_ctx*, void* stream, const void* recv, int W, int64_t src_stride, int64_t dist_off,
                         int64_t idx_off, int Q, int K, float absent, float* out_dist, int32_t* out_idx,
                         int16_t* out_rank);

/* Stable ranks of each row: rank[q][c] = #{c' : d[c'] < d[c] or (d[c'] == d[c] and c' < c)}
 * (== np.argsort(kind='stable').argsort(); the reference calls the unstable default,
 * GestureKNN.py:553,574).  out: [dev] i16 [Q][K]. */
int qpg_rank_rows_f64(qpg_ctx*, void* stream, const double* d, int Q, int K, int16_t* out);
int qpg_rank_rows_f32(qpg_ctx*, void* stream, const float* d, int Q, int K, int16_t* out);

/* ------------------------------------------------------------------------------------------
 * State-dependent tail of CodeKNN.search_code_knn + the window loop of predict_code_from_audio
 * (GestureKNN.py:501-664, 785-813).
 * ---------------------------------------------------------------------------------------- */

/* out[p][c] = |sig[p] - sig[c]|_2 in f32 (f64 accumulation, one rounding), +inf on the diagonal
 * (GestureKNN.py:531-536: `1e10000` for the current code).  sig: [dev] f32 [K][Dm]; out: [dev] f32 [K][K]. */
int qpg_l2_table_f32(qpg_ctx*, void* stream, const float* sig, int K, int Dm, float* out);

#define QPG_MODE_AUD_TXT 0 /* shipped flags: audio top-1 vs text top-1, phase gate (GestureKNN.py:627-657) */
#define QPG_MODE_AUD 1     /* audio only: top-2 audio candidates through the phase gate (:593-608)          */
#define QPG_MODE_TXT 2     /* text only: top-2 text candidates through the phase gate (:610-625)            */
#define QPG_MODE_SERIAL_WALK 0x100 /* OR-ed into mode: force the one-wave sequential walk (validation of the tabulated one) */
#define QPG_MODE_PREFUSED 0x200    /* OR-ed into QPG_MODE_AUD_TXT: gate_tables [0] / [1] were filled by qpg_fuse_best_ranked */

/* Walk all M windows x `steps` matching steps of a clip on the device.
 *   aud_rank/txt_rank: [dev] i16 [Q][K] stable ranks of the per-code minima (Q = M*steps);
 *   aud_idx/txt_idx:   [dev] i32 [Q][K] winning candidate index j*G+g (-1 = code absent);
 *   pos_rank:          [dev] i16 [K][K] stable ranks of qpg_l2_table_f32 rows;
 *   freq_rank:         [dev] i16 [K] rank of 1-count/total (GestureKNN.py:481-499, 544);
 *   code:              [dev] i32 [N][code_ld]; *_cidx [G]: code column of a grid position;
 *   *_pslot [G]:       phase start frame int(k/398*240) of a grid position (GestureKNN.py:632);
 *   phase:             [dev] f32 [N][Tp][2][8] (phase shift, amplitude);
 *   seed_code/seed_phase [dev f32 8x16]: init_code_phase() draw (GestureKNN.py:462-473);
 *   gate_tables:       [dev] i32 [3][Q][K] scratch: the two phase-gate candidates for every (step, previous
 *                      code), and the gate outcome for every (step, previous code, previous vote) — both
 *                      tabulated in parallel, so that the sequential part is Q dependent 2-byte lookups;
 *   out_codes [dev] i32 [M][30]; out_phase [dev] f32 [M][steps][8][16]; out_vote [dev] i32 [M][steps];
 *   out_status [dev] i32 [2]: [0] = 1 if a code absent from the DB won a rank fusion (the reference raises
 *   IndexError there, GestureKNN.py:631-632); [1] = *guard_flags (0 if NULL): the trouble word stats[1] of the
 *   sweeps / selects that produced the tables (list overflow, norms outside the error bound's range, cross-shard
 *   near tie), copied by the walk's last kernel so that it leaves the device in the same D2H copy as the codes —
 *   the host must not use codes whose status[1] != 0 (CodeKNN re-matches such a clip on the uncapped exact path).
 * combined = (pos_rank + freq_rank*0.05) + rank in float64 in that order; argmin = lowest index. */
int qpg_match_steps(qpg_ctx*, void* stream, const int16_t* aud_rank, const int32_t* aud_idx,
                    const int16_t* txt_rank, const int32_t* txt_idx, const int16_t* pos_rank,
                    const int16_t* freq_rank, const int32_t* code, int code_ld, const int32_t* aud_cidx,
                    const int32_t* aud_pslot, int Ga, const int32_t* txt_cidx, const int32_t* txt_pslot, int Gt,
                    const float* phase, int Tp, int mode, int M, int steps, int K, int seed_code,
                    const float* seed_phase, int32_t* gate_tables, int32_t* out_codes, float* out_phase,
                    int32_t* out_vote, int32_t* out_status, const int32_t* guard_flags);
/* The same for several INDEPENDENT clips (chains) of M windows each in one set of launches (BASELINE configs[4]: 16 clips
 * per sweep): the tables hold the chains' steps back to back ([n_chains x M x steps][K]); seed_codes [dev] i32 [n_chains],
 * seed_phase [dev] f32 [n_chains][8][16]; outputs [n_chains][...] in the single-clip shapes; chain c's status pair at
 * out_status + c x status_stride (>= 2).  gate_tables: 3 x n_chains x M x steps x K i32.  M > 0. */
int qpg_match_steps_batch(qpg_ctx*, void* stream, const int16_t* aud_rank, const int32_t* aud_idx, const int16_t* txt_rank,
                          const int32_t* txt_idx, const int16_t* pos_rank, const int16_t* freq_rank, const int32_t* code,
                          int code_ld, const int32_t* aud_cidx, const int32_t* aud_pslot, int Ga, const int32_t* txt_cidx,
                          const int32_t* txt_pslot, int Gt, const float* phase, int Tp, int mode, int M, int steps, int K,
                          int n_chains, const int32_t* seed_codes, const float* seed_phase, int32_t* gate_tables,
                          int32_t* out_codes, float* out_phase, int32_t* out_vote, int32_t* out_status,
                          int64_t status_stride, const int32_t* guard_flags);
/* One modality's half of the rank fusion in front of the walk (GestureKNN.py:540-545 + :574-576 for the audio order,
 * :553-555 for the text order - two independent argsorts): T[q][p] = idx[q][argmin_c (pos_rank[p][c] + 0.05 freq_rank[c])
 * + rank[q][c]] (f64, that order of operations; lowest code among equal scores).  rank [dev] i16 [Q][K] (a permutation per
 * row), idx [dev] i32 [Q][K], T [dev] i32 [Q][K]: region [0] (audio) or [1] (text) of the walk's gate_tables.  Launched
 * behind each modality's select on ITS stream, the walk (qpg_match_steps* with QPG_MODE_PREFUSED) then starts at the gate
 * table and the join of the two streams has half a rank fusion less in front of it.  K % 16 == 0, K <= 4096. */
int qpg_fuse_best_ranked(qpg_ctx*, void* stream, const int16_t* rank, const int32_t* idx, const int16_t* pos_rank,
                         const int16_t* freq_rank, int Q, int K, int32_t* T);
/* (From how many chains per launch qpg_match_steps_batch deduplicates the gate table by the previous step's winner - one
 * evaluation per DISTINCT winner instead of one per (previous code, vote) state; same table, bit for bit - is the context's
 * QPG_OPT_GATE_DEDUP_FROM_CHAINS: qpg_ctx_set_option.) */

/* ------------------------------------------------------------------------------------------
 * Library-owned collectives of the row-sharded matcher (round 5).  The reference has no
 * collective (its would-be sites are the dormant calls of codebook/models/bottleneck.py:45,75-77); the row shards and
 * their min + index exchange are this build's.  One RCCL communicator per (process, device), created from a 128-byte
 * unique id the host carries to the ranks once; every call below is stream-ordered on the `stream` it is handed and makes
 * no host round trip, so a sharded step - kernels and collectives - is capturable as ONE hipGraph.  RCCL is bound lazily
 * (dlopen): without it only these entry points fail.
 *   qpg_comm_unique_id        rank 0: id [host] >= 128 bytes (ncclGetUniqueId).
 *   qpg_comm_create           every rank, collectively: the id, its rank, the world size -> *out.
 *   qpg_comm_allgather        recv [dev] world x bytes: block w = rank w's send [dev] (bytes).
 *   qpg_comm_alltoall         send / recv [dev] world blocks of `bytes`: block w goes to / came from rank w (out of place).
 *   qpg_comm_allreduce_max_i32  in place MAX (the agreed trouble word of an all-to-all step).
 *   qpg_allreduce_min_u64     in place MIN of packed (order-preserving f32 distance key << 32 | global candidate index)
 *                             tables: the global per-code winner with first-wins ties (GestureKNN.py:686-689) in one
 *                             collective - for tables whose values decide every comparison (text f32, Levenshtein);
 *                             qpg_pack_min_u64 / qpg_unpack_min_u64 convert (dist f32, idx i32; -1 = absent <-> all ones). */
typedef struct qpg_comm qpg_comm;
int qpg_comm_unique_id(void* id, int64_t id_bytes);
int qpg_comm_create(qpg_ctx*, const void* id, int64_t id_bytes, int rank, int world, qpg_comm** out);
int qpg_comm_destroy(qpg_comm*);
int qpg_comm_allgather(qpg_ctx*, void* stream, qpg_comm*, const void* send, void* recv, int64_t bytes);
int qpg_comm_alltoall(qpg_ctx*, void* stream, qpg_comm*, const void* send, void* recv, int64_t bytes);
int qpg_comm_allreduce_max_i32(qpg_ctx*, void* stream, qpg_comm*, int32_t* buf, int64_t count);
int qpg_allreduce_min_u64(qpg_ctx*, void* stream, qpg_comm*, uint64_t* buf, int64_t count);
int qpg_pack_min_u64(qpg_ctx*, void* stream, const float* dist, const int32_t* idx, int64_t n, uint64_t* packed);
int qpg_unpack_min_u64(qpg_ctx*, void* stream, const uint64_t* packed, int64_t n, float absent, float* dist, int32_t* idx);

/* ------------------------------------------------------------------------------------------
 * Gesture VQ-VAE (codebook/models/{vqvae,encdec,resnet,bottleneck}.py).  Activations are channels-last
 * [B][T][C] f32 — the layout of the pose tensors at VQVAE.encode/decode's API (vqvae.py:132-136 permutes
 * to NCT and back; here nothing is permuted).
 * ---------------------------------------------------------------------------------------- */

/* One 1-D convolution as an implicit GEMM on the f32 matrix cores (exact f32 FMA chains):
 *   y[b][t*out_stride+out_offset][co] = act_out( bias[co] + sum_{tap,ci} w[tap][ci][co] *
 *                                         act_in(x[b][t*in_stride + in_offset + tap*dil][ci]) ) (+ residual)
 * for t < T_out; input rows outside [0,T_in) are zero padding.  Serves nn.Conv1d(k4,s2,p1) (encdec.py:20),
 * the dilated k3 and 1x1 convolutions of ResConv1DBlock with its ReLUs and residual (resnet.py:31-46),
 * nn.Conv1d(k3) (encdec.py:24,39,113), the x.k^T GEMM of BottleneckBlock.quantise (bottleneck.py:123) and
 * nn.ConvTranspose1d(k4,s2,p1) (encdec.py:45) as two 2-tap launches, one per output parity.
 * w: [dev] f32 [taps][Cin_pad][Cout_pad] repacked weights (Cin_pad % 16 == 0, Cout_pad % 128 == 0, zero padded);
 * bias: [dev] f32 [Cout_pad] or NULL; residual: indexed like y, or NULL; y: [dev] f32 [B][T_y][Cout].
 * ws / ws_floats: optional [dev] f32 scratch; when given and the launch would not fill the chip (short sequences),
 * the contraction is split over up to 8 block groups whose partial sums are added in a fixed order. */
int qpg_conv1d_f32(qpg_ctx*, void* stream, const float* x, int B, int T_in, int Cin, const float* w,
                   const float* bias, int taps, int Cin_pad, int Cout, int Cout_pad, int in_stride, int in_offset,
                   int dil, int T_out, int out_stride, int out_offset, int T_y, const float* residual, int relu_in,
                   int relu_out, float* y, float* ws, int64_t ws_floats);

/* The same convolution in the TRANSPOSED formulation (csrc/qpg_convt.hip): y^T = W^T . x^T on v_mfma_f32_16x16x4_f32,
 * a wave owns 16 positions, activations go straight from the channels-last rows to registers, the weights stream
 * through LDS by LDS-DMA from a pre-packed image.  Same argument meaning as qpg_conv1d_f32 except:
 *   x rows have pitch Cx floats (Cx % 4 == 0, Cx >= Cin_pad, 16-byte aligned base): pad 135-channel pose rows
 *   with qpg_pad_channels_f32 first;
 *   wt: [dev] f32 T-pack  wt[nb][kb][g][nl][j] = W[k = 16 kb + 4 g + j][n = 128 nb + nl],  k = tap*Cin_pad + ci,
 *   zero padded, (taps*Cin_pad) % 64 == 0, Cout_pad % 128 == 0. */
int qpg_convt_f32(qpg_ctx*, void* stream, const float* x, int B, int T_in, int Cx, const float* wt, const float* bias,
                  int taps, int Cin_pad, int Cout, int Cout_pad, int in_stride, int in_offset, int dil, int T_out,
                  int out_stride, int out_offset, int T_y, const float* residual, int relu_in, int relu_out, float* y);
/* Two qpg_convt_f32 convolutions of the SAME input and shape whose outputs interleave - the even / odd output frames
 * of ConvTranspose1d(k4, s2, p1) (encdec.py:112-115: y[2m] = x[m-1].W3 + x[m].W1, y[2m+1] = x[m].W2 + x[m+1].W0) -
 * issued together: one launch on short sequences (a clip's decode), two otherwise.  (wt0, bias0, in_offset0,
 * out_offset0) and (wt1, ...) are the two halves; everything else as in qpg_convt_f32, no residual, no ReLU. */
int qpg_convt_pair_f32(qpg_ctx*, void* stream, const float* x, int B, int T_in, int Cx, const float* wt0,
                       const float* bias0, int in_offset0, int out_offset0, const float* wt1, const float* bias1,
                       int in_offset1, int out_offset1, int taps, int Cin_pad, int Cout, int Cout_pad, int in_stride,
                       int dil, int T_out, int out_stride, int T_y, float* y);
/* T-pack of a convolution's weights on the device: w [dev] f32 [taps x Cin_pad][Cout_pad] (qpg_conv1d_f32's layout, the one
 * the optimiser updates) -> out [dev] f32, the wt of qpg_convt_f32 (nb = 128) or one half of qpg_resblock_f32's wpack
 * (nb = 512 for the dilated convolution): taps x Cin_pad x (Cout_pad rounded up to nb) floats. */
int qpg_tpack_f32(qpg_ctx*, void* stream, const float* w, int taps, int Cin_pad, int Cout_pad, int nb, float* out);
/* y[r][0..Cp) = x[r][0..C) zero-extended (rows of 135 floats are not 16-byte aligned). */
int qpg_pad_channels_f32(qpg_ctx*, void* stream, const float* x, int64_t R, int C, int Cp, float* y);
/* One ResConv1DBlock of width 512 in ONE launch (resnet.py:31-46):  y = x + W2 . relu(W1 (*) relu(x) + b1) + b2,
 * W1 the dilated k3 convolution, W2 the 1x1 one; the hidden activation stays in the MFMA accumulators (the C/D
 * register layout of the first GEMM is the B-operand layout of the second).  x, y: [dev] f32 [B][T][512], x != y;
 * wpack: [dev] f32, 96 stages (T-pack NB = 512 of W1, k = tap*512 + ci) then 32 stages (T-pack NB = 128 of W2) of
 * 8192 floats each; b1, b2: [dev] f32 [512]; hidden: optional [dev] f32 [B][T][512] copy of relu(W1(*)relu(x)+b1). */
int qpg_resblock_f32(qpg_ctx*, void* stream, const float* x, int B, int T, int dil, const float* wpack,
                     const float* b1, const float* b2, float* y, float* hidden);

/* BottleneckBlock.quantise (bottleneck.py:120-126) after the GEMM: ids[r] = argmin_c (|z_r|^2 - 2 dot[r][c]) + kk[c]
 * (f32, that operation order, lowest index on ties).  z: [dev] f32 [R][E]; dot: [dev] f32 [R][K]; kk: [dev] f32 [K];
 * ids: [dev] i64 [R]; dmin / dsecond: optional [dev] f32 [R] best and runner-up distance (parity margin). */
int qpg_vq_argmin_f32(qpg_ctx*, void* stream, const float* z, const float* dot, const float* kk, int64_t R, int E,
                      int K, int64_t* ids, float* dmin, float* dsecond);

/* BottleneckBlock.dequantise (bottleneck.py:128-130): out[r][:] = k[ids[r]][:].  status: optional [dev] i32, set to 1
 * if an id is outside [0,K) (torch's F.embedding raises IndexError). */
int qpg_vq_gather_f32(qpg_ctx*, void* stream, const float* k, const int64_t* ids, int64_t R, int E, int K, float* out,
                      int32_t* status);

/* The convolution of qpg_conv1d_f32's contract on the f16 matrix cores with SPLIT operands (round 5, csrc/qpg_conv16.hip):
 * every value is h + l (two f16 numbers; weights pre-scaled by a power of two per layer), a product is three
 * v_mfma_f32_16x16x32_f16 (h l', l h', h h') accumulated in f32 - 3/16 of the f32 matrix time per flop, within ~1e-6
 * (relative to sum |products|) of the f32 FMA chains, NOT bit-identical: the host uses it under a margin check and keeps
 * the f32 kernels as the referee (qpgesture_amd/vqvae.py encode_f16x3; encdec.py:8-51, resnet.py:31-46 are the layers).
 *   qpg_conv16_image_bytes   bytes of the weight image of a (taps, Cin, Cout) layer (Cin counted as the activation rows'
 *                            channel count, a multiple of 8); the i32 at byte (that - 64) is the layer's scale exponent.
 *   qpg_conv16_pack_weights  w [dev] f32 [taps][Cin_pad_w][Cout_pad_w] (qpg_conv1d_f32's layout) -> image [dev].
 *   qpg_conv16_f32           x [dev] f32 [B][T_in][Cx] (Cx %% 8 == 0, Cx >= Cin, Cin %% 8 == 0; 16-byte aligned), geometry
 *                            as qpg_conv1d_f32; y [dev] f32 [B][T_y][Cout]; status [dev] i32 |= 1 if an activation's
 *                            magnitude exceeds the f16 range (the output is then meaningless: redo on the f32 kernels).
 *                            w_exp: the image's scale exponent, or QPG_CONV16_WEXP_FROM_IMAGE - the kernel reads it from the
 *                            image itself (training: the weights change every step, no host read-back per layer). */
#define QPG_CONV16_WEXP_FROM_IMAGE 0x7fff
int64_t qpg_conv16_image_bytes(int taps, int Cin, int Cout);
int qpg_conv16_pack_weights(qpg_ctx*, void* stream, const float* w, int taps, int Cin, int Cin_pad_w, int Cout, int Cout_pad_w,
                            void* image, int64_t image_bytes);
int qpg_conv16_f32(qpg_ctx*, void* stream, const float* x, int B, int T_in, int Cx, int Cin, const void* image, int w_exp,
                   const float* bias, int taps, int Cout, int in_stride, int in_offset, int dil, int T_out, int out_stride,
                   int out_offset, int T_y, const float* residual, int relu_in, int relu_out, float* y, int32_t* status);

/* Whole-network entry points: VQVAE.encode / VQVAE.decode (vqvae.py:152-181) as ONE call each, the layer
 * sequence of encdec.py:53-136 issued from C on the caller's stream (no host round trips between layers).
 * The model is a plain descriptor of repacked device tensors filled by the caller (qpgesture_amd/vqvae.py
 * builds it from the reference checkpoint's state_dict). */
#define QPG_VQ_MAX_DOWN 4
#define QPG_VQ_MAX_DEPTH 4
typedef struct {
  const float* w;    /* [dev] [taps][cin_pad][cout_pad] */
  const float* b;    /* [dev] [cout_pad] or NULL */
  int32_t taps, cin, cin_pad, cout, cout_pad;
  const float* wt;   /* [dev] optional T-pack (NB = 128) of the same weights for qpg_convt_f32, or NULL */
} qpg_conv_desc;
typedef struct {
  int32_t in_dim, width, emb, bins, down_t, depth, growth, reverse_dec;
  qpg_conv_desc enc_down[QPG_VQ_MAX_DOWN];                       /* Conv1d k4 s2 p1              (encdec.py:20) */
  qpg_conv_desc enc_res[QPG_VQ_MAX_DOWN][QPG_VQ_MAX_DEPTH][2];   /* [..][d][0] k3 dilated, [1] 1x1 (resnet.py:31-46) */
  qpg_conv_desc enc_out;                                         /* Conv1d k3                    (encdec.py:24) */
  qpg_conv_desc dec_in;                                          /* Conv1d k3                    (encdec.py:39) */
  qpg_conv_desc dec_res[QPG_VQ_MAX_DOWN][QPG_VQ_MAX_DEPTH][2];
  qpg_conv_desc dec_up_even[QPG_VQ_MAX_DOWN];                    /* ConvTranspose1d k4 s2 p1, output parity 0/1 (encdec.py:45) */
  qpg_conv_desc dec_up_odd[QPG_VQ_MAX_DOWN];
  qpg_conv_desc dec_out;                                         /* Conv1d k3 -> in_dim          (encdec.py:113) */
  qpg_conv_desc kT;                                              /* codebook^T as a 1-tap conv   (bottleneck.py:123) */
  const float* k;                                                /* [dev] [bins][emb] */
  const float* kk;                                               /* [dev] [bins] sum_e k^2 */
  /* optional (width == emb == 512): fused ResConv1DBlock weight images for qpg_resblock_f32, or NULL */
  const float* enc_res_pack[QPG_VQ_MAX_DOWN][QPG_VQ_MAX_DEPTH];
  const float* dec_res_pack[QPG_VQ_MAX_DOWN][QPG_VQ_MAX_DEPTH];
} qpg_vq_model;

/* floats of scratch the two calls below need for a batch of B sequences of T pose frames */
int64_t qpg_vq_workspace_floats(const qpg_vq_model* m, int B, int T);
/* x: [dev] f32 [B][T][in_dim] (T %% 2^down_t == 0) -> ids [dev] i64 [B][T/2^down_t]; latent: optional [dev] f32
 * [B][T/2^down_t][emb] (pre-quantisation encoder output); margin: optional [dev] f32 [B][T/2^down_t] runner-up minus best. */
int qpg_vq_encode_f32(qpg_ctx*, void* stream, const qpg_vq_model* m, const float* x, int B, int T, float* ws,
                      int64_t ws_floats, int64_t* ids, float* latent, float* margin);
/* ids: [dev] i64 [B][L] -> out [dev] f32 [B][L*2^down_t][in_dim]; status [dev] i32: 1 if an id was out of range. */
int qpg_vq_decode_f32(qpg_ctx*, void* stream, const qpg_vq_model* m, const int64_t* ids, int B, int L, float* ws,
                      int64_t ws_floats, float* out, int32_t* status);

/* Post-decode pose conversion (VisualizeCodebook.py:148-149 + process/process_bvh.py:57-76): poses [dev]
 * f32 [T][J*9] normalised decoder output (J joints x row-major 3x3) -> de-normalise `p * stdc + mean` (f64; stdc already
 * clipped at 0.01) -> optional Savitzky-Golay smoothing in time (sg_mid [W], sg_head / sg_tail [W/2][W] coefficient tables
 * of scipy.signal.savgol_filter(x, W, 2, mode='interp'); NULL = no smoothing) -> orthogonalise each 3x3 like
 * scipy's Rotation.from_matrix (orthogonal polar factor) -> intrinsic Z-X-Y Euler angles in degrees, euler [dev] f64
 * [T][J*3].  status [dev] i32: raised to 1 if a matrix has a non-positive determinant (scipy raises ValueError). */
int qpg_pose_to_euler_f64(qpg_ctx*, void* stream, const float* poses, int64_t T, int J, const double* mean,
                          const double* stdc, const double* sg_mid, const double* sg_head, const double* sg_tail, int W,
                          double* euler, int32_t* status);

/* ---- VQ-VAE training step (codebook/train.py:120-148): VQVAE.forward's loss terms, the bottleneck statistics,
 * the EMA codebook update, the loss gradient and Adam.  Reductions are ordered two-stage sums in f64
 * (deterministic).  `ws` is a caller-owned scratch of at least qpg_vq_reduce_ws_bytes() bytes. ---- */
int64_t qpg_vq_reduce_ws_bytes(void);

/* vqvae.py:244-267.  x_out, x_target: [dev] f32 [B][T][C].  commit_loss: optional [dev] f32 scalar.
 * out6 [dev] f32 = {loss, recons (L1), regularization, velocity, acceleration, commit} with
 * loss = recons + w_commit*commit + w_reg*regularization + w_vel*velocity + w_acc*acceleration. */
int qpg_vq_loss_f32(qpg_ctx*, void* stream, const float* x_out, const float* x_target, int B, int T, int C,
                    const float* commit_loss, float w_commit, float w_reg, float w_vel, float w_acc, void* ws,
                    int64_t ws_bytes, float* out6);
/* d loss / d x_out of the same terms (what autograd produces for vqvae.py:244-267; sign(0) = 0), times `upstream`. */
int qpg_vq_loss_grad_f32(qpg_ctx*, void* stream, const float* x_out, const float* x_target, int B, int T, int C,
                         float w_reg, float w_vel, float w_acc, float upstream, float* d_x_out);

/* BottleneckBlock.forward statistics (bottleneck.py:96-118, 125, 176): z [dev] f32 [R][E] encoder output rows,
 * zq [dev] f32 [R][E] their dequantised codes (qpg_vq_gather_f32, taken BEFORE the EMA update like :169), dmin
 * optional [dev] f32 [R] (from qpg_vq_argmin_f32).
 * out3 [dev] f32 = {commit_loss = |zq-z|^2/(R E), fit = mean(dmin), prenorm = |z-mean(z)|/sqrt(R E)}. */
int qpg_vq_latent_stats_f32(qpg_ctx*, void* stream, const float* z, const float* zq, const float* dmin, int64_t R,
                            int E, void* ws, int64_t ws_bytes, float* out3);
/* d_z = d_zq (straight-through, bottleneck.py:179; optional) + scale * d commit_loss / d z (zq detached). */
int qpg_vq_commit_grad_f32(qpg_ctx*, void* stream, const float* z, const float* zq, int64_t R, int E, float scale,
                           const float* d_zq, float* d_z);

/* BottleneckBlock.update_k (bottleneck.py:63-94) in two halves so that the caller can all-reduce the batch sums
 * across ranks in between (bottleneck.py:73-75):
 *   qpg_vq_code_sums_f32: batch_sum[c][:] = sum of z rows assigned to c (ascending row order), batch_elem[c] = count;
 *   qpg_vq_ema_update_f32: k_sum/k_elem EMA (mu), k = k_sum/k_elem where k_elem >= threshold else k_rand; also
 *   refreshes kT ([E][ldkT] transposed copy, optional) and kk ([K] squared norms, optional) used by the quantiser,
 *   out4 [dev] f32 = {entropy, used_curr, usage, dk}.  ws: >= K doubles. */
int64_t qpg_vq_code_sums_ws_bytes(int64_t R, int E, int K);
int qpg_vq_code_sums_f32(qpg_ctx*, void* stream, const float* z, const int64_t* ids, int64_t R, int E, int K,
                         float* batch_sum, float* batch_elem, void* ws, int64_t ws_bytes);
int qpg_vq_ema_update_f32(qpg_ctx*, void* stream, float* k, float* k_sum, float* k_elem, const float* batch_sum,
                          const float* batch_elem, const float* k_rand, float mu, float threshold, int K, int E,
                          float* kT, int ldkT, float* kk, void* ws, int64_t ws_bytes, float* out4);

/* torch.optim.Adam single step as train.py:71 configures it (no weight decay / amsgrad), over a flat buffer. */
int qpg_adam_step_f32(qpg_ctx*, void* stream, float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                      int64_t n, float lr, float beta1, float beta2, float eps, int64_t step);

/* Backward of the convolution layers (what autograd computes for nn.Conv1d / nn.ConvTranspose1d in
 * encdec.py:20-45,113 and resnet.py:31-46), on the same channels-last activations and packed weights.
 *
 * qpg_conv1d_bwd_data_f32: dx = conv(dy, W^T) with the forward layer's packed weights read in place.
 *   dy [dev] f32 [B][T_in][C_dy] (C_dy = the forward layer's Cout); w_fwd [taps_fwd][fwd_Cin_pad][fwd_Cout_pad];
 *   input tap j of THIS convolution uses weight tap tap_base + j*tap_step (a stride-1 layer: taps, base taps-1,
 *   step -1; the strided down-convolution: one call per output parity with 2 taps, base 3 / 2, step -2; the
 *   transposed convolution's parity sets: base 1, step -1, dilation 2).  Geometry arguments as qpg_conv1d_f32.
 *   gate: optional, indexed like dx — result zeroed where gate <= 0 (the ReLU in front of the forward layer);
 *   residual: optional, added after the gate (skip connection / accumulation of a second parity set).
 * qpg_conv1d_bwd_weight_f32: dw [taps][Cin_pad][Cout_pad] (packed like the weights, fully overwritten) and
 *   db [Cout_pad] (optional; accumulate_bias != 0 adds to it: the two parity sets of a transposed convolution
 *   share one bias) from the layer input x (relu_in as in the forward call) and dy.  The position axis is
 *   split over thread blocks; ws holds the partial sums (qpg_conv1d_wgrad_ws_floats(taps,Cin_pad,Cout_pad,splits)
 *   floats for `splits` partials; more workspace = more parallelism, 1 partial is the minimum). */
int qpg_conv1d_bwd_data_f32(qpg_ctx*, void* stream, const float* dy, int B, int T_in, int C_dy, const float* w_fwd,
                            int taps, int fwd_Cin, int fwd_Cin_pad, int fwd_Cout_pad, int tap_base, int tap_step,
                            int in_stride, int in_offset, int dil, int T_out, int out_stride, int out_offset, int T_y,
                            const float* gate, const float* residual, float* dx, float* ws, int64_t ws_floats);
int64_t qpg_conv1d_wgrad_ws_floats(int taps, int Cin_pad, int Cout_pad, int splits);
int qpg_conv1d_bwd_weight_f32(qpg_ctx*, void* stream, const float* x, int B, int T_in, int Cin, const float* dy,
                              int taps, int Cin_pad, int Cout, int Cout_pad, int in_stride, int in_offset, int dil,
                              int T_out, int out_stride, int out_offset, int T_y, int relu_in, float* dw, float* db,
                              int accumulate_bias, float* ws, int64_t ws_floats);

/* ------------------------------------------------------------------------------------------
 * NOT part of the product library: measurement hooks of the kernel experiments.  They are compiled (and exported) only
 * with -DQPG_DEBUG_HOOKS (tools/build_variant.sh <source> <name> "-DQPG_DEBUG_HOOKS": a variant library the tools load
 * through QPG_LIB_PATH); in libqpg_hip.so every one of these knobs is a compile-time constant and the symbols do not
 * exist (tests/test_host_cpu.py asserts that).  Process-wide where they exist.
 * ---------------------------------------------------------------------------------------- */
#ifdef QPG_DEBUG_HOOKS
int qpg_debug_gemm64_waves(int nw);                  /* waves per block of qpg_hl_gemm_tilemin_h's kernel: 4 (default) or 8 */
int qpg_debug_convt_shape(int nq, int pd);           /* force the short-sequence convolution kernel's block shape */
int qpg_debug_convt_opts(int deep_ring, int xcd_map);/* deep fragment ring / XCD map of that kernel (measured, off) */
int qpg_debug_select_prof(long long* out);           /* -DQPG_SELECT_PROF section timers of the mixed select */
#endif

#ifdef __cplusplus
}
#endif
#endif /* QPG_H */
