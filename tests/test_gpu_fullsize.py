"""Full-size (BASELINE.json configs[1]/[2] scale) checks of the sweeps on the MI355X.

At N_db = 2048 the oracle's C port still finishes in seconds on the GPU box's host cores, so the
(Q,512) tables are compared directly; on top of that come size-independent properties of the path:
planted exact matches (distance 0 wins its code), exact duplicates (lowest index wins), row-shard
invariance, and empty / ragged shapes."""
import os

import numpy as np
import pytest

from tests.helpers import aud_tol

pytestmark = pytest.mark.gpu


def _db(N, seed, F=1024, dup=None, plant=None, speechlike=False):
    from qpgesture_amd import synth
    from qpgesture_amd.data_processing import interp_wavlm
    tr = synth.make_db(N, seed, F)
    if speechlike:
        synth.speechlike_transform(tr, seed + 7)
    interp = interp_wavlm(tr["wavlm"])
    ctx = np.ascontiguousarray(tr["context"].squeeze(2))
    code = synth.make_codes(N, seed + 1)
    return dict(interp=interp, ctx=ctx, code=code, phase=tr["phase_dense"], sig=synth.make_signature(seed + 2))


@pytest.mark.parametrize("speechlike", [False, True])
def test_fullsize_tables_vs_c_oracle(speechlike):
    """N_db = 2048, Q = 48 (the bench workload): every per-code winner equals the C port's (which is
    bit-identical to the reference), text distances bit-exact, audio distances to 1e-13 - on i.i.d. Gaussian features
    and on the speech-like ones (AR(1) in time on rank-64 mixtures, 10 % near-silent frames around one quiet vector,
    context rows repeating over code frames with 20 % silence embeddings: exact text ties, long near-tie bands), where
    the list populations of the capped selects are printed (the caps justified on realistic statistics)."""
    import torch
    from oracle import cref, knn_oracle as O
    from qpgesture_amd import synth
    from qpgesture_amd.code_knn import CodeKNN, GestureDB
    from qpgesture_amd.data_processing import interp_wavlm
    N, M = 2048, 6
    A = _db(N, 100, speechlike=speechlike)
    te = synth.make_db(M, 200)
    if speechlike:
        synth.speechlike_transform(te, 207)
    te_i = interp_wavlm(te["wavlm"])
    te_c = np.ascontiguousarray(te["context"].squeeze(2))
    db = GestureDB(A["code"], A["interp"], A["ctx"], A["phase"], A["sig"], device="cuda:0")
    knn = CodeKNN(db, rng=np.random.RandomState(1))
    T = knn.sweep_tables(torch.from_numpy(te_i).cuda(), torch.from_numpy(te_c).cuda(), M)
    cores = os.cpu_count() or 1
    q = np.stack([O.wavlm_feat_rows(te_i, w, [24 * s])[0] for w in range(M) for s in range(8)])
    d_ref, i_ref = cref.audio_scan(A["interp"], np.arange(26) * 6, A["code"], np.arange(26), q, n_threads=cores)
    assert np.array_equal(T["aud_idx"].cpu().numpy(), i_ref)
    assert np.abs(T["aud_d"].cpu().numpy() - d_ref).max() < aud_tol(knn)
    qt = np.stack([te_c[w][int(24 * s / 180 * 30)] for w in range(M) for s in range(8)])
    dt_ref, it_ref = cref.text_scan(A["ctx"], np.arange(26), A["code"], np.arange(26), qt, n_threads=cores)
    assert np.array_equal(T["txt_d"].cpu().numpy(), dt_ref)
    assert np.array_equal(T["txt_idx"].cpu().numpy(), it_ref)
    want = np.argsort(np.argsort(d_ref, axis=1, kind="stable"), axis=1, kind="stable")
    assert np.array_equal(T["aud_rank"].cpu().numpy(), want)
    ln = knn.tier1_list_lengths()
    st = knn.mixed_stats()
    print("%s: tier-1 list per query min / median / max = %d / %d / %d of 2048; tier-2 pairs %d; flags 0x%x"
          % ("speech-like" if speechlike else "gaussian", ln.min(), int(np.median(ln)), ln.max(), st["tier2_pairs"],
             st["flags"]))
    # the fast path decided this clip by itself or flagged it - either way the tables above are the oracle's
    codes = knn.match_clip(torch.from_numpy(te_i).cuda(), torch.from_numpy(te_c).cuda(), M)[0]
    kx = CodeKNN(db, rng=np.random.RandomState(1))
    kx.audio_precision = "exact"
    sc, sp = kx.init_code_phase()
    a = knn.match_clip(torch.from_numpy(te_i).cuda(), torch.from_numpy(te_c).cuda(), M, seed_code=sc, seed_phase=sp)[0]
    b = kx.match_clip(torch.from_numpy(te_i).cuda(), torch.from_numpy(te_c).cuda(), M, seed_code=sc, seed_phase=sp)[0]
    assert np.array_equal(a, b)


@pytest.mark.parametrize("speechlike", [False, True])
def test_fullsize_f16_track_one_plane_sweep_vs_c_oracle(speechlike):
    """BASELINE.json configs[4] "fp16 features" at the bench size (N_db = 2048, Q = 48): the track stored in f16 is swept
    by the ONE-plane split-f16 kernel (qpg_audio_cosine_hl1: the f16 values are their own h plane).  Tables equal the C
    port's on the f16-ROUNDED track (winners exact, ranks exact, distances inside the a-priori bound where nothing was
    re-evaluated), the f64 sweep of the same rounded track agrees to 1e-13, and the matched codes equal the uncapped
    path's."""
    import torch
    from oracle import cref, knn_oracle as O
    from qpgesture_amd import synth
    from qpgesture_amd.code_knn import CodeKNN, GestureDB
    from qpgesture_amd.data_processing import interp_wavlm
    N, M = 2048, 6
    A = _db(N, 120, speechlike=speechlike)
    te = synth.make_db(M, 220)
    if speechlike:
        synth.speechlike_transform(te, 227)
    te_i = interp_wavlm(te["wavlm"])
    te_c = np.ascontiguousarray(te["context"].squeeze(2))
    db = GestureDB(A["code"], A["interp"], A["ctx"], A["phase"], A["sig"], device="cuda:0", feature_dtype="f16")
    assert db.hl_image is not None and db.hl_planes == 1 and db.hl_image.numel() == N * 27 * 3 * 1024 * 2
    knn = CodeKNN(db, rng=np.random.RandomState(1))
    ti, tc = torch.from_numpy(te_i).cuda(), torch.from_numpy(te_c).cuda()
    T = knn.sweep_tables(ti, tc, M)
    assert knn._last_audio_hl and knn._last_audio_mixed
    rounded = A["interp"].astype(np.float16).astype(np.float32)
    cores = os.cpu_count() or 1
    q = np.stack([O.wavlm_feat_rows(te_i, w, [24 * s])[0] for w in range(M) for s in range(8)])
    d_ref, i_ref = cref.audio_scan(rounded, np.arange(26) * 6, A["code"], np.arange(26), q, n_threads=cores)
    assert np.array_equal(T["aud_idx"].cpu().numpy(), i_ref)
    assert np.abs(T["aud_d"].cpu().numpy() - d_ref).max() < aud_tol(knn)
    want = np.argsort(np.argsort(d_ref, axis=1, kind="stable"), axis=1, kind="stable")
    assert np.array_equal(T["aud_rank"].cpu().numpy(), want)
    # the sweep's own matrix against the f64 sweep of the same rounded track: inside the a-priori bound, everywhere
    D_hl = knn._last_D_aud.double()
    k64 = CodeKNN(db, rng=np.random.RandomState(1))
    k64.audio_precision = "f64"
    T64 = k64.sweep_tables(ti, tc, M)
    err = float((D_hl - k64._last_D_aud).abs().max())
    print("one-plane sweep: max |D - f64 sweep| = %.3g (bound 1.3e-6)" % err)
    assert err < 1.3e-6
    assert torch.equal(T64["aud_idx"], T["aud_idx"]) and torch.equal(T64["aud_rank"], T["aud_rank"])
    assert np.abs(T64["aud_d"].cpu().numpy() - d_ref).max() < 1e-13
    kx = CodeKNN(db, rng=np.random.RandomState(1))
    kx.audio_precision = "exact"
    sc, sp = kx.init_code_phase()
    a = knn.match_clip(ti, tc, M, seed_code=sc, seed_phase=sp)[0]
    b = kx.match_clip(ti, tc, M, seed_code=sc, seed_phase=sp)[0]
    assert np.array_equal(a, b) and knn.fallbacks == 0


def test_empty_clip_returns_empty_arrays():
    """n_windows = 0 used to read an uninitialised status word; the walk now always writes it and match_clip
    answers an empty clip without launching anything."""
    import torch
    from qpgesture_amd import synth
    from qpgesture_amd.code_knn import CodeKNN, GestureDB
    A = _db(16, 300)
    db = GestureDB(A["code"], A["interp"], A["ctx"], A["phase"], A["sig"], device="cuda:0")
    knn = CodeKNN(db, rng=np.random.RandomState(1))
    te_i = torch.zeros((0, 180, 1024), device="cuda:0")
    te_c = torch.zeros((0, 30, 384), device="cuda:0")
    codes, phases, votes = knn.match_clip(te_i, te_c, 0)
    assert codes.shape == (0, 30) and votes.shape[0] == 0 and phases.shape[0] == 0
    # and the walk entry point itself: M = 0 writes a defined status
    from qpgesture_amd import _lib
    st = torch.full((2,), 77, dtype=torch.int32, device="cuda:0")
    z16 = torch.zeros((1, 512), dtype=torch.int16, device="cuda:0")
    z32 = torch.zeros((1, 512), dtype=torch.int32, device="cuda:0")
    _lib.call("qpg_match_steps", db.device, z16, z32, z16, z32, db.pos_rank, db.freq_rank, db.code, db.code.shape[1],
              db.aud_cidx, db.aud_pslot, db.Ga, db.txt_cidx, db.txt_pslot, db.Gt, db.phase, db.Tp, 0, 0, 8, db.K, 0,
              torch.zeros((8, 16), device="cuda:0"), torch.zeros((3, 8, 512), dtype=torch.int32, device="cuda:0"),
              torch.zeros((1, 30), dtype=torch.int32, device="cuda:0"), torch.zeros((1, 8, 8, 16), device="cuda:0"),
              torch.zeros((1, 8), dtype=torch.int32, device="cuda:0"), st, None)
    assert st.cpu().tolist() == [0, 0]


def test_speaker1_class_db_8192_windows_vs_c_oracle():
    """BASELINE.json configs[3]'s workload on ONE GPU: N_db = 8192 windows (212 992 candidates, 6 GB resident base), one
    24 s clip.  Per-code winners of both sweeps equal the C port's, text distances bit-exact, audio <= 1e-13, ranks equal;
    and the two-way row-sharded tables merged by qpg_merge_select_* equal the unsharded ones (what --scaling strong does
    across GPUs, minus the collective)."""
    import torch
    from oracle import cref, knn_oracle as O
    from qpgesture_amd import _lib, synth
    from qpgesture_amd.code_knn import ABSENT_DIST, CodeKNN, GestureDB
    from qpgesture_amd.data_processing import interp_wavlm
    import bench
    N, M = 8192, 6
    interp, ctx = bench.chunked_db(N, 0, N, seed=0)
    code = synth.make_codes(N, 2)
    phase = np.zeros((N, 240, 4, 8), np.float32)
    sig = synth.make_signature(3)
    te = synth.make_db(M, 1000)
    te_i = interp_wavlm(te["wavlm"])
    te_c = np.ascontiguousarray(te["context"].squeeze(2))
    db = GestureDB(code, interp, ctx, phase, sig, device="cuda:0")
    knn = CodeKNN(db, rng=np.random.RandomState(1))
    ti, tc = torch.from_numpy(te_i).cuda(), torch.from_numpy(te_c).cuda()
    T = knn.sweep_tables(ti, tc, M)
    cores = os.cpu_count() or 1
    q = np.stack([O.wavlm_feat_rows(te_i, w, [24 * s])[0] for w in range(M) for s in range(8)])
    d_ref, i_ref = cref.audio_scan(interp, np.arange(26) * 6, code, np.arange(26), q, n_threads=cores)
    assert np.array_equal(T["aud_idx"].cpu().numpy(), i_ref) and np.abs(T["aud_d"].cpu().numpy() - d_ref).max() < aud_tol(knn)
    qt = np.stack([te_c[w][int(24 * s / 180 * 30)] for w in range(M) for s in range(8)])
    dt_ref, it_ref = cref.text_scan(ctx, np.arange(26), code, np.arange(26), qt, n_threads=cores)
    assert np.array_equal(T["txt_d"].cpu().numpy(), dt_ref) and np.array_equal(T["txt_idx"].cpu().numpy(), it_ref)
    assert np.array_equal(T["aud_rank"].cpu().numpy(), np.argsort(np.argsort(d_ref, axis=1, kind="stable"), axis=1, kind="stable"))
    tol = aud_tol(knn)
    del db, knn
    torch.cuda.empty_cache()
    # two row shards on the same GPU, merged by the HIP kernel (no collective: both halves are local here)
    parts = []
    for r in range(2):
        dbr = GestureDB(code, bench._ShardView(interp[r * 4096:(r + 1) * 4096], r * 4096, (r + 1) * 4096, N),
                        bench._ShardView(ctx[r * 4096:(r + 1) * 4096], r * 4096, (r + 1) * 4096, N), phase, sig,
                        device="cuda:0", rank=r, world=2)
        kr = CodeKNN(dbr, rng=np.random.RandomState(1))
        steps = kr.n_steps()
        q_win = np.repeat(np.arange(M), steps)
        q_t = np.tile(np.arange(steps) * 24, M)
        d_, i_ = kr.sweep_audio(ti, q_win, q_t, reduce=False)
        parts.append((d_.clone(), i_.clone()))
        del dbr, kr
        torch.cuda.empty_cache()
    Q, K = parts[0][0].shape
    buf = torch.cat([torch.cat((d_.reshape(-1).view(torch.uint8), i_.reshape(-1).view(torch.uint8))) for d_, i_ in parts])
    od = torch.empty((Q, K), dtype=torch.float64, device="cuda:0")
    oi = torch.empty((Q, K), dtype=torch.int32, device="cuda:0")
    _lib.call("qpg_merge_select_f64", torch.device("cuda:0"), buf, 2, buf.numel() // 2, 0, Q * K * 8, Q, K,
              float(ABSENT_DIST), od, oi, None, 0.0, None)
    # (sweep_audio(reduce=False) without an exchange layout runs the f64 sweep; the unsharded tables the mixed-precision one)
    assert torch.equal(oi, T["aud_idx"]) and float((od - T["aud_d"]).abs().max()) < tol


def test_planted_match_and_duplicates():
    """A query that IS a database candidate gets distance ~0 for that candidate's code and wins it; an exact
    duplicate of that window later in the DB ties and must lose (first wins == lowest index)."""
    import torch
    from qpgesture_amd.code_knn import CodeKNN, GestureDB
    N = 300
    A = _db(N, 300)
    A["interp"][250] = A["interp"][17]            # window 250 duplicates window 17
    A["ctx"][250] = A["ctx"][17]
    A["code"][250] = A["code"][17]
    db = GestureDB(A["code"], A["interp"], A["ctx"], A["phase"], A["sig"], device="cuda:0")
    knn = CodeKNN(db, rng=np.random.RandomState(1))
    # query window = DB window 17: query step s (t=24s) is candidate g=4s of window 17
    te_i = torch.from_numpy(A["interp"][17:18].copy()).cuda()
    te_c = torch.from_numpy(A["ctx"][17:18].copy()).cuda()
    T = knn.sweep_tables(te_i, te_c, 1)
    aud_d, aud_i = T["aud_d"].cpu().numpy(), T["aud_idx"].cpu().numpy()
    for s in range(7):                            # t = 24 s <= 150 is on the candidate grid
        g = 4 * s
        c = int(A["code"][17, g])
        assert aud_i[s, c] == 17 * 26 + g, (s, aud_i[s, c])
        assert abs(aud_d[s, c]) < 1e-15
    txt_d, txt_i = T["txt_d"].cpu().numpy(), T["txt_idx"].cpu().numpy()
    for s in range(7):
        r = int(24 * s / 180 * 30)                # query row r == candidate row r of window 17
        c = int(A["code"][17, r])
        assert txt_i[s, c] == 17 * 26 + r and txt_d[s, c] == 0.0


def test_ragged_and_small_shapes():
    """N not a multiple of any tile, a single DB window, one query window, and codes absent from the DB."""
    import torch
    from oracle import cref, knn_oracle as O
    from qpgesture_amd import synth
    from qpgesture_amd.code_knn import CodeKNN, GestureDB
    from qpgesture_amd.data_processing import interp_wavlm
    for N in (1, 3, 37):
        A = _db(N, 400 + N, F=128)
        te = synth.make_db(1, 500 + N, 128)
        te_i = interp_wavlm(te["wavlm"])
        te_c = np.ascontiguousarray(te["context"].squeeze(2))
        db = GestureDB(A["code"], A["interp"], A["ctx"], A["phase"], A["sig"], device="cuda:0")
        knn = CodeKNN(db, rng=np.random.RandomState(1))
        T = knn.sweep_tables(torch.from_numpy(te_i).cuda(), torch.from_numpy(te_c).cuda(), 1)
        q = np.stack([O.wavlm_feat_rows(te_i, 0, [24 * s])[0] for s in range(8)])
        d_ref, i_ref = cref.audio_scan(A["interp"], np.arange(26) * 6, A["code"], np.arange(26), q)
        assert np.array_equal(T["aud_idx"].cpu().numpy(), i_ref)
        absent = i_ref < 0
        assert absent.any() == (N * 26 < 512)                            # few windows: most codes never occur
        assert np.all(T["aud_d"].cpu().numpy()[absent] == 1e3)           # the reference's placeholder (:668)
        assert np.abs(T["aud_d"].cpu().numpy() - d_ref).max() < aud_tol(knn)
        qt = np.stack([te_c[0][int(24 * s / 180 * 30)] for s in range(8)])
        dt_ref, it_ref = cref.text_scan(A["ctx"], np.arange(26), A["code"], np.arange(26), qt)
        assert np.array_equal(T["txt_d"].cpu().numpy(), dt_ref) and np.array_equal(T["txt_idx"].cpu().numpy(), it_ref)


def test_absent_code_winning_raises_like_reference():
    """With one DB window almost every code is absent; the rank fusion then picks an absent code at some
    step and the reference fails with IndexError (aux[index] == [], GestureKNN.py:631-632) — so do we."""
    import torch
    from qpgesture_amd import synth
    from qpgesture_amd.code_knn import CodeKNN, GestureDB
    from qpgesture_amd.data_processing import interp_wavlm
    A = _db(1, 600, F=128)
    A["code"][:] = 7                                                       # a single code in the whole DB
    te = synth.make_db(1, 601, 128)
    db = GestureDB(A["code"], A["interp"], A["ctx"], A["phase"], A["sig"], device="cuda:0")
    knn = CodeKNN(db, rng=np.random.RandomState(3))
    with pytest.raises(IndexError):
        # previous code is 7 itself -> pos_dist[7] = inf -> its fused rank is worst -> an absent code wins
        knn.match_clip(torch.from_numpy(interp_wavlm(te["wavlm"])).cuda(),
                       torch.from_numpy(np.ascontiguousarray(te["context"].squeeze(2))).cuda(), 1)


def test_empty_inputs():
    import torch
    from qpgesture_amd import _lib
    dev = torch.device("cuda:0")
    z = torch.zeros((1,), device=dev)
    zi = torch.zeros((1,), dtype=torch.int32, device=dev)
    # Q = 0 / N = 0 are no-ops that succeed (they must not launch with a zero-sized grid)
    _lib.call("qpg_audio_cosine_f64", dev, z, 0, 180, 1024, zi, 26, 6, 2, z.double(), z, z.double(), 0, z.double(), 0)
    _lib.call("qpg_text_cosine_f32", dev, z, 0, 384, z, 0, z, 0)
    _lib.call("qpg_rank_rows_f32", dev, z, 0, 512, zi.to(torch.int16))
    with pytest.raises(RuntimeError, match="F % 128"):
        _lib.call("qpg_audio_cosine_f64", dev, z, 1, 180, 100, zi, 26, 6, 2, z.double(), z, z.double(), 1,
                  z.double(), 26)


def test_batch16_clips_tables_vs_c_oracle():
    """BASELINE.json configs[4] shape: 16 concurrent clips x 6 windows = 768 query steps in ONE pair of
    sweeps (query tiles loop over grid.y) — tables equal the C port's; each clip's walk equals the walk of
    that clip matched alone."""
    import torch
    from oracle import cref, knn_oracle as O
    from qpgesture_amd import synth
    from qpgesture_amd.code_knn import CodeKNN, GestureDB
    from qpgesture_amd.data_processing import interp_wavlm
    N, clips, M = 256, 16, 6
    A = _db(N, 700)
    te = synth.make_db(clips * M, 701)
    te_i = interp_wavlm(te["wavlm"])
    te_c = np.ascontiguousarray(te["context"].squeeze(2))
    db = GestureDB(A["code"], A["interp"], A["ctx"], A["phase"], A["sig"], device="cuda:0")
    knn = CodeKNN(db, rng=np.random.RandomState(1))
    ti, tc = torch.from_numpy(te_i).cuda(), torch.from_numpy(te_c).cuda()
    T = knn.sweep_tables(ti, tc, clips * M)
    assert T["aud_d"].shape == (768, 512)
    cores = os.cpu_count() or 1
    q = np.stack([O.wavlm_feat_rows(te_i, w, [24 * s])[0] for w in range(clips * M) for s in range(8)])
    d_ref, i_ref = cref.audio_scan(A["interp"], np.arange(26) * 6, A["code"], np.arange(26), q, n_threads=cores)
    assert np.array_equal(T["aud_idx"].cpu().numpy(), i_ref)
    assert np.abs(T["aud_d"].cpu().numpy() - d_ref).max() < aud_tol(knn)
    qt = np.stack([te_c[w][int(24 * s / 180 * 30)] for w in range(clips * M) for s in range(8)])
    dt_ref, it_ref = cref.text_scan(A["ctx"], np.arange(26), A["code"], np.arange(26), qt, n_threads=cores)
    assert np.array_equal(T["txt_d"].cpu().numpy(), dt_ref) and np.array_equal(T["txt_idx"].cpu().numpy(), it_ref)
    seed_code, seed_phase = knn.init_code_phase()
    for c in (0, 7, 15):
        got, _, _ = knn.walk(T, M, window_offset=c * M, seed_code=seed_code, seed_phase=seed_phase)
        alone = CodeKNN(db, rng=np.random.RandomState(1))
        want, _, _ = alone.match_clip(ti[c * M:(c + 1) * M], tc[c * M:(c + 1) * M], M, seed_code=seed_code,
                                      seed_phase=seed_phase)
        assert np.array_equal(got, want)


def test_f16_feature_storage_vs_c_oracle_on_rounded_track():
    """BASELINE.json configs[4] 'fp16 features': the interpolated WavLM base is stored in f16 and widened in registers
    (qpg_audio_cosine_f64_h).  Same f64 arithmetic on the ROUNDED values: the tables equal the C oracle run on the
    f16-rounded track (winners exact, distances <= 1e-13), for a 16-clip batch; and they differ from the f32 tables
    only through the rounding of the inputs."""
    import torch
    from oracle import cref, knn_oracle as O
    from qpgesture_amd import synth
    from qpgesture_amd.code_knn import CodeKNN, GestureDB
    from qpgesture_amd.data_processing import interp_wavlm
    N, clips, M = 256, 16, 6
    A = _db(N, 710)
    te = synth.make_db(clips * M, 711)
    te_i = interp_wavlm(te["wavlm"])
    te_c = np.ascontiguousarray(te["context"].squeeze(2))
    db = GestureDB(A["code"], A["interp"], A["ctx"], A["phase"], A["sig"], device="cuda:0", feature_dtype="f16")
    assert db.base.dtype == torch.float16
    knn = CodeKNN(db, rng=np.random.RandomState(1))
    ti, tc = torch.from_numpy(te_i).cuda(), torch.from_numpy(te_c).cuda()
    T = knn.sweep_tables(ti, tc, clips * M)
    assert knn._last_audio_hl and db.hl_planes == 1          # round 5: the one-plane split-f16 sweep, 16 query chunks
    rounded = A["interp"].astype(np.float16).astype(np.float32)
    q = np.stack([O.wavlm_feat_rows(te_i, w, [24 * s])[0] for w in range(clips * M) for s in range(8)])
    d_ref, i_ref = cref.audio_scan(rounded, np.arange(26) * 6, A["code"], np.arange(26), q, n_threads=os.cpu_count() or 1)
    assert np.array_equal(T["aud_idx"].cpu().numpy(), i_ref)
    assert np.abs(T["aud_d"].cpu().numpy() - d_ref).max() < aud_tol(knn)          # mixed-precision sweep on the f16 base
    want_rank = np.argsort(np.argsort(d_ref, axis=1, kind="stable"), axis=1, kind="stable")
    assert np.array_equal(T["aud_rank"].cpu().numpy(), want_rank)
    k64 = CodeKNN(db, rng=np.random.RandomState(1))
    k64.audio_precision = "f64"                                                    # f64 sweep + guard on the f16 base
    T64 = k64.sweep_tables(ti, tc, clips * M)
    assert torch.equal(T64["aud_idx"], T["aud_idx"]) and torch.equal(T64["aud_rank"], T["aud_rank"])
    assert np.abs(T64["aud_d"].cpu().numpy() - d_ref).max() < 1e-13
    db32 = GestureDB(A["code"], A["interp"], A["ctx"], A["phase"], A["sig"], device="cuda:0")
    T32 = CodeKNN(db32, rng=np.random.RandomState(1)).sweep_tables(ti, tc, clips * M)
    assert 1e-9 < float((T32["aud_d"] - T["aud_d"]).abs().max()) < 1e-2          # input rounding, nothing else
    assert torch.equal(T32["txt_d"], T["txt_d"])                                   # the text side is untouched


def test_captured_16_clip_step_with_the_encode_leg_equals_encode_then_match():
    """BASELINE.json configs[4] as ONE captured step (round 5): 16 clips x 6 windows against an f16-stored track, the
    VQ-VAE encode of a pose batch on a branch of the same hipGraph.  A replay's codes / votes / status of every clip equal
    the eager batched walk's from the same per-clip seeds, its ids equal VQVAE.encode's, a second replay with other
    seeds (the seed block is data) equals ITS eager walk, and a replay that was not collected can not be overwritten."""
    import torch
    from qpgesture_amd import synth
    from qpgesture_amd.code_knn import CodeKNN, GestureDB
    from qpgesture_amd.data_processing import interp_wavlm
    from qpgesture_amd.vqvae import VQVAE
    N, clips, M = 256, 16, 6
    A = _db(N, 730)
    te = synth.make_db(clips * M, 731)
    ti = torch.from_numpy(interp_wavlm(te["wavlm"])).cuda()
    tc = torch.from_numpy(np.ascontiguousarray(te["context"].squeeze(2))).cuda()
    db = GestureDB(A["code"], A["interp"], A["ctx"], A["phase"], A["sig"], device="cuda:0", feature_dtype="f16")
    enc = VQVAE(None, 135, device="cuda:0").load_state_dict(synth.make_vqvae_state_dict(7))
    x = torch.randn((24, 240, 135), device="cuda:0", generator=torch.Generator(device="cuda:0").manual_seed(5))
    want_ids = enc.encode(x)[0].cpu().numpy()
    knn = CodeKNN(db, rng=np.random.RandomState(1))
    rng = np.random.RandomState(9)
    kg = CodeKNN(db, rng=np.random.RandomState(1))
    cg = kg.capture_clip_graph(M, audio=ti, context=tc, n_clips=clips, encoder=enc, encode_input=x)
    for rep in range(2):
        seeds = rng.randint(0, 512, size=clips)
        phases = rng.standard_normal((clips, 8, 16)).astype(np.float32)
        T = knn.sweep_tables(ti, tc, clips * M, for_walk=True)
        knn.walk_batch(T, M, clips, seeds, phases)
        want = knn._last_ints.cpu().numpy()                                  # [clip][codes | votes | status]
        ints = cg.run_ints(seeds, phases)
        assert cg.captures == 1
        n_c = M * 30
        assert np.array_equal(cg.codes(ints).reshape(clips, -1), want[:, :n_c])
        assert np.array_equal(ints[clips * n_c:clips * (n_c + M * 8)].reshape(clips, -1), want[:, n_c:n_c + M * 8])
        assert np.array_equal(cg.statuses(ints), want[:, -2:]) and not want[:, -2:].any()
        assert np.array_equal(cg.encoded_ids(ints), want_ids)
    # one clip of the batch matched alone from its seed: the same codes
    alone = CodeKNN(db, rng=np.random.RandomState(1))
    c7, _, _ = alone.match_clip(ti[7 * M:8 * M], tc[7 * M:8 * M], M, seed_code=int(seeds[7]), seed_phase=phases[7])
    assert np.array_equal(c7, cg.codes(ints)[7])
    cg.launch(seeds, phases)
    with pytest.raises(RuntimeError, match="not been collected"):
        cg.launch(seeds, phases)
    cg.wait_ints()


def test_input_validation():
    import torch
    from qpgesture_amd.code_knn import CodeKNN, GestureDB
    A = _db(4, 800, F=128)
    db = GestureDB(A["code"], A["interp"], A["ctx"], A["phase"], A["sig"], device="cuda:0")
    knn = CodeKNN(db, rng=np.random.RandomState(1))
    ti = torch.from_numpy(A["interp"][:2].copy()).cuda()
    tc = torch.from_numpy(A["ctx"][:2].copy()).cuda()
    with pytest.raises(IndexError):
        knn.match_clip(ti, tc, 3)                       # more windows requested than supplied (reference: IndexError)
    with pytest.raises(ValueError):
        knn.match_clip(ti[:, :, :64].contiguous(), tc, 2)   # feature width differs from the database's
    with pytest.raises(ValueError):
        CodeKNN(db, use_wavlm=False, use_wavvq=True)    # DB built without a wavvq track


def test_graph_replays_stay_valid_after_eager_clips():
    """Regression (round 1): a captured clip replayed after eager clips of the same process returned all-absent tables
    / faulted, because the table initialisations were hipMemsetAsync nodes, which ROCm 7.2's graph replay let complete
    AFTER the kernels that followed them once other work had run in between.  They are fill kernels now.  Full-size DB
    (the failure needed the ~0.8 ms clip), eager clips on two streams, replays before and after."""
    import torch
    from qpgesture_amd import synth
    from qpgesture_amd.code_knn import CodeKNN, GestureDB
    N, M = 2048, 6
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    interp = torch.randn((N, 180, 1024), device=dev)
    ctx = rng.standard_normal((N, 30, 384)).astype(np.float32)
    phase = rng.standard_normal((N, 240, 4, 8)).astype(np.float32)
    db = GestureDB(synth.make_codes(N, 2), interp, ctx, phase, synth.make_signature(3), device=dev)
    knn = CodeKNN(db, rng=np.random.RandomState(123456))
    te_i = torch.randn((M, 180, 1024), device=dev)
    te_c = torch.randn((M, 30, 384), device=dev)
    sc, sp = knn.init_code_phase()
    spd = torch.from_numpy(sp).to(dev)

    def eager():
        T = knn.sweep_tables(te_i, te_c, M)
        return knn.walk(T, M, 0, seed_code=sc, seed_phase=spd, sync=False)[0].cpu()
    want = eager()
    g = knn.capture_clip_graph(M)
    assert torch.equal(g.run(te_i, te_c, sc, spd)[0].cpu(), want)
    for _ in range(40):
        assert torch.equal(eager(), want)
    torch.cuda.synchronize()
    for _ in range(20):
        out = g.run(te_i, te_c, sc, spd)
        assert torch.equal(out[0].cpu(), want) and out[3].cpu().tolist() == [0, 0]


def test_walk_batch_equals_one_walk_per_clip():
    """Several independent clips behind one batched sweep (BASELINE configs[4]): qpg_match_steps_batch walks them in one set
    of launches; codes, votes, phase blocks and status words must be those of one qpg_match_steps per clip (different seed
    codes and seed phase blocks per clip)."""
    import torch
    from qpgesture_amd import synth
    from qpgesture_amd.code_knn import CodeKNN, GestureDB
    A = _db(120, 300)
    db = GestureDB(A["code"], A["interp"], A["ctx"], A["phase"], A["sig"], device="cuda:0")
    knn = CodeKNN(db, rng=np.random.RandomState(4))
    CL, M = 5, 3
    g = torch.Generator(device="cpu").manual_seed(9)
    te_i = torch.randn((CL * M, 180, 1024), generator=g).cuda()
    te_c = torch.randn((CL * M, 30, 384), generator=g).cuda()
    T = knn.sweep_tables(te_i, te_c, CL * M)
    seeds, phases = [], []
    for c in range(CL):
        sc, sp = knn.init_code_phase()
        seeds.append(sc)
        phases.append(sp)
    from qpgesture_amd import _lib
    dev = "cuda:0"

    def gate_dedup(v):                                   # per-context knob (round 6; was a process-wide debug hook)
        _lib.set_option(dev, _lib.QPG_OPT_GATE_DEDUP_FROM_CHAINS, v)
        return 0
    one = []
    try:
        assert gate_dedup(0) == 0          # the one-clip references on round 4's plain gate table
        for c in range(CL):
            oc, op, ov, st = knn.walk(T, M, window_offset=c * M, seed_code=seeds[c], seed_phase=phases[c], sync=False)
            one.append((oc.cpu().numpy(), op.cpu().numpy(), ov.cpu().numpy(), st.cpu().numpy()))
        assert gate_dedup(1) == 0          # ... and the deduplicated table must give the same one-clip walk
        for c in range(CL):
            oc, op, ov, st = knn.walk(T, M, window_offset=c * M, seed_code=seeds[c], seed_phase=phases[c], sync=False)
            assert np.array_equal(oc.cpu().numpy(), one[c][0]) and np.array_equal(op.cpu().numpy(), one[c][1])
            assert np.array_equal(ov.cpu().numpy(), one[c][2]) and np.array_equal(st.cpu().numpy(), one[c][3])
    finally:
        gate_dedup(1)
    try:
        # round 5: the gate table is deduplicated by the previous winner (gate_table_dedup_kernel); 0 = the plain kernel,
        # 4 = the plain kernel for the one-clip walks above only - all must agree
        for dedup_from in (4, 0, 1):
            assert gate_dedup(dedup_from) == 0
            bc, bp, bv = knn.walk_batch(T, M, CL, seeds, np.stack(phases))
            ints = knn._last_ints.cpu().numpy()
            for c in range(CL):
                assert np.array_equal(bc[c].cpu().numpy(), one[c][0]), dedup_from
                assert np.array_equal(bp[c].cpu().numpy(), one[c][1]), dedup_from
                assert np.array_equal(bv[c].cpu().numpy(), one[c][2]), dedup_from
                assert np.array_equal(ints[c, -2:], one[c][3])
                assert np.array_equal(ints[c, :M * 30], one[c][0].reshape(-1))
    finally:
        gate_dedup(1)
    assert len({tuple(o[0].reshape(-1)) for o in one}) > 1              # the clips really differ


def test_rank_fusion_split_by_modality_equals_the_one_launch_fusion():
    """Round 5: with both modalities on, sweep_tables(for_walk=True) launches each side's half of the rank fusion behind
    that side's select on its own stream (qpg_fuse_best_ranked) and the walk takes the tables as they are
    (QPG_MODE_PREFUSED).  The two candidate tables must equal the two-table kernel's bit for bit, and the walks - one clip
    and a batch of clips, walk-relevance cut on - must return the same codes, votes, phase blocks and status words."""
    import torch
    from qpgesture_amd.code_knn import CodeKNN, GestureDB
    A = _db(110, 320)
    db = GestureDB(A["code"], A["interp"], A["ctx"], A["phase"], A["sig"], device="cuda:0")
    knn = CodeKNN(db, rng=np.random.RandomState(6))
    CL, M = 4, 3
    g = torch.Generator(device="cpu").manual_seed(19)
    te_i = torch.randn((CL * M, 180, 1024), generator=g).cuda()
    te_c = torch.randn((CL * M, 30, 384), generator=g).cuda()
    seeds, phases = [], []
    for c in range(CL):
        sc, sp = knn.init_code_phase()
        seeds.append(sc)
        phases.append(sp)
    res = {}
    knn.split_fuse_max_steps = 1 << 20
    for split in (False, True):
        knn.split_fuse = split
        # one clip of M windows: walk(); CL clips behind one sweep: walk_batch()
        T1 = knn.sweep_tables(te_i[:M], te_c[:M], M, for_walk=True)
        assert ("gate_tables" in T1) == split
        oc, op, ov, st = knn.walk(T1, M, seed_code=seeds[0], seed_phase=phases[0], sync=False)
        one = (oc.cpu().numpy(), op.cpu().numpy(), ov.cpu().numpy(), st.cpu().numpy())
        Tb = knn.sweep_tables(te_i, te_c, CL * M, for_walk=True)
        bc, bp, bv = knn.walk_batch(Tb, M, CL, seeds, np.stack(phases))
        res[split] = (one, bc.cpu().numpy(), bp.cpu().numpy(), bv.cpu().numpy(), knn._last_ints.cpu().numpy(),
                      knn._last_gate_tables[:2].cpu().numpy())      # both candidate tables, every (step, previous code)
        if split:
            assert knn._last_gate_tables.data_ptr() == Tb["gate_tables"].data_ptr()
    knn.split_fuse = True
    a, b = res[False], res[True]
    for x, y in zip(a[0], b[0]):
        assert np.array_equal(x, y)
    for x, y in zip(a[1:], b[1:]):
        assert np.array_equal(x, y)
    assert (a[4][:, -2:] == 0).all()


def test_walk_results_through_pinned_host_memory():
    """walk(sync="ints"): codes | votes | status written by the walk's last kernel straight into pinned host memory
    (zero-copy) == the device buffer of walk(sync=False); match_clip (sync=True) takes the same route."""
    import torch
    from qpgesture_amd.code_knn import CodeKNN, GestureDB
    A = _db(90, 310)
    db = GestureDB(A["code"], A["interp"], A["ctx"], A["phase"], A["sig"], device="cuda:0")
    knn = CodeKNN(db, rng=np.random.RandomState(2))
    M = 4
    g = torch.Generator(device="cpu").manual_seed(3)
    te_i = torch.randn((M, 180, 1024), generator=g).cuda()
    te_c = torch.randn((M, 30, 384), generator=g).cuda()
    sc, sp = knn.init_code_phase()
    T = knn.sweep_tables(te_i, te_c, M)
    knn.walk(T, M, seed_code=sc, seed_phase=sp, sync=False)
    dev_ints = knn._last_ints.cpu().numpy()
    for rep in range(200):                               # (the pinned buffer is reused from call to call; round 4: every
        host_ints = knn.walk(T, M, seed_code=sc, seed_phase=sp, sync="ints")     # word is a sentinel until it is written)
        assert host_ints.dtype == np.int32
        assert np.array_equal(host_ints, dev_ints), (rep, np.nonzero(host_ints != dev_ints)[0], host_ints[host_ints != dev_ints],
                                                     dev_ints[host_ints != dev_ints])
    codes, phases, votes = knn.walk(T, M, seed_code=sc, seed_phase=sp, sync=True)
    assert np.array_equal(codes.reshape(-1), dev_ints[:M * 30]) and np.array_equal(votes.reshape(-1), dev_ints[M * 30:-2])


def test_clips_in_flight_on_a_full_size_db_equal_serial_clips():
    """every lane of a ClipPipeline shares ONE GestureDB, and with the matrix-core text side the
    prefilter's scratch (column image, tile minima / masks) used to live on the shared SortedRows - lanes on unordered
    side streams overwrote each other's buffers.  The scratch belongs to the lane's matcher now.  Full-size DB (the text
    sides are long enough to overlap), three lanes, DIFFERENT clips, both modalities: every clip's codes, votes and phase
    blocks equal the same clip matched on its own."""
    import torch
    from qpgesture_amd import synth
    from qpgesture_amd.code_knn import ClipPipeline, CodeKNN, GestureDB
    N, M = 2048, 6
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    interp = torch.randn((N, 180, 1024), device=dev)
    ctx = rng.standard_normal((N, 30, 384)).astype(np.float32)
    phase = rng.standard_normal((N, 240, 4, 8)).astype(np.float32)
    db = GestureDB(synth.make_codes(N, 2), interp, ctx, phase, synth.make_signature(3), device=dev)
    knn = CodeKNN(db, rng=np.random.RandomState(5))
    g = torch.Generator(device="cpu").manual_seed(17)
    clips, seeds = [], []
    for k in range(9):
        m = M if k % 3 else 3
        clips.append((torch.randn((m, 180, 1024), generator=g).to(dev), torch.randn((m, 30, 384), generator=g).to(dev), m))
        seeds.append(knn.init_code_phase())
    want = [knn.match_clip(ti, tc, m, seed_code=s_[0], seed_phase=s_[1]) for (ti, tc, m), s_ in zip(clips, seeds)]
    assert knn._last_text_mfma and knn.fallbacks == 0
    pipe = ClipPipeline(db, depth=3, rng=np.random.RandomState(1))
    for rep in range(3):                                     # (a race shows up in some schedules only)
        got = pipe.match_clips(clips, seeds=seeds)
        for w, r in zip(want, got):
            for a, b in zip(w, r):
                assert np.array_equal(a, b)
    assert pipe.fallbacks == 0
    assert len({id(ln["knn"]._txt_scratch) for ln in pipe.lanes}) == 3


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("speechlike", [False, True])
def test_walk_relevance_cut_returns_the_same_walk(mode, speechlike):
    """sweep_tables(for_walk=True): the audio select settles in f64 only what the walk can read (codes whose rank is
    certainly above every step's winning fused score keep their sweep values: qpg_percode_select_mixed_f64_cut).  On a
    full-size DB with planted near-copies: far fewer f64 re-evaluations, and for many seeds - every previous code the first
    step can start from - codes, votes and phase blocks equal those of the exact tables; rank and candidate of every code
    the walk can read are the exact ones, every distance stays inside the sweep's bound and no rank moved by more than
    its tie neighbourhood allows."""
    import torch
    from qpgesture_amd.code_knn import CodeKNN, GestureDB
    N, M = 1024, 6
    d = _db(N, 31, speechlike=speechlike)
    rs = np.random.RandomState(5)
    for _ in range(60):                                        # near-copies: within-code and cross-code near-ties
        j, k = rs.choice(N, 2, replace=False)
        eps = 0.0 if rs.rand() < 0.2 else 10.0 ** rs.uniform(-7.5, -5.0)
        d["interp"][k] = (d["interp"][j] * (1.0 + eps * rs.standard_normal(d["interp"][j].shape))).astype(np.float32)
        if rs.rand() < 0.5:
            d["code"][k] = d["code"][j]
    dev = torch.device("cuda:0")
    db = GestureDB(d["code"], d["interp"], d["ctx"], d["phase"], d["sig"], device=dev)
    te = _db(M, 77, speechlike=speechlike)
    te_i, te_c = torch.from_numpy(te["interp"]).to(dev), torch.from_numpy(te["ctx"]).to(dev)
    knn = CodeKNN(db, rng=np.random.RandomState(1))
    assert knn.rank_cut
    s0 = knn.mixed_stats()
    T = knn.sweep_tables(te_i, te_c, M, mode=mode)
    s1 = knn.mixed_stats()
    Tc = knn.sweep_tables(te_i, te_c, M, mode=mode, for_walk=True)
    s2 = knn.mixed_stats()
    assert knn._last_rank_cut and knn._last_audio_hl and s2["flags"] == 0
    full, cut = s1["tier1_pairs"] - s0["tier1_pairs"], s2["tier1_pairs"] - s1["tier1_pairs"]
    print("tier-1 pairs: exact tables %d, cut %d" % (full, cut))
    assert 0 < cut < 0.6 * full
    # every code whose entry differs kept its sweep value (inside the bound) and its rank moved inside its neighbourhood
    dd = (T["aud_d"] - Tc["aud_d"]).abs()
    assert float(dd.max()) <= 1.3e-6
    moved = (T["aud_rank"].to(torch.int32) - Tc["aud_rank"].to(torch.int32)).abs()
    assert int(moved.max()) <= 8
    # a code the walk can read: the best fused score of some previous code - all of those rows are exact
    pr, fr = db.pos_rank.to(torch.float64), db.freq_rank.to(torch.float64)
    for q in range(T["aud_rank"].shape[0]):
        sc = pr + 0.05 * fr[None, :] + T["aud_rank"][q].to(torch.float64)[None, :]            # [prev][code]
        top = torch.topk(sc, 2 if mode == 1 else 1, dim=1, largest=False).indices.reshape(-1).unique()
        assert torch.equal(T["aud_rank"][q][top], Tc["aud_rank"][q][top])
        assert torch.equal(T["aud_idx"][q][top], Tc["aud_idx"][q][top])
        # (their DISTANCES are only equal inside the bound: the exact tables' grid test also re-evaluates codes whose
        # nearest neighbour is up to 2 eps1 away, the cut's sorted scan only those within eps1 - the walk reads ranks and
        # candidates, never distances)
    for seed_code in list(range(0, 512, 7)) + [511]:
        sp = rs.standard_normal((8, 16)).astype(np.float32)
        a = knn.walk(T, M, mode=mode, seed_code=seed_code, seed_phase=sp)
        b = knn.walk(Tc, M, mode=mode, seed_code=seed_code, seed_phase=sp)
        ph = [x.cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x) for x in (a[1], b[1])]
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2]) and np.array_equal(ph[0], ph[1]), seed_code
