"""GPU parity tests of the matching hot path: HIP kernels (through the C ABI) vs the oracle and
vs the reference-generated golden fixtures.  Run on the MI355X box: pytest -m gpu."""
import numpy as np
import pytest

from tests.helpers import fixture_arrays, load_golden

pytestmark = pytest.mark.gpu

GOLDENS = ["shipped_n48_m2_s0", "shipped_n64_m3_s10", "shipped_n256_m2_s70"]


def _build(meta, dev="cuda:0", freq_rank=None):
    import torch
    from qpgesture_amd.code_knn import CodeKNN, GestureDB
    ntr, nte, s0, s1, s2, s3, mf = [int(v) for v in meta]
    A = fixture_arrays(ntr, nte, s0, s1, s2, s3)
    db = GestureDB(A["code"], A["tr_interp"], A["tr_ctx"], A["tr_phase"], A["sig"], device=dev,
                   freq_rank=freq_rank)
    rs = np.random.RandomState(123456)
    knn = CodeKNN(db, rng=rs)
    te_i = torch.from_numpy(A["te_interp"]).to(dev)
    te_c = torch.from_numpy(np.ascontiguousarray(A["te_ctx"])).to(dev)
    return A, db, knn, te_i, te_c, nte


@pytest.mark.parametrize("name", GOLDENS)
def test_own_frequency_ranks_vs_the_reference_s(name):
    """The table tests hand the matcher the reference's captured frequency ranks (`freq_rank=g["step_freq_score"]`); this
    one checks the product's OWN ranking (GestureDB without freq_rank: code_to_freq on the host, qpg_rank_rows_f64 on the
    device; GestureKNN.py:481-499, :544): it is the stable rank of 1 - count / total (value, then code index); it equals
    the reference's captured rank wherever a frequency is unique, and inside every group of equal frequencies (NumPy's
    default argsort leaves their order open) both hold the same set of ranks.  And the clip matched with the product's own
    ranks returns the reference's codes."""
    g = load_golden(name)
    A, db, knn, te_i, te_c, M = _build(g["meta"])                       # freq_rank=None: the product's own
    code = np.asarray(A["code"]).reshape(-1)
    cnt = np.bincount(code, minlength=512)[:512]
    f = np.where(cnt > 0, 1 - cnt / cnt.sum(), 1.0)
    mine = db.freq_rank.cpu().numpy().astype(np.int64)
    ref = np.asarray(g["step_freq_score"]).astype(np.int64)
    assert np.array_equal(mine, np.argsort(np.argsort(f, kind="stable"), kind="stable"))
    assert sorted(mine.tolist()) == list(range(512)) and sorted(ref.tolist()) == list(range(512))
    for v in np.unique(f):                    # (small DBs: a handful of distinct counts, every one shared by many codes)
        ix = np.where(f == v)[0]
        assert sorted(mine[ix].tolist()) == sorted(ref[ix].tolist())
        if len(ix) == 1:
            assert mine[ix[0]] == ref[ix[0]]
    if np.array_equal(mine, ref):                                       # (ties broken alike: the whole clip must agree)
        codes, phases, votes = knn.match_clip(te_i, te_c, M)
        assert np.array_equal(codes, g["knn_pred"]) and np.array_equal(votes, g["vote"])


@pytest.mark.parametrize("prec", ["f64", "mixed"])
@pytest.mark.parametrize("name", GOLDENS)
def test_tables_vs_reference_golden(name, prec):
    """Per-(query, code) minima of both sweeps against what the REFERENCE returned:
    winners (argmin candidate) exact; text distances bit-exact (f32 arithmetic reproduced);
    audio distances to 1e-13 (f64, different but equivalent formula) with identical ordering — on the mixed-precision
    path (the default) the untouched minima carry the sweep's bounded error instead, winners and ordering unchanged."""
    from qpgesture_amd.code_knn import AUDIO_MX_ERR
    g = load_golden(name)
    A, db, knn, te_i, te_c, M = _build(g["meta"], freq_rank=g["step_freq_score"])
    knn.audio_precision = prec
    codes, phases, votes = knn.match_clip(te_i, te_c, M, return_tables=True)
    T = knn.tables
    aud_d = T["aud_d"].cpu().numpy()
    aud_idx = T["aud_idx"].cpu().numpy()
    txt_d = T["txt_d"].cpu().numpy()
    txt_idx = T["txt_idx"].cpu().numpy()
    # golden aux = [j, k]; candidate index = j*26 + k/step
    gj, gk = g["aud_aux"][..., 0], g["aud_aux"][..., 1]
    assert np.array_equal(aud_idx, gj * 26 + gk // 6)
    gj, gk = g["txt_aux"][..., 0], g["txt_aux"][..., 1]
    assert np.array_equal(txt_idx, gj * 26 + gk // 8)
    assert txt_d.dtype == np.float32 and np.array_equal(txt_d, g["txt_dist"])        # bit-exact
    assert np.abs(aud_d - g["aud_dist"]).max() < (1e-13 if prec == "f64" else AUDIO_MX_ERR)
    assert np.array_equal(np.argsort(aud_d, axis=1, kind="stable"),
                          np.argsort(g["aud_dist"], axis=1, kind="stable"))


@pytest.mark.parametrize("order", ["auto", "text_after_sweep", "text_first", "audio_first", "one_stream"])
@pytest.mark.parametrize("name", GOLDENS)
def test_knn_pred_vs_reference_golden(name, order):
    """End result of the clip: the (M,30) code indices the reference CLI wrote — bit-exact, whichever way the two
    sides are scheduled (default "auto": the text side on its own stream behind the first 80 % of the audio sweep)."""
    g = load_golden(name)
    A, db, knn, te_i, te_c, M = _build(g["meta"], freq_rank=g["step_freq_score"])
    if order != "auto":
        knn.text_after_sweep = order == "text_after_sweep"
        knn.audio_first = order == "audio_first"
    knn.overlap_sweeps = order != "one_stream"
    codes, phases, votes = knn.match_clip(te_i, te_c, M)
    assert codes.dtype == np.int64 and np.array_equal(codes, g["knn_pred"])
    assert np.array_equal(votes, g["vote"])
    assert np.array_equal(phases, g["phase_out"])


def test_unfused_entry_points_agree_with_fused():
    """The stand-alone C-ABI entry points (distance matrix + qpg_percode_select_*, qpg_rank_rows_*) give
    the same tables as the fused fast path, and the full distance matrices match the oracle's C port."""
    import torch
    from oracle import cref, knn_oracle as O
    g = load_golden(GOLDENS[0])
    A, db, knn, te_i, te_c, M = _build(g["meta"])
    steps = knn.n_steps()
    q_win, q_t = np.repeat(np.arange(M), steps), np.tile(np.arange(steps) * 24, M)
    knn.audio_precision = "f64"
    d0, i0, r0 = knn.sweep_audio(te_i, q_win, q_t, want_rank=True)
    from tests.helpers import sweep_audio_unfused, sweep_text_unfused
    d1, i1, D = sweep_audio_unfused(knn, te_i, q_win, q_t)
    assert torch.equal(d0, d1) and torch.equal(i0, i1)
    assert torch.equal(r0, knn.rank_rows(d1))
    knn.audio_precision = "mixed"           # the default: same winners and ranks, values inside the sweep's bound
    dm, im, rm = knn.sweep_audio(te_i, q_win, q_t, want_rank=True)
    assert torch.equal(im, i1) and torch.equal(rm, r0) and float((dm - d1).abs().max()) <= 2.05e-6
    rows = [int(i / 180 * 30) for i in q_t]
    qt = te_c[torch.as_tensor(q_win, device=te_c.device), torch.as_tensor(rows, device=te_c.device)].contiguous()
    t0, j0, s0 = knn.sweep_text(qt, want_rank=True)
    t1, j1, Dt = sweep_text_unfused(knn, qt)
    assert torch.equal(t0, t1) and torch.equal(j0, j1) and torch.equal(s0, knn.rank_rows(t1))
    # ranks == stable argsort-argsort
    want = np.argsort(np.argsort(d1.cpu().numpy(), axis=1, kind="stable"), axis=1, kind="stable")
    assert np.array_equal(r0.cpu().numpy(), want)


def test_reference_shaped_single_query_api():
    """CodeKNN.search_audio_cands / search_text_cands: same argument and return shapes as the reference
    (GestureKNN.py:666-691, 708-721), values equal to what the reference returned for the first step."""
    from oracle import knn_oracle as O
    g = load_golden(GOLDENS[0])
    A, db, knn, te_i, te_c, M = _build(g["meta"])
    clip = O.wavlm_feat_rows(A["te_interp"], 0, [0])[0]
    dist, pay, aux = knn.search_audio_cands(clip, mode="wavlm_feat")
    assert len(dist) == len(pay) == len(aux) == 512
    assert np.abs(np.array(dist) - g["aud_dist"][0]).max() < 1e-13
    assert all(list(aux[c]) == list(g["aud_aux"][0][c]) for c in range(512))
    assert all(np.array_equal(pay[c], g["aud_pay"][0][c]) for c in range(512))
    dist, pay, aux = knn.search_text_cands(A["te_ctx"][0][0])
    assert np.array_equal(np.array(dist, np.float32), g["txt_dist"][0])
    assert all(list(aux[c]) == list(g["txt_aux"][0][c]) for c in range(512))


@pytest.mark.parametrize("name", GOLDENS)
def test_cli_drop_in(name, tmp_path):
    """python -m qpgesture_amd.GestureKNN with the reference's flags on the same .npz files writes the
    same knn_pred bytes the reference CLI wrote."""
    from qpgesture_amd import GestureKNN as cli
    from qpgesture_amd import synth
    g = load_golden(name)
    ntr, nte, s0, s1, s2, s3, mf = [int(v) for v in g["meta"]]
    paths = synth.write_npz_set(str(tmp_path), ntr, nte, s0, s1, s2, s3)
    out = str(tmp_path / "result.npz")
    argv = []
    for k, v in paths.items():
        argv += ["--" + k, v]
    cli.main(argv + ["--out_knn_filename", out, "--max_frames", str(mf)])
    got = np.load(out)["knn_pred"]
    assert got.dtype == np.int64 and got.shape == g["knn_pred"].shape
    assert np.array_equal(got, g["knn_pred"])


WAVVQ = [("wavvq_aud_txt_n40_m2_s20", True), ("wavvq_aud_n40_m2_s20", False)]


@pytest.mark.parametrize("name,use_txt", WAVVQ)
def test_wavvq_mode(name, use_txt, tmp_path):
    """vq-wav2vec / Levenshtein audio (the flags the paper describes).  Integer distances and winners are
    compared with what the REFERENCE returned (bit-exact); the final codes with the oracle under the
    library's documented stable-rank contract (integer distances tie massively, and the reference's
    unstable argsort orders ties CPU-dependently), and with the reference itself when this host's NumPy
    happens to reproduce the reference's tie order."""
    import torch
    from oracle import knn_oracle as O
    from qpgesture_amd import synth
    from qpgesture_amd.code_knn import MODE_AUD, MODE_AUD_TXT, CodeKNN, GestureDB
    g = load_golden(name)
    ntr, nte, s0, s1, s2, s3, mf = [int(v) for v in g["meta"]]
    A = fixture_arrays(ntr, nte, s0, s1, s2, s3, wavlm_dim=8)
    db = GestureDB(A["code"], A["tr_interp"], A["tr_ctx"], A["tr_phase"], A["sig"], device="cuda:0",
                   wavvq=A["tr_wavvq"])
    knn = CodeKNN(db, use_wavlm=False, use_wavvq=True, rng=np.random.RandomState(2))
    assert knn.n_steps() == 8 and db.Gv == 26 and db.vq_taps == [-66, -53, -39, -26, -13, 0, 13, 26, 39, 53, 66]
    te_c = torch.from_numpy(np.ascontiguousarray(A["te_ctx"])).cuda()
    mode = MODE_AUD_TXT if use_txt else MODE_AUD
    codes, phases, votes = knn.match_clip(torch.from_numpy(A["te_wavvq"]).cuda(), te_c, nte, mode=mode,
                                          return_tables=True)
    T = knn.tables
    assert np.array_equal(T["aud_d"].cpu().numpy().astype(np.float64), g["aud_dist"])          # exact integers
    gj, gk = g["aud_aux"][..., 0], g["aud_aux"][..., 1]
    pos = {k: i for i, k in enumerate(db.vq_k)}
    want_idx = gj * 26 + np.vectorize(pos.get)(gk)
    assert np.array_equal(T["aud_idx"].cpu().numpy(), want_idx)
    if use_txt:
        assert np.array_equal(T["txt_d"].cpu().numpy(), g["txt_dist"])
    # oracle with the stable-rank contract
    with __import__("tempfile").TemporaryDirectory() as td:
        paths = synth.write_npz_set(td, ntr, nte, s0, s1, s2, s3, wavlm_dim=8)
        (want, wph, wv), ok = O.load_and_match_wavvq(paths, use_txt=use_txt, rank_kind="stable")
        (ref_like, _, _), ok2 = O.load_and_match_wavvq(paths, use_txt=use_txt, rank_kind="numpy")
    assert np.array_equal(codes, want)
    assert np.array_equal(phases, wph)
    if np.array_equal(ref_like, want):         # ties did not matter here: then the reference's output too
        assert np.array_equal(codes, g["knn_pred"])
    # reference-shaped single-query API
    vq_feat = O.wavvq_feat(A["te_wavvq"])
    d, pay, aux = knn.search_audio_cands(vq_feat[0, 0], mode="wavvq_feat")
    assert np.array_equal(np.array(d, np.float64), g["aud_dist"][0])
    assert all(list(aux[c]) == list(g["aud_aux"][0][c]) for c in range(512))


def test_wavvq_invalid_init_draw_raises():
    """The reference's own seed draws init frame 234 in wavvq mode -> short phase slice -> it fails; so do we."""
    from qpgesture_amd.code_knn import CodeKNN, GestureDB
    A = fixture_arrays(40, 2, 20, 21, 22, 23, wavlm_dim=8)
    db = GestureDB(A["code"], A["tr_interp"], A["tr_ctx"], A["tr_phase"], A["sig"], device="cuda:0",
                   wavvq=A["tr_wavvq"])
    knn = CodeKNN(db, use_wavlm=False, use_wavvq=True, rng=np.random.RandomState(123456))
    with pytest.raises(ValueError, match="phase slice"):
        knn.init_code_phase()


def test_device_resample_bitexact():
    """qpg_wavlm_resample_f32 == torch's CPU F.interpolate(linear, align_corners=True), bit for bit
    (data_processing.py:258-261: 199 -> 180 frames), incl. other lengths."""
    import torch
    from qpgesture_amd.data_processing import interp_wavlm, interp_wavlm_device
    rng = np.random.default_rng(5)
    for N, Tin, F in ((3, 199, 1024), (2, 60, 8), (1, 199, 6), (2, 398, 64)):
        x = (rng.standard_normal((N, Tin, F)) * 3).astype(np.float32)
        want = interp_wavlm(x)
        got = interp_wavlm_device(x, "cuda:0", chunk=2).cpu().numpy()
        assert got.shape == want.shape and np.array_equal(got, want)


def test_clip_graph_replay_equals_eager():
    """The per-clip launch sequence captured as one HIP graph replays to the same codes / phases / votes."""
    import torch
    g = load_golden(GOLDENS[1])
    A, db, knn, te_i, te_c, M = _build(g["meta"], freq_rank=g["step_freq_score"])
    sc, sp = knn.init_code_phase()
    cg = knn.capture_clip_graph(M)
    for rep in range(2):                    # second run reuses the captured graph
        codes, phases, votes, status = cg.run(te_i, te_c, sc, sp)
        assert status.cpu().tolist() == [0, 0]        # [absent code won, guard trouble word]
        assert np.array_equal(codes.cpu().numpy().astype(np.int64), g["knn_pred"])
        assert np.array_equal(phases.cpu().numpy(), g["phase_out"])
        assert np.array_equal(votes.cpu().numpy(), g["vote"])
    # round 4: nothing about a clip is baked into the capture - other seed codes / phase blocks / inputs replay the SAME
    # graph (the seed is data in pinned host memory, not a kernel argument) and equal the eager path
    for k in range(1, 5):
        sc2, sp2 = (sc + 37 * k) % 512, np.roll(sp, k, axis=0) * (1.0 + 0.25 * k)
        ti2 = te_i.roll(k, 0) + 0.01 * k
        want = knn.match_clip(ti2, te_c, M, seed_code=sc2, seed_phase=sp2)
        codes, phases, votes, status = cg.run(ti2, te_c, sc2, sp2)
        assert status.tolist() == [0, 0]
        assert np.array_equal(codes.numpy().astype(np.int64), want[0])
        assert np.array_equal(phases.cpu().numpy(), want[1]) and np.array_equal(votes.numpy(), want[2])
    assert cg.captures == 1
    with pytest.raises(ValueError):
        cg.run(te_i, te_c, 512, sp)
    # bound to the caller's resident tensors: no copy in front of a replay
    cb = knn.capture_clip_graph(M, audio=te_i, context=te_c)
    ints = cb.run_ints(sc, sp)
    assert np.array_equal(ints[:M * 30].reshape(M, 30), g["knn_pred"]) and ints[-2:].tolist() == [0, 0]


@pytest.mark.parametrize("depth", [1, 2, 3])
def test_clips_in_flight_equal_serial_clips(depth):
    """ClipPipeline: several clips in flight (one lane = CodeKNN + stream + pinned buffer each) return, in order,
    exactly what match_clip returns for each clip on its own - different clips, lengths and seeds, the golden included."""
    import torch
    from qpgesture_amd.code_knn import ClipPipeline
    g = load_golden(GOLDENS[1])
    A, db, knn, te_i, te_c, M = _build(g["meta"], freq_rank=g["step_freq_score"])
    sc, sp = knn.init_code_phase()
    gen = torch.Generator(device="cpu").manual_seed(5)
    clips, seeds = [(te_i, te_c, M)], [(sc, sp)]
    for k in range(6):
        m = 1 + k % M
        perm = torch.randperm(M, generator=gen)[:m].to(te_i.device)
        clips.append((te_i[perm] + 0.01 * k, te_c[perm], m))
        seeds.append(((sc + 17 * k) % 512, np.roll(sp, k, axis=0)))
    want = [knn.match_clip(ti, tc, m, seed_code=s_[0], seed_phase=s_[1]) for (ti, tc, m), s_ in zip(clips, seeds)]
    assert np.array_equal(want[0][0], g["knn_pred"])
    pipe = ClipPipeline(db, depth=depth, rng=np.random.RandomState(1))
    got = pipe.match_clips(clips, seeds=seeds)
    assert len(got) == len(want)
    for w, r in zip(want, got):
        for a, b in zip(w, r):
            assert a.dtype == b.dtype and np.array_equal(a, b)
    t = pipe.submit(*clips[0], seed_code=sc, seed_phase=sp)
    for _ in range(depth - 1):
        pipe.submit(*clips[1], seed_code=sc, seed_phase=sp)
    with pytest.raises(RuntimeError):
        pipe.submit(*clips[0], seed_code=sc, seed_phase=sp)          # every lane holds an uncollected clip
    assert np.array_equal(pipe.collect(t)[0], g["knn_pred"])


def test_one_launch_clip_pack_equals_separate_packs():
    """qpg_clip_pack_hl (round 4: the audio query pack AND the text side's gather / sklearn normalisation / column image in
    one launch) leaves bit-identical buffers and tables to the three separate launches of round 3."""
    import torch
    g = load_golden(GOLDENS[1])
    out = {}
    for fused in (True, False):
        A, db, knn, te_i, te_c, M = _build(g["meta"], freq_rank=g["step_freq_score"])
        knn.fused_pack = fused
        T = knn.sweep_tables(te_i, te_c, M)
        torch.cuda.synchronize()
        Qn = M * knn.n_steps()
        nb_c = int(__import__("qpgesture_amd")._lib.load().qpg_hl_cols_bytes(Qn, db.Dt)) - 4 * 96     # (image; the exponents follow)
        nb_q = int(__import__("qpgesture_amd")._lib.load().qpg_audio_hl_query_bytes(Qn, db.F)) - 4 * 48
        out[fused] = ({k: v.clone() for k, v in T.items() if v is not None}, knn._last_q32.clone(), knn._last_qn2.clone(),
                      knn._txt_scratch["cols"][:nb_c].clone(), knn._hl_qimage[:nb_q].clone(),
                      knn._txt_scratch["cols"][nb_c:nb_c + 4 * Qn].clone(), knn._hl_qimage[nb_q:nb_q + 4 * Qn].clone())
        assert knn._last_audio_hl and knn._last_text_mfma
    for k in out[True][0]:
        assert torch.equal(out[True][0][k], out[False][0][k]), k
    for i, (a, b) in enumerate(zip(out[True][1:], out[False][1:])):
        assert torch.equal(a, b), i


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_tabulated_walk_equals_sequential_walk(mode):
    """qpg_match_steps' tabulated walk (gate evaluated for every reachable (step, previous code, previous vote) in
    parallel, then Q two-byte lookups) vs the one-wave sequential walk (QPG_MODE_SERIAL_WALK): codes, carried phase
    blocks and votes bit-identical, for many seeds and in all three modality modes."""
    g = load_golden(GOLDENS[1])
    A, db, knn, te_i, te_c, M = _build(g["meta"], freq_rank=g["step_freq_score"])
    T = knn.sweep_tables(te_i, te_c, M, mode=mode)
    rs = np.random.RandomState(7)
    n_diff_votes = 0
    for trial in range(12):
        seed_code = int(rs.randint(0, 512))
        seed_phase = rs.standard_normal((8, 16)).astype(np.float32)
        knn.serial_walk = True
        try:
            ref = knn.walk(T, M, mode=mode, seed_code=seed_code, seed_phase=seed_phase)
        except IndexError:
            ref = None
        knn.serial_walk = False
        if ref is None:
            with pytest.raises(IndexError):
                knn.walk(T, M, mode=mode, seed_code=seed_code, seed_phase=seed_phase)
            continue
        got = knn.walk(T, M, mode=mode, seed_code=seed_code, seed_phase=seed_phase)
        assert np.array_equal(got[0], ref[0]) and np.array_equal(got[2], ref[2])
        assert np.array_equal(got[1], ref[1])
        n_diff_votes += int(ref[2].sum() > 0 and ref[2].sum() < ref[2].size)
    assert n_diff_votes > 0                                     # both gate outcomes occur in the sample


def test_long_clip_walk_tabulated_equals_sequential_and_oracle():
    """A 160 s clip (M = 40 windows, Q = 320 steps, window chaining 39 times): the tabulated walk, the sequential walk
    and the oracle's literal restatement of search_code_knn agree on every code."""
    import torch
    from oracle import knn_oracle as O
    from qpgesture_amd.code_knn import CodeKNN, GestureDB
    A = fixture_arrays(40, 40, 31, 32, 33, 34, wavlm_dim=128)
    db = GestureDB(A["code"], A["tr_interp"], A["tr_ctx"], A["tr_phase"], A["sig"], device="cuda:0")
    knn = CodeKNN(db, rng=np.random.RandomState(5))
    te_i = torch.from_numpy(A["te_interp"]).cuda()
    te_c = torch.from_numpy(np.ascontiguousarray(A["te_ctx"])).cuda()
    M = 40
    T = knn.sweep_tables(te_i, te_c, M)
    seed_code, seed_phase = knn.init_code_phase()
    got = knn.walk(T, M, seed_code=seed_code, seed_phase=seed_phase)
    knn.serial_walk = True
    ref = knn.walk(T, M, seed_code=seed_code, seed_phase=seed_phase)
    assert got[0].shape == (M, 30)
    for a, b in zip(got, ref):
        assert np.array_equal(a, b)
    orc = O.CodeKNNOracle(A["code"], A["sig"], A["tr_phase"], A["tr_ctx"], wavlm_interp=A["tr_interp"], scan="c",
                          rank_kind="stable", rng=np.random.RandomState(5))
    want, wph, wv = O.predict_code_from_audio(orc, test_interp=A["te_interp"], test_ctx=A["te_ctx"], n_windows=M)
    assert np.array_equal(got[0], want) and np.array_equal(got[2], wv)


# ---------------------------------------------------------------------------------------------------------------
# round 2: planted ties (fixtures captured from the reference by tests/golden/make_golden.py, variants of
# qpgesture_amd.synth.apply_variant)
# ---------------------------------------------------------------------------------------------------------------
def _build_variant(g, dev="cuda:0"):
    import torch
    from qpgesture_amd.code_knn import CodeKNN, GestureDB
    ntr, nte, s0, s1, s2, s3, mf = [int(v) for v in g["meta"]]
    A = fixture_arrays(ntr, nte, s0, s1, s2, s3, variant=str(g["variant"]))
    db = GestureDB(A["code"], A["tr_interp"], A["tr_ctx"], A["tr_phase"], A["sig"], device=dev,
                   freq_rank=g["step_freq_score"])
    knn = CodeKNN(db, rng=np.random.RandomState(123456))
    te_i = torch.from_numpy(A["te_interp"]).to(dev)
    te_c = torch.from_numpy(np.ascontiguousarray(A["te_ctx"])).to(dev)
    return A, db, knn, te_i, te_c, nte


def test_audio_near_ties_vs_reference_golden():
    """ulp-perturbed duplicate DB windows (and a query that IS a DB window): the reference sees distances 0 and
    ~6e-17, below the rounding noise of the sweep's dot-product form.  The near-tie guard re-evaluates the flagged
    pairs in the reference's arithmetic: winners, the full rank order of the 512 minima and knn_pred are the
    reference's; the refined distances are BIT-EXACT where the guard ran."""
    g = load_golden("shipped_neartie_n48_m2_s30")
    A, db, knn, te_i, te_c, M = _build_variant(g)
    knn.audio_precision = "f64"              # the f64 sweep + guard; the mixed path on this clip: test_gpu_mixed.py
    codes, phases, votes = knn.match_clip(te_i, te_c, M, return_tables=True)
    n_ref, overflow = knn.guard_stats()
    assert n_ref > 0 and not overflow
    T = knn.tables
    aud_d, aud_idx = T["aud_d"].cpu().numpy(), T["aud_idx"].cpu().numpy()
    gj, gk = g["aud_aux"][..., 0], g["aud_aux"][..., 1]
    want_idx = np.where(gj >= 0, gj * 26 + gk // 6, -1)
    assert np.array_equal(aud_idx, want_idx)                                           # winners incl. absent codes
    assert np.abs(aud_d - g["aud_dist"]).max() < 1e-13
    small = g["aud_dist"] < 1e-12
    assert small.sum() >= 2 * 14 and np.array_equal(aud_d[small], g["aud_dist"][small])   # refined values: bit-exact
    present = g["aud_dist"] != 1e3           # the 6 absent codes tie at 1e+3: NumPy's unstable order, not compared
    assert np.array_equal(T["aud_rank"].cpu().numpy()[present], g["step_aud_score"][present])   # the reference's ranks
    assert np.array_equal(T["txt_d"].cpu().numpy(), g["txt_dist"])
    assert np.array_equal(codes, g["knn_pred"]) and np.array_equal(votes, g["vote"])
    # without the guard the same clip is decided by rounding noise somewhere (recorded, not asserted: it is noise)
    knn2 = type(knn)(db, rng=np.random.RandomState(123456))
    knn2.tie_eps = 0.0
    knn2.match_clip(te_i, te_c, M, return_tables=True)
    r2 = knn2.tables["aud_rank"].cpu().numpy()
    print("rank rows differing without the guard: %d of %d"
          % (((r2 != g["step_aud_score"]) & present).any(axis=1).sum(), r2.shape[0]))


def test_text_exact_ties_vs_reference_golden():
    """Repeated context rows (silent frames share one embedding): 296 of 512 codes tie at distance exactly 0 in every
    step.  The reference ranks them with NumPy's unstable argsort; `host_ranks` (= the CLI's --tie_rule numpy) calls
    the same expression on the host, so on a machine whose NumPy sorts like the golden's the final codes are the
    reference's (checked first; otherwise the test is skipped - that order is CPU-dispatch dependent upstream)."""
    from qpgesture_amd.code_knn import CodeKNN
    g = load_golden("shipped_texttie_n48_m2_s40")
    here = np.stack([np.array(list(r.astype(np.float64))).argsort().argsort() for r in g["txt_dist"]])
    if not np.array_equal(here, g["step_txt_score"]):
        pytest.skip("this host's NumPy orders exact ties differently from the machine that captured the golden")
    A, db, knn, te_i, te_c, M = _build_variant(g)
    knn.host_ranks = True
    codes, phases, votes = knn.match_clip(te_i, te_c, M, return_tables=True)
    T = knn.tables
    assert np.array_equal(T["txt_d"].cpu().numpy(), g["txt_dist"]) and (g["txt_dist"] == 0).sum(axis=1).min() > 200
    gj, gk = g["txt_aux"][..., 0], g["txt_aux"][..., 1]
    assert np.array_equal(T["txt_idx"].cpu().numpy(), np.where(gj >= 0, gj * 26 + gk // 8, -1))
    assert np.array_equal(T["txt_rank"].cpu().numpy(), g["step_txt_score"])
    assert np.array_equal(T["aud_rank"].cpu().numpy(), g["step_aud_score"])
    assert np.array_equal(codes, g["knn_pred"]) and np.array_equal(votes, g["vote"])
    # the deterministic rule gives the same minima and winners, only the tied ranks differ
    knn2 = CodeKNN(db, rng=np.random.RandomState(123456))
    knn2.match_clip(te_i, te_c, M, return_tables=True)
    assert np.array_equal(knn2.tables["txt_idx"].cpu().numpy(), T["txt_idx"].cpu().numpy())
    r2 = knn2.tables["txt_rank"].cpu().numpy()
    uniq = np.stack([np.isin(r, np.unique(r)[np.unique(r, return_counts=True)[1] == 1]) for r in g["txt_dist"]])
    assert uniq.sum() > 0 and np.array_equal(r2[uniq], g["step_txt_score"][uniq])
    zero = g["txt_dist"] == 0
    assert (r2[zero] < zero.sum(axis=1).max()).all()


def test_second_device_when_present():
    """a GestureDB on cuda:1 while the caller's current device is cuda:0 (every C entry point launches on
    the CURRENT device: _lib.call switches).  Skipped on 1-GPU boxes."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    g = load_golden(GOLDENS[0])
    torch.cuda.set_device(0)
    A, db, knn, te_i, te_c, M = _build(g["meta"], dev="cuda:1", freq_rank=g["step_freq_score"])
    assert torch.cuda.current_device() == 0
    codes, _, _ = knn.match_clip(te_i, te_c, M)
    assert np.array_equal(codes, g["knn_pred"]) and torch.cuda.current_device() == 0


def test_long_wavvq_clip_chunks_the_query_strings():
    """more than 1489 query strings used to exceed the LDS of one launch; qpg_wavvq_lev_f32 now walks the
    queries in chunks of 1024.  2000 queries against a small DB: identical to the same queries issued 500 at a time."""
    import torch
    from qpgesture_amd import synth
    from qpgesture_amd.code_knn import CodeKNN, GestureDB
    tr = synth.make_db(24, 90, 8)
    te = synth.make_db(250, 91, 8)
    from oracle import knn_oracle as O
    db = GestureDB(synth.make_codes(24, 92), O.interp_wavlm(tr["wavlm"]), tr["context"].squeeze(2), tr["phase_dense"],
                   synth.make_signature(93), device="cuda:0", wavvq=tr["wavvq"])
    knn = CodeKNN(db, use_wavlm=False, use_wavvq=True, rng=np.random.RandomState(2))
    q_win = np.repeat(np.arange(250), 8)
    q_t = np.tile(np.array([int(i) for i in knn.query_positions()]), 250)
    d_all, i_all = knn.sweep_audio_wavvq(te["wavvq"], q_win, q_t)
    assert d_all.shape == (2000, 512)
    for lo in range(0, 2000, 500):
        d, i = knn.sweep_audio_wavvq(te["wavvq"], q_win[lo:lo + 500], q_t[lo:lo + 500])
        assert torch.equal(d, d_all[lo:lo + 500]) and torch.equal(i, i_all[lo:lo + 500])
