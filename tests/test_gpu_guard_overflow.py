"""The near-tie guard must never be silently unguarded: every capped re-evaluation list of the fast
audio paths (guarded f64 select: 256 entries; mixed-precision select: 2048 / 256; cross-shard requests) raises a trouble
word when it overflows, the word leaves the device with the codes, and the host re-matches the clip on the UNCAPPED path
(audio_precision "exact": qpg_percode_select_exact_f64, across shards a reference-arithmetic request round).  These tests
force every overflow and hold the results to the reference itself (a golden captured from /root/reference on a
near-silent stretch: 780 candidates within ~1e-14 of each other) and to the C oracle (reference arithmetic) at sizes
no list can hold."""
import numpy as np
import pytest

from tests.helpers import fixture_arrays, load_golden

pytestmark = pytest.mark.gpu

NEARSILENT = "shipped_nearsilent_n48_m2_s50"


def _build(g, dev="cuda:0"):
    import torch
    from qpgesture_amd.code_knn import CodeKNN, GestureDB
    ntr, nte, s0, s1, s2, s3, mf = [int(v) for v in g["meta"]]
    variant = (str(g["variant"]) or None) if "variant" in g.files else None
    A = fixture_arrays(ntr, nte, s0, s1, s2, s3, variant=variant)
    db = GestureDB(A["code"], A["tr_interp"], A["tr_ctx"], A["tr_phase"], A["sig"], device=dev,
                   freq_rank=g["step_freq_score"])
    knn = CodeKNN(db, rng=np.random.RandomState(123456))
    te_i = torch.from_numpy(A["te_interp"]).to(dev)
    te_c = torch.from_numpy(np.ascontiguousarray(A["te_ctx"])).to(dev)
    return A, db, knn, te_i, te_c, nte


def _unique_mask(d):
    """True where a row's value occurs once: the rank of such an entry does not depend on how ties are ordered."""
    out = np.zeros(d.shape, bool)
    for r, row in enumerate(d):
        u, inv, cnt = np.unique(row, return_inverse=True, return_counts=True)
        out[r] = cnt[inv] == 1
    return out


def _check_against_golden(g, T):
    aud_d, aud_idx = T["aud_d"].cpu().numpy(), T["aud_idx"].cpu().numpy()
    gj, gk = g["aud_aux"][..., 0], g["aud_aux"][..., 1]
    assert np.array_equal(aud_idx, np.where(gj >= 0, gj * 26 + gk // 6, -1))           # the REFERENCE's winners
    assert np.abs(aud_d - g["aud_dist"]).max() < 1e-13
    small = g["aud_dist"] < 1e-12
    assert small.sum() >= 8 * 170 and np.array_equal(aud_d[small], g["aud_dist"][small])   # reference arithmetic: bit-exact
    uniq = _unique_mask(g["aud_dist"])             # (five codes tie EXACTLY in the quiet steps: NumPy's unstable order)
    assert np.array_equal(T["aud_rank"].cpu().numpy()[uniq], g["step_aud_score"][uniq])
    assert np.array_equal(T["txt_d"].cpu().numpy(), g["txt_dist"])


@pytest.mark.parametrize("prec", ["mixed", "f64", "exact"])
def test_near_silent_stretch_vs_reference_golden(prec):
    """30 DB windows and one query window of near-identical quiet frames: 780 candidates (about 130 per crowded code,
    176 codes' minima) within ~1e-14 of each other.  The capped selects overflow (flag), the clip is re-matched on the
    uncapped path, and winners / refined distances / ranks are the reference's."""
    g = load_golden(NEARSILENT)
    A, db, knn, te_i, te_c, M = _build(g)
    knn.audio_precision = prec
    codes, phases, votes = knn.match_clip(te_i, te_c, M, return_tables=True)
    assert knn.fallbacks == (0 if prec == "exact" else 1), "the capped lists must overflow on this clip"
    assert knn.audio_precision == prec and knn.mixed_stats()["flags"] == 0          # restored, word cleared
    _check_against_golden(g, knn.tables)
    # the final codes also depend on the order of the EXACT ties (NumPy's unstable sort upstream): --tie_rule numpy
    here = np.stack([np.array(list(r)).argsort().argsort() for r in g["aud_dist"]])
    if np.array_equal(here, g["step_aud_score"]):
        knn2 = type(knn)(db, rng=np.random.RandomState(123456))
        knn2.audio_precision, knn2.host_ranks = prec, True
        codes2, _, votes2 = knn2.match_clip(te_i, te_c, M)
        assert np.array_equal(codes2, g["knn_pred"]) and np.array_equal(votes2, g["vote"])


def test_walk_raises_instead_of_returning_unguarded_codes():
    """Callers below match_clip (sweep_tables + walk) get GuardOverflow, never codes, for a flagged clip."""
    from qpgesture_amd.code_knn import FLAG_LIST_OVERFLOW, GuardOverflow
    g = load_golden(NEARSILENT)
    A, db, knn, te_i, te_c, M = _build(g)
    T = knn.sweep_tables(te_i, te_c, M)
    sc, sp = knn.init_code_phase()
    with pytest.raises(GuardOverflow) as e:
        knn.walk(T, M, 0, seed_code=sc, seed_phase=sp)
    assert e.value.flags & FLAG_LIST_OVERFLOW
    # asynchronous callers find the word in the status they must check
    oc, op, ov, st = knn.walk(T, M, 0, seed_code=sc, seed_phase=sp, sync=False)
    assert st.cpu().tolist()[1] != 0


def test_exact_path_equals_guarded_path_on_ordinary_data():
    """Where nothing overflows the uncapped select returns the guarded select's tables bit for bit."""
    import torch
    for name in ("shipped_n48_m2_s0", "shipped_neartie_n48_m2_s30"):
        g = load_golden(name)
        out = {}
        for prec in ("f64", "exact"):
            A, db, knn, te_i, te_c, M = _build(g)
            knn.audio_precision = prec
            out[prec] = (knn.match_clip(te_i, te_c, M, return_tables=True), knn.tables, knn.guard_stats(), knn.fallbacks)
        (ca, _, va), Ta, sa, fa = out["f64"]
        (cb, _, vb), Tb, sb, fb = out["exact"]
        assert fa == 0 and fb == 0 and not sa[1] and not sb[1] and sa[0] == sb[0]
        for k in ("aud_d", "aud_idx", "aud_rank"):
            assert torch.equal(Ta[k], Tb[k]), (name, k)
        assert np.array_equal(ca, cb) and np.array_equal(va, vb) and np.array_equal(cb, g["knn_pred"])


def _crowded(n_copies, scale=1.0, seed=80):
    """A 64-window DB plus `n_copies` eps-perturbed copies of window 5 carrying window 5's codes; test window 0 IS
    window 5.  Every query step of that window then has n_copies + 1 candidates of ONE code within ~1e-12."""
    A = fixture_arrays(64, 2, seed, seed + 1, seed + 2, seed + 3)
    rng = np.random.Generator(np.random.PCG64(seed + 4))
    src = A["tr_interp"][5]
    extra = np.empty((n_copies,) + src.shape, np.float32)
    for lo in range(0, n_copies, 256):
        n = min(256, n_copies - lo)
        eps = (10.0 ** rng.uniform(-7.3, -6.3, size=(n, 1, 1))).astype(np.float32)
        extra[lo:lo + n] = src * (1.0 + eps * rng.standard_normal((n,) + src.shape, dtype=np.float32))
    A["tr_interp"] = np.concatenate((A["tr_interp"], extra)) * np.float32(scale)
    A["code"] = np.concatenate((A["code"], np.repeat(A["code"][5:6], n_copies, axis=0)))
    A["tr_ctx"] = np.concatenate((A["tr_ctx"], rng.standard_normal((n_copies,) + A["tr_ctx"].shape[1:], dtype=np.float32)))
    A["tr_phase"] = np.concatenate((A["tr_phase"],
                                    rng.standard_normal((n_copies,) + A["tr_phase"].shape[1:], dtype=np.float32)))
    A["te_interp"][0] = src
    A["te_interp"] = A["te_interp"] * np.float32(scale)
    return A


def _oracle_tables(A, M):
    import os
    from oracle import cref, knn_oracle as O
    q = np.stack([O.wavlm_feat_rows(A["te_interp"], w, [24 * s])[0] for w in range(M) for s in range(8)])
    return cref.audio_scan(A["tr_interp"], np.arange(26) * 6, A["code"], np.arange(26), q,
                           n_threads=min(32, os.cpu_count() or 1))


def _match(A, prec, dev="cuda:0", kernel="hl"):
    import torch
    from qpgesture_amd.code_knn import CodeKNN, GestureDB
    db = GestureDB(A["code"], A["tr_interp"], A["tr_ctx"], A["tr_phase"], A["sig"], device=dev)
    knn = CodeKNN(db, rng=np.random.RandomState(7))
    knn.audio_precision = prec
    knn.audio_kernel = kernel
    te_i = torch.from_numpy(A["te_interp"]).to(dev)
    te_c = torch.from_numpy(np.ascontiguousarray(A["te_ctx"])).to(dev)
    out = knn.match_clip(te_i, te_c, 2, return_tables=True)
    return knn, out


@pytest.mark.parametrize("prec", ["mixed", "mixed-mx", "f64"])
def test_three_thousand_near_copies_of_one_window_vs_c_oracle(prec):
    """>= 3 000 eps-perturbed copies of one window under one code (VERDICT r2 next #1a): every list cap is exceeded
    (tier 1: 2 048, tier 2 / guarded: 256).  After the re-match the tables are the C oracle's (reference arithmetic):
    winners, the stable rank order of the 512 minima, refined distances bit for bit; codes == the exact path's own."""
    A = _crowded(3000)
    d_ref, i_ref = _oracle_tables(A, 2)
    knn, (codes, _, votes) = _match(A, prec.split("-")[0], kernel="mx" if prec.endswith("mx") else "hl")
    assert knn.fallbacks == 1 and knn.mixed_stats()["flags"] == 0
    T = knn.tables
    assert np.array_equal(T["aud_idx"].cpu().numpy(), i_ref)
    aud_d = T["aud_d"].cpu().numpy()
    assert np.abs(aud_d - d_ref).max() < 1e-13
    small = d_ref < 1e-12
    assert np.array_equal(aud_d[small], d_ref[small])
    want_rank = np.argsort(np.argsort(d_ref, axis=1, kind="stable"), axis=1, kind="stable")
    assert np.array_equal(T["aud_rank"].cpu().numpy(), want_rank)
    kx, (codes_x, _, votes_x) = _match(A, "exact")
    assert kx.fallbacks == 0 and np.array_equal(codes, codes_x) and np.array_equal(votes, votes_x)
    assert kx.guard_stats()[0] >= 8 * 2000            # the band of the crowded code went through reference arithmetic


def test_norms_below_the_error_bound_s_range_rematch():
    """f32-matrix-core sweep (audio_kernel "mx"): operand norms scaled to ~1e-10 (|q||c| < 1e-16) void its a-priori bound
    (f32 products may underflow): flag 2, re-match on the exact path, tables == the C oracle's on the scaled data (cosine
    distance is scale-free).  The split-f16 sweep scales by powers of two first: the same data is inside ITS range."""
    from qpgesture_amd.code_knn import FLAG_SMALL_NORMS, GuardOverflow
    A = _crowded(8, scale=1e-11)
    d_ref, i_ref = _oracle_tables(A, 2)
    khl, _ = _match(A, "mixed")
    assert khl.fallbacks == 0 and khl._last_audio_hl and np.array_equal(khl.tables["aud_idx"].cpu().numpy(), i_ref)
    knn, (codes, _, _) = _match(A, "mixed", kernel="mx")
    assert knn.fallbacks == 1
    assert np.array_equal(knn.tables["aud_idx"].cpu().numpy(), i_ref)
    assert np.abs(knn.tables["aud_d"].cpu().numpy() - d_ref).max() < 1e-12
    # and the flag that caused it is the norm check's
    import torch
    te_i = torch.from_numpy(A["te_interp"]).cuda()
    te_c = torch.from_numpy(np.ascontiguousarray(A["te_ctx"])).cuda()
    T = knn.sweep_tables(te_i, te_c, 2)
    sc, sp = knn.init_code_phase()
    with pytest.raises(GuardOverflow) as e:
        knn.walk(T, 2, 0, seed_code=sc, seed_phase=sp)
    assert e.value.flags & FLAG_SMALL_NORMS
    knn.clear_flags()


def test_split_f16_sweep_rematches_what_is_outside_its_range():
    """Split-f16 sweep: a database window 1e-7 of the loudest value has a scaled norm < 1, where the representation bound
    does not hold: flag 2, re-match, tables == the C oracle's."""
    A = _crowded(8)
    A["tr_interp"][20] *= np.float32(1e-7)
    A["tr_interp"][33, 40:90] *= np.float32(50.0)
    d_ref, i_ref = _oracle_tables(A, 2)
    knn, _ = _match(A, "mixed")
    assert knn.fallbacks == 1
    assert np.array_equal(knn.tables["aud_idx"].cpu().numpy(), i_ref)
    assert np.abs(knn.tables["aud_d"].cpu().numpy() - d_ref).max() < 1e-12


def test_clips_in_flight_rematch_flagged_clips():
    """ClipPipeline.collect never hands out a flagged clip either: it re-matches it from the inputs the lane kept."""
    from qpgesture_amd.code_knn import ClipPipeline
    g = load_golden(NEARSILENT)
    A, db, knn, te_i, te_c, M = _build(g)
    g0 = load_golden("shipped_n48_m2_s0")
    sc, sp = knn.init_code_phase()
    want = knn.match_clip(te_i, te_c, M, seed_code=sc, seed_phase=sp)
    assert knn.fallbacks == 1
    pipe = ClipPipeline(db, depth=2, rng=np.random.RandomState(1))
    clips = [(te_i, te_c, M), (te_i[1:], te_c[1:], 1), (te_i, te_c, M)]      # quiet, ordinary, quiet
    got = pipe.match_clips(clips, seeds=[(sc, sp)] * 3)
    assert pipe.fallbacks == 2
    for a, b in zip(got[0], want):
        assert np.array_equal(a, b)
    for a, b in zip(got[2], want):
        assert np.array_equal(a, b)
    alone = knn.match_clip(te_i[1:], te_c[1:], 1, seed_code=sc, seed_phase=sp)
    assert knn.fallbacks == 1                                                  # the ordinary clip needs no re-match
    for a, b in zip(got[1], alone):
        assert np.array_equal(a, b)


# ---- row shards: the cross-shard protocol with the byte exchanges done by hand (all-gather form, owner = shard 0) ----
def _shard_tables(A, W, te_i, prec, q_win, q_t, Q):
    import torch
    from qpgesture_amd.code_knn import CodeKNN, ExchangeLayout, GestureDB
    dev = torch.device("cuda:0")
    shards, lays = [], []
    for r in range(W):
        db = GestureDB(A["code"], A["tr_interp"], A["tr_ctx"], A["tr_phase"], A["sig"], device=dev, rank=r, world=W)
        knn = CodeKNN(db, rng=np.random.RandomState(1))
        knn.audio_precision = prec
        knn.sharded_mixed_min_gflop = 0.0
        lay = ExchangeLayout(Q, db.K, 1, ["aud"], True, dev)
        knn.sweep_audio(te_i, q_win, q_t, reduce=False, out=lay.views("aud"))
        shards.append(knn)
        lays.append(lay)
    return shards, lays


def _protocol(shards, lays, Q, band, R, fl_cap, reference_arithmetic, eps2):
    import torch
    from qpgesture_amd import _lib
    from qpgesture_amd.code_knn import ABSENT_DIST
    dev = torch.device("cuda:0")
    W, K = len(shards), shards[0].db.K
    recv = torch.cat([l.send for l in lays])
    src_stride = lays[0].send.numel()
    req_stride, resp_stride = 8 + 8 * R, 8 + 8 * R
    req = torch.zeros((W * req_stride,), dtype=torch.uint8, device=dev)
    ws = torch.empty((int(_lib.load().qpg_merge_mixed_ws_bytes(Q, K, fl_cap)),), dtype=torch.uint8, device=dev)
    stats = torch.zeros((4,), dtype=torch.int32, device=dev)
    _lib.call("qpg_merge_mixed_phase1_f64", dev, recv, W, src_stride, lays[0].off["aud_d"], lays[0].off["aud_i"], Q, K,
              float(ABSENT_DIST), band, R, req, req_stride, ws, ws.numel(), stats, fl_cap, -1)
    counts = [int(req[w * req_stride:w * req_stride + 4].view(torch.int32)[0]) for w in range(W)]   # header: count | flags
    resp_recv = torch.zeros((W * resp_stride,), dtype=torch.uint8, device=dev)
    for w in range(W):
        req_recv = torch.full((W * req_stride,), 255, dtype=torch.uint8, device=dev)      # (unused slots / blocks: ~0)
        for b in range(W):
            req_recv[b * req_stride:b * req_stride + 8] = 0                       # headers: count | flags
        req_recv[:req_stride] = req[w * req_stride:(w + 1) * req_stride]
        resp = torch.zeros((W * resp_stride,), dtype=torch.uint8, device=dev)
        k, db = shards[w], shards[w].db
        _lib.call("qpg_shard_refine_f64", dev, req_recv, W, req_stride, R, 0, db.idx_base * db.Ga, db.base, 0, db.T, db.F,
                  db.aud_t, db.Ga, 6, db.tap_stride, k._last_q32, k._last_qn2, db.cn2, resp, resp_stride,
                  int(reference_arithmetic), None, R // Q)
        resp_recv[w * resp_stride:(w + 1) * resp_stride] = resp[:resp_stride]
    d = torch.empty((Q, K), dtype=torch.float64, device=dev)
    ix = torch.empty((Q, K), dtype=torch.int32, device=dev)
    rk = torch.empty((Q, K), dtype=torch.int16, device=dev)
    _lib.call("qpg_merge_mixed_phase2_f64", dev, recv, W, src_stride, lays[0].off["aud_i"], Q, K, float(ABSENT_DIST), ws,
              ws.numel(), resp_recv, resp_stride, d, ix, rk, stats, fl_cap, eps2)
    torch.cuda.synchronize()
    return d, ix, rk, stats.cpu().numpy(), counts, recv


def test_cross_shard_tier2_on_the_near_silent_golden():
    """The uncapped path across two row shards (windows 0..23 | 24..47: the quiet stretch 8..37 straddles the boundary):
    per-shard uncapped select, then the request round with band = 1e-12 and REFERENCE-ARITHMETIC responses (cross-shard
    tier 2).  Winners, refined distances (bit-exact) and ranks are the reference's - what round 2 left per shard."""
    import torch
    g = load_golden(NEARSILENT)
    A, db, knn, te_i, te_c, M = _build(g)
    steps = knn.n_steps()
    q_win, q_t = np.repeat(np.arange(M), steps), np.tile(np.arange(steps) * 24, M)
    Q, K, W = M * steps, db.K, 2
    shards, lays = _shard_tables(A, W, te_i, "exact", q_win, q_t, Q)
    assert all(s._last_audio_exact for s in shards)
    d, ix, rk, st, counts, _ = _protocol(shards, lays, Q, 1e-12, Q * K, K * W, True, 0.0)
    print("reference-arithmetic requests per shard:", counts)
    assert st[1] == 0 and min(counts) > 0 and sum(counts) > 1000
    _check_against_golden(g, dict(aud_d=d, aud_idx=ix, aud_rank=rk,
                                  txt_d=torch.from_numpy(g["txt_dist"])))          # (text side not under test here)


def test_sharded_fast_paths_flag_the_near_silent_clip():
    """The fast sharded paths must notice what they cannot decide: mixed-precision shards + dot-product responses raise
    the trouble word (a shard's capped list overflows, and / or contenders from different shards stay within 1e-12);
    f64 shards + plain merge raise FLAG_CROSS_SHARD_TIE."""
    import torch
    from qpgesture_amd import _lib
    from qpgesture_amd.code_knn import ABSENT_DIST, AUDIO_MX_BAND, FLAG_CROSS_SHARD_TIE
    g = load_golden(NEARSILENT)
    A, db, knn, te_i, te_c, M = _build(g)
    steps = knn.n_steps()
    q_win, q_t = np.repeat(np.arange(M), steps), np.tile(np.arange(steps) * 24, M)
    Q, K, W = M * steps, db.K, 2
    shards, lays = _shard_tables(A, W, te_i, "mixed", q_win, q_t, Q)
    assert all(s._last_audio_mixed for s in shards)
    shard_flags = int(np.bitwise_or.reduce([s.mixed_stats()["flags"] for s in shards]))
    d, ix, rk, st, counts, _ = _protocol(shards, lays, Q, AUDIO_MX_BAND, 4096, 1024, False, 1e-12)
    print("shard flags 0x%x, owner flags 0x%x, requests %s" % (shard_flags, int(st[1]), counts))
    assert (shard_flags | int(st[1])) != 0
    assert int(st[1]) & FLAG_CROSS_SHARD_TIE
    # f64 shards, one-exchange merge
    shards, lays = _shard_tables(A, W, te_i, "f64", q_win, q_t, Q)
    recv = torch.cat([l.send for l in lays])
    dev = recv.device
    od = torch.empty((Q, K), dtype=torch.float64, device=dev)
    oi = torch.empty((Q, K), dtype=torch.int32, device=dev)
    ork = torch.empty((Q, K), dtype=torch.int16, device=dev)
    stats = torch.zeros((4,), dtype=torch.int32, device=dev)
    _lib.call("qpg_merge_select_f64", dev, recv, W, lays[0].send.numel(), lays[0].off["aud_d"], lays[0].off["aud_i"],
              Q, K, float(ABSENT_DIST), od, oi, ork, 1e-12, stats)
    assert int(stats.cpu()[1]) & FLAG_CROSS_SHARD_TIE
    # ... and stays silent on an ordinary clip
    g0 = load_golden("shipped_n48_m2_s0")
    A0, db0, knn0, te0, _, M0 = _build(g0)
    shards, lays = _shard_tables(A0, W, te0, "f64", q_win, q_t, Q)
    recv = torch.cat([l.send for l in lays])
    stats.zero_()
    _lib.call("qpg_merge_select_f64", dev, recv, W, lays[0].send.numel(), lays[0].off["aud_d"], lays[0].off["aud_i"],
              Q, K, float(ABSENT_DIST), od, oi, ork, 1e-12, stats)
    assert int(stats.cpu()[1]) == 0
    gj, gk = g0["aud_aux"][..., 0], g0["aud_aux"][..., 1]
    assert np.array_equal(oi.cpu().numpy(), gj * 26 + gk // 6)


def test_request_overflow_is_flagged_and_leaves_no_stale_entries():
    """with tiny request / flag lists the owner used to count slots it never wrote and phase 2
    decoded stale workspace words.  Now every counted entry is written, the surplus is dropped, flag 4 is raised, and
    every output index is still one of the shards' candidates for that (query, code)."""
    import torch
    from qpgesture_amd.code_knn import AUDIO_MX_BAND, FLAG_REQUEST_OVERFLOW
    ntr, nte, W = 120, 2, 2
    A = fixture_arrays(ntr, nte, 70, 71, 72, 73)
    rng = np.random.Generator(np.random.PCG64(9))
    x, code = A["tr_interp"], A["code"]
    for i in range(30):
        src, dst = i % 6, 60 + i
        eps = 0.0 if i % 7 == 0 else 10.0 ** rng.uniform(-7.2, -4.0)
        x[dst] = (x[src] * (1.0 + eps * rng.standard_normal(x[src].shape))).astype(np.float32)
        if i % 2 == 0:
            code[dst] = code[src]
    te_i = torch.from_numpy(A["te_interp"]).cuda()
    q_win, q_t = np.repeat(np.arange(nte), 8), np.tile(np.arange(8) * 24, nte)
    Q = nte * 8
    shards, lays = _shard_tables(A, W, te_i, "mixed", q_win, q_t, Q)
    K = shards[0].db.K
    for R, fl_cap in ((16, 1024), (4096, 4), (16, 4)):          # (R = Q x slots per (query, shard))
        # poison the workspace the way torch.empty may: stale words must never be decoded
        d, ix, rk, st, counts, recv = _protocol(shards, lays, Q, AUDIO_MX_BAND, R, fl_cap, False, 1e-12)
        assert int(st[1]) & FLAG_REQUEST_OVERFLOW, (R, fl_cap)
        cand = torch.stack([recv[w * lays[0].send.numel():][lays[0].off["aud_i"]:lays[0].off["aud_i"] + Q * K * 4]
                            .view(torch.int32).view(Q, K) for w in range(W)])
        ok = (ix.unsqueeze(0) == cand).any(dim=0) | ((ix < 0) & (cand < 0).all(dim=0))
        assert bool(ok.all()), (R, fl_cap)
        assert bool(((rk >= 0) & (rk < K)).all())
