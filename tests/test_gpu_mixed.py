"""Mixed-precision audio path (qpg_audio_cosine_mx + qpg_percode_select_mixed_f64, the default of CodeKNN.sweep_audio
on one GPU): (1) the sweep's a-priori error bound holds, measured; (2) winners, ranks and knn_pred equal the f64 path's
— on the reference goldens, on the planted near-tie golden, and on a crowded DB built to fill the re-evaluation band."""
import numpy as np
import pytest

from tests.helpers import fixture_arrays, load_golden

pytestmark = pytest.mark.gpu


def _sweeps(N=96, Q=48, seed=0, half=False):
    import torch
    from qpgesture_amd import _lib
    dev = torch.device("cuda:0")
    T, F, G = 180, 1024, 26
    g = torch.Generator(device="cpu").manual_seed(seed)
    base = torch.randn((N, T, F), generator=g).to(dev)
    base[3] = 0.0                                              # an all-zero window (sklearn's degenerate-row rule)
    base[5, 100:] = 0.0
    q32 = torch.randn((Q, 6 * F), generator=g).to(dev)
    if Q > 11:
        q32[7] = 0.0
        q32[9] = base[11, 12:24:2].reshape(-1)                 # a query that IS a candidate (distance ~ 0)
        q32[10] *= 1e-3                                        # scale must not matter
        q32[11] *= 1e3
    cand_t = (torch.arange(G, dtype=torch.int32) * 6).to(dev)
    fn2 = torch.empty((N, T), dtype=torch.float64, device=dev)
    if half:                                                   # f16 storage: everything is defined on the rounded track
        base_h = base.to(torch.float16).contiguous()
        base = base_h.float()
    _lib.call("qpg_frame_norm2_f64", dev, base, N * T, F, fn2)
    cn2 = torch.empty((N, G), dtype=torch.float64, device=dev)
    _lib.call("qpg_audio_cand_norm2", dev, fn2, N, T, cand_t, G, 6, 2, cn2)
    qn2 = (q32.double() ** 2).sum(1)
    D64 = torch.empty((Q, N * G), dtype=torch.float64, device=dev)
    Dmx = torch.empty_like(D64)
    stats = torch.zeros((4,), dtype=torch.int32, device=dev)
    _lib.call("qpg_audio_cosine_f64", dev, base, N, T, F, cand_t, G, 6, 2, cn2, q32, qn2, Q, D64, D64.stride(0))
    if half:
        _lib.call("qpg_audio_cosine_mx_h", dev, base_h, N, T, F, cand_t, G, 6, 2, cn2, q32, qn2, Q, Dmx, 0, Dmx.stride(0), stats)
    else:
        _lib.call("qpg_audio_cosine_mx", dev, base, N, T, F, cand_t, G, 6, 2, cn2, q32, qn2, Q, Dmx, 0, Dmx.stride(0), stats)
    # the f32-stored matrix (what CodeKNN uses): the same values rounded once
    D32 = torch.empty((Q, N * G), dtype=torch.float32, device=dev)
    _lib.call("qpg_audio_cosine_mx_h" if half else "qpg_audio_cosine_mx", dev, base_h if half else base, N, T, F, cand_t, G,
              6, 2, cn2, q32, qn2, Q, D32, 1, D32.stride(0), stats)
    torch.cuda.synchronize()
    assert torch.equal(D32, Dmx.float())
    return D64.cpu().numpy(), D32.double().cpu().numpy(), stats.cpu().numpy()


def test_mixed_sweep_stays_inside_its_error_bound():
    from qpgesture_amd.code_knn import AUDIO_MX_ERR
    # N = 96: split-K organisation only (four query-tile shapes + ragged); Q = 200: the LDS-shared-query organisation
    # (mx2) with a ragged last query tile; N = 700: both in one call (256 mx2 blocks + a split-K remainder)
    # the last three: the base stored in f16 (qpg_audio_cosine_mx_h), each organisation
    for Q, N, half in ((48, 96, False), (16, 96, False), (64, 96, False), (5, 96, False), (200, 96, False),
                       (48, 700, False), (100, 333, False), (48, 96, True), (200, 96, True), (48, 700, True)):
        D64, Dmx, stats = _sweeps(N=N, Q=Q, seed=Q, half=half)
        err = np.abs(D64 - Dmx)
        print("N=%d Q=%d%s: max |D_mx - D_f64| = %.3g (bound %.3g), mean %.3g"
              % (N, Q, " f16 base" if half else "", err.max(), AUDIO_MX_ERR, err.mean()))
        assert err.max() <= AUDIO_MX_ERR
        assert stats[1] == 0
        if Q > 11:
            assert np.array_equal(D64[7], Dmx[7])              # zero query row: exact in both (0.5 is an f32 number)
        assert np.array_equal(D64[:, 3 * 26:4 * 26], Dmx[:, 3 * 26:4 * 26])   # zero candidate rows: exact in both


def _build(A, freq_rank, precision, dev="cuda:0"):
    import torch
    from qpgesture_amd.code_knn import CodeKNN, GestureDB
    db = GestureDB(A["code"], A["tr_interp"], A["tr_ctx"], A["tr_phase"], A["sig"], device=dev, freq_rank=freq_rank)
    knn = CodeKNN(db, rng=np.random.RandomState(123456))
    knn.audio_precision = "mixed" if precision == "mixed1" else precision
    knn.mixed_single_launch = precision == "mixed1"
    te_i = torch.from_numpy(A["te_interp"]).to(dev)
    te_c = torch.from_numpy(np.ascontiguousarray(A["te_ctx"])).to(dev)
    return knn, te_i, te_c


def _compare(A, freq_rank, M, want_activity):
    from qpgesture_amd.code_knn import AUDIO_MX_ERR
    out = {}
    for prec in ("f64", "mixed", "mixed1"):      # mixed1: the select as ONE launch (no workspace)
        knn, te_i, te_c = _build(A, freq_rank, prec)
        codes, phases, votes = knn.match_clip(te_i, te_c, M, return_tables=True)
        T = knn.tables
        out[prec] = dict(codes=codes, votes=votes, d=T["aud_d"].cpu().numpy(), idx=T["aud_idx"].cpu().numpy(),
                         rank=T["aud_rank"].cpu().numpy(), stats=knn.mixed_stats())
    a, b, b1 = out["f64"], out["mixed"], out["mixed1"]
    for key in ("idx", "rank", "codes", "votes", "d"):
        assert np.array_equal(b[key], b1[key]), key          # three launches == one launch, bit for bit
    assert b["stats"] == b1["stats"]
    print("mixed stats:", b["stats"], " max |d_mixed - d_f64| = %.3g" % np.abs(a["d"] - b["d"]).max())
    assert b["stats"]["flags"] == 0 and a["stats"]["tier1_pairs"] == 0
    assert b["stats"]["tier1_pairs"] >= want_activity
    assert np.array_equal(a["idx"], b["idx"])
    assert np.array_equal(a["rank"], b["rank"])
    assert np.array_equal(a["codes"], b["codes"]) and np.array_equal(a["votes"], b["votes"])
    assert np.abs(a["d"] - b["d"]).max() <= AUDIO_MX_ERR
    return out


@pytest.mark.parametrize("name", ["shipped_n48_m2_s0", "shipped_n64_m3_s10"])
def test_mixed_equals_f64_path_and_reference_on_goldens(name):
    g = load_golden(name)
    ntr, nte, s0, s1, s2, s3, mf = [int(v) for v in g["meta"]]
    A = fixture_arrays(ntr, nte, s0, s1, s2, s3)
    out = _compare(A, g["step_freq_score"], nte, want_activity=0)
    m = out["mixed"]
    gj, gk = g["aud_aux"][..., 0], g["aud_aux"][..., 1]
    assert np.array_equal(m["idx"], gj * 26 + gk // 6)                       # the REFERENCE's winners
    assert np.array_equal(m["codes"], g["knn_pred"]) and np.array_equal(m["votes"], g["vote"])
    assert np.array_equal(np.argsort(m["d"], axis=1, kind="stable"), np.argsort(g["aud_dist"], axis=1, kind="stable"))


def test_mixed_on_the_planted_near_tie_golden():
    """Sub-1e-16 near-ties: tier 1 (f64 dot) cannot decide them, tier 2 (reference arithmetic) does."""
    g = load_golden("shipped_neartie_n48_m2_s30")
    ntr, nte, s0, s1, s2, s3, mf = [int(v) for v in g["meta"]]
    A = fixture_arrays(ntr, nte, s0, s1, s2, s3, variant=str(g["variant"]))
    out = _compare(A, g["step_freq_score"], nte, want_activity=1)
    m = out["mixed"]
    assert m["stats"]["tier2_pairs"] > 0
    gj, gk = g["aud_aux"][..., 0], g["aud_aux"][..., 1]
    assert np.array_equal(m["idx"], np.where(gj >= 0, gj * 26 + gk // 6, -1))
    present = g["aud_dist"] != 1e3
    assert np.array_equal(m["rank"][present], g["step_aud_score"][present])
    small = g["aud_dist"] < 1e-12
    assert np.array_equal(m["d"][small], g["aud_dist"][small])               # reference-arithmetic values: bit-exact
    assert np.array_equal(m["codes"], g["knn_pred"])


def test_mixed_on_a_crowded_database():
    """A DB built to fill the band: 40 windows are copies of 4 source windows perturbed by relative noise from 6e-8
    (one f32 ulp) to 1e-4, half of them with their source's codes (candidate-level near-ties), half with their own
    (rank-level near-ties); plus exact duplicates.  Every decision must still equal the f64 path's."""
    ntr, nte = 96, 2
    A = fixture_arrays(ntr, nte, 60, 61, 62, 63)
    rng = np.random.Generator(np.random.PCG64(7))
    x = A["tr_interp"]
    code = A["code"]
    for i in range(40):
        src, dst = i % 4, 8 + i
        eps = 10.0 ** rng.uniform(-7.2, -4.0)
        x[dst] = (x[src] * (1.0 + eps * rng.standard_normal(x[src].shape))).astype(np.float32)
        if i % 2 == 0:
            code[dst] = code[src]
    x[60], code[60] = x[1], code[1]                                          # exact duplicates: lowest index wins
    x[61] = x[2]
    out = _compare(A, None, nte, want_activity=200)
    assert out["mixed"]["stats"]["tier1_pairs"] > out["mixed"]["stats"]["tier2_pairs"]


def test_sharded_mixed_merge_protocol_on_one_gpu():
    """The cross-shard merge of mixed-precision tables (qpg_merge_mixed_phase1 / qpg_shard_refine / qpg_merge_mixed_phase2),
    with the two byte exchanges done by hand: two row shards of a DB whose near-ties straddle the shard boundary
    (perturbed copies of shard-0 windows live in shard 1, half of them with the same codes).  Winners and ranks must equal
    the unsharded f64 tables."""
    import torch
    from qpgesture_amd import _lib
    from qpgesture_amd.code_knn import ABSENT_DIST, AUDIO_MX_BAND, CodeKNN, ExchangeLayout, GestureDB
    ntr, nte, W = 120, 2, 2
    A = fixture_arrays(ntr, nte, 70, 71, 72, 73)
    rng = np.random.Generator(np.random.PCG64(9))
    x, code = A["tr_interp"], A["code"]
    for i in range(30):                                      # shard 0 = windows 0..59, shard 1 = 60..119
        src, dst = i % 6, 60 + i
        eps = 0.0 if i % 7 == 0 else 10.0 ** rng.uniform(-7.2, -4.0)
        x[dst] = (x[src] * (1.0 + eps * rng.standard_normal(x[src].shape))).astype(np.float32)
        if i % 2 == 0:
            code[dst] = code[src]
    dev = torch.device("cuda:0")
    te_i = torch.from_numpy(A["te_interp"]).to(dev)
    te_c = torch.from_numpy(np.ascontiguousarray(A["te_ctx"])).to(dev)
    full = GestureDB(code, x, A["tr_ctx"], A["tr_phase"], A["sig"], device=dev)
    k_full = CodeKNN(full, rng=np.random.RandomState(1))
    k_full.audio_precision = "f64"
    T = k_full.sweep_tables(te_i, te_c, nte)
    want_idx, want_rank = T["aud_idx"], T["aud_rank"]
    steps = k_full.n_steps()
    q_win, q_t = np.repeat(np.arange(nte), steps), np.tile(np.arange(steps) * 24, nte)
    Q, K = nte * steps, full.K
    shards, lays = [], []
    for r in range(W):
        db = GestureDB(code, x, A["tr_ctx"], A["tr_phase"], A["sig"], device=dev, rank=r, world=W)
        knn = CodeKNN(db, rng=np.random.RandomState(1))
        knn.sharded_mixed_min_gflop = 0.0                        # (these shards are far below the default work threshold)
        lay = ExchangeLayout(Q, K, 1, ["aud"], True, dev)
        knn.sweep_audio(te_i, q_win, q_t, reduce=False, out=lay.views("aud"))
        assert knn._last_audio_mixed
        shards.append(knn)
        lays.append(lay)
    recv = torch.cat([l.send for l in lays])                 # what the all-gather leaves on every rank
    src_stride = lays[0].send.numel()
    R = 4096
    req_stride, resp_stride = 8 + 8 * R, 8 + 8 * R
    owner = shards[0]
    req = torch.zeros((W * req_stride,), dtype=torch.uint8, device=dev)
    ws = torch.empty((int(_lib.load().qpg_merge_mixed_ws_bytes(Q, K, 1024)),), dtype=torch.uint8, device=dev)
    stats = torch.zeros((4,), dtype=torch.int32, device=dev)
    _lib.call("qpg_merge_mixed_phase1_f64", dev, recv, W, src_stride, lays[0].off["aud_d"], lays[0].off["aud_i"], Q, K,
              float(ABSENT_DIST), AUDIO_MX_BAND, R, req, req_stride, ws, ws.numel(), stats, 1024, -1)
    counts = [int(req[w * req_stride:w * req_stride + 4].view(torch.int32)[0]) for w in range(W)]   # header: count | flags
    assert min(counts) > 0 and max(counts) <= R              # both shards are asked (the near-ties straddle the boundary)
    resp_recv = torch.zeros((W * resp_stride,), dtype=torch.uint8, device=dev)
    for w in range(W):                                       # all-to-all by hand: owner 0's block w -> shard w's block 0
        req_recv = torch.full((W * req_stride,), 255, dtype=torch.uint8, device=dev)      # (unused slots / blocks: ~0)
        for b in range(W):
            req_recv[b * req_stride:b * req_stride + 8] = 0                       # headers: count | flags
        req_recv[:req_stride] = req[w * req_stride:(w + 1) * req_stride]
        resp = torch.zeros((W * resp_stride,), dtype=torch.uint8, device=dev)
        k, db = shards[w], shards[w].db
        _lib.call("qpg_shard_refine_f64", dev, req_recv, W, req_stride, R, 0, db.idx_base * db.Ga, db.base, 0, db.T, db.F,
                  db.aud_t, db.Ga, 6, db.tap_stride, k._last_q32, k._last_qn2, db.cn2, resp, resp_stride, 0, None, R // Q)
        resp_recv[w * resp_stride:(w + 1) * resp_stride] = resp[:resp_stride]
    d = torch.empty((Q, K), dtype=torch.float64, device=dev)
    ix = torch.empty((Q, K), dtype=torch.int32, device=dev)
    rk = torch.empty((Q, K), dtype=torch.int16, device=dev)
    _lib.call("qpg_merge_mixed_phase2_f64", dev, recv, W, src_stride, lays[0].off["aud_i"], Q, K, float(ABSENT_DIST), ws,
              ws.numel(), resp_recv, resp_stride, d, ix, rk, stats, 1024, 1e-12)
    torch.cuda.synchronize()
    st = stats.cpu().numpy()
    print("requests per shard", counts, " cross-shard re-evaluations", int(st[3]), " flags", int(st[1]))
    # (flag 8 = FLAG_CROSS_SHARD_TIE: the planted EXACT duplicates across the shard boundary tie below 1e-12, which the
    # dot-product responses cannot order like the reference for sure - the host would re-match this clip on the exact
    # path, tests/test_gpu_guard_overflow.py; the tables are right here because equal values fall back to the index)
    assert (st[1] & ~8) == 0 and (st[1] & 8) and st[3] == sum(counts)
    assert torch.equal(ix, want_idx)
    assert torch.equal(rk, want_rank)
    from qpgesture_amd.code_knn import AUDIO_MX_ERR
    assert float((d - T["aud_d"]).abs().max()) <= AUDIO_MX_ERR
