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
