"""Split-operand f16 audio sweep (qpg_audio_cosine_hl, csrc/qpg_audio_hl.hip): (1) the one measured constant of its
a-priori bound - how far a v_mfma_f32_16x16x32_f16 block sum is from the exact sum of its 32 products; (2) the bound
itself, measured against the f64 sweep on awkward data; (3) CodeKNN on this kernel returns the reference's tables and
codes (goldens) and the same winners / ranks as on the f32-matrix-core kernel."""
import numpy as np
import pytest

from tests.helpers import fixture_arrays, load_golden

pytestmark = pytest.mark.gpu


def _probe(a, b, c=None):
    import torch
    from qpgesture_amd import _lib
    dev = torch.device("cuda:0")
    tiles = a.shape[0]
    ad = torch.from_numpy(a).to(dev).contiguous()
    bd = torch.from_numpy(b).to(dev).contiguous()
    cd = None if c is None else torch.from_numpy(c).to(dev).contiguous()
    out = torch.empty((tiles, 16, 16), dtype=torch.float32, device=dev)
    _lib.call("qpg_probe_mfma_f16_tile", dev, ad, bd, cd, tiles, out)
    return out.cpu().numpy()


def test_f16_matrix_core_block_sum_error():
    """kappa: |MFMA(A, B, 0) - exact| in units of 2^-24 * sum |products|, over random blocks, blocks with a wide dynamic
    range, blocks that cancel, and blocks built against the hardware's octet-wise chop (model: 7 chopped terms per octet +
    the final adder + one rounding = 8.5), then the same for the chains of two instructions the sweep runs; the bound of the
    sweep ASSUMES 13 for a chain; the measured values are printed."""
    rng = np.random.default_rng(0)
    tiles = 4096
    worst = {}
    for name in ("normal", "wide", "cancel", "scaled"):
        a = rng.standard_normal((tiles, 16, 32))
        b = rng.standard_normal((tiles, 16, 32))
        if name == "wide":
            a *= 2.0 ** rng.integers(-10, 11, size=a.shape)
            b *= 2.0 ** rng.integers(-10, 11, size=b.shape)
        if name == "cancel":                                      # pairs (x, -x) against equal partners + small terms
            a[:, :, 1::2] = -a[:, :, 0::2]
            b[:, :, 1::2] = b[:, :, 0::2] * (1 + 2.0 ** -9 * rng.integers(-2, 3, size=b[:, :, 0::2].shape))
        if name == "scaled":
            a *= 2.0 ** 13
            b *= 2.0 ** 13
        a16, b16 = a.astype(np.float16), b.astype(np.float16)
        got = _probe(a16, b16).astype(np.float64)
        A, B = a16.astype(np.float64), b16.astype(np.float64)
        exact = np.einsum("tik,tjk->tij", A, B)                   # f16 x f16 products are exact in f64; 32 terms
        mag = np.einsum("tik,tjk->tij", np.abs(A), np.abs(B))
        kappa = np.abs(got - exact) / (2.0 ** -24 * mag)
        worst[name] = float(kappa.max())
    # adversarial for the octet-chop the probe found (tools/probe_mfma_f16.py): ONE dominant product per block, the other
    # 31 positive and just small enough to lose most of their bits against it
    for rep in range(8):
        a = np.abs(rng.standard_normal((tiles, 16, 32)))
        b = np.abs(rng.standard_normal((tiles, 16, 32)))
        k0 = int(rng.integers(0, 32))
        a[:, :, k0] *= 2.0 ** int(rng.integers(8, 12))
        b[:, :, k0] *= 2.0 ** int(rng.integers(8, 12))
        a16, b16 = a.astype(np.float16), b.astype(np.float16)
        got = _probe(a16, b16).astype(np.float64)
        A, B = a16.astype(np.float64), b16.astype(np.float64)
        kappa = np.abs(got - np.einsum("tik,tjk->tij", A, B)) / (2.0 ** -24 * np.einsum("tik,tjk->tij", np.abs(A), np.abs(B)))
        worst["dominant"] = max(worst.get("dominant", 0.0), float(kappa.max()))
    print("kappa (f32 roundings of the block's sum |products|):", worst)
    assert max(worst.values()) <= 9.0                             # (the model says 8.5; the bound ASSUMES 13 for a chain of two)
    # The sweep runs h h' in chains of TWO instructions (the second starts from the first's result) before it adds the
    # sum to its f64 accumulator: kappa_2 in units of 2^-24 x sum |64 products| - model kappa + 0.5, the bound assumes 13
    worst2 = {}

    def chain(name, a1, b1, a2, b2):
        a1, b1, a2, b2 = (x.astype(np.float16) for x in (a1, b1, a2, b2))
        got = _probe(a2, b2, _probe(a1, b1)).astype(np.float64)
        A1, B1, A2, B2 = (x.astype(np.float64) for x in (a1, b1, a2, b2))
        exact = np.einsum("tik,tjk->tij", A1, B1) + np.einsum("tik,tjk->tij", A2, B2)
        mag = np.einsum("tik,tjk->tij", np.abs(A1), np.abs(B1)) + np.einsum("tik,tjk->tij", np.abs(A2), np.abs(B2))
        worst2[name] = max(worst2.get(name, 0.0), float((np.abs(got - exact) / (2.0 ** -24 * mag)).max()))

    rnd = lambda sc=1.0: rng.standard_normal((tiles, 16, 32)) * sc
    for rep in range(3):
        chain("normal", rnd(), rnd(), rnd(), rnd())
        chain("wide", rnd() * 2.0 ** rng.integers(-10, 11, size=(tiles, 16, 32)), rnd(),
              rnd() * 2.0 ** rng.integers(-10, 11, size=(tiles, 16, 32)), rnd())
        for sc in (2.0 ** -6, 2.0 ** -12):
            chain("big, small", np.abs(rnd()), np.abs(rnd()), np.abs(rnd(sc)), np.abs(rnd()))
            chain("small, big", np.abs(rnd(sc)), np.abs(rnd()), np.abs(rnd()), np.abs(rnd()))
        for first in (True, False):
            a, b = np.abs(rnd()), np.abs(rnd())
            k0 = int(rng.integers(0, 32))
            a[:, :, k0] *= 2.0 ** int(rng.integers(8, 12))
            b[:, :, k0] *= 2.0 ** int(rng.integers(8, 12))
            if first:
                chain("dominant in first", a, b, np.abs(rnd()), np.abs(rnd()))
            else:
                chain("dominant in second", np.abs(rnd()), np.abs(rnd()), a, b)
    print("kappa_2 (chains of two):", worst2)
    assert max(worst2.values()) <= 10.5
    # chained form (C != 0), for the record: the cross-term chains of the sweep are bounded without this number
    a16 = rng.standard_normal((256, 16, 32)).astype(np.float16)
    b16 = rng.standard_normal((256, 16, 32)).astype(np.float16)
    c = (rng.standard_normal((256, 16, 16)) * 100).astype(np.float32)
    got = _probe(a16, b16, c).astype(np.float64)
    exact = np.einsum("tik,tjk->tij", a16.astype(np.float64), b16.astype(np.float64)) + c
    mag = np.einsum("tik,tjk->tij", np.abs(a16.astype(np.float64)), np.abs(b16.astype(np.float64))) + np.abs(c)
    print("with C: kappa = %.3f" % float((np.abs(got - exact) / (2.0 ** -24 * mag)).max()))


def _sweeps(N, Q, seed, F=1024, plant=True):
    import torch
    from qpgesture_amd import _lib
    dev = torch.device("cuda:0")
    T, G = 180, 26
    g = torch.Generator(device="cpu").manual_seed(seed)
    base = torch.randn((N, T, F), generator=g)
    if plant and N > 12 and F >= 1024:
        base[3] = 0.0                                              # an all-zero window
        base[5, 100:] = 0.0
        base[7] *= 0.05                                            # a quiet window (inside the bound's range: its scaled
        base[9, 40:90] *= 30.0                                     # norm^2 is ~2.5e5 >= 2^16) and a loud stretch, which sets
        #                                                            the scale exponent of the whole image
    base = base.to(dev)
    q32 = torch.randn((Q, 6 * F), generator=g).to(dev)
    if plant and Q > 11:
        q32[7] = 0.0
        q32[9] = base[min(11, N - 1), 12:24:2].reshape(-1)         # a query that IS a candidate (distance ~ 0)
        q32[10] *= 1e-3
        q32[11] *= 1e3
    cand_t = (torch.arange(G, dtype=torch.int32) * 6).to(dev)
    fn2 = torch.empty((N, T), dtype=torch.float64, device=dev)
    _lib.call("qpg_frame_norm2_f64", dev, base, N * T, F, fn2)
    cn2 = torch.empty((N, G), dtype=torch.float64, device=dev)
    _lib.call("qpg_audio_cand_norm2", dev, fn2, N, T, cand_t, G, 6, 2, cn2)
    qn2 = (q32.double() ** 2).sum(1)
    D64 = torch.empty((Q, N * G), dtype=torch.float64, device=dev)
    _lib.call("qpg_audio_cosine_f64", dev, base, N, T, F, cand_t, G, 6, 2, cn2, q32, qn2, Q, D64, D64.stride(0))
    lib = _lib.load()
    assert lib.qpg_audio_hl_supported(T, F, G, 6, 2, 6)
    img = torch.empty((int(lib.qpg_audio_hl_db_bytes(N, F)),), dtype=torch.uint8, device=dev)
    _lib.call("qpg_audio_hl_pack_db", dev, base, N, T, F, G, 6, 2, 6, img, img.numel())
    qi = torch.empty((int(lib.qpg_audio_hl_query_bytes(Q, F)),), dtype=torch.uint8, device=dev)
    _lib.call("qpg_audio_hl_pack_queries", dev, q32, Q, F, qi, qi.numel())
    stats = torch.zeros((4,), dtype=torch.int32, device=dev)
    Dhl = torch.full((Q, N * G), float("nan"), dtype=torch.float64, device=dev)
    _lib.call("qpg_audio_cosine_hl", dev, img, N, F, G, cn2, qi, qn2, Q, Dhl, 0, Dhl.stride(0), stats)
    D32 = torch.full((Q, N * G), float("nan"), dtype=torch.float32, device=dev)
    _lib.call("qpg_audio_cosine_hl", dev, img, N, F, G, cn2, qi, qn2, Q, D32, 1, D32.stride(0), stats)
    torch.cuda.synchronize()
    assert torch.equal(D32, Dhl.float())
    return D64.cpu().numpy(), D32.double().cpu().numpy(), stats.cpu().numpy()


def test_hl_sweep_stays_inside_the_error_bound():
    from qpgesture_amd.code_knn import AUDIO_HL_ERR as AUDIO_MX_ERR
    for Q, N, F in ((48, 96, 1024), (16, 37, 1024), (5, 8, 1024), (100, 50, 1024), (768, 24, 1024), (48, 700, 1024),
                    (48, 33, 256)):
        D64, Dhl, stats = _sweeps(N, Q, seed=Q + N, F=F)
        assert not np.isnan(Dhl).any()                             # every (query, candidate) was written
        err = np.abs(D64 - Dhl)
        print("N=%d Q=%d F=%d: max |D_hl - D_f64| = %.3g (bound %.3g), mean %.3g" % (N, Q, F, err.max(), AUDIO_MX_ERR, err.mean()))
        assert err.max() <= AUDIO_MX_ERR
        assert stats[1] == 0
        if N > 12 and F >= 1024:
            assert np.array_equal(D64[:, 3 * 26:4 * 26], Dhl[:, 3 * 26:4 * 26])   # zero candidate rows: exact in both
        if Q > 11:
            assert np.array_equal(D64[7], Dhl[7])                                 # zero query row: exact


@pytest.mark.parametrize("quiet", [1e-7, 3e-4])
def test_hl_sweep_flags_operands_outside_its_range(quiet):
    """A window far below the loudest value of the database: the representation bound does not cover it -> stats[1] |= 2
    (the host re-matches such a clip); everything else stays inside the bound.  1e-7: scaled norm < 1 (round 3's guard);
    3e-4: scaled norm ~100 - the l planes' f16 subnormals alone could cost such a row 2e-8 .. 2e-6 of the budget, so since
    round 5 the guard asks for a scaled norm^2 >= 2^16."""
    import torch
    from qpgesture_amd.code_knn import AUDIO_HL_ERR as AUDIO_MX_ERR
    orig = torch.randn

    def planted(*a, **k):
        x = orig(*a, **k)
        if x.dim() == 3 and x.shape[0] == 40:
            x[20] *= quiet
        return x
    torch.randn = planted
    try:
        D64, Dhl, stats = _sweeps(40, 16, seed=5, plant=False)
    finally:
        torch.randn = orig
    assert stats[1] & 2
    ok = np.ones(D64.shape[1], bool)
    ok[20 * 26:21 * 26] = False
    assert np.abs(D64 - Dhl)[:, ok].max() <= AUDIO_MX_ERR


def _build(A, freq_rank, kernel, dev="cuda:0"):
    import torch
    from qpgesture_amd.code_knn import CodeKNN, GestureDB
    db = GestureDB(A["code"], A["tr_interp"], A["tr_ctx"], A["tr_phase"], A["sig"], device=dev, freq_rank=freq_rank)
    assert db.hl_image is not None
    knn = CodeKNN(db, rng=np.random.RandomState(123456))
    knn.audio_kernel = kernel
    te_i = torch.from_numpy(A["te_interp"]).to(dev)
    te_c = torch.from_numpy(np.ascontiguousarray(A["te_ctx"])).to(dev)
    return knn, te_i, te_c


@pytest.mark.parametrize("name", ["shipped_n48_m2_s0", "shipped_n64_m3_s10", "shipped_neartie_n48_m2_s30",
                                  "shipped_speechlike_n48_m2_s60"])
def test_matcher_on_the_hl_kernel_vs_reference_goldens(name):
    from qpgesture_amd.code_knn import AUDIO_HL_ERR as AUDIO_MX_ERR
    g = load_golden(name)
    ntr, nte, s0, s1, s2, s3, mf = [int(v) for v in g["meta"]]
    variant = (str(g["variant"]) or None) if "variant" in g.files else None
    A = fixture_arrays(ntr, nte, s0, s1, s2, s3, variant=variant)
    out = {}
    for kernel in ("hl", "mx"):
        knn, te_i, te_c = _build(A, g["step_freq_score"], kernel)
        codes, _, votes = knn.match_clip(te_i, te_c, nte, return_tables=True)
        assert knn._last_audio_hl == (kernel == "hl") and knn.fallbacks == 0
        T = knn.tables
        out[kernel] = (codes, votes, T["aud_idx"].cpu().numpy(), T["aud_rank"].cpu().numpy(), T["aud_d"].cpu().numpy())
    for a, b in zip(out["hl"][:4], out["mx"][:4]):
        assert np.array_equal(a, b)
    codes, votes, idx, rank, d = out["hl"]
    gj, gk = g["aud_aux"][..., 0], g["aud_aux"][..., 1]
    assert np.array_equal(idx, np.where(gj >= 0, gj * 26 + gk // 6, -1))                  # the REFERENCE's winners
    assert np.abs(d - g["aud_dist"]).max() <= AUDIO_MX_ERR
    present = g["aud_dist"] != 1e3
    if "step_aud_score" in g.files:
        assert np.array_equal(rank[present], g["step_aud_score"][present])
    else:
        assert np.array_equal(np.argsort(d, axis=1, kind="stable"), np.argsort(g["aud_dist"], axis=1, kind="stable"))
    assert np.array_equal(codes, g["knn_pred"]) and np.array_equal(votes, g["vote"])


def test_load_time_selfcheck_and_routing_when_it_fails(monkeypatch):
    """selfcheck.mfma_bound_ok: the measured kappa_2 of this device is inside the assumption (and is what GestureDB
    records); a device that fails the check gets no bounded path - the audio sweep runs in f64, the text side on the
    exact-order kernel - and still returns the reference's tables and codes."""
    import torch
    from qpgesture_amd import selfcheck
    from qpgesture_amd.code_knn import CodeKNN, GestureDB
    selfcheck._cache.clear()
    ok, rep = selfcheck.mfma_bound_ok("cuda:0")
    print("selfcheck:", rep)
    assert ok and not rep["skipped"] and 1.0 < rep["kappa2"] <= selfcheck.KAPPA2_LIMIT < selfcheck.KAPPA2_ASSUMED
    assert 1.0 < rep["kappa4"] <= selfcheck.KAPPA6_LIMIT and 1.0 < rep["kappa6"] <= selfcheck.KAPPA6_LIMIT
    g = load_golden("shipped_n48_m2_s0")
    ntr, nte, s0, s1, s2, s3, mf = [int(v) for v in g["meta"]]
    A = fixture_arrays(ntr, nte, s0, s1, s2, s3)

    def run():
        db = GestureDB(A["code"], A["tr_interp"], A["tr_ctx"], A["tr_phase"], A["sig"], device="cuda:0",
                       freq_rank=g["step_freq_score"])
        knn = CodeKNN(db, rng=np.random.RandomState(123456))
        te_i = torch.from_numpy(A["te_interp"]).cuda()
        te_c = torch.from_numpy(np.ascontiguousarray(A["te_ctx"])).cuda()
        codes, _, _ = knn.match_clip(te_i, te_c, nte, return_tables=True)
        return db, knn, codes
    db, knn, codes = run()
    assert db.hl_bound_ok and db.hl_image is not None and knn._last_audio_hl and knn._last_text_mfma
    assert np.array_equal(codes, g["knn_pred"])
    monkeypatch.setitem(selfcheck._cache, 0, (False, dict(rep, kappa2=14.2)))
    db, knn, codes = run()
    assert not db.hl_bound_ok and db.hl_image is None and db.txt_sorted is None
    assert not knn._last_audio_mixed and not knn._last_text_mfma
    assert np.array_equal(codes, g["knn_pred"])
    assert np.abs(knn.tables["aud_d"].cpu().numpy() - g["aud_dist"]).max() < 1e-13
    assert np.array_equal(knn.tables["txt_d"].cpu().numpy(), g["txt_dist"])
