"""The matcher's text side on the bounded matrix-core prefilter (CodeKNN.text_kernel = "mfma": sorted_rows.SortedRows,
csrc/qpg_sorted.hip) against the exact-order sweep of every pair (text_kernel = "valu", qpg_text_cosine_f32): tables, ranks
and matched codes must be identical bit for bit - on dense rows, on rows that repeat (the reference's per-frame text
embeddings repeat: /root/reference/process/make_beat_dataset.py:556-565), on all-zero rows and for an all-zero query
(band overflow -> trouble word -> the clip is re-matched on the exact path).  Reference: GestureKNN.py:708-721."""
import numpy as np
import pytest

from tests.helpers import fixture_arrays, load_golden

pytestmark = pytest.mark.gpu


def _build(n_train=96, n_test=3, seed=5, mutate=None):
    import torch
    from qpgesture_amd.code_knn import CodeKNN, GestureDB
    A = fixture_arrays(n_train, n_test, seed, seed + 1, seed + 2, seed + 3)
    if mutate is not None:
        mutate(A)
    db = GestureDB(A["code"], A["tr_interp"], A["tr_ctx"], A["tr_phase"], A["sig"], device="cuda:0")
    knn = CodeKNN(db, rng=np.random.RandomState(3))
    te_i = torch.from_numpy(np.ascontiguousarray(A["te_interp"], np.float32)).cuda()
    te_c = torch.from_numpy(np.ascontiguousarray(A["te_ctx"], np.float32)).cuda()
    return A, db, knn, te_i, te_c


def _text_tables(knn, te_c, n_windows, kernel):
    import torch
    from qpgesture_amd import _lib
    db, dev = knn.db, knn.db.device
    steps = knn.n_steps()
    pos = knn.query_positions()
    rows_ = [int(i / knn.n_db_frm * 30) for i in pos] * n_windows
    qw = np.repeat(np.arange(n_windows), steps)
    q = te_c[torch.as_tensor(qw, device=dev), torch.as_tensor(np.asarray(rows_), device=dev)].contiguous()
    knn.text_kernel = kernel
    knn.clear_flags()
    d, i, r = knn.sweep_text(q, want_rank=True)
    torch.cuda.synchronize()
    return d.cpu().numpy(), i.cpu().numpy(), r.cpu().numpy(), knn.mixed_stats()["flags"]


def test_text_prefilter_tables_equal_the_exact_sweep():
    A, db, knn, te_i, te_c = _build()
    assert db.txt_sorted is not None and db.txt_sorted.R % 32 == 0
    dv, iv, rv, fv = _text_tables(knn, te_c, 3, "valu")
    assert not knn._last_text_mfma
    dm, im, rm, fm = _text_tables(knn, te_c, 3, "mfma")
    assert knn._last_text_mfma and fm == 0 and fv == 0
    assert np.array_equal(dv.view(np.uint32), dm.view(np.uint32))
    assert np.array_equal(iv, im) and np.array_equal(rv, rm)


def test_text_prefilter_repeated_and_zero_rows():
    """Context rows that repeat inside and across windows (exact ties, first-wins by candidate index) and all-zero rows
    (sklearn leaves them at zero: distance 0.5 |q^|^2, not 1 - <x^, q^>)."""
    def mutate(A):
        c = A["tr_ctx"]
        c[5:40, 3:20] = c[4, 7]                       # one embedding over 35 windows x 17 rows
        c[50:60, :, :] = 0.0                          # silent windows
        c[70, 10:14] = 0.0
        A["te_ctx"][1, 5:9] = c[4, 7]           # a query that hits the repeated embedding exactly
    A, db, knn, te_i, te_c = _build(mutate=mutate)
    assert db.txt_sorted.n_zero_rows > 0
    assert db.txt_sorted.n_rows_kept < 96 * 26 - db.txt_sorted.n_zero_rows - 100      # repeats inside a code were dropped
    dv, iv, rv, fv = _text_tables(knn, te_c, 3, "valu")
    dm, im, rm, fm = _text_tables(knn, te_c, 3, "mfma")
    assert fm == 0
    assert np.array_equal(dv.view(np.uint32), dm.view(np.uint32))
    assert np.array_equal(iv, im) and np.array_equal(rv, rm)


def test_zero_query_overflows_the_band_and_the_clip_is_rematched():
    """An all-zero text query is at 0.5 |x^|^2 from every row: all rows of a code tie inside the band.  With enough rows
    the band list overflows, the trouble word leaves with the codes, and match_clip answers from the exact path."""
    import torch
    from qpgesture_amd.code_knn import FLAG_TEXT_OVERFLOW
    def mutate(A):
        A["te_ctx"][0, :, :] = 0.0
    A, db, knn, te_i, te_c = _build(n_train=700, mutate=mutate)             # 700 x 26 = 18 200 rows > the lists' 8 x 2 048
    _, _, _, fm = _text_tables(knn, te_c, 3, "mfma")
    assert fm == FLAG_TEXT_OVERFLOW                # (its own bit: only the text side has to run again)
    knn.clear_flags()
    knn.text_kernel = "mfma"
    knn.rng = np.random.RandomState(11)
    c1, p1, v1 = knn.match_clip(te_i, te_c, 3)
    assert knn.fallbacks == 1 and knn.text_fallbacks == 1 and knn.text_kernel == "mfma"
    assert knn._last_audio_mixed                   # the re-match kept the mixed-precision audio path
    knn.text_kernel = "valu"
    knn.rng = np.random.RandomState(11)
    c2, p2, v2 = knn.match_clip(te_i, te_c, 3)
    assert knn.fallbacks == 1
    assert np.array_equal(c1, c2) and np.array_equal(v1, v2) and np.array_equal(p1, p2)


@pytest.mark.parametrize("name", ["shipped_n48_m2_s0", "shipped_texttie_n48_m2_s40", "shipped_speechlike_n48_m2_s60"])
def test_goldens_on_both_text_kernels(name):
    """Reference-captured text tables and codes (tests/golden, made by tests/golden/make_golden.py from the reference's
    CodeKNN) on both text kernels; `texttie` holds 296-way exact ties (first-wins by candidate index)."""
    import torch
    from qpgesture_amd.code_knn import CodeKNN, GestureDB
    g = load_golden(name)
    ntr, nte, s0, s1, s2, s3, mf = [int(v) for v in g["meta"]]
    variant = (str(g["variant"]) or None) if "variant" in g.files else None
    A = fixture_arrays(ntr, nte, s0, s1, s2, s3, variant=variant)
    db = GestureDB(A["code"], A["tr_interp"], A["tr_ctx"], A["tr_phase"], A["sig"], device="cuda:0",
                   freq_rank=g["step_freq_score"])
    te_i = torch.from_numpy(A["te_interp"]).cuda()
    te_c = torch.from_numpy(np.ascontiguousarray(A["te_ctx"])).cuda()
    gj, gk = g["txt_aux"][..., 0], g["txt_aux"][..., 1]
    out = []
    for kernel in ("mfma", "valu"):
        knn = CodeKNN(db, rng=np.random.RandomState(123456))
        knn.text_kernel = kernel
        codes, _, votes = knn.match_clip(te_i, te_c, nte, return_tables=True)
        assert knn._last_text_mfma == (kernel == "mfma") and knn.fallbacks == 0
        T = knn.tables
        d, idx = T["txt_d"].cpu().numpy(), T["txt_idx"].cpu().numpy()
        assert np.array_equal(d, g["txt_dist"].astype(np.float32)), kernel            # bit-exact f32 (sklearn's arithmetic)
        assert np.array_equal(idx, np.where(gj >= 0, gj * 26 + gk // 8, -1)), kernel      # the REFERENCE's winners
        if "texttie" not in name:       # (exact ties BETWEEN codes: NumPy's unstable order, test_gpu_matching.py covers it)
            assert np.array_equal(codes, g["knn_pred"]) and np.array_equal(votes, g["vote"]), kernel
        out.append((codes, votes, T["txt_rank"].cpu().numpy()))
    assert all(np.array_equal(a, b) for a, b in zip(out[0], out[1]))
