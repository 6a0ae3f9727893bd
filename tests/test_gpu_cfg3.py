"""BASELINE.json configs[2]: synthetic DB of 100 000 codes x 512-d, 1 000 query windows, code ids
uniform in [0,512), validity mask Bernoulli(0.9) — the generic cosine per-code-min kernel pair (qpg_text_pack_candidates_f32
/ qpg_text_cosine_f32 / qpg_percode_select_f32, sklearn-exact f32 arithmetic) at that size.
Parity: a slice of the queries bit-exact against the C oracle (the oracle needs ~0.3 s per query on one core);
all 1 000 queries through size-independent properties (planted exact matches win their code with distance 0 and the
lowest index, masked rows never win, the global argmin is the planted row)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N, D, Q, K = 100_000, 512, 1000, 512


def _inputs():
    X = np.random.Generator(np.random.PCG64(0)).standard_normal((N, D), dtype=np.float32)
    code = np.random.Generator(np.random.PCG64(1)).integers(0, K, size=N).astype(np.int32)
    valid = np.random.Generator(np.random.PCG64(2)).random(N) < 0.9
    q = np.random.Generator(np.random.PCG64(3)).standard_normal((Q, D), dtype=np.float32)
    # plant: query i (i < 200) is a positive multiple of valid row r_i, and an identical copy of that row sits at a
    # HIGHER index with the same code (first-wins must return the lower one)
    vidx = np.flatnonzero(valid)
    rows = vidx[np.random.Generator(np.random.PCG64(4)).choice(len(vidx) // 2, size=200, replace=False)]
    for i, r in enumerate(rows):
        q[i] = X[r] * np.float32(2.0)
        dup = vidx[len(vidx) // 2 + i]
        X[dup] = X[r]
        code[dup] = code[r]
    return X, code, valid, q, rows


def _run(X, code, valid, q):
    import torch
    from qpgesture_amd import _lib
    dev = torch.device("cuda:0")
    n, d = X.shape
    nq = q.shape[0]
    xd = torch.from_numpy(X).to(dev).view(n, 1, d)
    cand_r = torch.zeros((1,), dtype=torch.int32, device=dev)
    xt = torch.zeros((((n + 63) // 64) * 64 * d,), dtype=torch.float32, device=dev)
    _lib.call("qpg_text_pack_candidates_f32", dev, xd, n, 1, d, cand_r, 1, xt)
    cm = torch.from_numpy(np.where(valid, code, -1).astype(np.int16)).to(dev).contiguous()
    qd = torch.from_numpy(q).to(dev)
    qn = torch.empty_like(qd)
    _lib.call("qpg_l2_normalize_rows_f32", dev, qd, nq, d, qn)
    Dm = torch.empty((nq, n), dtype=torch.float32, device=dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    ev[0].record()
    _lib.call("qpg_text_cosine_f32", dev, xt, n, d, qn, nq, Dm, Dm.stride(0))
    ev[1].record()
    dist = torch.empty((nq, K), dtype=torch.float32, device=dev)
    idx = torch.empty((nq, K), dtype=torch.int32, device=dev)
    _lib.call("qpg_percode_select_f32", dev, Dm, Dm.stride(0), nq, cm, n, K, 1000.0, 0, dist, idx, None, 0, 0)
    ev[2].record()
    torch.cuda.synchronize()
    return dist.cpu().numpy(), idx.cpu().numpy(), ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])


def test_cfg3_100k_by_512_per_code_min():
    from oracle import cref
    X, code, valid, q, rows = _inputs()
    dist, idx, t_sweep, t_min = _run(X, code, valid, q)
    assert dist.shape == (Q, K) and idx.dtype == np.int32
    # (1) bit-exact slice vs the C restatement of the reference arithmetic (planted + plain queries)
    sel = np.r_[0:4, 200:204, Q - 2:Q]
    cm = np.where(valid, code, -1).astype(np.int32).reshape(N, 1)
    od, oi = cref.text_scan(X.reshape(N, 1, D), [0], cm, [0], q[sel], K=K, n_threads=8)
    assert np.array_equal(idx[sel], oi)
    assert np.array_equal(dist[sel], od)
    # (2) properties over all queries
    assert valid[idx[idx >= 0]].all()                                    # a masked row never wins
    assert np.array_equal(code[idx[idx >= 0]], np.nonzero(idx >= 0)[1])  # winners carry the code of their column
    for i, r in enumerate(rows):
        c = code[r]
        assert idx[i, c] == r, (i, idx[i, c], r)                         # lower index of the two identical rows
        assert dist[i, c] <= 2e-7 and dist[i].argmin() == c              # cos distance of a positive multiple ~ 0
    present = np.bincount(code[valid], minlength=K) > 0
    assert np.array_equal(idx >= 0, np.broadcast_to(present, (Q, K)))
    assert np.all(dist[:, ~present] == 1000.0)
    flops = 2.0 * Q * N * D
    print("cfg-3 (matrix path): sweep %.2f ms (%.1f TFLOP-equivalent/s, D write %.0f GB/s), per-code min+finalize %.2f ms"
          % (t_sweep, flops / t_sweep / 1e9, Q * N * 4 / t_sweep / 1e6, t_min))
    # (3) the fused path (qpg_text_percode_f32: no Q x C matrix in HBM) returns the SAME tables, bit for bit, plus the
    # global nearest neighbour of every query
    import torch
    from qpgesture_amd.cfg3 import CosineIndex
    index = CosineIndex(X, code, valid, n_codes=K, method="valu")
    qd = torch.from_numpy(q).cuda()
    fd, fi, nn = index.query(qd)
    for tpc in (64, 7, 2000, 1):                                   # work granularity must not matter
        index.tiles_per_chunk = tpc
        index._ws = None
        d2, i2, n2 = index.query(qd)
        assert torch.equal(d2, fd) and torch.equal(i2, fi) and torch.equal(n2, nn)
    assert np.array_equal(fd.cpu().numpy(), dist) and np.array_equal(fi.cpu().numpy(), idx)
    nnh = nn.cpu().numpy()
    best_code = dist.argmin(axis=1)
    assert np.array_equal(nnh, idx[np.arange(Q), best_code])     # (ties between codes at the global minimum: none here)
    assert np.array_equal(nnh[:200], rows)                       # the planted rows are the nearest neighbours
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    index.tiles_per_chunk = 64
    index.query(qd)
    ev[0].record()
    for _ in range(5):
        index.query(qd)
    ev[1].record()
    torch.cuda.synchronize()
    print("cfg-3 (fused path): %.2f ms per 1000-query batch" % (ev[0].elapsed_time(ev[1]) / 5))


def test_cfg3_f16_storage_equals_reference_arithmetic_on_rounded_rows():
    """fp16-storage variant: the DB rows live in HBM as f16 + an f32 norm each; the sweep widens, normalises with
    sklearn's division and runs the same f32 arithmetic, so its tables are the C oracle's on the f16-ROUNDED database,
    bit for bit (a 10-query slice; the oracle needs ~0.3 s per query), and every winner carries its column's code."""
    import torch
    from oracle import cref
    from qpgesture_amd.cfg3 import CosineIndex
    X, code, valid, q, rows = _inputs()
    Xr = X.astype(np.float16).astype(np.float32)
    index = CosineIndex(X, code, valid, n_codes=K, feature_dtype="f16")
    assert index.xt.dtype == torch.float16 and index.xt.numel() * 2 == ((N + 63) // 64) * 64 * D * 2
    qd = torch.from_numpy(q).cuda()
    fd, fi, nn = index.query(qd)
    dist, idx = fd.cpu().numpy(), fi.cpu().numpy()
    sel = np.r_[0:4, 200:204, Q - 2:Q]
    cm = np.where(valid, code, -1).astype(np.int32).reshape(N, 1)
    od, oi = cref.text_scan(Xr.reshape(N, 1, D), [0], cm, [0], q[sel], K=K, n_threads=8)
    assert np.array_equal(idx[sel], oi)
    assert np.array_equal(dist[sel], od)
    assert valid[idx[idx >= 0]].all()
    assert np.array_equal(code[idx[idx >= 0]], np.nonzero(idx >= 0)[1])
    # against the f32-stored index: same winners almost everywhere, distances moved by the rounding of the inputs only
    d32, i32, _ = CosineIndex(X, code, valid, n_codes=K, method="valu").query(qd)
    assert 1e-6 < float((d32 - fd).abs().max()) < 1e-2
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    index.query(qd)
    ev[0].record()
    for _ in range(5):
        index.query(qd)
    ev[1].record()
    torch.cuda.synchronize()
    print("cfg-3 (f16 storage): %.2f ms per 1000-query batch" % (ev[0].elapsed_time(ev[1]) / 5))


def test_cfg3_bounded_prefilter_equals_the_exact_sweep():
    """Round 3: CosineIndex(method="mfma") - rows sorted by code, split-f16 matrix-core prefilter with an a-priori bound
    against sklearn's f32 value, exact-order evaluation of the band members only - returns the exact sweep's tables bit
    for bit: distances, first-wins indices (the planted identical rows: lower index), global nearest neighbours, absent
    codes; a slice against the C oracle; and the prefilter's error stays inside the band it is given."""
    import torch
    from oracle import cref
    from qpgesture_amd import _lib
    from qpgesture_amd.cfg3 import CosineIndex, prefilter_band
    X, code, valid, q, rows = _inputs()
    qd = torch.from_numpy(q).cuda()
    ref = CosineIndex(X, code, valid, n_codes=K, method="valu")
    rd, ri, rn = ref.query(qd)
    index = CosineIndex(X, code, valid, n_codes=K)
    assert index.method == "mfma" and index.R % 32 == 0
    index.sorted.by_code = False       # (this test: the by-query select of rounds 3 / 4; round 5's by-code path: next test)
    fd, fi, fn = index.query(qd)
    assert index.fallbacks == 0
    assert torch.equal(fd, rd) and torch.equal(fi, ri) and torch.equal(fn, rn)
    sel = np.r_[0:3, 200:203, Q - 2:Q]
    cm = np.where(valid, code, -1).astype(np.int32).reshape(N, 1)
    od, oi = cref.text_scan(X.reshape(N, 1, D), [0], cm, [0], q[sel], K=K, n_threads=8)
    assert np.array_equal(fi.cpu().numpy()[sel], oi) and np.array_equal(fd.cpu().numpy()[sel], od)
    # round 4: the product path hands the select tile minima + row masks; the same batch through the round-3 form (the
    # Q x R prefilter matrix) must give the same tables bit for bit
    assert index.sorted.use_masks and index._scratch.get("Dm") is None
    index.sorted.use_masks = False
    md, mi, mn = index.query(qd)
    index.sorted.use_masks = True
    assert torch.equal(fd, md) and torch.equal(fi, mi) and torch.equal(fn, mn)
    # the prefilter matrix against the exact f32 distances of the same (sorted) rows: inside half the band
    Dm = index._scratch["Dm"][:8].double().cpu().numpy()
    # ... and the masks against it: bit r of a tile = row r within the band of the tile's minimum (f32 compare)
    tmin = index._scratch["tmin"][:8].cpu().numpy()
    index.query(qd)                                            # (masks of this batch)
    tmask = index._scratch["tmask"][:8].cpu().numpy().view(np.uint16)
    D32 = index._scratch["Dm"][:8].cpu().numpy().reshape(8, -1, 16)
    want = (D32 <= (D32.min(axis=2, keepdims=True) + np.float32(index.band))).astype(np.uint16)
    want = (want << np.arange(16, dtype=np.uint16)).sum(axis=2).astype(np.uint16)
    assert np.array_equal(tmask, want) and np.array_equal(tmin, D32.min(axis=2))
    rows_ok = (index.sorted.row_index >= 0).cpu().numpy()
    xs = index.sorted.xs.double().cpu().numpy()
    qn = torch.empty_like(qd)
    _lib.call("qpg_l2_normalize_rows_f32", qd.device, qd, Q, D, qn)
    exact = 1.0 - qn[:8].double().cpu().numpy() @ xs[:-1].T                      # (xs row R: the zero row)
    err = np.abs(Dm - exact)[:, rows_ok].max()
    print("prefilter: max |d~ - (1 - <x^, q^>)| = %.3g; band %.3g" % (err, prefilter_band(D)))
    assert err <= 1.3e-6
    # ragged batch sizes (chunks of 96) and timing
    for nq in (1, 95, 97, 200):
        d2, i2, n2 = index.query(qd[:nq])
        assert torch.equal(d2, rd[:nq]) and torch.equal(i2, ri[:nq]) and torch.equal(n2, rn[:nq])
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    index.check_flags = False
    index.query(qd)
    ev[0].record()
    for _ in range(5):
        index.query(qd)
    ev[1].record()
    torch.cuda.synchronize()
    assert int(index._stats[1].item()) == 0
    print("cfg-3 (bounded prefilter + exact refine): %.2f ms per 1000-query batch" % (ev[0].elapsed_time(ev[1]) / 5))


def test_cfg3_prefilter_falls_back_when_a_band_overflows():
    """9 000 NEAR-copies of one row under one code (relative perturbations of 1e-7: all within 1e-6 of each other for every
    query, far inside the band): more than the select's lists hold.  The index notices (stats flag) and answers from the
    exact sweep: same tables.  (9 000 EXACT copies no longer get that far: the builder keeps the first of identical rows of
    a code - second half of the test - and the lowest index still wins the tie.)"""
    import torch
    from qpgesture_amd.cfg3 import CosineIndex
    rng = np.random.Generator(np.random.PCG64(11))
    n, d, k, nq = 20_000, 512, 64, 40
    X = rng.standard_normal((n, d), dtype=np.float32)
    code = rng.integers(0, k, size=n).astype(np.int32)
    X[5_000:14_000] = X[4_999] * (1.0 + 1e-7 * rng.standard_normal((9_000, d))).astype(np.float32)
    code[4_999:14_000] = 7
    q = rng.standard_normal((nq, d), dtype=np.float32)
    q[3] = 1.5 * X[4_999]                                   # a query whose nearest rows ARE the copies
    qd = torch.from_numpy(q).cuda()
    ref = CosineIndex(X, code, None, n_codes=k, method="valu").query(qd)
    index = CosineIndex(X, code, None, n_codes=k)
    got = index.query(qd)
    assert index.fallbacks == 1
    for a, b in zip(got, ref):
        assert torch.equal(a, b)
    assert (got[1][:, 7].cpu().numpy() != -1).all()
    # exact copies: dropped at build time, no fallback, the first copy wins
    X[5_000:14_000] = X[4_999]
    ref = CosineIndex(X, code, None, n_codes=k, method="valu").query(qd)
    index = CosineIndex(X, code, None, n_codes=k)
    assert index.sorted.n_rows_kept <= n - 9_000
    got = index.query(qd)
    assert index.fallbacks == 0
    for a, b in zip(got, ref):
        assert torch.equal(a, b)
    assert (got[1][:, 7].cpu().numpy() != -1).all()


def test_cfg3_h_plane_prefilter_and_by_code_select_equal_the_exact_sweep():
    """Round 5: batches of >= 256 queries take the prefilter on the h planes alone (one f16 MFMA per block, a-priori bound
    sorted_rows.gemm_h_err) and the exact-order evaluations BY CODE (four lanes per pair on chain-permuted rows): the exact
    sweep's tables bit for bit - all 1 000 queries, batch sizes around the chunk / threshold edges, a slice against the C
    oracle; the tile minima stay inside the bound; the permutation is the one the kernel's comment states; all-zero rows,
    exact duplicates and an overflowing pair list (fallback) on a small index."""
    import torch
    from oracle import cref
    from qpgesture_amd import _lib
    from qpgesture_amd.cfg3 import CosineIndex
    from qpgesture_amd.sorted_rows import gemm_h_err, prefilter_band_h
    X, code, valid, q, rows = _inputs()
    qd = torch.from_numpy(q).cuda()
    ref = CosineIndex(X, code, valid, n_codes=K, method="valu")
    rd, ri, rn = ref.query(qd)
    index = CosineIndex(X, code, valid, n_codes=K)
    assert index.method == "mfma" and index.sorted.by_code and index.R % 64 == 0 and index.sorted.uses_by_code(Q)
    fd, fi, fn = index.query(qd)
    assert index.fallbacks == 0 and "tmin_t" in index._scratch
    assert torch.equal(fd, rd) and torch.equal(fi, ri) and torch.equal(fn, rn)
    sel = np.r_[0:3, 200:203, Q - 2:Q]
    cm = np.where(valid, code, -1).astype(np.int32).reshape(N, 1)
    od, oi = cref.text_scan(X.reshape(N, 1, D), [0], cm, [0], q[sel], K=K, n_threads=8)
    assert np.array_equal(fi.cpu().numpy()[sel], oi) and np.array_equal(fd.cpu().numpy()[sel], od)
    # the permutation: position 32 G + 8 k + j holds element 16 (2 G + (j >> 2)) + 4 (3 - (j & 3)) + k
    xs, xp = index.sorted.xs[:64].cpu().numpy(), index.sorted.xs_perm()[:64].cpu().numpy()
    want = xs.reshape(64, D // 32, 2, 4, 4)[:, :, :, ::-1, :].transpose(0, 1, 4, 2, 3).reshape(64, D)
    assert np.array_equal(xp, want)
    # tile minima of the h-plane GEMM against the exact 1 - <x^, q^> of the same (sorted) rows: inside its bound
    qn = torch.empty_like(qd)
    _lib.call("qpg_l2_normalize_rows_f32", qd.device, qd, Q, D, qn)
    xsd = index.sorted.xs[:-1].double()
    exact = (1.0 - qn[:8].double() @ xsd.T).cpu().numpy().reshape(8, -1, 16)
    tmin_t = index._scratch["tmin_t"][:, :8].double().cpu().numpy().T                 # [8][tiles]
    live = (index.sorted.row_code.cpu().numpy().reshape(-1, 16)[:, 0] & 0x1fff) != 0x1fff
    err = np.abs(tmin_t - exact.min(axis=2))[:, live].max()
    print("h-plane prefilter: max |tile min - exact tile min| = %.3g; bound %.3g; band %.3g" % (err, gemm_h_err(D), prefilter_band_h(D)))
    assert err <= gemm_h_err(D)
    # batch sizes: below the threshold (by-query path), at it, ragged chunks of 96
    for nq in (95, 256, 257, 300, 999):
        d2, i2, n2 = index.query(qd[:nq])
        assert torch.equal(d2, rd[:nq]) and torch.equal(i2, ri[:nq]) and torch.equal(n2, rn[:nq]), nq
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    index.check_flags = False
    index.query(qd)
    ev[0].record()
    for _ in range(5):
        index.query(qd)
    ev[1].record()
    torch.cuda.synchronize()
    assert int(index._stats[1].item()) == 0
    print("cfg-3 (h-plane prefilter + by-code select): %.3f ms per 1000-query batch" % (ev[0].elapsed_time(ev[1]) / 5))
    # a small index with all-zero rows, exact duplicates and masked rows
    rng = np.random.Generator(np.random.PCG64(21))
    n, k, nq = 6_000, 40, 300
    Xs = rng.standard_normal((n, D), dtype=np.float32)
    cs = rng.integers(0, k, size=n).astype(np.int32)
    Xs[100:130] = 0.0                                       # all-zero embeddings (several codes)
    Xs[2_000:2_050] = Xs[1_999]                             # exact duplicates of one row under one code
    cs[1_999:2_050] = 5
    vs = rng.random(n) < 0.8
    qs = rng.standard_normal((nq, D), dtype=np.float32)
    qs[7] = 3.0 * Xs[1_999]
    qsd = torch.from_numpy(qs).cuda()
    want3 = CosineIndex(Xs, cs, vs, n_codes=k, method="valu").query(qsd)
    small = CosineIndex(Xs, cs, vs, n_codes=k)
    assert small.sorted.uses_by_code(nq) and small.sorted.n_zero_rows > 0
    got3 = small.query(qsd)
    assert small.fallbacks == 0
    for a, b in zip(got3, want3):
        assert torch.equal(a, b)
    # D = 384 (the matcher's text width: three stages of four k-blocks - the one-stage-per-trip form of the GEMM)
    X3 = rng.standard_normal((30_000, 384), dtype=np.float32)
    c3 = rng.integers(0, 128, size=30_000).astype(np.int32)
    q3 = torch.from_numpy(rng.standard_normal((400, 384), dtype=np.float32)).cuda()
    want6 = CosineIndex(X3, c3, None, n_codes=128, method="valu").query(q3)
    i384 = CosineIndex(X3, c3, None, n_codes=128)
    assert i384.sorted.uses_by_code(400)
    got6 = i384.query(q3)
    assert i384.fallbacks == 0
    for a, b in zip(got6, want6):
        assert torch.equal(a, b)
    # 9 000 NEAR-copies under one code.  One query next to them: 9 000 exact evaluations, no overflow (the by-code lists
    # are per 16-row tile).  A hundred such queries: 1 600 pairs in every one of those tiles - the list overflows, the
    # exact sweep answers.
    n2, k2 = 20_000, 64
    X2 = rng.standard_normal((n2, D), dtype=np.float32)
    c2 = rng.integers(0, k2, size=n2).astype(np.int32)
    X2[5_000:14_000] = X2[4_999] * (1.0 + 1e-7 * rng.standard_normal((9_000, D))).astype(np.float32)
    c2[4_999:14_000] = 7
    q2 = rng.standard_normal((nq, D), dtype=np.float32)
    q2[3] = 1.5 * X2[4_999]
    q2d = torch.from_numpy(q2).cuda()
    want4 = CosineIndex(X2, c2, None, n_codes=k2, method="valu").query(q2d)
    idx2 = CosineIndex(X2, c2, None, n_codes=k2)
    got4 = idx2.query(q2d)
    assert idx2.fallbacks == 0
    for a, b in zip(got4, want4):
        assert torch.equal(a, b)
    q2[100:200] = X2[4_999][None, :] * rng.uniform(0.5, 2.0, size=(100, 1)).astype(np.float32)
    q2d = torch.from_numpy(q2).cuda()
    want5 = CosineIndex(X2, c2, None, n_codes=k2, method="valu").query(q2d)
    got5 = idx2.query(q2d)
    assert idx2.fallbacks == 1
    for a, b in zip(got5, want5):
        assert torch.equal(a, b)


@pytest.mark.parametrize("Qn,Dn", [(1000, 512), (257, 384), (96, 128), (5, 1024)])
def test_fused_query_prepare_equals_the_three_launches(Qn, Dn):
    """Round 6: qpg_hl_prepare_queries (sklearn's normalisation + the split-f16 column image + the chain-permuted copy in one
    launch on the raw queries) against qpg_l2_normalize_rows_f32 + qpg_hl_pack_cols + qpg_perm32_rows_f32: every output byte
    equal - ragged last chunk of 96, an all-zero query (sklearn leaves it at zero), a tiny-norm and a huge-norm query."""
    import torch
    from qpgesture_amd import _lib
    dev = torch.device("cuda:0")
    rng = np.random.Generator(np.random.PCG64(Qn * 7 + Dn))
    q = rng.standard_normal((Qn, Dn), dtype=np.float32)
    q[1] = 0.0
    q[2] *= np.float32(1e-20)
    q[3] *= np.float32(1e15)
    qd = torch.from_numpy(q).to(dev)
    nb = int(_lib.load().qpg_hl_cols_bytes(Qn, Dn))
    qn_a, qp_a = torch.empty_like(qd), torch.empty_like(qd)
    img_a = torch.full((nb,), 0x5a, dtype=torch.uint8, device=dev)
    _lib.call("qpg_l2_normalize_rows_f32", dev, qd, Qn, Dn, qn_a)
    _lib.call("qpg_hl_pack_cols", dev, qn_a, Qn, Dn, img_a, nb)
    _lib.call("qpg_perm32_rows_f32", dev, qn_a, Qn, Dn, qp_a)
    qn_b, qp_b = torch.empty_like(qd), torch.empty_like(qd)
    img_b = torch.full((nb,), 0xa5, dtype=torch.uint8, device=dev)
    _lib.call("qpg_hl_prepare_queries", dev, qd, Qn, Dn, qn_b, img_b, nb, qp_b)
    assert torch.equal(qn_a.view(torch.int32), qn_b.view(torch.int32))
    assert torch.equal(qp_a.view(torch.int32), qp_b.view(torch.int32))
    chunks = (Qn + 95) // 96
    frag = chunks * (Dn // 32) * 6 * 2 * 1024
    assert torch.equal(img_a[:frag], img_b[:frag])                              # fragments incl. the zero padding queries
    assert torch.equal(img_a[frag:frag + 4 * Qn], img_b[frag:frag + 4 * Qn])    # the scale exponents of the live queries
    # optional outputs may be left out
    img_c = torch.empty_like(img_b)
    _lib.call("qpg_hl_prepare_queries", dev, qd, Qn, Dn, None, img_c, nb, None)
    assert torch.equal(img_c[:frag], img_a[:frag])


def test_cfg3_fused_prepare_and_three_launch_paths_return_the_same_tables():
    import torch
    from qpgesture_amd.cfg3 import CosineIndex
    X, code, valid, q, rows = _inputs()
    qd = torch.from_numpy(q).cuda()
    index = CosineIndex(X, code, valid, n_codes=K)
    assert index.sorted.uses_by_code(Q)
    a = index.query(qd)
    index.fused_prepare = False
    b = index.query(qd)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
