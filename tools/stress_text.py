"""Randomised stress of the text side's bounded prefilter (CodeKNN.text_kernel = "mfma": sorted_rows.SortedRows,
csrc/qpg_sorted.hip) against the exact-order sweep of every pair ("valu"), which the goldens and the C oracle pin: random DB
sizes, context rows that repeat inside and across windows (the reference's per-frame embeddings do), all-zero rows, near
copies (1e-7 .. 1e-4 relative), queries that ARE database rows, 1-5 clips per sweep.  Distances (bitwise), candidates, ranks
and the walk's codes must be identical; a raised trouble word (band overflow) must lead to the same codes through
match_clip's re-match.  Not part of the test suite; run on the GPU box: python tools/stress_text.py [trials]."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from qpgesture_amd import synth
from qpgesture_amd.code_knn import CodeKNN, GestureDB
from qpgesture_amd.data_processing import interp_wavlm

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rs = np.random.RandomState(91)
bad = 0
t0 = time.time()
for t in range(trials):
    N = int(rs.choice([rs.randint(20, 200), rs.randint(300, 900), rs.randint(900, 2100)]))
    M = int(rs.randint(1, 7)) * int(rs.randint(1, 4))
    if t % 3 == 2:                                                         # round 5: >= 256 queries take the h-plane prefilter + by-code select
        M = int(rs.randint(32, 61))
    tr = synth.make_db(N, int(rs.randint(0, 10000)))
    ctx = np.ascontiguousarray(tr["context"].squeeze(2))                   # (N, 30, 384)
    code = synth.make_codes(N, int(rs.randint(0, 10000)))
    kind = rs.randint(0, 4)
    if kind >= 1:                                                          # repeats inside windows
        for j in range(N):
            r = 0
            while r < 30:
                span = int(rs.randint(1, 6))
                ctx[j, r:r + span] = ctx[j, r]
                r += span
    if kind >= 2:                                                          # one embedding shared by many windows, zero rows
        sil = rs.standard_normal(384).astype(np.float32)
        m = rs.rand(N, 30) < rs.uniform(0.02, 0.3)
        ctx[m] = sil
        ctx[rs.rand(N, 30) < 0.03] = 0.0
    if kind == 3:                                                          # near copies
        for _ in range(int(rs.randint(1, 40))):
            j, k = rs.choice(N, 2, replace=False)
            eps = 10.0 ** rs.uniform(-7.3, -4.0)
            ctx[k] = (ctx[j] * (1.0 + eps * rs.standard_normal(ctx[j].shape))).astype(np.float32)
    db = GestureDB(code, interp_wavlm(tr["wavlm"]), ctx, tr["phase_dense"], synth.make_signature(3), device="cuda:0")
    te = synth.make_db(M, int(rs.randint(0, 10000)))
    ti = torch.from_numpy(interp_wavlm(te["wavlm"])).cuda()
    tcn = np.ascontiguousarray(te["context"].squeeze(2))
    if rs.rand() < 0.6:                                                    # queries that are database rows / zero
        tcn[0, :10] = ctx[int(rs.randint(0, N)), :10]
    if rs.rand() < 0.2:
        tcn[-1, 5:9] = 0.0
    tc = torch.from_numpy(tcn).cuda()
    out = {}
    for kern in ("valu", "mfma"):
        knn = CodeKNN(db, rng=np.random.RandomState(5)); knn.text_kernel = kern
        T = knn.sweep_tables(ti, tc, M)
        torch.cuda.synchronize()
        flags = knn.mixed_stats()["flags"]
        knn.clear_flags()
        knn.rng = np.random.RandomState(6)
        codes, _, votes = knn.match_clip(ti, tc, M)
        out[kern] = (T["txt_d"].cpu().numpy(), T["txt_idx"].cpu().numpy(), T["txt_rank"].cpu().numpy(), codes, votes, flags,
                     knn.fallbacks)
    a, b = out["valu"], out["mfma"]
    tables_ok = (np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)) and np.array_equal(a[1], b[1]) and
                 np.array_equal(a[2], b[2]))
    # (an overflowing band list raises FLAG_TEXT_OVERFLOW = 16 since round 4 - bit 1 before: the tables of that sweep are
    # then unguarded and match_clip re-runs the text side on the exact sweep; the CODES must be equal in every case)
    codes_ok = np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4])
    ok = codes_ok and (tables_ok or (b[5] & (1 | 16))) and a[5] == 0
    bad += not ok
    print("trial %2d N=%4d M=%2d%s kind=%d kept %6d of %6d rows, %3d zero  flags=%d rematched=%d  %s"
          % (t, N, M, " (by code)" if db.txt_sorted.uses_by_code(8 * M) else "", kind, db.txt_sorted.n_rows_kept, N * 26,
             db.txt_sorted.n_zero_rows, b[5], b[6],
             "ok" if ok else "MISMATCH (codes %s, tables %s)" % (codes_ok, tables_ok)), flush=True)
    del db
    torch.cuda.empty_cache()
print("%d trials, %d mismatches, %.0f s" % (trials, bad, time.time() - t0))
sys.exit(1 if bad else 0)
