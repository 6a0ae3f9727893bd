"""Randomised stress of the walk-relevance cut (qpg_percode_select_mixed_f64_cut): random DB sizes, planted
exact / near copies inside and across codes, random frequency ranks, both modality modes that use the audio tables (0: audio +
text, 1: audio alone - the two best codes are read), 1-6 windows per clip.  For every trial the walk over the cut tables
must return the same codes, votes and phase blocks as the walk over the fully settled tables, from 64 seeds (every 8th
previous code) - and every (step, previous code) entry of the fusion tables must be the same candidate.
    python tools/stress_cut.py [trials]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from qpgesture_amd import synth
from qpgesture_amd.code_knn import CodeKNN, GestureDB
from qpgesture_amd.data_processing import interp_wavlm

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rs = np.random.RandomState(404)
dev = torch.device("cuda:0")
bad, t0 = 0, time.time()
for t in range(trials):
    N = int(rs.choice([rs.randint(30, 120), rs.randint(200, 700), rs.randint(800, 1600)]))
    M = int(rs.randint(1, 7))
    mode = int(rs.randint(0, 2))
    tr = synth.make_db(N, int(rs.randint(0, 10000)))
    if rs.rand() < 0.4:
        synth.speechlike_transform(tr, int(rs.randint(0, 10000)))
    x = interp_wavlm(tr["wavlm"])
    code = synth.make_codes(N, int(rs.randint(0, 10000)))
    for _ in range(int(rs.randint(0, 80))):
        j, k = rs.choice(N, 2, replace=False)
        eps = 0.0 if rs.rand() < 0.2 else 10.0 ** rs.uniform(-8.0, -4.5)
        x[k] = (x[j] * (1.0 + eps * rs.standard_normal(x[j].shape))).astype(np.float32)
        if rs.rand() < 0.5:
            code[k] = code[j]
    ctx = np.ascontiguousarray(tr["context"].squeeze(2))
    freq = rs.permutation(512).astype(np.int16) if rs.rand() < 0.5 else None
    db = GestureDB(code, x, ctx, tr["phase_dense"], synth.make_signature(int(rs.randint(0, 100))), device=dev, freq_rank=freq)
    te = synth.make_db(M, int(rs.randint(0, 10000)))
    ti = torch.from_numpy(interp_wavlm(te["wavlm"])).to(dev)
    if rs.rand() < 0.5:                                       # a query that IS (nearly) a database window
        ti[0] = torch.from_numpy(x[int(rs.randint(0, N))]).to(dev)
    tc = torch.from_numpy(np.ascontiguousarray(te["context"].squeeze(2))).to(dev)
    knn = CodeKNN(db, rng=np.random.RandomState(1))
    s0 = knn.mixed_stats()
    T = knn.sweep_tables(ti, tc, M, mode=mode)
    s1 = knn.mixed_stats()
    Tc = knn.sweep_tables(ti, tc, M, mode=mode, for_walk=True)
    s2 = knn.mixed_stats()
    used = bool(knn._last_rank_cut)
    ok = True
    if s2["flags"] == 0:
        for seed_code in range(int(rs.randint(0, 8)), 512, 8):
            sp = rs.standard_normal((8, 16)).astype(np.float32)
            try:
                a = knn.walk(T, M, mode=mode, seed_code=seed_code, seed_phase=sp)
            except IndexError:
                a = None
            try:
                b = knn.walk(Tc, M, mode=mode, seed_code=seed_code, seed_phase=sp)
            except IndexError:
                b = None
            if (a is None) != (b is None):
                ok = False
            elif a is not None:
                pa, pb = [np.asarray(v.cpu()) if isinstance(v, torch.Tensor) else np.asarray(v) for v in (a[1], b[1])]
                ok = ok and np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2]) and np.array_equal(pa, pb)
    knn.clear_flags()
    bad += not ok
    print("trial %2d N=%4d M=%d mode=%d cut=%s  f64 pairs %5d -> %5d  flags=%d  %s"
          % (t, N, M, mode, used, s1["tier1_pairs"] - s0["tier1_pairs"], s2["tier1_pairs"] - s1["tier1_pairs"], s2["flags"],
             "ok" if ok else "MISMATCH"), flush=True)
    del db, knn
    torch.cuda.empty_cache()
print("%d trials, %d mismatches, %.0f s" % (trials, bad, time.time() - t0))
sys.exit(1 if bad else 0)
