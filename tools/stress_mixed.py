"""Randomised stress of the mixed-precision audio path against the f64 path (which the oracle tests pin): random DB
sizes that mix both sweep organisations (LDS-shared-query blocks + split-K remainder), 1-5 clips per sweep, f32 / f16
base, planted duplicates and ulp..1e-4 perturbed copies.  Winners, ranks and the walk's codes must be identical.
Not part of the test suite; run on the GPU box: python tools/stress_mixed.py [trials]."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from qpgesture_amd import synth
from qpgesture_amd.code_knn import AUDIO_MX_ERR, CodeKNN, GestureDB
from qpgesture_amd.data_processing import interp_wavlm

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rs = np.random.RandomState(77)
bad = 0
t0 = time.time()
for t in range(trials):
    N = int(rs.choice([rs.randint(40, 300), rs.randint(630, 1400), rs.randint(1400, 2300)]))
    clips = int(rs.randint(1, 6)); M = int(rs.randint(1, 7))
    half = bool(rs.rand() < 0.35)
    tr = synth.make_db(N, int(rs.randint(0, 10000)))
    interp = interp_wavlm(tr["wavlm"])
    code = synth.make_codes(N, int(rs.randint(0, 10000)))
    for _ in range(int(rs.randint(0, 12))):               # perturbed / exact copies, same or own codes
        j, k = rs.choice(N, 2, replace=False)
        eps = 0.0 if rs.rand() < 0.3 else 10.0 ** rs.uniform(-7.3, -4.0)
        interp[k] = (interp[j] * (1.0 + eps * rs.standard_normal(interp[j].shape))).astype(np.float32)
        if rs.rand() < 0.5:
            code[k] = code[j]
    db = GestureDB(code, interp, np.ascontiguousarray(tr["context"].squeeze(2)), tr["phase_dense"], synth.make_signature(3),
                   device="cuda:0", feature_dtype="f16" if half else "f32")
    te = synth.make_db(M * clips, int(rs.randint(0, 10000)))
    ti = torch.from_numpy(interp_wavlm(te["wavlm"])).cuda()
    if rs.rand() < 0.5:                                   # a query window that IS a DB window
        ti[0] = torch.from_numpy(interp[int(rs.randint(0, N))]).cuda()
    tc = torch.from_numpy(np.ascontiguousarray(te["context"].squeeze(2))).cuda()
    out = {}
    for prec in ("f64", "mixed"):
        knn = CodeKNN(db, rng=np.random.RandomState(5)); knn.audio_precision = prec
        T = knn.sweep_tables(ti, tc, M * clips)
        sc, sp = knn.init_code_phase()
        codes = knn.walk(T, M, 0, seed_code=sc, seed_phase=sp)[0]
        out[prec] = (T["aud_idx"].cpu().numpy(), T["aud_rank"].cpu().numpy(), T["aud_d"].cpu().numpy(), codes, knn.mixed_stats())
    a, b = out["f64"], out["mixed"]
    ok = (np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[3], b[3])
          and np.abs(a[2] - b[2]).max() <= AUDIO_MX_ERR and b[4]["flags"] == 0)
    bad += not ok
    print("trial %2d N=%4d clips=%d M=%d %s  tier1=%d tier2=%d  %s" % (t, N, clips, M, "f16" if half else "f32",
          b[4]["tier1_pairs"], b[4]["tier2_pairs"], "ok" if ok else "MISMATCH idx=%d rank=%d"
          % ((a[0] != b[0]).sum(), (a[1] != b[1]).sum())), flush=True)
    del db
    torch.cuda.empty_cache()
print("%d trials, %d mismatches, %.0f s" % (trials, bad, time.time() - t0))
sys.exit(1 if bad else 0)
