"""Randomised parity stress of GraphPipeline (round 6): random database sizes, clips per replay, lanes, clip lengths, Gaussian
and speech-like statistics (long near-tie bands, exact text ties); every clip of every group must come back with the codes and
votes CodeKNN.match_clip returns for that clip alone with the same seed.   python tools/stress_pipeline.py [trials]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from qpgesture_amd import synth
from qpgesture_amd.code_knn import CodeKNN, GestureDB, GraphPipeline
from qpgesture_amd.data_processing import interp_wavlm

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(20260930)
bad = rematched = clips_checked = 0
t0 = time.time()
for t in range(trials):
    N = int(rng.choice([96, 160, 320, 512]))
    G, depth, M = int(rng.integers(1, 5)), int(rng.integers(1, 4)), int(rng.integers(1, 4))
    speech = bool(rng.integers(0, 2))
    tr = synth.make_db(N, 9000 + t)
    if speech:
        synth.speechlike_transform(tr, 9100 + t)
    db = GestureDB(synth.make_codes(N, 9200 + t), interp_wavlm(tr["wavlm"]), np.ascontiguousarray(tr["context"].squeeze(2)),
                   tr["phase_dense"], synth.make_signature(9300 + t), device="cuda:0")
    knn = CodeKNN(db, rng=np.random.RandomState(t))
    pipe = GraphPipeline(db, M, clips_per_replay=G, depth=depth, rng=np.random.RandomState(t + 1))
    groups, seeds, want = [], [], []
    for g in range(int(rng.integers(3, 7))):
        te = synth.make_db(G * M, 9400 + 31 * t + g)
        if speech:
            synth.speechlike_transform(te, 9500 + 31 * t + g)
        ti = torch.from_numpy(interp_wavlm(te["wavlm"])).cuda()
        tc = torch.from_numpy(np.ascontiguousarray(te["context"].squeeze(2))).cuda()
        sc, sp = zip(*[knn.init_code_phase() for _ in range(G)])
        for c in range(G):
            want.append(knn.match_clip(ti[c * M:(c + 1) * M], tc[c * M:(c + 1) * M], M, seed_code=sc[c], seed_phase=sp[c]))
        groups.append((ti, tc))
        seeds.append((list(sc), np.stack(sp)))
    got = pipe.match_groups(groups, seeds)
    nb = sum(0 if (np.array_equal(a[0], w[0]) and np.array_equal(a[1], w[2])) else 1 for a, w in zip(got, want))
    bad += nb
    rematched += pipe.rematched
    clips_checked += len(want)
    print("trial %2d: N_db %3d, %d clips x %d lanes, M %d, %s: %d clips, %d mismatches, %d re-matched" % (
        t, N, G, depth, M, "speech-like" if speech else "gaussian", len(want), nb, pipe.rematched), flush=True)
print("done: %d trials, %d clips, %d mismatches, %d re-matched, %.0f s" % (trials, clips_checked, bad, rematched, time.time() - t0))
sys.exit(1 if bad else 0)
