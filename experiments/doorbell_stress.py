"""Stress of ClipGraph(doorbell=True) / SerialReplayer (round 6): N steps with a DIFFERENT seed per step, variants
  plain     : doorbell captures, every step launched after the previous one returned (no pre-launch)
  prelaunch : two captures taking turns, the next one pre-launched behind the doorbell
For every step: did the results arrive, and do the codes belong to THIS step's seed (a replay that ran early would carry the
previous seed's codes)?  python experiments/doorbell_stress.py [steps]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from qpgesture_amd import synth
from qpgesture_amd.code_knn import CodeKNN, GestureDB, SerialReplayer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
N, M = 2048, 6
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
interp = torch.randn((N, 180, 1024), device=dev)
ctx = rng.standard_normal((N, 30, 384)).astype(np.float32)
phase = rng.standard_normal((N, 240, 4, 8)).astype(np.float32)
db = GestureDB(synth.make_codes(N, 2), interp, ctx, phase, synth.make_signature(3), device=dev)
knn = CodeKNN(db, rng=np.random.RandomState(123456))
te_i = torch.randn((M, 180, 1024), device=dev)
te_c = torch.randn((M, 30, 384), device=dev)
seeds = [knn.init_code_phase() for _ in range(7)]
want = [knn.match_clip(te_i, te_c, M, seed_code=c, seed_phase=p)[0].reshape(-1) for c, p in seeds]
assert len({tuple(w) for w in want}) > 1
for variant in ("plain", "prelaunch"):
    cgs = [CodeKNN(db, rng=np.random.RandomState(1 + i)).capture_clip_graph(M, audio=te_i, context=te_c, doorbell=True)
           for i in range(2)]
    sr = SerialReplayer(cgs)
    bad = missing = stale = 0
    for i in range(steps):
        k = i % len(seeds)
        r0 = sr.recovered
        got, _ = sr.step(seeds[k][0], seeds[k][1], more=(variant == "prelaunch" and i + 1 < steps))
        if sr.recovered != r0:
            missing += 1
            print(variant, "step", i, ": the launch's kernels never ran (recovered by a plain replay)", flush=True)
        codes = got[:M * 30].astype(np.int64)
        if not np.array_equal(codes, want[k]):
            bad += 1
            stale += int(np.array_equal(codes, want[(k - 1) % len(seeds)]))
    print("%s: %d steps, %d lost launches, %d wrong codes (%d of them = the previous seed's)" % (variant, steps, missing, bad, stale), flush=True)
