"""GraphPipeline sweep: ms per clip for (clips per replay, lanes) combinations on the bench workload (N_db = 2048, M = 6,
random features).  python tools/bench_graph_pipeline.py [G:lanes ...]   default: 1:1 1:2 2:1 2:2 4:1 4:2 4:3 8:1 8:2 16:1"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from qpgesture_amd import synth
from qpgesture_amd.code_knn import CodeKNN, GestureDB, GraphPipeline

N, M = 2048, 6
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
interp = torch.randn((N, 180, 1024), device=dev)
ctx = rng.standard_normal((N, 30, 384)).astype(np.float32)
phase = rng.standard_normal((N, 240, 4, 8)).astype(np.float32)
db = GestureDB(synth.make_codes(N, 2), interp, ctx, phase, synth.make_signature(3), device=dev)
knn = CodeKNN(db, rng=np.random.RandomState(123456))
sc, sp = knn.init_code_phase()
# G:lanes[:stagger]   stagger 1 (default): a lane's replay waits for the previous lane's SWEEP; 0: launched at once
combos = [(tuple(int(x) for x in a.split(":")) + (1,))[:3] for a in sys.argv[1:]] or \
    [(1, 1, 1), (2, 2, 0), (2, 2, 1), (2, 3, 1), (4, 1, 1), (4, 2, 0), (4, 2, 1), (4, 3, 1), (8, 2, 0), (8, 2, 1), (16, 1, 1)]
Gmax = max(c[0] for c in combos)
g_ = torch.Generator(device="cpu").manual_seed(7)
te_i = torch.randn((Gmax * M, 180, 1024), generator=g_).to(dev)
te_c = torch.randn((Gmax * M, 30, 384), generator=g_).to(dev)
want = [knn.match_clip(te_i[c * M:(c + 1) * M], te_c[c * M:(c + 1) * M], M, seed_code=sc, seed_phase=sp)[0] for c in range(Gmax)]
for G, depth, stag in combos:
    pipe = GraphPipeline(db, M, clips_per_replay=G, depth=depth, rng=np.random.RandomState(1), stagger=bool(stag))
    for ln in range(depth):
        a_, c_ = pipe.buffers(ln)
        a_.copy_(te_i[:G * M])
        c_.copy_(te_c[:G * M])

    def run(n):
        pend, res = [], None
        for _ in range(n):
            if len(pend) == depth:
                res = pipe.collect(pend.pop(0))
            pend.append(pipe.submit(None, None, sc, sp))
        while pend:
            res = pipe.collect(pend.pop(0))
        return res
    n = max(8, 240 // G)
    run(max(4, n // 4))
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = run(n)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / (n * G))
    ok = all(np.array_equal(res[c][0], want[c]) for c in range(G))
    print("clips/replay %2d lanes %d stagger %d: %.4f ms per clip (%.2f M frames/s), %.3f ms per replay, codes equal one-clip path: %s, "
          "rematched %d, stagger timeouts %d" % (G, depth, int(pipe.stagger), best * 1e3, 240 * M / best / 1e6, best * 1e3 * G, ok,
                                                 pipe.rematched, pipe.stagger_timeouts), flush=True)
    del pipe
