"""Eager vs hipGraph replay of the per-clip launch sequence (N_db=2048, M=6)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from qpgesture_amd import synth
from qpgesture_amd.code_knn import CodeKNN, GestureDB
N, M = int(sys.argv[1]) if len(sys.argv) > 1 else 2048, 6
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
interp = torch.randn((N, 180, 1024), device=dev)
ctx = rng.standard_normal((N, 30, 384)).astype(np.float32)
phase = rng.standard_normal((N, 240, 4, 8)).astype(np.float32)
db = GestureDB(synth.make_codes(N, 2), interp, ctx, phase, synth.make_signature(3), device=dev)
knn = CodeKNN(db, rng=np.random.RandomState(123456))
te_i = torch.randn((M, 180, 1024), device=dev)
te_c = torch.randn((M, 30, 384), device=dev)
sc, sp = knn.init_code_phase()
spd = torch.from_numpy(sp).to(dev)
def eager():
    T = knn.sweep_tables(te_i, te_c, M)
    return knn.walk(T, M, 0, seed_code=sc, seed_phase=spd, sync=False)[0].cpu()
g = knn.capture_clip_graph(M, audio=te_i, context=te_c)
def graph():
    return torch.from_numpy(g.run_ints(sc, sp)[:M * 30]).view(M, 30)
a, b = eager(), graph()
print("graph == eager:", torch.equal(a, b))
def t(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / iters * 1e3
print("eager %.4f ms/clip   graph %.4f ms/clip" % (t(eager), t(graph)))
