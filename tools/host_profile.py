import os, sys, time, cProfile, pstats
sys.path.insert(0, "/root/repo")
import numpy as np, torch, bench
from qpgesture_amd import synth
from qpgesture_amd.code_knn import CodeKNN, GestureDB
from qpgesture_amd.data_processing import interp_wavlm
dev = torch.device("cuda:0"); N, M = 2048, 6
code = synth.make_codes(N, 2); sig = synth.make_signature(3)
phase = np.random.Generator(np.random.PCG64(5)).standard_normal((N, 240, 4, 8)).astype(np.float32)
interp, ctx = bench.chunked_db(N, 0, N, seed=0)
db = GestureDB(code, interp, ctx, phase, sig, device=dev)
clip = synth.make_db(M, 1000)
te_i = torch.from_numpy(interp_wavlm(clip["wavlm"])).to(dev); te_c = torch.from_numpy(clip["context"].squeeze(2)).to(dev)
knn = CodeKNN(db, rng=np.random.RandomState(123456))
sc, sp = knn.init_code_phase(); spd = torch.from_numpy(sp).to(dev)
def enq():
    T = knn.sweep_tables(te_i, te_c, M)
    return knn.walk(T, M, 0, seed_code=sc, seed_phase=spd, sync=False)[0]
for _ in range(10): enq().cpu()
torch.cuda.synchronize()
te = ts = 0.0
for _ in range(200):
    t0 = time.perf_counter(); o = enq(); t1 = time.perf_counter(); o.cpu(); t2 = time.perf_counter()
    te += t1 - t0; ts += t2 - t1
print("host enqueue %.1f us, wait+D2H %.1f us per clip" % (te / 200 * 1e6, ts / 200 * 1e6))
# enqueue cost with the GPU idle in between (pure host cost)
te = 0.0
for _ in range(100):
    torch.cuda.synchronize(); t0 = time.perf_counter(); o = enq(); te += time.perf_counter() - t0
print("host enqueue (GPU idle at start) %.1f us" % (te / 100 * 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(200): enq().cpu()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
