"""Host side of one matching step (bench geometry): where the ~60 us between a step's last GPU event and the next step's
first go.  perf_counter stamps around every _lib.call and around the final .cpu()."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from qpgesture_amd import synth, _lib
from qpgesture_amd.code_knn import CodeKNN, GestureDB
N, M = 2048, 6
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
db = GestureDB(synth.make_codes(N, 2), torch.randn((N, 180, 1024), device=dev), rng.standard_normal((N, 30, 384)).astype(np.float32),
               rng.standard_normal((N, 240, 4, 8)).astype(np.float32), synth.make_signature(3), device=dev)
knn = CodeKNN(db, rng=np.random.RandomState(123456))
te_i = torch.randn((M, 180, 1024), device=dev); te_c = torch.randn((M, 30, 384), device=dev)
sc, sp = knn.init_code_phase(); spd = torch.from_numpy(sp).to(dev)
log = []
orig = _lib.call
def call(name, *a):
    t0 = time.perf_counter(); r = orig(name, *a); log.append((name, t0, time.perf_counter())); return r
def step():
    T = knn.sweep_tables(te_i, te_c, M)
    o = knn.walk(T, M, 0, seed_code=sc, seed_phase=spd, sync=False)[0]
    t0 = time.perf_counter(); r = o.cpu(); log.append(("<.cpu()>", t0, time.perf_counter())); return r
for _ in range(20): step()
_lib.call = call
import qpgesture_amd.code_knn as ck, qpgesture_amd.sorted_rows as sr
torch.cuda.synchronize()
rows = []
for it in range(50):
    log.clear(); t0 = time.perf_counter(); step(); t1 = time.perf_counter()
    rows.append([(n, a - t0, b - a) for n, a, b in log] + [("<step>", 0.0, t1 - t0)])
med = rows[len(rows) // 2]
print("%-40s %9s %9s" % ("call", "start us", "dur us"))
for i, (n, a, d) in enumerate(rows[25]):
    ds = sorted(r[i][2] for r in rows if len(r) == len(rows[25]))
    print("%-40s %9.1f %9.1f (median %.1f)" % (n, a * 1e6, d * 1e6, ds[len(ds) // 2] * 1e6))
