"""Prepared-database cache: save / load timing at bench size (N_db = 2048), by reader threads and staging geometry."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from qpgesture_amd import synth, db_cache
from qpgesture_amd.code_knn import GestureDB
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
interp = torch.randn((N, 180, 1024), device=dev)
ctx = rng.standard_normal((N, 30, 384)).astype(np.float32)
phase = rng.standard_normal((N, 240, 4, 8)).astype(np.float32)
t0 = time.perf_counter()
db = GestureDB(synth.make_codes(N, 2), interp, ctx, phase, synth.make_signature(3), device=dev)
torch.cuda.synchronize(); print("build %.3f s" % (time.perf_counter() - t0))
td = tempfile.mkdtemp(prefix="qpg_cache_bench_")
p = os.path.join(td, "db.qpgdb")
t0 = time.perf_counter(); db.save(p, "k"); print("save %.3f s, %.2f GB" % (time.perf_counter() - t0, os.path.getsize(p) / 1e9))
del db, interp
torch.cuda.empty_cache()
gb = os.path.getsize(p) / 1e9
for readers in (1, 2, 4, 8, 4):
    for rep in range(2):
        t0 = time.perf_counter()
        d2 = db_cache.load(p, dev, "k", n_readers=readers)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("load readers=%d rep=%d: %.3f s = %.1f GB/s" % (readers, rep, dt, gb / dt))
        del d2
t0 = time.perf_counter(); d2 = GestureDB.load(p, dev, "k"); torch.cuda.synchronize()
print("GestureDB.load (incl. the selfcheck of this process): %.3f s" % (time.perf_counter() - t0))
t0 = time.perf_counter(); d3 = GestureDB.load(p, dev, "k"); torch.cuda.synchronize()
print("GestureDB.load again: %.3f s" % (time.perf_counter() - t0))
import shutil; shutil.rmtree(td, ignore_errors=True)
