"""Debug helper: the crowded DB of tests/test_gpu_guard_overflow.py through sweep_audio with a sync after every stage."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.test_gpu_guard_overflow import _crowded
from qpgesture_amd.code_knn import CodeKNN, GestureDB
from qpgesture_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
prec = sys.argv[2] if len(sys.argv) > 2 else "mixed"
A = _crowded(n)
dev = "cuda:0"
db = GestureDB(A["code"], A["tr_interp"], A["tr_ctx"], A["tr_phase"], A["sig"], device=dev)
torch.cuda.synchronize(); print("db ok", db.N, flush=True)
knn = CodeKNN(db, rng=np.random.RandomState(7))
knn.audio_precision = prec
te_i = torch.from_numpy(A["te_interp"]).to(dev)
te_c = torch.from_numpy(np.ascontiguousarray(A["te_ctx"])).to(dev)
orig = _lib.call
def call(name, device, *args):
    r = orig(name, device, *args)
    torch.cuda.synchronize()
    print("  ok", name, flush=True)
    return r
_lib.call = call
import qpgesture_amd.code_knn as ck
ck._lib.call = call
knn.overlap_sweeps = False
T = knn.sweep_tables(te_i, te_c, 2)
print("tables ok, flags", knn.mixed_stats(), flush=True)
sc, sp = knn.init_code_phase()
try:
    print(knn.walk(T, 2, 0, seed_code=sc, seed_phase=sp)[0][0][:8])
except Exception as e:
    print("walk:", repr(e))
knn.clear_flags()
knn.audio_precision = "exact"
T = knn.sweep_tables(te_i, te_c, 2)
print("exact tables ok", knn.mixed_stats(), flush=True)
print(knn.walk(T, 2, 0, seed_code=sc, seed_phase=sp)[0][0][:8])
