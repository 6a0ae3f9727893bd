"""experiments: phase timing of percode_select_mixed_f64_kernel (library built with -DQPG_SEL_TIMING=1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from qpgesture_amd import synth
from qpgesture_amd.code_knn import CodeKNN, GestureDB
from qpgesture_amd.data_processing import interp_wavlm
N, M = 2048, 6
dev = "cuda:0"
code = synth.make_codes(N, 2); sig = synth.make_signature(3)
phase = np.zeros((N, 240, 4, 8), np.float32)
interp, ctx = bench.chunked_db(N, 0, N, seed=0)
db = GestureDB(code, interp, ctx, phase, sig, device=dev)
knn = CodeKNN(db, rng=np.random.RandomState(1))
knn._guard_stats = torch.zeros((128,), dtype=torch.int32, device=dev)
te = synth.make_db(M, 1000)
ti = torch.from_numpy(interp_wavlm(te["wavlm"])).to(dev); tc = torch.from_numpy(np.ascontiguousarray(te["context"].squeeze(2))).to(dev)
for _ in range(3):
    knn.sweep_tables(ti, tc, M)
torch.cuda.synchronize()
t = knn._guard_stats.cpu().numpy()[8:8 + 16].view(np.uint64)
names = ["init", "pass1", "pass2", "list a", "rank+list b", "tier1 dots", "tier1 merge+tier2", "store+rank"]
print("stats", knn.mixed_stats())
print("tier-1 list length per query:", knn._guard_stats.cpu().numpy()[64:64 + 48])
for i in range(1, 8):
    print("%-18s %8.1f us (at 100 MHz ticks: %d)" % (names[i], (int(t[i]) - int(t[i - 1])) / 100.0, int(t[i]) - int(t[i - 1])))
