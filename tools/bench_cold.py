"""Cold path: host arrays (as load_db_codebook leaves them) -> resident GestureDB, and clip H2D + match (N_db=2048)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from qpgesture_amd import synth
from qpgesture_amd.code_knn import CodeKNN, GestureDB
from qpgesture_amd.data_processing import interp_wavlm_device
N, M = 2048, 6
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
wavlm = rng.standard_normal((N, 199, 1024), dtype=np.float32)           # raw WavLM track of the DB (1.63 GB)
ctx = rng.standard_normal((N, 30, 384), dtype=np.float32)
phase = rng.standard_normal((N, 240, 4, 8), dtype=np.float32)
code = synth.make_codes(N, 2)
sig = synth.make_signature(3)
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    interp = interp_wavlm_device(wavlm, dev)                # chunked H2D + 199 -> 180 resample on the device
    torch.cuda.synchronize(); t2 = time.perf_counter()
    db = GestureDB(code, interp, ctx, phase, sig, device=dev)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print("DB build rep %d: H2D + resample of the WavLM track %.1f ms (%.1f GB/s from pageable host memory), norms / packing / rank tables %.1f ms, total %.1f ms" % (
        rep, (t2 - t0) * 1e3, wavlm.nbytes / (t2 - t0) / 1e9, (t3 - t2) * 1e3, (t3 - t0) * 1e3))
knn = CodeKNN(db, rng=np.random.RandomState(1))
clip_w = rng.standard_normal((M, 199, 1024), dtype=np.float32)
clip_c = rng.standard_normal((M, 30, 384), dtype=np.float32)
sc, sp = knn.init_code_phase()
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ti = interp_wavlm_device(clip_w, dev)
    tc = torch.from_numpy(clip_c).to(dev)
    codes = knn.match_clip(ti, tc, M, seed_code=sc, seed_phase=sp)[0]
    t1 = time.perf_counter()
    print("clip rep %d: H2D + resample + match + D2H = %.3f ms" % (rep, (t1 - t0) * 1e3))
