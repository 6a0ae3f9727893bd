"""experiments: per-step wall-time distribution of the default matching step (one clip at a time) - is a slow run a
uniform shift or a few long steps?  python tools/step_jitter.py [text-first]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from qpgesture_amd import synth
from qpgesture_amd.code_knn import CodeKNN, GestureDB
from qpgesture_amd.data_processing import interp_wavlm
dev = torch.device("cuda:0"); N, M = 2048, 6
code = synth.make_codes(N, 2); sig = synth.make_signature(3)
phase = np.random.Generator(np.random.PCG64(5)).standard_normal((N, 240, 4, 8)).astype(np.float32)
interp, ctx = bench.chunked_db(N, 0, N, seed=0)
if "like-bench" in sys.argv[1:]:       # the DB built the way bench.py builds it
    import torch.distributed as dist    # noqa: F401
    torch.cuda.set_device(dev)
    db = GestureDB(code, bench._ShardView(interp, 0, N, N), bench._ShardView(ctx, 0, N, N), phase, sig, device=dev, rank=0,
                   world=1, feature_dtype="f32")
else:
    db = GestureDB(code, interp, ctx, phase, sig, device=dev)
clip = synth.make_db(M, 1000)
te_i = torch.from_numpy(interp_wavlm(clip["wavlm"])).to(dev); te_c = torch.from_numpy(clip["context"].squeeze(2)).to(dev)
knn = CodeKNN(db, rng=np.random.RandomState(123456))
knn.text_after_sweep = "text-first" not in sys.argv[1:]
sc, sp = knn.init_code_phase(); spd = torch.from_numpy(sp).to(dev)
def step():
    t0 = time.perf_counter()
    T = knn.sweep_tables(te_i, te_c, M)
    o = knn.walk(T, M, 0, seed_code=sc, seed_phase=spd, sync=False)[0]
    t1 = time.perf_counter()
    o.cpu()
    return t1 - t0, time.perf_counter() - t1
for _ in range(10): step()
torch.cuda.synchronize()
if "events" in sys.argv[1:]:
    knn.kernel_events = []
for rep in range(3):
    t0 = time.perf_counter()
    r = np.array([step() for _ in range(400)]) * 1e6
    tot = (time.perf_counter() - t0) / 400 * 1e6
    w = r.sum(1)
    print("%s rep %d: mean %.0f us (loop %.0f)  median %.0f  p90 %.0f  p99 %.0f  max %.0f | enqueue median %.0f p99 %.0f | wait median %.0f p99 %.0f | "
          "time in steps > 1.5x median: %.0f%%" % ("after" if knn.text_after_sweep else "text-first", rep, w.mean(), tot, np.median(w), np.percentile(w, 90),
                                                 np.percentile(w, 99), w.max(), np.median(r[:, 0]), np.percentile(r[:, 0], 99),
                                                 np.median(r[:, 1]), np.percentile(r[:, 1], 99),
                                                 100 * w[w > 1.5 * np.median(w)].sum() / w.sum()), flush=True)
