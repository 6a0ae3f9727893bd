"""Why does the FIRST pre-launched replay of a fresh capture lose its results?  Probe: B's seed block holds seed 3 when B is
pre-launched; its doorbell is rung later with seed 5.  Snapshots of B's pinned result block tell whether B ran EARLY (codes of
seed 3 appear before the ring) or never wrote at all."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from qpgesture_amd import synth
from qpgesture_amd.code_knn import CodeKNN, GestureDB
N, M = 512, 2
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
db = GestureDB(synth.make_codes(N, 2), torch.randn((N, 180, 1024), device=dev), rng.standard_normal((N, 30, 384)).astype(np.float32),
               rng.standard_normal((N, 240, 4, 8)).astype(np.float32), synth.make_signature(3), device=dev)
knn = CodeKNN(db, rng=np.random.RandomState(1))
te_i = torch.randn((M, 180, 1024), device=dev); te_c = torch.randn((M, 30, 384), device=dev)
seeds = [knn.init_code_phase() for _ in range(7)]
want = [knn.match_clip(te_i, te_c, M, seed_code=c, seed_phase=p)[0].reshape(-1) for c, p in seeds]
def who(buf):
    codes = buf[:M * 30].astype(np.int64)
    for k, w in enumerate(want):
        if np.array_equal(codes, w): return "codes of seed %d" % k
    return "sentinel" if (buf[:M * 30] == -1234567).all() else "other"
for trial, delay in enumerate((0.0, 0.0, 0.05)):
    A, B = [CodeKNN(db, rng=np.random.RandomState(7 + i)).capture_clip_graph(M, audio=te_i, context=te_c, doorbell=True) for i in range(2)]
    A.launch(*seeds[0])                      # captures A, rings, replays
    B._set_seed(*seeds[3])
    B._capture()
    torch.cuda.synchronize()
    print("trial %d: after B's capture: B holds %s, counter %d, go %d, host seq %d" % (trial, who(B._pin_np), int(B._db_cnt.item()), int(B._db_go_np[0]), B._db_seq))
    B._pin_np.fill(-1234567)
    B.prelaunch()
    A.wait_ints()
    time.sleep(0.02 + delay)
    print("   pre-launched, not rung, 20+ ms later: B holds %s" % who(B._pin_np))
    B.launch(*seeds[5])
    try:
        got = B.wait_ints()
        print("   rung with seed 5: B returned %s" % who(got))
    except RuntimeError as e:
        print("   rung with seed 5: MISSING (%s); B holds %s" % (str(e)[-90:], who(B._pin_np)))
    torch.cuda.synchronize()
