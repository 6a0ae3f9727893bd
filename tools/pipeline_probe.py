"""experiments: clips in flight.  The default bench step (one 24 s clip vs the 2048-window DB) issued serially, or with
D lanes (one CodeKNN + HIP stream + pinned result buffer each) so that clip i+1's sweep runs under clip i's select / walk /
D2H.  python tools/pipeline_probe.py [graph]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from qpgesture_amd import synth
from qpgesture_amd.code_knn import CodeKNN, GestureDB
from qpgesture_amd.data_processing import interp_wavlm

use_graph = len(sys.argv) > 1 and sys.argv[1] == "graph"
dev = torch.device("cuda:0")
N, M = 2048, 6
code = synth.make_codes(N, 2)
sig = synth.make_signature(3)
phase = np.random.Generator(np.random.PCG64(5)).standard_normal((N, 240, 4, 8)).astype(np.float32)
interp, ctx = bench.chunked_db(N, 0, N, seed=0)
db = GestureDB(code, interp, ctx, phase, sig, device=dev)
clip = synth.make_db(M, 1000)
te_i = torch.from_numpy(interp_wavlm(clip["wavlm"])).to(dev)
te_c = torch.from_numpy(clip["context"].squeeze(2)).to(dev)


class Lane:
    def __init__(self):
        self.knn = CodeKNN(db, rng=np.random.RandomState(123456))
        self.stream = torch.cuda.Stream(dev)
        self.host = torch.empty((M * 30,), dtype=torch.int32).pin_memory()
        self.event = torch.cuda.Event()
        self.busy = False
        self.graph = None

    def submit(self, sc, sp):
        with torch.cuda.stream(self.stream):
            if use_graph:
                if self.graph is None:
                    self.graph = self.knn.capture_clip_graph(M)
                oc = self.graph.run(te_i, te_c, sc, sp)[0]
            else:
                T = self.knn.sweep_tables(te_i, te_c, M)
                oc = self.knn.walk(T, M, 0, seed_code=sc, seed_phase=sp, sync=False)[0]
            self.host.copy_(oc.reshape(-1), non_blocking=True)
            self.event.record(self.stream)
        self.busy = True

    def collect(self):
        self.event.synchronize()
        self.busy = False
        return self.host


def run(D, steps=300, warm=20):
    lanes = [Lane() for _ in range(D)]
    sc, sp = lanes[0].knn.init_code_phase()
    spd = torch.from_numpy(sp).to(dev)
    torch.cuda.synchronize()
    res = None
    for phase_ in (warm, steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(phase_):
            ln = lanes[i % D]
            if ln.busy:
                res = ln.collect().clone()
            ln.submit(sc, spd)
        for ln in lanes:
            if ln.busy:
                res = ln.collect().clone()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return dt / steps * 1e3, res


base = None
for D in (1, 2, 3):
    ms, res = run(D)
    if base is None:
        base = res
    print("%s lanes=%d: %.4f ms per clip (%.2f M frames/s)  same codes: %s" % ("graph" if use_graph else "eager", D, ms, 1440 / ms / 1e3,
                                                                          bool(torch.equal(res, base))), flush=True)
