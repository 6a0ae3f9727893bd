"""cfg-3 (bounded prefilter path): HIP-event time of every launch of CosineIndex.query for the 1 000-query batch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from qpgesture_amd import _lib, cfg3
X, code, valid, q = cfg3.make_inputs()
index = cfg3.CosineIndex(X, code, valid)
index.check_flags = False
qd = torch.from_numpy(q).cuda()
times = {}
orig = _lib.call
def call(name, dev, *a):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = orig(name, dev, *a); e1.record()
    times.setdefault(name, []).append((e0, e1))
    return r
for _ in range(3): index.query(qd)
cfg3._lib.call = call
for _ in range(10): index.query(qd)
torch.cuda.synchronize()
for k, v in times.items():
    ts = sorted(a.elapsed_time(b) for a, b in v)
    print("%-34s min %.3f  median %.3f ms" % (k, ts[0], ts[len(ts) // 2]))
print("R =", index.R, " band =", index.band)
# ablation: band < 0 makes the select skip its list and refine (phase 1 only: the segmented per-code minimum)
times.clear()
index.band = -1.0
for _ in range(5): index.query(qd)
torch.cuda.synchronize()
ts = sorted(a.elapsed_time(b) for a, b in times["qpg_percode_select_sorted_f32"])
print("select, phase 1 only (band < 0): median %.3f ms" % ts[len(ts) // 2])
