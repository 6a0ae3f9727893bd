"""The h-plane prefilter GEMM of cfg-3 alone (100 000 x 512 rows, 1 000 queries): HIP events, 20 launches, min / median."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from qpgesture_amd import _lib
from qpgesture_amd.cfg3 import CosineIndex, make_inputs
X, code, valid, q = make_inputs()
dev = torch.device("cuda:0")
ix = CosineIndex(X, code, valid, device=dev)
sr = ix.sorted
qd = torch.from_numpy(q).to(dev)
Q, d = qd.shape
ix.check_flags = False
ix.query(qd)
sc = ix._scratch
ldq = sc["tmin_t"].shape[1]
fn = lambda: _lib.call("qpg_hl_gemm_tilemin_h", dev, sr.image, sr.R, d, sc["cols"], Q, sr.band_h, sc["tmin_t"], sc["tmask_t"], ldq)
for _ in range(3): fn()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
for a, b in ev:
    a.record(); fn(); b.record()
torch.cuda.synchronize()
ts = sorted(a.elapsed_time(b) for a, b in ev)
fl = 2.0 * sr.R * ((Q + 95) // 96 * 96) * d
print("gemm64h R=%d Q=%d D=%d: min %.1f med %.1f us = %.0f TFLOP/s f16 issued" % (sr.R, Q, d, ts[0] * 1e3, ts[10] * 1e3, fl / ts[10] / 1e9))
