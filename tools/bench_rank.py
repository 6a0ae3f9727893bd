"""HIP-event time of the rank kernels (qpg_rank_rows_f32 / _f64) for Q rows of K values, back to back."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qpgesture_amd import _lib
dev = torch.device("cuda:0")
for Q, K in ((48, 512), (1000, 512), (48, 2048)):
    for dt, name in ((torch.float32, "qpg_rank_rows_f32"), (torch.float64, "qpg_rank_rows_f64")):
        d = torch.rand((Q, K), device=dev, dtype=dt)
        out = torch.empty((Q, K), dtype=torch.int16, device=dev)
        for _ in range(5):
            _lib.call(name, dev, d, Q, K, out)
        e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        e[0].record()
        for _ in range(50):
            _lib.call(name, dev, d, Q, K, out)
        e[1].record()
        torch.cuda.synchronize()
        ref = torch.argsort(torch.argsort(d, dim=1, stable=True), dim=1).to(torch.int16)
        print("%-18s Q=%4d K=%4d  %.1f us per launch   %s" % (name, Q, K, e[0].elapsed_time(e[1]) * 20,
                                                             "ok" if torch.equal(ref, out) else "MISMATCH"))
