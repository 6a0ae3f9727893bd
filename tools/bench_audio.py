"""Micro-benchmark of the audio sweep kernel alone (for rocprofv3 --pmc passes)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from qpgesture_amd import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 48
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = torch.device("cuda:0")
T, F, G = 180, 1024, 26
PADF, PADQ = int(os.environ.get("QPG_PADF", 0)), int(os.environ.get("QPG_PADQ", 0))   # experiments/audio_mx strides
base = torch.randn((N, T, F + PADF), device=dev)
q32 = torch.randn((Q, 6 * F + PADQ), device=dev)
qn2 = (q32.double() ** 2).sum(1)
cn2 = torch.rand((N, G), device=dev, dtype=torch.float64) + 6000
cand_t = torch.arange(G, device=dev, dtype=torch.int32) * 6
D = torch.empty((Q, N * G), device=dev, dtype=torch.float64)
MX = len(sys.argv) > 4 and sys.argv[4] == "mx"       # the mixed-precision sweep (f32 matrix cores, bounded error)
STATS = torch.zeros((8,), dtype=torch.int64, device=dev) if os.environ.get("QPG_TIMING") else None
def run():
    if MX:
        _lib.call("qpg_audio_cosine_mx", dev, base, N, T, F, cand_t, G, 6, 2, cn2, q32, qn2, Q, D, 0, D.stride(0), STATS)
    else:
        _lib.call("qpg_audio_cosine_f64", dev, base, N, T, F, cand_t, G, 6, 2, cn2, q32, qn2, Q, D, D.stride(0))
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
print("N=%d Q=%d  %.1f us  %.2f TF %s" % (N, Q, ms * 1e3, 2.0 * Q * N * G * 6 * F / ms / 1e9, "f32-mx" if MX else "f64"))
if STATS is not None:      # QPG_MX_TIMING build: per-wave cycle counters (s_memtime ticks) summed over all launches
    st = STATS.cpu().numpy()
    print("timing: waves=%d  loop ticks/wave=%.0f  wait ticks/wave=%.0f (%.1f%%)" % (st[2], st[0] / st[2], st[1] / st[2], 100.0 * st[1] / st[0]))
