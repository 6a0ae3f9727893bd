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
base = torch.randn((N, T, F), device=dev)
q32 = torch.randn((Q, 6 * F), device=dev)
qn2 = (q32.double() ** 2).sum(1)
cn2 = torch.rand((N, G), device=dev, dtype=torch.float64) + 6000
cand_t = torch.arange(G, device=dev, dtype=torch.int32) * 6
D = torch.empty((Q, N * G), device=dev, dtype=torch.float64)
def run():
    _lib.call("qpg_audio_cosine_f64", dev, base, N, T, F, cand_t, G, 6, 2, cn2, q32, qn2, Q, D, D.stride(0))
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
print("N=%d Q=%d  %.1f us  %.2f TF f64" % (N, Q, ms * 1e3, 2.0 * Q * N * G * 6 * F / ms / 1e9))
