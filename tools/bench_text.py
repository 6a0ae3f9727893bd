"""Micro-benchmark of the text sweep kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qpgesture_amd import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 48
dev = torch.device("cuda:0")
Dm, G = 384, 26
C = N * G
xt = torch.randn((((C + 63) // 64) * 64 * Dm,), device=dev)
qn = torch.randn((Q, Dm), device=dev)
D = torch.empty((Q, C), device=dev)
def t(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
u = lambda: _lib.call("qpg_text_cosine_f32", dev, xt, C, Dm, qn, Q, D, D.stride(0))
print("N=%d Q=%d  %.1f us" % (N, Q, t(u)))
