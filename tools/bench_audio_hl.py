"""Kernel timing of the split-operand f16 audio sweep vs the f32-matrix-core one (HIP events, N_db = 2048, Q = 48)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from qpgesture_amd import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 48
dev = torch.device("cuda:0"); T, F, G = 180, 1024, 26
base = torch.randn((N, T, F), device=dev); q32 = torch.randn((Q, 6 * F), device=dev)
cand_t = (torch.arange(G, dtype=torch.int32) * 6).to(dev)
fn2 = torch.empty((N, T), dtype=torch.float64, device=dev)
_lib.call("qpg_frame_norm2_f64", dev, base, N * T, F, fn2)
cn2 = torch.empty((N, G), dtype=torch.float64, device=dev)
_lib.call("qpg_audio_cand_norm2", dev, fn2, N, T, cand_t, G, 6, 2, cn2)
qn2 = (q32.double() ** 2).sum(1)
lib = _lib.load()
img = torch.empty((int(lib.qpg_audio_hl_db_bytes(N, F)),), dtype=torch.uint8, device=dev)
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record(); _lib.call("qpg_audio_hl_pack_db", dev, base, N, T, F, G, 6, 2, 6, img, img.numel()); t1.record()
torch.cuda.synchronize(); print("pack_db %.3f ms" % t0.elapsed_time(t1))
qi = torch.empty((int(lib.qpg_audio_hl_query_bytes(Q, F)),), dtype=torch.uint8, device=dev)
stats = torch.zeros((4,), dtype=torch.int32, device=dev)
D = torch.empty((Q, N * G), dtype=torch.float32, device=dev)
def timed(fn, n=30):
    for _ in range(5): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[0], ts[len(ts) // 2]
pq = timed(lambda: _lib.call("qpg_audio_hl_pack_queries", dev, q32, Q, F, qi, qi.numel()))
hl = timed(lambda: _lib.call("qpg_audio_cosine_hl", dev, img, N, F, G, cn2, qi, qn2, Q, D, 1, D.stride(0), stats))
mx = timed(lambda: _lib.call("qpg_audio_cosine_mx", dev, base, N, T, F, cand_t, G, 6, 2, cn2, q32, qn2, Q, D, 1, D.stride(0), stats))
byt = N * 81 * F * 4 + N * G * 8 + Q * 6 * F * 4 + Q * N * G * 4
print("N=%d Q=%d: pack_queries min %.1f us | hl sweep min %.1f / med %.1f us = %.2f TB/s algorithmic (%.0f MB) | mx sweep min %.1f / med %.1f us"
      % (N, Q, pq[0] * 1e3, hl[0] * 1e3, hl[1] * 1e3, byt / hl[1] / 1e9, byt / 1e6, mx[0] * 1e3, mx[1] * 1e3))
if hasattr(lib, "qpg_audio_cosine_hl1"):          # round 5: the one-plane image of an f16-stored track
    b16 = base.to(torch.float16)
    _lib.call("qpg_frame_norm2_f64", dev, b16.float(), N * T, F, fn2)
    _lib.call("qpg_audio_cand_norm2", dev, fn2, N, T, cand_t, G, 6, 2, cn2)
    img1 = torch.empty((int(lib.qpg_audio_hl1_db_bytes(N, F)),), dtype=torch.uint8, device=dev)
    _lib.call("qpg_audio_hl1_pack_db", dev, b16, N, T, F, G, 6, 2, 6, img1, img1.numel())
    h1 = timed(lambda: _lib.call("qpg_audio_cosine_hl1", dev, img1, N, F, G, cn2, qi, qn2, Q, D, 1, D.stride(0), stats))
    byt1 = N * 81 * F * 2 + N * G * 8 + Q * 6 * F * 4 + Q * N * G * 4
    chunks = (Q + 47) // 48
    mf = 2 * N * 2 * 6 * 96 * chunks * 16 * 16 * 32 * 2
    print("N=%d Q=%d: hl1 (one-plane f16 track) sweep min %.1f / med %.1f us = %.2f TB/s algorithmic (%.0f MB), %.0f TFLOP/s f16 issued"
          % (N, Q, h1[0] * 1e3, h1[1] * 1e3, byt1 / h1[1] / 1e9, byt1 / 1e6, mf / h1[1] / 1e9))
