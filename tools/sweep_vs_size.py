"""The two-plane audio sweep (audio_cosine_hl2_kernel<2, 2, true>, Q = 48) at N_db = 2048 / 4096 / 8192 - warm (back-to-back
launches: what a replay loop sees) and cold (a 768 MB buffer rewritten between launches: nothing of the image or the matrix
left in the Infinity Cache / L2) - VERDICT r5 weak #6: the N = 2048 figure (0.67 of 8 TB/s) against N = 8192 (0.56).
    python tools/sweep_vs_size.py [N ...]          prints one markdown table row per (N, mode)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from qpgesture_amd import _lib

Ns = [int(a) for a in sys.argv[1:]] or [2048, 4096, 8192]
Q = 48
dev = torch.device("cuda:0")
T, F, G = 180, 1024, 26
lib = _lib.load()
trash = torch.zeros((768 << 20) // 4, dtype=torch.float32, device=dev)
print("| N_db | image MB | algorithmic MB | mode | min us | median us | TB/s (median) | frac of 8 TB/s |")
print("|---|---|---|---|---|---|---|---|")
for N in Ns:
    base = torch.randn((N, T, F), device=dev)
    q32 = torch.randn((Q, 6 * F), device=dev)
    cand_t = (torch.arange(G, dtype=torch.int32) * 6).to(dev)
    fn2 = torch.empty((N, T), dtype=torch.float64, device=dev)
    _lib.call("qpg_frame_norm2_f64", dev, base, N * T, F, fn2)
    cn2 = torch.empty((N, G), dtype=torch.float64, device=dev)
    _lib.call("qpg_audio_cand_norm2", dev, fn2, N, T, cand_t, G, 6, 2, cn2)
    qn2 = (q32.double() ** 2).sum(1)
    img = torch.empty((int(lib.qpg_audio_hl_db_bytes(N, F)),), dtype=torch.uint8, device=dev)
    _lib.call("qpg_audio_hl_pack_db", dev, base, N, T, F, G, 6, 2, 6, img, img.numel())
    del base, fn2
    qi = torch.empty((int(lib.qpg_audio_hl_query_bytes(Q, F)),), dtype=torch.uint8, device=dev)
    _lib.call("qpg_audio_hl_pack_queries", dev, q32, Q, F, qi, qi.numel())
    stats = torch.zeros((4,), dtype=torch.int32, device=dev)
    D = torch.empty((Q, N * G), dtype=torch.float32, device=dev)

    def sweep():
        _lib.call("qpg_audio_cosine_hl", dev, img, N, F, G, cn2, qi, qn2, Q, D, 1, D.stride(0), stats)
    byt = N * 81 * F * 4 + N * G * 8 + Q * 6 * F * 4 + Q * N * G * 4
    for mode in ("warm", "cold"):
        for _ in range(5):
            sweep()
        n = 30 if mode == "warm" else 16
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for a, b in ev:
            if mode == "cold":
                trash.add_(1.0)                      # 768 MB read + written: evicts the 256 MB Infinity Cache and the L2s
            a.record()
            sweep()
            b.record()
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in ev)
        mn, med = ts[0] * 1e3, ts[len(ts) // 2] * 1e3
        print("| %d | %.0f | %.0f | %s | %.1f | %.1f | %.2f | %.3f |" % (N, img.numel() / 1e6, byt / 1e6, mode, mn, med,
                                                                     byt / med / 1e6, byt / med / 1e6 / 8.0), flush=True)
    del img, D
    torch.cuda.empty_cache()
