"""Micro-benchmark of VQ-VAE encode / decode."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qpgesture_amd import synth
from qpgesture_amd.vqvae import VQVAE
dev = torch.device("cuda:0")
m = VQVAE(None, 135, device=dev).load_state_dict(synth.make_vqvae_state_dict(7))
def t(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / iters
for B in (1, 16, 64, 256):
    x = torch.randn((B, 240, 135), device=dev)
    te = t(lambda: m.encode(x))
    ids = torch.randint(0, 512, (B, 30), device=dev)
    td = t(lambda: m.decode([ids]))
    print("B=%3d encode %.3f ms (%.1f TF, %.2fM frames/s)   decode %.3f ms (%.1f TF)" % (
        B, te * 1e3, 1.639e9 * B / te / 1e12, 240 * B / te / 1e6, td * 1e3, 1.908e9 * B / td / 1e12))
