"""Micro-benchmark of one VQ-VAE training step (forward / backward / Adam) at the reference's batch size."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qpgesture_amd import synth
from qpgesture_amd.optim import Adam
from qpgesture_amd.vqvae import VQVAE
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
m = VQVAE(dict(vel=1, acc=1), 135, device=dev).load_state_dict(synth.make_vqvae_state_dict(7)).train()
m.train_fused = os.environ.get("QPG_TRAIN_FUSED", "1") == "1"       # forward on the transposed-formulation kernels
opt = Adam(m.parameters(), lr=3e-5, betas=(0.5, 0.999))
x = torch.randn((B, 240, 135), device=dev)
FWD = (1.639e9 + 1.908e9) * B           # encoder + decoder flops per window
def t(fn, iters=5):
    for _ in range(6): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / iters
def fwd(): m(x)
def fb(): m(x); m.backward()
def full(): m(x); m.backward(); opt.step()
tf = t(fwd); tfb = t(fb); tall = t(full)
print("B=%d forward %.2f ms (%.1f TF)  fwd+bwd %.2f ms (bwd %.2f ms, %.1f TF)  step %.2f ms (adam %.2f ms)  %.0f windows/s" % (
    B, tf * 1e3, FWD / tf / 1e12, tfb * 1e3, (tfb - tf) * 1e3, 2 * FWD / (tfb - tf) / 1e12, tall * 1e3, (tall - tfb) * 1e3, B / tall))
if os.environ.get("QPG_TRAIN_F16X3", "1") == "1":
    # round 5: the forward convolutions on the split-f16 kernels (opt-in: VQVAE.train_precision = "f16x3"); backward in f32
    m.train_precision = "f16x3"
    tf = t(fwd); tfb = t(fb); tall = t(full)
    print("B=%d train_precision=f16x3: forward %.2f ms (%.1f TF f32-equivalent)  fwd+bwd %.2f ms (bwd %.2f ms)  step %.2f ms  %.0f windows/s"
          % (B, tf * 1e3, FWD / tf / 1e12, tfb * 1e3, (tfb - tf) * 1e3, tall * 1e3, B / tall))
    m.train_precision = "f32"
print("params", m.param.numel(), "peak mem GB", torch.cuda.max_memory_allocated() / 1e9)
