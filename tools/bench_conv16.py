"""Split-f16 convolutions against the f32 kernels: per-layer timing at the encoder's shapes and the B = 256 encode."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from qpgesture_amd import _lib, synth
from qpgesture_amd.vqvae import VQVAE
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
m = VQVAE(None, 135, device=dev).load_state_dict(synth.make_vqvae_state_dict(7))
x = torch.randn((B, 240, 135), device=dev)


def timed(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


gf = 1.639 * B
t_fused = timed(lambda: m.encode_fused(x))
t_l32 = timed(lambda: m.encode_latent(x))
t_l16 = timed(lambda: m.encode_latent(x, precision="f16x3"))
t_e16 = timed(lambda: m.encode_f16x3(x))
ids, st = m.encode_f16x3(x, return_stats=True)
print("B=%d encode: fused f32 %.3f ms (%.0f TF/s) | layerwise f32 %.3f | layerwise f16x3 latents %.3f ms (%.0f TF/s f32-equivalent) | "
      "encode_f16x3 incl. margin check %.3f ms -> %.1f M frames/s; %s; ids equal: %s"
      % (B, t_fused, gf / t_fused, t_l32, t_l16, gf / t_l16, t_e16, 240 * B / t_e16 / 1e3, st,
         bool(torch.equal(ids, m.encode(x)[0]))))
# one k3 512 -> 512 layer at T = 120
c3 = m.enc_down[0][1][0][0]
h = torch.randn((B, 120, 512), device=dev)
t32 = timed(lambda: m._conv(c3, h, B, 120, 120, in_offset=-1, dil=1, relu_in=True, relu_out=True))
t16 = timed(lambda: m._conv16(c3, h, B, 120, 120, in_offset=-1, dil=1, relu_in=True, relu_out=True))
fl = 2.0 * B * 120 * 1536 * 512 / 1e9
print("k3 512->512 T=120 B=%d: LDS-tiled f32 %.3f ms (%.0f TF/s) | conv16 %.3f ms (%.0f TF/s f32-equivalent, %.0f TF/s f16 issued)"
      % (B, t32, fl / t32, t16, fl / t16, 3 * fl / t16))
