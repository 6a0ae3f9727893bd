"""Where does the conv kernel's time go?  Builds libqpg_hip.so variants with parts of the K loop compiled out
(-DQPG_CONV_PROBE=1: no global fetch; =2: no fetch, no LDS commit, no barriers) and times the B=256 encode.
Results are numerically meaningless for the probe builds; only the time matters."""
import os, subprocess, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
variant = sys.argv[1]
import qpgesture_amd._lib as L
if variant.endswith(".so"):
    L.LIB_PATH = os.path.abspath(variant)
elif variant not in ("0", "a0"):
    so = os.path.join(HERE, "libqpg_probe%s.so" % variant)
    csrc = os.path.join(ROOT, "qpgesture_amd", "csrc")
    srcs = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".hip"))
    if not os.path.exists(so):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                               "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
                               "-DQPG_AUDIO_PROBE=1" if variant == "a" else "-DQPG_CONV_PROBE=" + variant]
                              + srcs + ["-o", so])
    L.LIB_PATH = so
import torch
if variant in ("a", "a0"):
    if variant == "a0":
        pass
    N, Q, T, F, G = 2048, 48, 180, 1024, 26
    dev = torch.device("cuda:0")
    base = torch.randn((N, T, F), device=dev)
    q32 = torch.randn((Q, 6 * F), device=dev)
    qn2 = (q32.double() ** 2).sum(1)
    cn2 = torch.rand((N, G), device=dev, dtype=torch.float64) + 6000
    cand_t = torch.arange(G, device=dev, dtype=torch.int32) * 6
    D = torch.empty((Q, N * G), device=dev, dtype=torch.float64)
    run = lambda: L.call("qpg_audio_cosine_f64", dev, base, N, T, F, cand_t, G, 6, 2, cn2, q32, qn2, Q, D, D.stride(0))
    for _ in range(3): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): run()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 10 * 1e3
    print("audio variant %s: %.1f us (%.1f TF-equivalent)" % (variant, ms * 1e3, 2.0 * Q * N * G * 6 * F / ms / 1e9))
    sys.exit(0)
from qpgesture_amd import synth
from qpgesture_amd.vqvae import VQVAE
dev = torch.device("cuda:0")
m = VQVAE(None, 135, device=dev).load_state_dict(synth.make_vqvae_state_dict(7))
x = torch.randn((256, 240, 135), device=dev)
for _ in range(2): m.encode(x)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): m.encode(x)
torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 10
print("variant %s: encode B=256 %.3f ms (%.1f TF-equivalent)" % (variant, te * 1e3, 1.639e9 * 256 / te / 1e12))
