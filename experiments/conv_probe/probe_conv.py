"""Where does the conv kernel's time go?  Builds libqpg_hip.so variants with parts of the K loop compiled out
(-DQPG_CONV_PROBE=1: no global fetch; =2: no fetch, no LDS commit, no barriers) and times the B=256 encode.
Results are numerically meaningless for the probe builds; only the time matters."""
import os, subprocess, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
variant = sys.argv[1]
import qpgesture_amd._lib as L
if variant != "0":
    so = os.path.join(HERE, "libqpg_probe%s.so" % variant)
    csrc = os.path.join(ROOT, "qpgesture_amd", "csrc")
    srcs = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".hip"))
    if not os.path.exists(so):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                               "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-DQPG_CONV_PROBE=" + variant]
                              + srcs + ["-o", so])
    L.LIB_PATH = so
import torch
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
