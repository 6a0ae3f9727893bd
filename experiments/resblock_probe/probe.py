"""Where does the fused ResConv1DBlock kernel's time go?  Builds libqpg_hip.so variants with parts of the stage
compiled out (-DQPG_RES_PROBE=<bits>, hooks in csrc/qpg_convt.hip) and times qpg_resblock_f32 alone at
B = 256, T = 60 (240 workgroups: one per CU) and T = 120 (480).  Probe builds compute garbage; only time matters.
    python experiments/resblock_probe/probe.py <bits> [sched]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
bits = int(sys.argv[1])
os.environ["QPG_RES_SCHED"] = sys.argv[2] if len(sys.argv) > 2 else "1"
import qpgesture_amd._lib as L  # noqa: E402

if bits:
    so = os.path.join(HERE, "libqpg_probe%d.so" % bits)
    csrc = os.path.join(ROOT, "qpgesture_amd", "csrc")
    srcs = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".hip"))
    if not os.path.exists(so):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                               "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
                               "-DQPG_RES_PROBE=%d" % bits] + srcs + ["-o", so])
    L.LIB_PATH = so
import torch  # noqa: E402

dev = torch.device("cuda:0")
pack = torch.randn((128 * 8192,), device=dev) * 0.03
b1 = torch.randn((512,), device=dev) * 0.05
b2 = torch.randn((512,), device=dev) * 0.05
for T in (60, 120):
    x = torch.randn((256, T, 512), device=dev)
    y = torch.empty_like(x)
    for _ in range(3):
        L.call("qpg_resblock_f32", dev, x, 256, T, 3, pack, b1, b2, y, None)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for e0, e1 in ev:
        e0.record()
        L.call("qpg_resblock_f32", dev, x, 256, T, 3, pack, b1, b2, y, None)
        e1.record()
    torch.cuda.synchronize()
    ms = sorted(e0.elapsed_time(e1) for e0, e1 in ev)
    med = ms[len(ms) // 2]
    print("probe %2d sched %s T=%3d: median %.1f us  min %.1f us  (%.1f TFLOP/s-equivalent)"
          % (bits, os.environ["QPG_RES_SCHED"], T, med * 1e3, ms[0] * 1e3, 2.0 * 256 * T * 512 * 2048 / med / 1e9))
