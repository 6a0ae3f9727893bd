#!/usr/bin/env python
"""Per-layer view of the VQ-VAE encode (B windows of 240 frames): `run` executes `iters` encodes (for
rocprofv3 --kernel-trace), `table <trace.csv>` folds the trace into one row per launch position of the
encoder with the layer's ideal time at the 155 TFLOP/s a register-only f32 MFMA loop sustains.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/enc -- python tools/prof_encode.py run 256 12
    python tools/prof_encode.py table gpurun_out/enc 256 12
"""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def layer_list(B):
    """(name, GFLOP) of every launch of one encode in issue order (codebook.yml architecture)."""
    L = []
    T, cin = 240, 135
    for lvl in range(3):
        T //= 2
        L.append(("down%d k4s2 %d->512 T=%d" % (lvl, cin, T), 2e-9 * B * T * 512 * 4 * cin))
        for d in range(3):
            L.append(("res%d.%d k3 d=%d T=%d" % (lvl, d, 3 ** d, T), 2e-9 * B * T * 512 * 3 * 512))
            L.append(("res%d.%d 1x1 T=%d" % (lvl, d, T), 2e-9 * B * T * 512 * 512))
        cin = 512
    L.append(("out k3 T=%d" % T, 2e-9 * B * T * 512 * 3 * 512))
    L.append(("quantise x.kT", 2e-9 * B * T * 512 * 512))
    L.append(("argmin", 0.0))
    return L


def run(B, iters):
    import torch
    from qpgesture_amd import synth
    from qpgesture_amd.vqvae import VQVAE
    dev = torch.device("cuda", 0)
    model = VQVAE(None, 135, device=dev).load_state_dict(synth.make_vqvae_state_dict(7))
    x = torch.randn((B, 240, 135), device=dev)
    for _ in range(3):
        model.encode(x)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for e0, e1 in ev:
        e0.record()
        model.encode(x)
        e1.record()
    torch.cuda.synchronize()
    ms = sorted(e0.elapsed_time(e1) for e0, e1 in ev)
    print("encode B=%d: min %.3f  median %.3f  max %.3f ms  (%.1f TFLOP/s at the median)"
          % (B, ms[0], ms[len(ms) // 2], ms[-1], 1.639 * B / ms[len(ms) // 2]))


def table(d, B, iters):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                             int(r.get("Grid_Size", 0) or 0) // max(int(r.get("Workgroup_Size", 1) or 1), 1)))
    rows.sort()
    names = ("conv1d_mfma", "vq_argmin", "conv_splitk", "resblock", "sub_inplace", "convt_f32", "convt_small", "pad_channels",
             "vq_gather")
    rows = [r for r in rows if any(n in r[2] for n in names)]
    per = len(rows) // (iters + 3)
    rows = rows[-per * iters:]
    # flops by launch shape: fused block = 64 rows x 512 x 2048 x 2 per workgroup; convt = rows x 128 x K x 2
    print("launches per encode: %d" % per)
    tot = 0.0
    print("| # | kernel | blocks | avg us | gap-before us |")
    print("|---|---|---|---|---|")
    for i in range(per):
        durs = [(rows[k * per + i][1] - rows[k * per + i][0]) / 1e3 for k in range(iters)]
        gaps = [(rows[k * per + i][0] - rows[k * per + i - 1][1]) / 1e3 for k in range(iters)] if i else [0.0]
        du = sum(durs) / len(durs)
        tot += du + max(sum(gaps) / len(gaps), 0.0)
        print("| %d | %s | %d | %.1f | %.1f |" % (i, rows[i][2][:48], rows[i][3], du, sum(gaps) / len(gaps)))
    print("sum (kernels + gaps): %.3f ms  ->  %.1f TFLOP/s" % (tot / 1e3, 1.639 * B / (tot / 1e3)))


def run_decode(L, iters):
    import torch
    from qpgesture_amd import synth
    from qpgesture_amd.vqvae import VQVAE
    dev = torch.device("cuda", 0)
    model = VQVAE(None, 135, device=dev).load_state_dict(synth.make_vqvae_state_dict(7))
    ids = torch.randint(0, 512, (1, L), device=dev)
    for _ in range(3 + iters):
        model.decode([ids])
    torch.cuda.synchronize()


if __name__ == "__main__":
    if sys.argv[1] == "decode":
        run_decode(int(sys.argv[2]), int(sys.argv[3]))
    elif sys.argv[1] == "run":
        run(int(sys.argv[2]), int(sys.argv[3]))
    else:
        table(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
