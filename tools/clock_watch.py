"""What the power manager does under the sweep (VERDICT r5 weak #6 / DESIGN 4.1c): the two-plane sweep at N_db = 2048 in three
launch patterns - back to back, with a 768 MB elementwise kernel between launches, with a 0.3 ms host sleep between launches -
for ~2.5 s each, while a thread samples rocm-smi (sclk / mclk / socket power) every ~0.25 s.  Prints per pattern: the sweep's
median / min time by HIP events and the samples."""
import os
import subprocess
import sys
import threading
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qpgesture_amd import _lib

N, Q = 2048, 48
dev = torch.device("cuda:0")
T, F, G = 180, 1024, 26
lib = _lib.load()
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
qi = torch.empty((int(lib.qpg_audio_hl_query_bytes(Q, F)),), dtype=torch.uint8, device=dev)
_lib.call("qpg_audio_hl_pack_queries", dev, q32, Q, F, qi, qi.numel())
stats = torch.zeros((4,), dtype=torch.int32, device=dev)
D = torch.empty((Q, N * G), dtype=torch.float32, device=dev)
trash = torch.zeros((768 << 20) // 4, dtype=torch.float32, device=dev)


def sweep():
    _lib.call("qpg_audio_cosine_hl", dev, img, N, F, G, cn2, qi, qn2, Q, D, 1, D.stride(0), stats)


samples, stop = [], [False]


def watch():
    while not stop[0]:
        try:
            o = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--csv"], stdout=subprocess.PIPE,
                               stderr=subprocess.DEVNULL, timeout=5).stdout.decode()
            samples.append((time.perf_counter(), o.strip().splitlines()[-1][:200]))
        except Exception as e:          # noqa: BLE001
            samples.append((time.perf_counter(), "rocm-smi failed: %r" % e))
            return
        time.sleep(0.25)


for pattern in ("back_to_back", "elementwise_between", "sleep_between"):
    samples.clear()
    stop[0] = False
    th = threading.Thread(target=watch, daemon=True)
    th.start()
    ts, t_end = [], time.perf_counter() + 2.5
    while time.perf_counter() < t_end:
        ev = []
        for _ in range(20):
            if pattern == "elementwise_between":
                trash.add_(1.0)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            sweep()
            b.record()
            ev.append((a, b))
            if pattern == "sleep_between":
                torch.cuda.synchronize()
                time.sleep(0.0003)
        torch.cuda.synchronize()
        ts += [a.elapsed_time(b) * 1e3 for a, b in ev]
    stop[0] = True
    th.join()
    ts.sort()
    print("== %s: %d sweeps, min %.1f us, p25 %.1f, median %.1f, p75 %.1f, max %.1f" % (
        pattern, len(ts), ts[0], ts[len(ts) // 4], ts[len(ts) // 2], ts[3 * len(ts) // 4], ts[-1]))
    for _, s in samples[1:-1][:8]:
        print("   ", s)
