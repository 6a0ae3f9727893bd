"""Numerics + timing of the int8 digit-plane audio sweep against the f64 MFMA sweep."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qpgesture_amd import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 48
dev = torch.device("cuda:0")
T, F, G = 180, 1024, 26
torch.manual_seed(0)
base = torch.randn((N, T, F), device=dev)
q32 = torch.randn((Q, 6 * F), device=dev)
cand_t = torch.arange(G, device=dev, dtype=torch.int32) * 6
fn2 = torch.empty((N, T), dtype=torch.float64, device=dev)
cn2 = torch.empty((N, G), dtype=torch.float64, device=dev)
_lib.call("qpg_frame_norm2_f64", dev, base, N * T, F, fn2)
_lib.call("qpg_audio_cand_norm2", dev, fn2, N, T, cand_t, G, 6, 2, cn2)
qn2 = (q32.double() ** 2).sum(1).contiguous()
D64 = torch.empty((Q, N * G), device=dev, dtype=torch.float64)
_lib.call("qpg_audio_cosine_f64", dev, base, N, T, F, cand_t, G, 6, 2, cn2, q32, qn2, Q, D64, D64.stride(0))
A = torch.empty((4, N * T, F), dtype=torch.int8, device=dev)
sA = torch.empty((N * T,), dtype=torch.float64, device=dev)
_lib.call("qpg_i8_slice_rows", dev, base, N * T, F, A, sA, None)
Bq = torch.empty((4, Q * 6, F), dtype=torch.int8, device=dev)
sQ = torch.empty((Q * 6,), dtype=torch.float64, device=dev)
_lib.call("qpg_i8_slice_rows", dev, q32, Q * 6, F, Bq, sQ, None)
D8 = torch.empty_like(D64)
run = lambda: _lib.call("qpg_audio_cosine_i8", dev, A, sA, N, T, F, cand_t, G, 6, 2, cn2, Bq, sQ, qn2, Q, D8, D8.stride(0))
run(); torch.cuda.synchronize()
err = (D8 - D64).abs()
print("N=%d Q=%d  max |D8-D64| = %.3e  mean %.3e   D range [%.4f, %.4f]" % (N, Q, err.max().item(), err.mean().item(), D64.min().item(), D64.max().item()))
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print("int8 sweep %.1f us  (%.1f int8 TOPS over 13 digit pairs; f64-equivalent %.1f TFLOP/s)" % (ms * 1e3, 13 * 2.0 * Q * N * G * 6 * F / ms / 1e9, 2.0 * Q * N * G * 6 * F / ms / 1e9))
