"""Numerics + timing of the int8 digit-plane audio sweep against the product's f64 MFMA sweep.
Builds experiments/int8_sweep/qpg_audio_i8.hip into its own shared object (it is not part of libqpg_hip.so)."""
import ctypes, subprocess, sys, os
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import torch
from qpgesture_amd import _lib
SO = os.path.join(HERE, "libqpg_i8_exp.so")
if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(os.path.join(HERE, "qpg_audio_i8.hip")):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                           "-ffp-contract=off", "-I", os.path.join(ROOT, "qpgesture_amd", "csrc"),
                           os.path.join(HERE, "qpg_audio_i8.hip"), os.path.join(ROOT, "qpgesture_amd", "csrc", "qpg_core.hip"),
                           "-o", SO])
exp = ctypes.CDLL(SO)
P, I, L = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
exp.qpg_i8_slice_rows.argtypes = [P, P, P, L, I, P, P, P]
exp.qpg_audio_cosine_i8.argtypes = [P, P, P, P, I, I, I, P, I, I, I, P, P, P, P, I, P, L]
exp.qpg_ctx_create.argtypes = [I, ctypes.POINTER(P)]
_ectx = P(); assert exp.qpg_ctx_create(0, ctypes.byref(_ectx)) == 0
def ecall(name, dev, *args):
    st = P(torch.cuda.current_stream(dev).cuda_stream)
    conv = [P(a.data_ptr()) if isinstance(a, torch.Tensor) else a for a in args]
    rc = getattr(exp, name)(_ectx, st, *conv)
    assert rc == 0, name
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
ecall("qpg_i8_slice_rows", dev, base, N * T, F, A, sA, None)
Bq = torch.empty((4, Q * 6, F), dtype=torch.int8, device=dev)
sQ = torch.empty((Q * 6,), dtype=torch.float64, device=dev)
ecall("qpg_i8_slice_rows", dev, q32, Q * 6, F, Bq, sQ, None)
D8 = torch.empty_like(D64)
run = lambda: ecall("qpg_audio_cosine_i8", dev, A, sA, N, T, F, cand_t, G, 6, 2, cn2, Bq, sQ, qn2, Q, D8, D8.stride(0))
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
