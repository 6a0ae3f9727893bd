"""f32-MFMA approximate audio sweep vs the f64 sweep: time and max |difference|."""
import ctypes, subprocess, sys, os
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import numpy as np, torch
from qpgesture_amd import _lib
MT = os.environ.get("F32MM_MT", "2")
SO = os.path.join(HERE, "libqpg_f32mm_exp_mt%s.so" % MT)
if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(os.path.join(HERE, f)) for f in ("qpg_audio_f32mm.hip", "qpg_audio_f32lds.hip")):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                           "-ffp-contract=off", "-DF32MM_MT=" + MT, "-I", os.path.join(ROOT, "qpgesture_amd", "csrc"),
                           os.path.join(HERE, "qpg_audio_f32mm.hip"), os.path.join(HERE, "qpg_audio_f32lds.hip"),
                           os.path.join(ROOT, "qpgesture_amd", "csrc", "qpg_core.hip"), "-o", SO])
exp = ctypes.CDLL(SO)
P_, I_, L_ = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
exp.qpg_ctx_create.argtypes = [I_, ctypes.POINTER(P_)]
exp.qpg_audio_cosine_approx_f32.argtypes = [P_, P_, P_, I_, I_, I_, P_, I_, I_, I_, P_, P_, P_, I_, P_, L_]
exp.qpg_audio_cosine_approx_lds.argtypes = [P_, P_, P_, I_, I_, I_, P_, I_, I_, I_, P_, P_, P_, I_, P_, L_]
_ectx = P_(); assert exp.qpg_ctx_create(0, ctypes.byref(_ectx)) == 0
def ecall(name, dev, *args):
    st = P_(torch.cuda.current_stream(dev).cuda_stream)
    conv = [P_(a.data_ptr()) if isinstance(a, torch.Tensor) else a for a in args]
    assert getattr(exp, name)(_ectx, st, *conv) == 0, name
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 48
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
torch.manual_seed(0)
dev = torch.device("cuda:0")
T, F, G = 180, 1024, 26
base = torch.randn((N, T, F), device=dev)
q32 = torch.randn((Q, 6 * F), device=dev)
qn2 = (q32.double() ** 2).sum(1)
cand_t = torch.arange(G, device=dev, dtype=torch.int32) * 6
fn2 = torch.empty((N, T), dtype=torch.float64, device=dev)
cn2 = torch.empty((N, G), dtype=torch.float64, device=dev)
_lib.call("qpg_frame_norm2_f64", dev, base, N * T, F, fn2)
_lib.call("qpg_audio_cand_norm2", dev, fn2, N, T, cand_t, G, 6, 2, cn2)
D = torch.empty((Q, N * G), device=dev, dtype=torch.float64)
D32 = torch.empty((Q, N * G), device=dev, dtype=torch.float32)
def run64():
    _lib.call("qpg_audio_cosine_f64", dev, base, N, T, F, cand_t, G, 6, 2, cn2, q32, qn2, Q, D, D.stride(0))
def run32():
    ecall("qpg_audio_cosine_approx_f32", dev, base, N, T, F, cand_t, G, 6, 2, cn2, q32, qn2, Q, D32, D32.stride(0))
def t(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
D32b = torch.empty((Q, N * G), device=dev, dtype=torch.float32)
def runlds():
    ecall("qpg_audio_cosine_approx_lds", dev, base, N, T, F, cand_t, G, 6, 2, cn2, q32, qn2, Q, D32b, D32b.stride(0))
tl = t(runlds)
print("LDS-shared-query variant: %.1f us (%.1f TF)  max|d - d64| = %.3e" % (tl * 1e3, 2.0 * Q * N * G * 6 * F / tl / 1e9, 0.0))
t64, t32 = t(run64), t(run32)
print("LDS variant max|d32 - d64| = %.3e" % (D32b.double() - D).abs().max().item())
err = (D32.double() - D).abs().max().item()
print("N=%d Q=%d  f64 %.1f us  f32mm %.1f us (%.1f TF)  max|d32-d64| = %.3e  (f32 storage ulp ~6e-8)" % (
    N, Q, t64 * 1e3, t32 * 1e3, 2.0 * Q * N * G * 6 * F / t32 / 1e9, err))
