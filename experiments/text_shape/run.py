"""experiments: the clip-size text sweep (48 queries x 53 248 candidates x 384) with other (queries per lane, waves per
block) shapes; the variant libraries are built by build.sh.  python experiments/text_shape/run.py"""
import ctypes, glob, os, sys
import torch
from ctypes import c_void_p, c_int, c_int64
dev = torch.device("cuda:0")
Q, Dm, C = 48, 384, 2048 * 26
xt = torch.randn((((C + 63) // 64) * 64 * Dm,), device=dev)
qn = torch.randn((Q, Dm), device=dev)
ref = None
for so in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libtext_*.so"))):
    lib = ctypes.CDLL(so)
    h = c_void_p()
    assert lib.qpg_ctx_create(0, ctypes.byref(h)) == 0
    D = torch.zeros((Q, C), device=dev)
    def call():
        st = c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        rc = lib.qpg_text_cosine_f32(h, st, c_void_p(xt.data_ptr()), c_int64(C), c_int(Dm), c_void_p(qn.data_ptr()), c_int(Q),
                                     c_void_p(D.data_ptr()), c_int64(D.stride(0)))
        assert rc == 0
    for _ in range(3): call()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20): call()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    if ref is None: ref = D.clone()
    print("%-22s %.1f us  identical=%s" % (os.path.basename(so), e0.elapsed_time(e1) / 100 * 1e3, bool(torch.equal(D, ref))), flush=True)
