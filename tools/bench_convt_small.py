"""experiments: time the short-sequence convolution kernel (convt_small_f32_kernel) for every block shape
(qpg_debug_convt_shape(nq, pd)) on the layer shapes of a clip decode.  python tools/bench_convt_small.py"""
# needs a -DQPG_DEBUG_HOOKS variant of the library (the product exports no qpg_debug_* setters since round 6):
#   tools/build_variant.sh qpg_convt hooks "-DQPG_DEBUG_HOOKS" && QPG_LIB_PATH=experiments/variants/libqpg_hooks.so python tools/bench_convt_small.py
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qpgesture_amd import _lib
from qpgesture_amd.vqvae import tpack
dev = torch.device("cuda:0")
def run(T, taps, dil, relu_in, shape):
    nq_, pd_ = (int(v) for v in shape.split(","))
    assert _lib.load().qpg_debug_convt_shape(nq_, pd_) == 0
    cin = cout = 512
    x = torch.randn((1, T, cin), device=dev)
    w = torch.randn((taps, cin, cout), device=dev) * 0.02
    wt = tpack(w, cin, 128)
    b = torch.randn((cout,), device=dev)
    y = torch.empty((1, T, cout), device=dev)
    def call():
        _lib.call("qpg_convt_f32", dev, x, 1, T, cin, wt, b, taps, cin, cout, cout, 1, -dil * (taps // 2), dil, T, 1, 0, T, None,
                  int(relu_in), 0, y)
    for _ in range(3): call()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()                      # host launch cost (~10 us per ctypes call) out of the picture
    with torch.cuda.graph(g):
        for _ in range(40): call()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 200 * 1e3
for T in (30, 180, 360, 720, 1440):
    for taps in (3, 1):
        row = []
        for nq in (4, 2, 1):
            for pd in (0, 4):
                row.append(("%d,%d" % (nq, pd), run(T, taps, 1, taps == 3, "%d,%d" % (nq, pd))))
        best = min(row, key=lambda r: r[1])
        print("T=%4d k%d: " % (T, taps) + "  ".join("%s=%.1f" % r for r in row) + "   best %s" % best[0], flush=True)
