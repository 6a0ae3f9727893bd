"""Which convolution kernel for which number of rows: the short-sequence kernel (launcher's own block shape) against the
64 x 128 kernel (qpg_debug_convt_shape(8, 0)) on 512 -> 512 layers, M = B x T rows, L2-cold weights (8 distinct layers
cycled).  python tools/bench_convt_m.py"""
# needs a -DQPG_DEBUG_HOOKS variant of the library (the product exports no qpg_debug_* setters since round 6):
#   tools/build_variant.sh qpg_convt hooks "-DQPG_DEBUG_HOOKS" && QPG_LIB_PATH=experiments/variants/libqpg_hooks.so python tools/bench_convt_m.py
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qpgesture_amd import _lib
from qpgesture_amd.vqvae import tpack
dev = torch.device("cuda:0")
lib = _lib.load()
NL = 8
def run(M, taps, force):
    assert lib.qpg_debug_convt_shape(force, 0) == 0
    cin = cout = 512
    x = torch.randn((1, M, cin), device=dev)
    wts = [tpack(torch.randn((taps, cin, cout), device=dev) * 0.02, cin, 128) for _ in range(NL)]
    b = torch.randn((cout,), device=dev)
    ys = [torch.empty((1, M, cout), device=dev) for _ in range(2)]
    def call(i):
        _lib.call("qpg_convt_f32", dev, x if i % 2 == 0 else ys[0], 1, M, cin, wts[i % NL], b, taps, cin, cout, cout, 1, -(taps // 2), 1, M, 1, 0, M,
                  None, int(taps == 3), 0, ys[i % 2] if i % 2 == 0 else ys[1])
    for i in range(NL): call(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(40): call(i)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 200 * 1e3
for M in (180, 360, 720, 1440, 1920, 2880, 3840, 5760, 7680, 11520):
    for taps in (3, 1):
        a, r = run(M, taps, 0), run(M, taps, 8)
        gf = 2.0 * M * taps * 512 * 512 / 1e9
        print("M=%5d k%d: launcher's choice %.1f us (%.0f TF/s) | 64x128 kernel %.1f us (%.0f TF/s)" % (M, taps, a, gf / a * 1e3 / 1e3, r, gf / r * 1e3 / 1e3), flush=True)
lib.qpg_debug_convt_shape(0, 0)
