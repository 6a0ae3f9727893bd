"""Decode of one 24 s clip (180 codes -> 1440 frames, one pass) and a single-window encode: HIP-event timings."""
# needs a -DQPG_DEBUG_HOOKS variant of the library (the product exports no qpg_debug_* setters since round 6):
#   tools/build_variant.sh qpg_convt hooks "-DQPG_DEBUG_HOOKS" && QPG_LIB_PATH=experiments/variants/libqpg_hooks.so python tools/bench_decode.py
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qpgesture_amd import synth
from qpgesture_amd.vqvae import VQVAE

dev = torch.device("cuda:0")
m = VQVAE(None, 135, device=dev).load_state_dict(synth.make_vqvae_state_dict(7))
ids = torch.randint(0, 512, (1, 180), device=dev)
x1 = torch.randn((1, 240, 135), device=dev)


def t(fn, n=50):
    for _ in range(8):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for e0, e1 in ev:
        e0.record()
        fn()
        e1.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    return ms[0], ms[len(ms) // 2]


from qpgesture_amd import _lib
lib = _lib.load()
ref = m.decode([ids]).clone()
hooks = hasattr(lib, "qpg_debug_convt_opts")           # (a -DQPG_DEBUG_HOOKS variant library; the product has none)
for deep, xcd in ((0, 0), (1, 0), (0, 1), (1, 1), (0, 0), (1, 1)) if hooks else ():
    lib.qpg_debug_convt_opts(deep, xcd)
    out = m.decode([ids])
    print("deep_ring=%d xcd_map=%d: decode 24 s clip min %.3f median %.3f ms; max |out - first| = %.3g"
          % ((deep, xcd) + t(lambda: m.decode([ids])) + (float((out - ref).abs().max()),)))
if hooks:
    lib.qpg_debug_convt_opts(0, 0)
print("decode 24 s clip: min %.3f median %.3f ms" % t(lambda: m.decode([ids])))
print("encode 1 window:  min %.3f median %.3f ms" % t(lambda: m.encode(x1)))
for L in (30, 60, 720):
    idl = torch.randint(0, 512, (1, L), device=dev)
    print("decode L=%d: min %.3f median %.3f ms" % ((L,) + t(lambda: m.decode([idl]))))
