"""Decode of one 24 s clip (180 codes -> 1440 frames) on the split-f16 convolutions beside the f32 kernels: HIP events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qpgesture_amd import synth
from qpgesture_amd.vqvae import VQVAE
dev = torch.device("cuda:0")
m = VQVAE(None, 135, device=dev).load_state_dict(synth.make_vqvae_state_dict(7))
def t(fn, n=50):
    for _ in range(8): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for e0, e1 in ev:
        e0.record(); fn(); e1.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    return ms[0], ms[len(ms) // 2]
for B, L in ((1, 180), (1, 30), (1, 720), (16, 180), (512, 30)):
    ids = torch.randint(0, 512, (B, L), device=dev)
    a = m.decode([ids]); b, st = m.decode_f16x3([ids], return_stats=True, force=True)
    d = float((a - b).abs().max()); s = float(a.abs().max())
    print("B=%d L=%d: f32 min %.3f med %.3f ms | f16x3 min %.3f med %.3f ms | max |diff| %.3g (max |pose| %.3g) %s"
          % ((B, L) + t(lambda: m.decode([ids])) + t(lambda: m.decode_f16x3([ids], force=True)) + (d, s, st)))
# the same launches as ONE hipGraph replay (static ids / output buffers): what the kernels themselves take
for B, L in ((1, 180), (1, 30)):
    ids = torch.randint(0, 512, (B, L), device=dev)
    ref = m.decode([ids])
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        zq = torch.empty((B * L, m.emb), dtype=torch.float32, device=dev)
        def body():
            from qpgesture_amd import _lib
            _lib.call("qpg_vq_gather_f32", dev, m.k, ids, B * L, m.emb, m.bins, zq, m._dec_status)
            m._force16 = True
            try:
                return m.decode_latent(zq.view(B, L, m.emb), B, L)
            finally:
                m._force16 = False
        body(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            out = body()
    torch.cuda.current_stream().wait_stream(s)
    g.replay(); torch.cuda.synchronize()
    print("B=%d L=%d graph replay of the f16x3 decode: min %.3f med %.3f ms; max |diff| vs f32 %.3g; status %d"
          % ((B, L) + t(lambda: g.replay()) + (float((out - ref).abs().max()), int(m._c16_status.item()))))
