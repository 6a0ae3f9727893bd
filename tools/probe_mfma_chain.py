"""kappa of a TWO-instruction chain of v_mfma_f32_16x16x32_f16 (the second starts from the first's result):
|MFMA(A2, B2, MFMA(A1, B1, 0)) - exact sum of the 64 products| in units of 2^-24 x sum |64 products|, over the families of
tests/test_gpu_audio_hl.py plus chains built against the accumulator (a dominant first block, a tiny second one and the
reverse).  Decides whether the sweep may add every SECOND block sum to its f64 accumulators."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from qpgesture_amd import _lib
dev = torch.device("cuda:0")


def probe(a, b, c=None):
    tiles = a.shape[0]
    ad = torch.from_numpy(a).to(dev).contiguous(); bd = torch.from_numpy(b).to(dev).contiguous()
    cd = None if c is None else torch.from_numpy(c).to(dev).contiguous()
    out = torch.empty((tiles, 16, 16), dtype=torch.float32, device=dev)
    _lib.call("qpg_probe_mfma_f16_tile", dev, ad, bd, cd, tiles, out)
    return out.cpu().numpy()


rng = np.random.default_rng(1)
tiles = 4096
worst = {}


def chain(name, a1, b1, a2, b2):
    a1, b1, a2, b2 = (x.astype(np.float16) for x in (a1, b1, a2, b2))
    r1 = probe(a1, b1)
    got = probe(a2, b2, r1).astype(np.float64)
    A1, B1, A2, B2 = (x.astype(np.float64) for x in (a1, b1, a2, b2))
    exact = np.einsum("tik,tjk->tij", A1, B1) + np.einsum("tik,tjk->tij", A2, B2)
    mag = np.einsum("tik,tjk->tij", np.abs(A1), np.abs(B1)) + np.einsum("tik,tjk->tij", np.abs(A2), np.abs(B2))
    k = np.abs(got - exact) / (2.0 ** -24 * mag)
    worst[name] = max(worst.get(name, 0.0), float(k.max()))


def rnd(scale=1.0):
    return rng.standard_normal((tiles, 16, 32)) * scale


for rep in range(4):
    chain("normal", rnd(), rnd(), rnd(), rnd())
    chain("scaled", rnd(2.0 ** 13), rnd(2.0 ** 13), rnd(2.0 ** 13), rnd(2.0 ** 13))
    w = lambda: rnd() * 2.0 ** rng.integers(-10, 11, size=(tiles, 16, 32))
    chain("wide", w(), w(), w(), w())
    for s in (2.0 ** -6, 2.0 ** -10, 2.0 ** -12):
        chain("big first, small second", np.abs(rnd()), np.abs(rnd()), np.abs(rnd(s)), np.abs(rnd()))
        chain("small first, big second", np.abs(rnd(s)), np.abs(rnd()), np.abs(rnd()), np.abs(rnd()))
    # dominant product in the first block, positive small ones in both
    a1, b1 = np.abs(rnd()), np.abs(rnd())
    k0 = int(rng.integers(0, 32))
    a1[:, :, k0] *= 2.0 ** int(rng.integers(8, 12)); b1[:, :, k0] *= 2.0 ** int(rng.integers(8, 12))
    chain("dominant in first", a1, b1, np.abs(rnd()), np.abs(rnd()))
    a2, b2 = np.abs(rnd()), np.abs(rnd())
    a2[:, :, k0] *= 2.0 ** int(rng.integers(8, 12)); b2[:, :, k0] *= 2.0 ** int(rng.integers(8, 12))
    chain("dominant in second", np.abs(rnd()), np.abs(rnd()), a2, b2)
    # cancellation: the first block's sum is ~0 against large products
    a = rnd(); b = rnd()
    a[:, :, 1::2] = -a[:, :, 0::2]; b[:, :, 1::2] = b[:, :, 0::2]
    chain("cancelling first", a, b, rnd(), rnd())
for k, v in worst.items():
    print("%-28s kappa_2 = %.3f" % (k, v))
print("max %.3f" % max(worst.values()))
