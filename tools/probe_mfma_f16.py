"""Hardware probe: how does v_mfma_f32_16x16x32_f16 sum its 32 products?  (error model behind qpg_audio_hl.hip's bound)
Families: one big product 2^24 plus n small products of magnitude c * 2^s at k positions; sign patterns; C != 0."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from qpgesture_amd import _lib
dev = torch.device("cuda:0")

def run(a, b, c=None):
    a16 = torch.from_numpy(a.astype(np.float16)).to(dev); b16 = torch.from_numpy(b.astype(np.float16)).to(dev)
    cd = None if c is None else torch.from_numpy(c.astype(np.float32)).to(dev)
    out = torch.empty((a.shape[0], 16, 16), dtype=torch.float32, device=dev)
    _lib.call("qpg_probe_mfma_f16_tile", dev, a16, b16, cd, a.shape[0], out)
    return out.cpu().numpy().astype(np.float64)

def one(avals, bvals, c=0.0):
    """row 0 of A x row 0 of B (col 0): a single dot product of 32 terms."""
    a = np.zeros((1, 16, 32)); b = np.zeros((1, 16, 32))
    a[0, 0, :len(avals)] = avals; b[0, 0, :len(bvals)] = bvals
    cc = None
    if c != 0.0:
        cc = np.zeros((1, 16, 16)); cc[0, 0, 0] = c
    return run(a, b, cc)[0, 0, 0]

print("== big 2^24 at position P + 31 small terms of value v (exact products): result - 2^24")
for pos in (0, 7, 8, 15, 16, 31):
    for v in (0.25, 0.5, 0.75, 1.0, 1.5, 2.0, 3.0):
        av = np.full(32, 1.0); bv = np.full(32, v)
        av[pos] = 4096.0; bv[pos] = 4096.0
        r = one(av, bv)
        print("  pos %2d small %.2f: got %+.2f  exact %+.2f" % (pos, v, r - 2.0 ** 24, 31 * v), end=" |")
    print()
print("== big 2^24 + ONE small term v at distance d from it")
for d in (1, 4, 8, 16, 31):
    for v in (0.25, 0.5, 0.75, 1.0, 1.25, 1.5, 2.5, 3.0):
        av = np.zeros(32); bv = np.zeros(32)
        av[0] = 4096.0; bv[0] = 4096.0; av[d] = 1.0; bv[d] = v
        print("  d=%2d v=%.2f -> %+.2f" % (d, v, one(av, bv) - 2.0 ** 24), end=" |")
    print()
print("== big 2^24 - small: negative small terms")
for v in (0.25, 0.5, 0.75, 1.0, 1.5):
    av = np.full(32, 1.0); bv = np.full(32, -v); av[0] = 4096.0; bv[0] = 4096.0
    print("  31 x -%.2f -> %+.2f (exact %+.2f)" % (v, one(av, bv) - 2.0 ** 24, -31 * v), end=" |")
print()
print("== C = 2^24, products small: 32 x v")
for v in (0.25, 0.5, 0.75, 1.0, 1.5):
    print("  32 x %.2f + C -> %+.2f (exact %+.2f)" % (v, one(np.full(32, 1.0), np.full(32, v), c=2.0 ** 24) - 2.0 ** 24, 32 * v), end=" |")
print()
print("== staircase: terms 2^24, 2^12, 2^0, 2^-12 repeated (8 each), exact sum vs got")
av = np.array([4096.0, 64.0, 1.0, 2.0 ** -6] * 8); bv = av.copy()
ex = float(np.sum(av * bv)); print("  got - exact = %+.4f  (exact %.4f)" % (one(av, bv) - ex, ex))
rng = np.random.default_rng(1)
worst = 0
for trial in range(200):
    tiles = 512
    e1 = rng.integers(-12, 13, size=(tiles, 16, 32)); e2 = rng.integers(-12, 13, size=(tiles, 16, 32))
    a = (1 + rng.random((tiles, 16, 32))) * 2.0 ** e1 * rng.choice([-1, 1], size=(tiles, 16, 32))
    b = (1 + rng.random((tiles, 16, 32))) * 2.0 ** e2
    a16 = a.astype(np.float16).astype(np.float64); b16 = b.astype(np.float16).astype(np.float64)
    got = run(a16, b16)
    ex = np.einsum("tik,tjk->tij", a16, b16); mag = np.einsum("tik,tjk->tij", np.abs(a16), np.abs(b16))
    worst = max(worst, float((np.abs(got - ex) / (2.0 ** -24 * mag)).max()))
print("random wide-range (exponents +-12 each side, mixed signs): worst kappa %.3f" % worst)
worst = 0
for trial in range(100):
    tiles = 512
    a = np.abs(rng.standard_normal((tiles, 16, 32))); b = np.abs(rng.standard_normal((tiles, 16, 32)))
    big = rng.integers(0, 32, size=(tiles, 16))
    for t in range(0):
        pass
    a16 = a.astype(np.float16).astype(np.float64); b16 = b.astype(np.float16).astype(np.float64)
    # one dominant product per row pair: scale column `k0` of both operands
    k0 = rng.integers(0, 32)
    a16[:, :, k0] *= 2.0 ** rng.integers(8, 12); b16[:, :, k0] *= 2.0 ** rng.integers(8, 12)
    a16 = a16.astype(np.float16).astype(np.float64); b16 = b16.astype(np.float16).astype(np.float64)
    got = run(a16, b16)
    ex = np.einsum("tik,tjk->tij", a16, b16); mag = np.einsum("tik,tjk->tij", np.abs(a16), np.abs(b16))
    worst = max(worst, float((np.abs(got - ex) / (2.0 ** -24 * mag)).max()))
print("one dominant product + 31 positive small ones: worst kappa %.3f" % worst)
