"""Load-time check of the one MEASURED constant the mixed-precision paths rest on.

The a-priori bound of the split-operand f16 sweep (QPG_AUDIO_HL_ERR = 1.3e-6, reused by the text prefilter as
sorted_rows.HL_GEMM_ERR) assumes that a chain of two v_mfma_f32_16x16x32_f16 instructions is within
kappa_2 * 2^-24 * sum|64 products| of the exact sum, with kappa_2 <= 13.  Nothing in the ISA documents how the matrix core
aligns and rounds its 32 products; the constant was probed (tools/probe_mfma_f16.py, tools/probe_mfma_chain.py: worst
9.72).  So the product re-measures it once per process and device, on blocks built against the accumulator (dominant
products, wide dynamic range, big-then-small chains), through the same instruction (qpg_probe_mfma_f16_tile), and a
device that does not honour the assumption is NOT given the bounded paths: CodeKNN then sweeps in f64 and the text
side runs the exact-order sweep (both still HIP kernels; nothing falls back to the CPU).  The exact sums the measured
values are compared with are 64-term f64 sums of exactly representable f16 x f16 products, computed on the host with
NumPy: ~10 ms, once.
"""
import os
import warnings

import numpy as np
import torch

from . import _lib

KAPPA2_ASSUMED = 13.0          # what QPG_AUDIO_HL_ERR budgets for a chain of two (csrc/qpg_audio_hl.hip)
KAPPA2_LIMIT = 12.0            # the check fails ABOVE this: one unit of slack against families the probe does not build
KAPPA16_ASSUMED = 40.0         # a chain of <= 16 full-size blocks in one accumulator (the prefilter GEMMs): sorted_rows.gemm32_err
KAPPA16_LIMIT = 36.0
KAPPA6_ASSUMED = 13.05         # chains of six (audio_cosine_hl2_kernel: the four cross-term blocks first, then the two h h' blocks)
KAPPA6_LIMIT = 12.0
_cache = {}


def _probe(dev, a16, b16, c=None):
    tiles = a16.shape[0]
    ad = torch.from_numpy(a16).to(dev).contiguous()
    bd = torch.from_numpy(b16).to(dev).contiguous()
    cd = None if c is None else torch.from_numpy(np.ascontiguousarray(c, np.float32)).to(dev)
    out = torch.empty((tiles, 16, 16), dtype=torch.float32, device=dev)
    _lib.call("qpg_probe_mfma_f16_tile", dev, ad, bd, cd, tiles, out)
    return out.cpu().numpy()


def measure_kappa(device, tiles=512, seed=20260929):
    """Worst |chain of two MFMAs - exact| / (2^-24 sum |products|) over the adversarial families; also the single
    instruction's kappa.  Returns {"kappa": .., "kappa2": .., "families": {...}}."""
    dev = torch.device(device)
    rng = np.random.default_rng(seed)
    fam = {}

    def rnd(scale=1.0):
        return rng.standard_normal((tiles, 16, 32)) * scale

    def chain(name, a1, b1, a2, b2):
        a1, b1, a2, b2 = (np.ascontiguousarray(x.astype(np.float16)) for x in (a1, b1, a2, b2))
        r1 = _probe(dev, a1, b1)
        got = _probe(dev, a2, b2, r1).astype(np.float64)
        A1, B1, A2, B2 = (x.astype(np.float64) for x in (a1, b1, a2, b2))
        e1, e2 = np.einsum("tik,tjk->tij", A1, B1), np.einsum("tik,tjk->tij", A2, B2)
        m1 = np.einsum("tik,tjk->tij", np.abs(A1), np.abs(B1))
        m2 = np.einsum("tik,tjk->tij", np.abs(A2), np.abs(B2))
        k1 = np.abs(r1.astype(np.float64) - e1) / np.maximum(2.0 ** -24 * m1, 1e-300)
        k2 = np.abs(got - (e1 + e2)) / np.maximum(2.0 ** -24 * (m1 + m2), 1e-300)
        fam[name] = (max(fam.get(name, (0, 0))[0], float(k1.max())), max(fam.get(name, (0, 0))[1], float(k2.max())))

    def dominant():
        a, b = np.abs(rnd()), np.abs(rnd())
        k0 = int(rng.integers(0, 32))
        a[:, :, k0] *= 2.0 ** int(rng.integers(8, 12))
        b[:, :, k0] *= 2.0 ** int(rng.integers(8, 12))
        return a, b

    chain("normal", rnd(), rnd(), rnd(), rnd())
    chain("scaled", rnd(2.0 ** 13), rnd(2.0 ** 13), rnd(2.0 ** 13), rnd(2.0 ** 13))
    wide = lambda: rnd() * 2.0 ** rng.integers(-10, 11, size=(tiles, 16, 32))
    chain("wide", wide(), wide(), wide(), wide())
    for s in (2.0 ** -6, 2.0 ** -12):
        chain("big, small", np.abs(rnd()), np.abs(rnd()), np.abs(rnd(s)), np.abs(rnd()))
        chain("small, big", np.abs(rnd(s)), np.abs(rnd()), np.abs(rnd()), np.abs(rnd()))
    for _ in range(3):
        a, b = dominant()
        chain("dominant in first", a, b, np.abs(rnd()), np.abs(rnd()))
        a, b = dominant()
        chain("dominant in second", np.abs(rnd()), np.abs(rnd()), a, b)
        a1, b1 = dominant()
        a2, b2 = dominant()
        chain("dominant in both", a1, b1, a2, b2)
    a, b = rnd(), rnd()
    a[:, :, 1::2] = -a[:, :, 0::2]
    b[:, :, 1::2] = b[:, :, 0::2]
    chain("cancelling first", a, b, rnd(), rnd())
    # round 4: the sweep's chains are SIX instructions through one accumulator - the four cross-term blocks (2^-11 of
    # the h h' ones) FIRST, the two h h' blocks last: kappa_6 in units of 2^-24 x sum |the two big blocks' products|
    def cross_first(n_cross):
        """chain of n_cross small blocks (2^-11 of the others) followed by two full-size blocks"""
        worst = 0.0
        BIG = (n_cross, n_cross + 1)
        for rep in range(4):
            blocks = []
            for i in range(n_cross + 2):
                sc = 1.0 if i in BIG else 2.0 ** -11
                if rep in (1, 3) and i == BIG[rep // 2]:
                    a_, b_ = dominant()
                else:
                    a_, b_ = np.abs(rnd(sc)) if rep < 2 else rnd(sc), np.abs(rnd()) if rep < 2 else rnd()
                blocks.append((np.ascontiguousarray(a_.astype(np.float16)), np.ascontiguousarray(b_.astype(np.float16))))
            run, exact, mag = None, 0.0, 0.0
            for i, (a16, b16) in enumerate(blocks):
                run = _probe(dev, a16, b16, run)
                A, B = a16.astype(np.float64), b16.astype(np.float64)
                exact = exact + np.einsum("tik,tjk->tij", A, B)
                if i in BIG:
                    mag = mag + np.einsum("tik,tjk->tij", np.abs(A), np.abs(B))
            worst = max(worst, float((np.abs(run.astype(np.float64) - exact) / (2.0 ** -24 * mag)).max()))
        return worst
    k6 = cross_first(4)

    # round 5: the prefilter GEMMs (hl_gemm32_kernel, hl_gemm64h_kernel) keep the h h' products of the WHOLE
    # contraction in one accumulator - a chain of d / 32 FULL-SIZE blocks (16 at d = 512): kappa_16 in units of 2^-24 x
    # sum |all products|, which sorted_rows.gemm32_err / gemm_h_err budget as KAPPA16_ASSUMED
    def full_chain(n):
        worst = 0.0
        for rep in range(3):
            run, exact, mag = None, 0.0, 0.0
            for i in range(n):
                if rep == 2 and i in (0, n - 1):
                    a_, b_ = dominant()
                else:
                    a_, b_ = (np.abs(rnd()), np.abs(rnd())) if rep == 0 else (rnd(), rnd())
                a16, b16 = np.ascontiguousarray(a_.astype(np.float16)), np.ascontiguousarray(b_.astype(np.float16))
                run = _probe(dev, a16, b16, run)
                A, B = a16.astype(np.float64), b16.astype(np.float64)
                exact = exact + np.einsum("tik,tjk->tij", A, B)
                mag = mag + np.einsum("tik,tjk->tij", np.abs(A), np.abs(B))
            worst = max(worst, float((np.abs(run.astype(np.float64) - exact) / (2.0 ** -24 * mag)).max()))
        return worst
    k16 = full_chain(16)
    # round 5: the one-plane (f16 track) sweep's chains are FOUR instructions - the two l' h blocks, then the two h h' blocks
    k4 = cross_first(2)
    # f16 SUBNORMAL operands (the audio images' l planes hold them): the products must come out exact
    sub = (rng.integers(1, 1024, size=(tiles, 16, 32)).astype(np.float64) * 2.0 ** -24).astype(np.float16)
    big = rng.integers(1, 2048, size=(tiles, 16, 32)).astype(np.float16)
    big[:, :, 1:] = 0                                       # one product per dot: exact in f32 whatever the order
    got = _probe(dev, np.ascontiguousarray(sub), np.ascontiguousarray(big)).astype(np.float64)
    want = np.einsum("tik,tjk->tij", sub.astype(np.float64), big.astype(np.float64))
    sub_ok = bool(np.array_equal(got, want))
    return {"kappa": max(v[0] for v in fam.values()), "kappa2": max(v[1] for v in fam.values()), "kappa6": k6,
            "kappa4": k4, "kappa16": k16,
            "subnormals_exact": sub_ok,
            "families": {k: [round(v[0], 3), round(v[1], 3)] for k, v in fam.items()}}


def mfma_bound_ok(device):
    """(ok, report) for `device`, measured once per process: ok = the matrix core honours the kappa_2 the bounded
    sweeps assume.  QPG_SKIP_SELFCHECK=1 skips the measurement (report["skipped"])."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    hit = _cache.get(idx)
    if hit is not None:
        return hit
    if os.environ.get("QPG_SKIP_SELFCHECK", "") == "1":
        rep = {"skipped": True, "kappa2_assumed": KAPPA2_ASSUMED}
        _cache[idx] = (True, rep)
        return _cache[idx]
    rep = measure_kappa(torch.device("cuda", idx))
    rep.update(kappa2_assumed=KAPPA2_ASSUMED, kappa2_limit=KAPPA2_LIMIT, skipped=False)
    rep.update(kappa6_assumed=KAPPA6_ASSUMED, kappa6_limit=KAPPA6_LIMIT, kappa16_assumed=KAPPA16_ASSUMED)
    ok = (rep["kappa2"] <= KAPPA2_LIMIT and rep["kappa"] <= KAPPA2_LIMIT and rep["kappa6"] <= KAPPA6_LIMIT and
          rep["kappa4"] <= KAPPA6_LIMIT and rep["kappa16"] <= KAPPA16_LIMIT and rep["subnormals_exact"])
    if not ok:
        warnings.warn("qpgesture_amd: this device's f16 matrix core measured kappa_2 = %.2f (limit %.1f), kappa_6 / kappa_4 = %.2f "
                      "(limit %.1f), f16 subnormals exact: %s - the a-priori bound of the split-f16 sweeps does not hold "
                      "here; audio sweeps run in f64 and the text side on the exact-order kernel"
                      % (rep["kappa2"], KAPPA2_LIMIT, max(rep["kappa6"], rep["kappa4"]), KAPPA6_LIMIT, rep["subnormals_exact"]),
                      RuntimeWarning)
    _cache[idx] = (ok, rep)
    return _cache[idx]
