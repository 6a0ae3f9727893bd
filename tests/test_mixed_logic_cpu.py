"""The DECISION LOGIC of the mixed-precision select, restated in NumPy and property-tested on the CPU (no kernels):
given any table whose entries are within E of the exact distances, re-evaluating (a) every candidate within the band of
its code's approximate minimum when there are two or more, and (b) the winner of every code whose approximate minimum
lies within the band of its rank neighbour, and then ranking refined and untouched values together, yields exactly the
winners and ranks of the exact table — provided the band is >= 2E.  (csrc/qpg_select.hip implements this on the device;
tests/test_gpu_mixed.py checks the kernels; this file checks the argument itself, including adversarial errors.)"""
import numpy as np
import pytest


def exact_tables(D, code, K):
    """per-code minimum (first index among equals) and stable ranks of the exact matrix row D."""
    best = np.full(K, np.inf)
    idx = np.full(K, -1)
    for c in np.argsort(D, kind="stable"):            # ascending value, ties by index
        k = code[c]
        if idx[k] < 0:
            best[k], idx[k] = D[c], c
    v = np.where(idx >= 0, best, 1e3)
    rank = np.argsort(np.argsort(v, kind="stable"), kind="stable")
    return idx, rank


def mixed_tables(Dapprox, Dexact, code, K, band):
    """The tiered procedure: only Dapprox is read, except for the pairs the rules select, which read Dexact."""
    n_refined = 0
    m = np.full(K, np.inf)
    for c, k in enumerate(code):
        m[k] = min(m[k], Dapprox[c])
    present = np.isfinite(m)
    idx = np.full(K, -1)
    v = np.full(K, 1e3)
    touched = np.zeros(K, bool)
    members = {k: [c for c in np.flatnonzero(code == k) if Dapprox[c] <= m[k] + band] for k in np.flatnonzero(present)}
    for k, mem in members.items():
        if len(mem) >= 2:                              # (a) candidate level
            vals = [(Dexact[c], c) for c in mem]
            n_refined += len(mem)
            v[k], idx[k] = min(vals)
            touched[k] = True
        else:
            idx[k], v[k] = mem[0], Dapprox[mem[0]]
    order = sorted(np.flatnonzero(present), key=lambda k: (m[k], k))      # neighbours by APPROXIMATE minima
    for a, b in zip(order[:-1], order[1:]):
        if m[b] - m[a] < band:                         # (b) rank level
            for k in (a, b):
                if not touched[k]:
                    v[k] = Dexact[idx[k]]
                    touched[k] = True
                    n_refined += 1
    rank = np.argsort(np.argsort(v, kind="stable"), kind="stable")
    return idx, rank, n_refined


@pytest.mark.parametrize("seed", range(12))
def test_band_of_twice_the_bound_is_sufficient(seed):
    rs = np.random.RandomState(seed)
    K, C = 40, 600
    E = 1e-3
    code = rs.randint(0, K + 3, size=C) % K if seed % 3 else rs.randint(0, K - 5, size=C)     # some codes absent
    # crowded exact distances: clusters much tighter than E, exact duplicates, and isolated values
    base = rs.choice([0.2, 0.5, 0.5004, 0.5011, 0.9], size=C)
    D = base + rs.choice([0.0, 1e-9, 1e-6, 3e-4, 8e-4, 2e-3], size=C) * rs.standard_normal(C)
    D[rs.randint(0, C, 20)] = D[rs.randint(0, C, 20)]
    # errors: random inside the bound, or adversarial (push every value towards its neighbours: +-E at the extremes)
    err = rs.uniform(-E, E, size=C) if seed % 2 else E * np.sign(rs.standard_normal(C))
    Da = D + err
    want_idx, want_rank = exact_tables(D, code, K)
    got_idx, got_rank, n = mixed_tables(Da, D, code, K, band=2.0 * E * 1.0000001)
    assert np.array_equal(got_idx, want_idx)
    assert np.array_equal(got_rank, want_rank)
    assert 0 < n < C


def test_a_band_below_twice_the_bound_can_fail():
    """The factor 2 is tight: two codes whose exact minima are 1.5 E apart can be swapped by errors of +-E, and a band of
    E does not catch it."""
    E = 1e-3
    D = np.array([0.5, 0.5 + 1.5 * E])
    code = np.array([0, 1])
    Da = D + np.array([+E, -E])                         # approximate order is reversed, 0.5 E apart
    want_idx, want_rank = exact_tables(D, code, 2)
    ok_idx, ok_rank, _ = mixed_tables(Da, D, code, 2, band=2.0 * E * 1.0000001)
    assert np.array_equal(ok_rank, want_rank)
    bad_idx, bad_rank, n = mixed_tables(Da, D, code, 2, band=0.4 * E)
    assert n == 0 and not np.array_equal(bad_rank, want_rank)


def sharded_tables(Dapprox, Dexact, code, K, band, shard_of):
    """Row-sharded form: every shard reports, per code, its winner and that winner's APPROXIMATE value —
    near-ties inside the shard already settled with exact values by the shard's own select — and the owner (a) asks
    every shard within the band of the merged minimum for the exact value of its winner when there are two or more,
    (b) asks the winning shard when the merged minimum is within the band of a rank neighbour."""
    W = int(shard_of.max()) + 1
    val = np.full((W, K), np.inf)
    win = np.full((W, K), -1)
    for w in range(W):
        loc = np.flatnonzero(shard_of == w)
        idx, _, _ = mixed_tables(Dapprox[loc], Dexact[loc], code[loc], K, band)      # the shard's own select
        for k in range(K):
            if idx[k] >= 0:
                win[w, k] = loc[idx[k]]
                val[w, k] = Dapprox[win[w, k]]
    m = val.min(axis=0)
    present = np.isfinite(m)
    idx = np.full(K, -1)
    v = np.full(K, 1e3)
    touched = np.zeros(K, bool)
    for k in np.flatnonzero(present):
        cont = [w for w in range(W) if win[w, k] >= 0 and val[w, k] <= m[k] + band]
        if len(cont) >= 2:
            v[k], idx[k] = min((Dexact[win[w, k]], win[w, k]) for w in cont)
            touched[k] = True
        else:
            idx[k], v[k] = win[cont[0], k], m[k]
    order = sorted(np.flatnonzero(present), key=lambda k: (m[k], k))
    for a, b in zip(order[:-1], order[1:]):
        if m[b] - m[a] < band:
            for k in (a, b):
                if not touched[k]:
                    v[k] = Dexact[idx[k]]
                    touched[k] = True
    return idx, np.argsort(np.argsort(v, kind="stable"), kind="stable")


@pytest.mark.parametrize("seed", range(8))
def test_sharded_rule_equals_the_exact_tables(seed):
    rs = np.random.RandomState(100 + seed)
    K, C, W = 30, 480, 2 + seed % 4
    E = 1e-3
    code = rs.randint(0, K, size=C)
    base = rs.choice([0.2, 0.5, 0.5004, 0.5011, 0.9], size=C)
    D = base + rs.choice([0.0, 1e-9, 1e-6, 3e-4, 8e-4, 2e-3], size=C) * rs.standard_normal(C)
    D[rs.randint(0, C, 20)] = D[rs.randint(0, C, 20)]                    # exact duplicates, also across shards
    err = rs.uniform(-E, E, size=C) if seed % 2 else E * np.sign(rs.standard_normal(C))
    shard_of = np.sort(rs.randint(0, W, size=C))                          # contiguous row blocks
    shard_of[0], shard_of[-1] = 0, W - 1
    want_idx, want_rank = exact_tables(D, code, K)
    got_idx, got_rank = sharded_tables(D + err, D, code, K, 2.0 * E * 1.0000001, shard_of)
    assert np.array_equal(got_idx, want_idx)
    assert np.array_equal(got_rank, want_rank)
