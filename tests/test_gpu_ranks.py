"""Stable ranks by sorting (qpg_rank_rows_f32 / _f64 -> block_sorted_ranks, csrc/qpg_common.h) against their definition,
rank[c] = #{c' : d[c'] < d[c] or (d[c'] == d[c] and c' < c)} (np.argsort(kind='stable').argsort(); the reference ranks its
(512,) distance rows with np.array(x).argsort().argsort(), GestureKNN.py:540,544,553,574): exact ties in bulk (absent
codes all carry 1000.0), -0.0 against +0.0, K that is not a power of two, K on every path of the sort (one wave, four
waves + LDS stages, the LDS fallback), and the rank row a select emits."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ranks(d):
    import torch
    from qpgesture_amd import _lib
    dev = torch.device("cuda:0")
    t = torch.from_numpy(np.ascontiguousarray(d)).to(dev)
    out = torch.empty(t.shape, dtype=torch.int16, device=dev)
    name = "qpg_rank_rows_f64" if d.dtype == np.float64 else "qpg_rank_rows_f32"
    _lib.call(name, dev, t, t.shape[0], t.shape[1], out)
    return out.cpu().numpy()


@pytest.mark.parametrize("K", [512, 500, 64, 96, 128, 200, 1024, 1500, 2048, 3000, 7])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_rank_rows_equal_stable_argsort_argsort(K, dtype):
    rng = np.random.default_rng(K)
    Q = 37
    d = rng.standard_normal((Q, K)).astype(dtype)
    d[1] = np.round(d[1] * 4) / 4                            # many exact ties
    d[2, :] = 1000.0                                         # every code absent: ranks = code order
    d[3, ::3] = 1000.0
    d[4, : K // 2] = 0.0
    d[4, 1: K // 2: 2] = -0.0                                # -0.0 == +0.0: the tie goes to the lower code
    d[5] = np.sort(d[5])[::-1]                               # descending
    want = np.argsort(np.argsort(d, axis=1, kind="stable"), axis=1, kind="stable").astype(np.int16)
    got = _ranks(d)
    assert np.array_equal(got, want)
    for row in got:                                          # a permutation of 0..K-1 (what the rank fusion's scan relies on)
        assert np.array_equal(np.sort(row), np.arange(K))


def test_text_select_ranks_are_those_of_its_table():
    """The fused rank output of the exact text select (its key tables double as the sort's scratch)."""
    import torch
    from qpgesture_amd import _lib
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(3)
    Q, C, K = 9, 5000, 512
    D = torch.from_numpy(rng.standard_normal((Q, C)).astype(np.float32)).to(dev)
    code = torch.from_numpy(rng.integers(0, 400, size=C).astype(np.int16)).to(dev)      # codes 400..511 absent
    dist = torch.empty((Q, K), dtype=torch.float32, device=dev)
    idx = torch.empty((Q, K), dtype=torch.int32, device=dev)
    rank = torch.empty((Q, K), dtype=torch.int16, device=dev)
    _lib.call("qpg_percode_select_f32", dev, D, C, Q, code, C, K, 1000.0, 0, dist, idx, rank, 0, 0)
    d = dist.cpu().numpy()
    want = np.argsort(np.argsort(d, axis=1, kind="stable"), axis=1, kind="stable").astype(np.int16)
    assert np.array_equal(rank.cpu().numpy(), want)
    assert (d[:, 400:] == 1000.0).all() and (idx.cpu().numpy()[:, 400:] == -1).all()
