"""pose -> ZXY Euler channels on the device (qpg_pose_to_euler_f64) against what the REFERENCE's own
make_bvh_GENEA2020_BT computed (tests/golden/bvh_euler_s50.npz, captured by tests/golden/make_golden_bvh.py).
Bar: 1e-4 degrees."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from tests.helpers import load_golden

pytestmark = pytest.mark.gpu


def _wrap_diff(a, b):
    d = np.abs(a - b)
    return np.minimum(d, 360.0 - d)                        # +-180 is one angle


def test_pose_to_euler_vs_reference_golden(tmp_path):
    import make_golden_bvh as G
    from qpgesture_amd import bvh
    g = load_golden("bvh_euler_s50")
    poses_n, mean, std = G.make_inputs(int(g["meta"][0]), int(g["meta"][1]))
    e = bvh.poses_to_euler(poses_n, mean, std)
    assert e.shape == (96, 45) and e.dtype == np.float64
    assert _wrap_diff(e, g["euler"]).max() < 1e-4
    es = bvh.poses_to_euler(poses_n, mean, std, smoothing=True)
    assert _wrap_diff(es, g["euler_smooth"]).max() < 1e-4
    assert _wrap_diff(es, e).max() > 1e-3                   # the smoothing option does something
    # ranges of scipy's as_euler: first / third in [-180, 180], second in [-90, 90]
    assert np.abs(e[:, 1::3]).max() <= 90.0 and np.abs(e).max() <= 180.0
    # minimal BVH carrying the channels
    path = str(tmp_path / "g.bvh")
    order = bvh.write_bvh(path, e)
    txt = open(path).read().splitlines()
    assert txt[0] == "HIERARCHY" and "Frames: 96" in txt and sum(1 for l in txt if "CHANNELS 3" in l) == 15
    row = np.array(txt[-1].split(), float)
    cols = np.concatenate([np.arange(3 * i, 3 * i + 3) for i in order])
    assert np.abs(row - e[-1, cols]).max() < 1e-5


def test_pose_to_euler_errors():
    from qpgesture_amd import bvh
    rng = np.random.default_rng(0)
    eye = np.tile(np.eye(3).reshape(9), (20, 15)).astype(np.float32)
    e = bvh.poses_to_euler(eye, np.zeros(135), np.ones(135))
    assert np.abs(e).max() < 1e-12                          # identity -> zero angles
    with pytest.raises(ValueError):                         # scipy: window longer than the sequence
        bvh.poses_to_euler(eye[:10], np.zeros(135), np.ones(135), smoothing=True)
    bad = eye.copy()
    bad[3, 9:18] = np.diag([1.0, 1.0, -1.0]).reshape(9)     # a reflection: scipy raises on det <= 0
    with pytest.raises(ValueError):
        bvh.poses_to_euler(bad, np.zeros(135), np.ones(135))
    with pytest.raises(ValueError):
        bvh.poses_to_euler(rng.standard_normal((4, 10)), np.zeros(10), np.ones(10))
