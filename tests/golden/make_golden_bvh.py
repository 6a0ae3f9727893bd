#!/usr/bin/env python
"""Golden vectors for SURVEY.md §8 f-4 (pose -> ZXY Euler channels) by running the REFERENCE's own function
process/process_bvh.py::make_bvh_GENEA2020_BT (rot-matrix -> `R.from_matrix` -> `as_euler('ZXY', degrees=True)`, with
and without its Savitzky-Golay option) in the build container.

The function also calls the third-party `pymo` inverse pipeline and BVH writer, which are absent here and out of scope:
`pymo.*` are stubbed as empty modules, `joblib.load` returns an object whose `inverse_transform` CAPTURES the Euler
array the reference computed (exactly the array it would hand to pymo), and `BVHWriter.write` is a no-op.  Inputs are
seeded: decoder-like poses = noisy rotation matrices (the VQ-VAE output is only approximately orthogonal), de-normalised
with a seeded mean / std as VisualizeCodebook.py:148-149 does.  Only inputs' seeds and OUTPUTS are committed.

Usage: python tests/golden/make_golden_bvh.py
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)


def make_inputs(seed=50, T=96):
    """(normalised poses f32 [T,135], mean f64 [135], std f64 [135]): de-normalised they are rotation matrices of random
    ZXY angles plus 2 % noise (not orthogonal), one joint near gimbal lock (X up to 89.9 deg)."""
    from scipy.spatial.transform import Rotation as R
    rng = np.random.Generator(np.random.PCG64(seed))
    # smooth in time (a gesture, and the Savitzky-Golay option must not destroy the matrices): a few slow sinusoids
    tt = np.arange(T)[:, None, None] / 60.0
    amp = rng.uniform(0.2, 1.0, size=(1, 15, 3)) * np.array([120.0, 60.0, 120.0])
    ang = amp * np.sin(2 * np.pi * rng.uniform(0.2, 1.5, size=(1, 15, 3)) * tt + rng.uniform(0, 6.28, size=(1, 15, 3)))
    ang[:, 3, 1] = 89.9 * np.cos(0.02 * np.arange(T))                # a joint that stays near gimbal lock
    mats = R.from_euler("ZXY", ang.reshape(-1, 3), degrees=True).as_matrix().reshape(T, 15, 9)
    mats = mats + rng.normal(0, 0.02, size=mats.shape)
    mean = rng.normal(0, 0.3, size=135)
    std = rng.uniform(0.005, 0.6, size=135)                     # some below the 0.01 clip of VisualizeCodebook.py:136
    stdc = np.clip(std, a_min=0.01, a_max=None)
    poses_n = ((mats.reshape(T, 135) - mean) / stdc).astype(np.float32)
    return poses_n, mean, std


def run_reference(out_poses, smoothing):
    cap = {}
    for name in ("pymo", "pymo.parsers", "pymo.preprocessing", "pymo.viz_tools", "pymo.writers"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["pymo.parsers"].BVHParser = object

    class _Writer:
        def write(self, data, f):
            pass
    sys.modules["pymo.writers"].BVHWriter = _Writer
    sys.path.insert(0, "/root/reference")
    import importlib
    pb = importlib.import_module("process.process_bvh")

    class _Pipe:
        def inverse_transform(self, xs):
            cap["euler"] = np.array(xs[0])
            return [None]
    pb.jl.load = lambda path: _Pipe()
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        pb.make_bvh_GENEA2020_BT(td, "g", out_poses, smoothing=smoothing, pipeline_path="unused")
    return cap["euler"], list(pb.target_joints)


def main():
    poses_n, mean, std = make_inputs()
    stdc = np.clip(std, a_min=0.01, a_max=None)
    out_poses = np.multiply(poses_n, stdc) + mean                 # VisualizeCodebook.py:148-149 (f32 * f64 -> f64)
    e0, joints = run_reference(out_poses, False)
    e1, _ = run_reference(out_poses, True)
    np.savez_compressed(os.path.join(HERE, "bvh_euler_s50.npz"), euler=e0, euler_smooth=e1, denorm=out_poses,
                        joints=np.array(joints), meta=np.array([50, 96], np.int64))
    print("euler", e0.shape, e0.dtype, "range", e0.min(), e0.max(), "joints", joints)


if __name__ == "__main__":
    main()
