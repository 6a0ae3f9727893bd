"""Post-decode step of the reference's inference: decoded poses -> the Euler channel table its BVH
writer receives (codebook/VisualizeCodebook.py:148-149, 361-365; process/process_bvh.py:57-83).

    euler = poses_to_euler(poses_normalised, data_mean, data_std)          # (T, 45) float64, degrees, ZXY per joint
    write_bvh(path, euler)                                                  # minimal BVH carrying those channels

The conversion (de-normalise, optional Savitzky-Golay, orthogonalise, matrix -> intrinsic ZXY Euler) is ONE HIP kernel
(qpg_pose_to_euler_f64).  What the reference does AFTER it - `pymo`'s inverse feature pipeline (un-mirroring, constant
channels, root transform, the recorded skeleton's offsets) and `pymo.writers.BVHWriter` - is third-party code fitted
on the BEAT recordings and stays out of scope: `write_bvh` emits a self-contained BVH whose hierarchy is the 15
selected joints with placeholder offsets, so the channels can be inspected in any BVH viewer, not the reference's
retargeted skeleton."""
import numpy as np
import torch

from . import _lib

TARGET_JOINTS = ['Spine', 'Spine1', 'Spine2', 'Spine3', 'Neck', 'Neck1', 'Head',                # process_bvh.py:17-19
                 'RightShoulder', 'RightArm', 'RightForeArm', 'RightHand',
                 'LeftShoulder', 'LeftArm', 'LeftForeArm', 'LeftHand']
PARENTS = [-1, 0, 1, 2, 3, 4, 5, 3, 7, 8, 9, 3, 11, 12, 13]                                     # the BEAT upper-body chain


def savgol_tables(window=15, polyorder=2):
    """Coefficient tables of scipy.signal.savgol_filter(x, window, polyorder, mode='interp') as plain least squares:
    mid [W] (interior frames), head / tail [W//2][W] (the fitted polynomial of the first / last W samples evaluated at
    the edge positions)."""
    h = window // 2
    x = np.arange(window, dtype=np.float64)
    A = np.vander(x, polyorder + 1, increasing=True)              # [W][p+1]
    P = np.linalg.pinv(A)                                         # coefficients = P @ y
    ev = lambda pos: np.vander(np.atleast_1d(np.float64(pos)), polyorder + 1, increasing=True) @ P
    mid = ev(h)[0]
    head = np.concatenate([ev(i) for i in range(h)])
    tail = np.concatenate([ev(window - h + i) for i in range(h)])
    return np.ascontiguousarray(mid), np.ascontiguousarray(head), np.ascontiguousarray(tail)


def poses_to_euler(poses, data_mean, data_std, smoothing=False, device="cuda:0", window=15):
    """poses: (T, 9 J) normalised decoder output (array or device tensor).  Returns a float64 NumPy array (T, 3 J):
    per joint the intrinsic Z, X, Y angles in degrees - `out_euler` of make_bvh_GENEA2020_BT."""
    dev = torch.device(device)
    p = torch.as_tensor(poses).to(dev, torch.float32).contiguous()
    T, C = p.shape
    if C % 9:
        raise ValueError("poses must have 9 values per joint (got %d columns)" % C)
    J = C // 9
    mean = torch.as_tensor(np.asarray(data_mean, np.float64).reshape(-1)).to(dev)
    stdc = torch.as_tensor(np.clip(np.asarray(data_std, np.float64).reshape(-1), a_min=0.01, a_max=None)).to(dev)
    if mean.numel() != C or stdc.numel() != C:
        raise ValueError("data_mean / data_std must have %d entries" % C)
    tabs = [None, None, None]
    if smoothing:
        if T < window:
            raise ValueError("If mode is 'interp', window_length must be less than or equal to the size of x.")  # scipy's
        tabs = [torch.from_numpy(t_).to(dev) for t_ in savgol_tables(window, 2)]
    out = torch.empty((T, J * 3), dtype=torch.float64, device=dev)
    status = torch.zeros((1,), dtype=torch.int32, device=dev)
    _lib.call("qpg_pose_to_euler_f64", dev, p, T, J, mean, stdc, tabs[0], tabs[1], tabs[2], window, out, status)
    if int(status.item()):
        raise ValueError("Non-positive determinant (left-handed or null coordinate frame) in a rotation matrix")
    return out.cpu().numpy()


def write_bvh(path, euler, joints=None, parents=None, frame_time=1.0 / 60.0, offset=(0.0, 10.0, 0.0)):
    """Minimal BVH: one joint per rotation triple, channels `Zrotation Xrotation Yrotation` (the order of the ZXY
    angles), placeholder offsets.  NOT the reference's skeleton (see the module docstring)."""
    euler = np.asarray(euler, np.float64)
    J = euler.shape[1] // 3
    joints = list(joints or TARGET_JOINTS[:J])
    parents = list(parents or PARENTS[:J])
    kids = {i: [k for k in range(J) if parents[k] == i] for i in range(-1, J)}
    lines = ["HIERARCHY"]

    def emit(i, depth):
        ind = "\t" * depth
        lines.append("%s%s %s" % (ind, "ROOT" if parents[i] < 0 else "JOINT", joints[i]))
        lines.append(ind + "{")
        lines.append("%s\tOFFSET %.6f %.6f %.6f" % ((ind,) + (tuple(offset) if parents[i] >= 0 else (0.0, 0.0, 0.0))))
        lines.append("%s\tCHANNELS 3 Zrotation Xrotation Yrotation" % ind)
        if kids[i]:
            for k in kids[i]:
                emit(k, depth + 1)
        else:
            lines.extend([ind + "\tEnd Site", ind + "\t{", "%s\t\tOFFSET %.6f %.6f %.6f" % ((ind,) + tuple(offset)),
                          ind + "\t}"])
        lines.append(ind + "}")
    order = []

    def walk(i):
        order.append(i)
        for k in kids[i]:
            walk(k)
    for r in kids[-1]:
        emit(r, 0)
        walk(r)
    lines += ["MOTION", "Frames: %d" % euler.shape[0], "Frame Time: %.7f" % frame_time]
    cols = np.concatenate([np.arange(3 * i, 3 * i + 3) for i in order])
    for row in euler[:, cols]:
        lines.append(" ".join("%.6f" % v for v in row))
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
    return order
