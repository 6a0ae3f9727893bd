"""ctypes wrapper of the oracle's C restatement (oracle/sweep_ref.c).  Test infrastructure only."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libqpg_ref.so")
_lib = None


def build(force=False):
    src = os.path.join(HERE, "sweep_ref.c")
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "-B", "libqpg_ref.so"], stdout=subprocess.DEVNULL)
    return LIB


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = ctypes.CDLL(LIB)
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def audio_scan(base, cand_t, code, cand_cidx, q, K=512, ntaps=6, stride=2, n_threads=1):
    """base f32 (N,T,F); q f64 (Q, ntaps*F).  Returns (dist f64 (Q,K), idx i32 (Q,K))."""
    base = np.ascontiguousarray(base, np.float32)
    q = np.ascontiguousarray(q, np.float64)
    code = np.ascontiguousarray(code, np.int32)
    cand_t = np.ascontiguousarray(cand_t, np.int32)
    cand_cidx = np.ascontiguousarray(cand_cidx, np.int32)
    N, T, F = base.shape
    Q = q.shape[0]
    dist = np.empty((Q, K), np.float64)
    idx = np.empty((Q, K), np.int32)
    load().qpg_ref_audio_scan(_p(base), N, T, F, _p(cand_t), len(cand_t), ntaps, stride, _p(code), code.shape[1],
                              _p(cand_cidx), _p(q), Q, K, _p(dist), _p(idx), n_threads)
    return dist, idx


def text_scan(ctx, cand_r, code, cand_cidx, q, K=512, n_threads=1):
    """ctx f32 (N,R,Dm); q f32 (Q,Dm).  Returns (dist f32 (Q,K), idx i32 (Q,K))."""
    ctx = np.ascontiguousarray(ctx, np.float32)
    q = np.ascontiguousarray(q, np.float32)
    code = np.ascontiguousarray(code, np.int32)
    cand_r = np.ascontiguousarray(cand_r, np.int32)
    cand_cidx = np.ascontiguousarray(cand_cidx, np.int32)
    N, R, Dm = ctx.shape
    Q = q.shape[0]
    dist = np.empty((Q, K), np.float32)
    idx = np.empty((Q, K), np.int32)
    load().qpg_ref_text_scan(_p(ctx), N, R, Dm, _p(cand_r), len(cand_r), _p(code), code.shape[1], _p(cand_cidx),
                             _p(q), Q, K, _p(dist), _p(idx), n_threads)
    return dist, idx
