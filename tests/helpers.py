"""Shared fixtures for the parity tests: regenerate the synthetic inputs of a golden fixture."""
import os
import tempfile

import numpy as np

from qpgesture_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def fixture_arrays(n_train, n_test, s_train, s_test, s_code, s_sig, wavlm_dim=1024, variant=None):
    """In-memory version of synth.write_npz_set (same seeds -> same bytes), already windowed the way
    data_processing.load_db_codebook leaves them: interpolated WavLM, squeezed context, dense phase."""
    from oracle import knn_oracle as O
    tr = synth.make_db(n_train, s_train, wavlm_dim)
    te = synth.make_db(n_test, s_test, wavlm_dim)
    code = synth.make_codes(n_train, s_code)
    synth.apply_variant(tr, te, code, variant)
    return dict(
        code=code, sig=synth.make_signature(s_sig),
        tr_interp=O.interp_wavlm(tr["wavlm"]), te_interp=O.interp_wavlm(te["wavlm"]),
        tr_ctx=tr["context"].squeeze(2), te_ctx=te["context"].squeeze(2),
        tr_phase=tr["phase_dense"], te_phase=te["phase_dense"],
        tr_wavvq=tr["wavvq"], te_wavvq=te["wavvq"])


def aud_tol(knn):
    """Tolerance of the audio distance TABLE against the reference's f64 values: 1e-13 for the f64 sweep; the sweep's
    a-priori bound for the mixed-precision path (untouched minima keep the sweep value; winners and ranks are exact
    either way and asserted separately)."""
    from qpgesture_amd.code_knn import AUDIO_MX_ERR
    mixed = knn.audio_precision == "mixed" and knn.db.world == 1 and knn.tie_eps > 0
    return AUDIO_MX_ERR if mixed else 1e-13


# ---- the stand-alone C-ABI entry points, driven by hand (test scaffolding: the product calls the fused forms) ----------
def sweep_audio_unfused(knn, qbase, q_win, q_t):
    """Distance matrix (qpg_audio_cosine_f64), then the unguarded qpg_percode_select_f64: same tables as CodeKNN.sweep_audio's
    f64 path on tie-free data."""
    import torch
    from qpgesture_amd import _lib
    from qpgesture_amd.code_knn import _i32
    from qpgesture_amd.constant import ABSENT_DIST, NUM_AUDIO_FEAT_FRAMES
    db, dev = knn.db, knn.db.device
    Q = len(q_win)
    qbase = qbase.contiguous()
    M, T, F = qbase.shape
    q32 = torch.empty((Q, NUM_AUDIO_FEAT_FRAMES * F), dtype=torch.float32, device=dev)
    qn2 = torch.empty((Q,), dtype=torch.float64, device=dev)
    _lib.call("qpg_audio_pack_queries", dev, qbase, M, T, F, _i32(q_win, dev), _i32(q_t, dev), Q,
              NUM_AUDIO_FEAT_FRAMES, db.tap_stride, q32, qn2)
    D = torch.empty((Q, max(db.n_local * db.Ga, 1)), dtype=torch.float64, device=dev)
    _lib.call("qpg_audio_cosine_f64", dev, db.base, db.n_local, db.T, db.F, db.aud_t, db.Ga,
              NUM_AUDIO_FEAT_FRAMES, db.tap_stride, db.cn2, q32, qn2, Q, D, D.stride(0))
    dist = torch.empty((Q, db.K), dtype=torch.float64, device=dev)
    idx = torch.empty((Q, db.K), dtype=torch.int32, device=dev)
    _lib.call("qpg_percode_select_f64", dev, D, D.stride(0), Q, db.aud_cand_code, db.n_local * db.Ga, db.K,
              float(ABSENT_DIST), db.idx_base * db.Ga, dist, idx, None, 0, 0)
    return dist, idx, D


def sweep_text_unfused(knn, queries):
    import torch
    from qpgesture_amd import _lib
    from qpgesture_amd.constant import ABSENT_DIST
    db, dev = knn.db, knn.db.device
    Q = queries.shape[0]
    qn = torch.empty_like(queries)
    _lib.call("qpg_l2_normalize_rows_f32", dev, queries, Q, db.Dt, qn)
    D = torch.empty((Q, max(db.Ct, 1)), dtype=torch.float32, device=dev)
    _lib.call("qpg_text_cosine_f32", dev, db.ctxt, db.Ct, db.Dt, qn, Q, D, D.stride(0))
    dist = torch.empty((Q, db.K), dtype=torch.float32, device=dev)
    idx = torch.empty((Q, db.K), dtype=torch.int32, device=dev)
    _lib.call("qpg_percode_select_f32", dev, D, D.stride(0), Q, db.txt_cand_code, db.Ct, db.K, float(ABSENT_DIST),
              db.idx_base * db.Gt, dist, idx, None, 0, 0)
    return dist, idx, D


def decode_layers(m, ids):
    """Layer-by-layer decode through the per-layer entry points (qpg_vq_gather_f32 + qpg_conv1d_f32): same result as
    VQVAE.decode(), which makes ONE C call."""
    import torch
    from qpgesture_amd import _lib
    ids = torch.as_tensor(ids).to(m.device, torch.int64).contiguous()
    B, L = ids.shape
    status = torch.zeros((1,), dtype=torch.int32, device=m.device)
    x = torch.empty((B, L, m.emb), dtype=torch.float32, device=m.device)
    _lib.call("qpg_vq_gather_f32", m.device, m.k, ids, B * L, m.emb, m.bins, x, status)
    T = L
    x = m._conv(m.dec_in, x, B, T, T, in_offset=-1)
    for res, even, odd in m.dec_up:
        x = m._resnet(res, x, B, T, m.reverse)
        y = torch.empty((B, 2 * T, even.cout), dtype=torch.float32, device=m.device)
        m._conv(even, x, B, T, T, in_offset=-1, out=y, out_stride=2, out_offset=0, T_y=2 * T)
        m._conv(odd, x, B, T, T, in_offset=0, out=y, out_stride=2, out_offset=1, T_y=2 * T)
        x, T = y, 2 * T
    return m._conv(m.dec_out, x, B, T, T, in_offset=-1)
