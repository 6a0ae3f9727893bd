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
