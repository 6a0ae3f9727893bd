#!/usr/bin/env python
"""Generate golden vectors by running the REFERENCE (imported from /root/reference).

Runs only in the build container (the reference does not exist on the GPU box).
Inputs are synthetic and regenerated from PCG64 seeds by qpgesture_amd.synth, so
only OUTPUTS (and captured intermediates) are committed, as small .npz files.

What is captured, per fixture (shipped mode = wavlm_feat + text + phase,
GestureKNN.py:838-843):
  knn_pred                       final (M,30) int64, exactly what the CLI saves   (:845)
  aud_dist / aud_aux / aud_pay   return of CodeKNN.search_audio_cands per step   (:666-691)
  txt_dist / txt_aux / txt_pay   return of CodeKNN.search_text_cands per step    (:708-721)
  pos_score / freq_score         locals of search_code_knn at the end of a step  (:540-545)
  comb_aud / comb_txt            combined_score / combined_score_                (:575, :554)
  vote, phase_out                final_index per step (:657), result_phase       (:656)
  init_code, init_phase          init_code_phase() draw                          (:462-473)

Usage: python tests/golden/make_golden.py [--only NAME]
"""
import argparse
import os
import runpy
import sys
import tempfile
import time
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
REF_DIR = "/root/reference/codebook/Speech2GestureMatching"

from qpgesture_amd import synth  # noqa: E402


def _lev(a, b):
    """Unit-cost edit distance: stand-in for python-Levenshtein (absent, no network)."""
    la, lb = len(a), len(b)
    prev = list(range(lb + 1))
    for i in range(1, la + 1):
        cur = [i] + [0] * lb
        for j in range(1, lb + 1):
            cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (a[i - 1] != b[j - 1]))
        prev = cur
    return prev[lb]


def run_reference(paths, mode, max_frames=0):
    """Run the reference on the npz set; returns dict of captured arrays."""
    stub = types.ModuleType("Levenshtein")
    stub.distance = _lev
    sys.modules["Levenshtein"] = stub
    argv = ["GestureKNN.py"]
    for k, v in paths.items():
        argv += ["--" + k, v]
    out = os.path.join(os.path.dirname(paths["train_database"]), "ref_out.npz")
    argv += ["--out_knn_filename", out, "--max_frames", str(max_frames)]
    old_argv, old_cwd, old_path = sys.argv, os.getcwd(), list(sys.path)
    sys.argv = argv
    os.chdir(REF_DIR)
    sys.path.insert(0, REF_DIR)
    cap = dict(aud=[], txt=[], steps=[], init=[], ret=[])
    try:
        g = runpy.run_path(os.path.join(REF_DIR, "GestureKNN.py"), run_name="ref")
        K = g["CodeKNN"]
        orig_aud, orig_txt, orig_init, orig_knn = (K.search_audio_cands, K.search_text_cands,
                                                   K.init_code_phase, K.search_code_knn)

        def wrap_aud(self, clip_input, mode="audio"):
            r = orig_aud(self, clip_input, mode)
            cap["aud"].append(r)
            return r

        def wrap_txt(self, clip_input, mode="wavvq_feat"):
            r = orig_txt(self, clip_input, mode)
            cap["txt"].append(r)
            return r

        def wrap_init(self):
            r = orig_init(self)
            cap["init"].append(r)
            return r

        code_obj = orig_knn.__code__
        # last statement of the while body: `i += STEP_SZ * self.step_sz` (GestureKNN.py:659)
        step_line = 659

        def local_trace(frame, event, arg):
            if event == "line" and frame.f_lineno == step_line:
                L = frame.f_locals
                cap["steps"].append({k: np.array(L[k]).copy() for k in
                                     ("pos_score", "freq_score", "combined_score", "combined_score_",
                                      "final_index", "aud_score", "txt_score") if k in L})
            return local_trace

        def global_trace(frame, event, arg):
            if frame.f_code is code_obj:
                return local_trace
            return None

        def wrap_knn(self, *a, **kw):
            sys.settrace(global_trace)
            try:
                r = orig_knn(self, *a, **kw)
            finally:
                sys.settrace(None)
            cap["ret"].append(r)
            return r

        K.search_audio_cands, K.search_text_cands = wrap_aud, wrap_txt
        K.init_code_phase, K.search_code_knn = wrap_init, wrap_knn
        t0 = time.time()
        if mode == "shipped":
            g["main_codebook"](maxFrames=max_frames)
            knn_pred = np.load(out)["knn_pred"]
        elif mode in ("wavvq_aud_txt", "wavvq_aud"):
            # main_codebook's body (GestureKNN.py:816-845) with the vq-wav2vec flags the paper describes
            # instead of the hard-coded ones at :842-843
            a = g["args"]
            sys.path.insert(0, REF_DIR)
            from data_processing import load_db_codebook, calc_data_stats
            L = load_db_codebook(a.train_database, a.train_codebook, a.test_data, a.train_wavlm, a.test_wavlm,
                                 a.train_wavvq, a.test_wavvq)
            (train_mfcc, train_code, test_mfcc, train_feat, test_feat, train_wavlm, test_wavlm, train_wavlm_feat,
             test_wavlm_feat, speech_features, test_speech_features, train_speech_features_feat,
             test_speech_features_feat, train_wavvq_feat, test_wavvq_feat, train_phase, test_phase, train_context,
             test_context) = L
            T = lambda x: x.transpose((0, 2, 1))
            st = {}
            for nm, (x, y) in dict(mfcc=(train_mfcc, test_mfcc), feat=(train_feat, test_feat),
                                   speech_features=(speech_features, test_speech_features),
                                   speech_features_feat=(train_speech_features_feat,
                                                         test_speech_features_feat)).items():
                m_, s_, _, _ = calc_data_stats(T(x), T(y))
                st[nm + "_train_mean"], st[nm + "_train_std"] = m_, s_
            # With the seed the reference sets at import (123456) the first draw is init_j = 234 > 232, the
            # 8-frame phase slice comes back short and np.array(result_phase) raises on NumPy >= 1.24
            # (SURVEY.md §7.6).  Re-seed so the draw is valid; the tests seed the same way.
            np.random.seed(2)
            knn_pred = g["predict_code_from_audio"](
                train_mfcc, train_code, test_mfcc, st, train_feat, test_feat, train_wavlm, test_wavlm,
                train_wavlm_feat, test_wavlm_feat, speech_features, test_speech_features,
                train_speech_features_feat, test_speech_features_feat, train_wavvq_feat, test_wavvq_feat,
                train_phase, test_phase, train_context, test_context, use_feature=True, use_wavlm=False,
                use_freq=False, use_speechfeat=False, use_wavvq=True, use_phase=True,
                use_txt=(mode == "wavvq_aud_txt"), use_aud=True, frames=max_frames)
        else:
            raise ValueError(mode)
        wall = time.time() - t0
    finally:
        sys.argv = old_argv
        os.chdir(old_cwd)
        sys.path[:] = old_path
    res = {"knn_pred": knn_pred, "ref_wall_s": np.float64(wall)}

    def pack(trip):
        dist = np.array([t[0] for t in trip])
        pay = np.full((len(trip), 512, 4), -1, np.int64)
        aux = np.full((len(trip), 512, 2), -1, np.int64)
        for s, t in enumerate(trip):
            for c in range(512):
                if len(t[1][c]):
                    p = np.asarray(t[1][c])
                    pay[s, c, :len(p)] = p
                    aux[s, c] = t[2][c]
        return dist, pay, aux

    if cap["aud"]:
        d, p, a = pack(cap["aud"])
        d = d.astype(np.float64)
        res.update(aud_dist=d, aud_pay=p.astype(np.int16), aud_aux=a.astype(np.int32))
    if cap["txt"]:
        d, p, a = pack(cap["txt"])
        if d.dtype != np.float32:
            # codes absent from the DB keep the Python float 1e+3 placeholder (GestureKNN.py:709), which makes
            # np.array(list) float64; every real entry is still a float32 value
            d32 = d.astype(np.float32)
            assert np.array_equal(d32.astype(np.float64), d), "text distances are not float32 values"
            d = d32
        res.update(txt_dist=d, txt_pay=p.astype(np.int16), txt_aux=a.astype(np.int32))
    if cap["steps"]:
        for k in cap["steps"][0]:
            res["step_" + k] = np.array([s[k] for s in cap["steps"]])
        res["step_freq_score"] = res["step_freq_score"][0].astype(np.int16)
    if cap["init"]:
        res["init_code"] = np.int64(cap["init"][0][0])
        res["init_phase"] = np.asarray(cap["init"][0][1], np.float32)
    if cap["ret"]:
        res["phase_out"] = np.array([r[1] for r in cap["ret"]]).astype(np.float32)
        res["vote"] = np.array([r[2] for r in cap["ret"]]).astype(np.int8)
    return res


FIXTURES = {
    # name: (n_train, n_test, seeds(train,test,code,sig), max_frames, mode[, variant of synth.apply_variant])
    "shipped_n48_m2_s0": (48, 2, (0, 1, 2, 3), 0, "shipped"),
    "shipped_n64_m3_s10": (64, 3, (10, 11, 12, 13), 0, "shipped"),
    # planted audio near-ties (ulp-perturbed duplicate windows) / exact text ties (repeated context rows)
    "shipped_neartie_n48_m2_s30": (48, 2, (30, 31, 32, 33), 0, "shipped", "neartie"),
    "shipped_texttie_n48_m2_s40": (48, 2, (40, 41, 42, 43), 0, "shipped", "texttie"),
    # a near-silent stretch: 780 candidates within ~1e-14 of each other (more than any capped near-tie list holds)
    "shipped_nearsilent_n48_m2_s50": (48, 2, (50, 51, 52, 53), 0, "shipped", "nearsilent"),
    # AR(1) / rank-64 / 10 % near-silent frames / repeating context rows
    "shipped_speechlike_n48_m2_s60": (48, 2, (60, 61, 62, 63), 0, "shipped", "speechlike"),
    # mid-size DB from the reference itself (about 80 s of reference time): a DB larger than the tier-1 list caps see
    "shipped_n256_m2_s70": (256, 2, (70, 71, 72, 73), 0, "shipped"),
    # vq-wav2vec + Levenshtein audio (the mode the paper describes); wavlm_dim=8 keeps the unused WavLM small
    "wavvq_aud_txt_n40_m2_s20": (40, 2, (20, 21, 22, 23), 0, "wavvq_aud_txt"),
    "wavvq_aud_n40_m2_s20": (40, 2, (20, 21, 22, 23), 0, "wavvq_aud"),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    for name, spec in FIXTURES.items():
        ntr, nte, seeds, mf, mode = spec[:5]
        variant = spec[5] if len(spec) > 5 else None
        if a.only and a.only != name:
            continue
        with tempfile.TemporaryDirectory() as td:
            paths = synth.write_npz_set(td, ntr, nte, *seeds, wavlm_dim=1024 if mode == "shipped" else 8,
                                        variant=variant)
            res = run_reference(paths, mode, mf)
        res["meta"] = np.array([ntr, nte, *seeds, mf], np.int64)
        res["variant"] = np.array(variant or "")
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **res)
        print(name, "knn_pred", res["knn_pred"].shape, "ref wall %.1fs" % res["ref_wall_s"],
              {k: (v.shape, str(v.dtype)) for k, v in res.items()})


if __name__ == "__main__":
    main()
