"""Drop-in for the reference's codebook/Speech2GestureMatching/GestureKNN.py command line.

Same flags (GestureKNN.py:25-39), same .npz inputs, same output file
(`np.savez_compressed(out, knn_pred=int64 (M,30))`, :845), same seeding (:19-22) — the matching
itself runs on the MI355X through libqpg_hip.so.  Run as
    python -m qpgesture_amd.GestureKNN --train_database ... --out_knn_filename result.npz

Additive flags: --device (default cuda:0), --mode (default `shipped` = the flags hard-coded at
GestureKNN.py:842-843: wavlm_feat + text + phase; `audio` / `text` = the single-modality phase
branches :593-625; `wavvq` / `wavvq_audio` = vq-wav2vec Levenshtein audio, the flags the paper describes,
with / without text), --seed (default 123456 as :19; in wavvq modes that seed draws an invalid initial
phase slice in the reference as well), --tie_rule (how EQUAL values are ranked - code frequencies, and the per-code
audio / text minima, which tie exactly on real text data where silent frames share one embedding: `numpy` = the
reference's own `argsort().argsort()` call on the host (NumPy's unstable sort, like the reference), `stable` = lowest
code first, deterministic, ranks taken on the device).
"""
import argparse
import os
import random
import sys
import time

import numpy as np

seed_value = 123456


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument('-d', '--train_database', type=str, default="/path/to/training_db_data.npz")
    p.add_argument('-c', '--train_codebook', type=str, default="/path/to/training_db_data.npz")
    p.add_argument('-w', '--train_wavlm', type=str, default="/path/to/training_db_data.npz")
    p.add_argument('-wvq', '--train_wavvq', type=str, default="/path/to/training_db_data.npz")
    p.add_argument('-s', '--codebook_signature', type=str, default="/path/to/training_db_data.npz")
    p.add_argument('-e', '--test_data', type=str, default="/path/to/test_data.npz")
    p.add_argument('-tw', '--test_wavlm', type=str, default="/path/to/training_db_data.npz")
    p.add_argument('-twvq', '--test_wavvq', type=str, default="/path/to/training_db_data.npz")
    p.add_argument('-om', '--out_knn_filename', type=str, default="/path/to/knn_pred.npz")
    p.add_argument('-ov', '--out_video_path', type=str, default="/path/to/video/")
    p.add_argument('-k', '--desired_k', type=int, default=0)
    p.add_argument('-f', '--fake', type=bool, default=False)
    p.add_argument('-of', '--out_fake_knn_filename', type=str, default="/path/to/knn_pred.npz")
    p.add_argument('--max_frames', type=int, default=0)
    # additive
    p.add_argument('--device', type=str, default="cuda:0")
    p.add_argument('--mode', choices=["shipped", "audio", "text", "wavvq", "wavvq_audio"], default="shipped")
    p.add_argument('--seed', type=int, default=seed_value)
    p.add_argument('--tie_rule', choices=["numpy", "stable"], default="numpy")
    p.add_argument('--audio_precision', choices=["mixed", "f64", "exact"], default="mixed",
                   help="mixed: f32 matrix-core sweep with an a-priori error bound + exact re-evaluation of every "
                        "undecided comparison (same output); f64: the f64 matrix-core sweep; exact: f64 sweep + the uncapped "
                        "near-tie guard.  A clip on which a capped guard of mixed / f64 raises its trouble word is "
                        "re-matched on `exact` automatically: unguarded codes are never written")
    p.add_argument('--db_cache', choices=["auto", "off", "refresh"], default="auto",
                   help="prepared-database cache (qpgesture_amd/db_cache.py): the device-resident database this command "
                        "builds from the five database-side files is written once, keyed by their paths, sizes and mtimes, "
                        "and restored by later invocations without re-reading / re-packing anything.  auto: use or create; "
                        "refresh: rebuild and overwrite; off: never touch the cache directory")
    p.add_argument('--db_cache_dir', type=str, default=None, help="default: $QPG_DB_CACHE_DIR or ~/.cache/qpgesture_amd")
    return p


def _tables_have_exact_ties(T):
    """Device-side: does any row of the audio / text minima hold two EQUAL values?  Only then can NumPy's unstable
    argsort (the reference's rank expression, GestureKNN.py:553, 574) differ from the stable ranks the device took."""
    import torch
    flag = None
    for k in ("aud_d", "txt_d"):
        if T.get(k) is None:
            continue
        sd = torch.sort(T[k], dim=1).values
        f = (sd[:, 1:] == sd[:, :-1]).any()
        flag = f if flag is None else (flag | f)
    return flag


def main_codebook(args, maxFrames=0):
    """main_codebook (GestureKNN.py:816-845)."""
    import torch
    from .code_knn import MODE_AUD, MODE_AUD_TXT, MODE_TXT, CodeKNN, GestureDB
    from .data_processing import load_db_codebook

    from . import db_cache
    from .data_processing import load_test_side

    t0 = time.time()
    vq = args.mode.startswith("wavvq")
    db, cpath, ckey, csrc = None, None, None, None
    if args.db_cache != "off":
        files = [args.train_database, args.train_codebook, args.train_wavlm, args.codebook_signature] + \
            ([args.train_wavvq] if vq else [])
        copts = {"tie_rule": args.tie_rule, "wavvq": vq, "device_kind": "hip"}
        ckey, csrc = db_cache.file_key(files, copts), db_cache.sources_id(files, copts)
        cpath = db_cache.cache_path(ckey, args.db_cache_dir)
        if args.db_cache == "auto":
            db = GestureDB.load(cpath, args.device, ckey)
    from_cache = db is not None
    if db is None:
        L = load_db_codebook(args.train_database, args.train_codebook, args.test_data, args.train_wavlm,
                             args.test_wavlm, args.train_wavvq, args.test_wavvq, device=args.device)
        signature = np.load(args.codebook_signature)['signature']                 # :476
        freq_rank = None
        if args.tie_rule == "numpy":
            cnt = np.bincount(np.asarray(L.code).reshape(-1), minlength=512)[:512]
            freq = np.where(cnt > 0, 1 - cnt / cnt.sum(), 1.0)                   # :481-499
            freq_rank = np.array(list(freq)).argsort().argsort()                 # :544, the reference's own call
        db = GestureDB(L.code, L.train_wavlm, L.train_context, L.train_phase, signature, device=args.device,
                       freq_rank=freq_rank, wavvq=L.train_wavvq if vq else None)
    else:
        L = load_test_side(args.test_data, args.test_wavlm, args.test_wavvq, device=args.device)
    knn = CodeKNN(db, use_wavlm=not vq, use_wavvq=vq)                            # draws from np.random like :463-464
    knn.audio_precision = args.audio_precision
    n_test_seq = maxFrames if maxFrames != 0 else L.test_wavvq.shape[0]          # :740
    dev = db.device
    te_i = (torch.from_numpy(np.ascontiguousarray(L.test_wavvq[:n_test_seq])).to(dev) if vq
            else L.test_wavlm[:n_test_seq].contiguous())
    te_c = torch.from_numpy(L.test_context[:n_test_seq]).to(dev)
    t1 = time.time()
    print('begin search...')
    mode = {"shipped": MODE_AUD_TXT, "audio": MODE_AUD, "text": MODE_TXT, "wavvq": MODE_AUD_TXT,
            "wavvq_audio": MODE_AUD}[args.mode]
    seed_code, seed_phase = knn.init_code_phase()                                 # (drawn once: :462-473)
    # --tie_rule numpy: the reference ranks both (Q,512) tables with NumPy's UNSTABLE argsort, whose order differs from a
    # stable one only among EQUAL values.  So the clip is matched with the device's stable ranks, the tables are checked for
    # exact ties on the device, and only a clip that has one is ranked again by the reference's own NumPy call on the host
    # (CodeKNN.host_ranks: both tables through the host in the middle of the step).  Real text tracks tie (silent frames
    # share one embedding); continuous features do not.
    pred_seqs, _, _ = knn.match_clip(te_i, te_c, n_test_seq, mode=mode, seed_code=seed_code, seed_phase=seed_phase,
                                     return_tables=True)                           # (re-matches on the uncapped path if flagged)
    if args.tie_rule == "numpy" and bool(_tables_have_exact_ties(knn.tables)):
        knn.host_ranks = True
        pred_seqs, _, _ = knn.match_clip(te_i, te_c, n_test_seq, mode=mode, seed_code=seed_code, seed_phase=seed_phase)
    t2 = time.time()
    if knn.fallbacks:
        print('near-tie guard: a capped re-evaluation list overflowed; the clip was re-matched on the uncapped path')
    print(pred_seqs.shape)
    np.savez_compressed(args.out_knn_filename, knn_pred=pred_seqs)               # :845
    print('load+prepare %.2fs%s, match %.4fs (%.0f frames/s)' % (t1 - t0, ' (prepared-database cache)' if from_cache else '',
                                                                   t2 - t1, 240 * n_test_seq / (t2 - t1)))
    if cpath is not None and not from_cache:
        try:                                    # behind the result: a later invocation finds the database prepared
            db.save(cpath, ckey, sources=csrc)
        except Exception as e:                  # noqa: BLE001 (the result is written: a cache failure must not fail the command)
            print('prepared-database cache not written: %r' % (e,))
    return pred_seqs


def main(argv=None):
    args = build_parser().parse_args(argv)
    os.environ['PYTHONHASHSEED'] = str(seed_value)                               # :19-22
    random.seed(args.seed)
    np.random.seed(args.seed)
    return main_codebook(args, maxFrames=args.max_frames)


if __name__ == "__main__":
    main(sys.argv[1:])
