"""Synthetic inputs with the reference's on-disk schema.

No real BEAT database or checkpoint ships with the reference, so every test,
golden vector and bench line is driven by these generators.  Everything is
drawn from numpy.random.Generator(PCG64(seed)) so that the GPU box can
regenerate byte-identical inputs from a seed instead of shipping them.
"""
import os
import numpy as np

from .constant import codebook_size, num_frames, num_frames_code


def _rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def make_codes(n_seq, seed, force_all_present=True):
    """(n_seq, 30) int64 code ids; every code forced present on grid rows 0..25 when it fits.

    Presence has to hold on the 26 grid positions the scans visit
    (GestureKNN.py:673-675, 713-714), not merely somewhere in the row.
    """
    rng = _rng(seed)
    code = rng.integers(0, codebook_size, size=(n_seq, num_frames_code), dtype=np.int64)
    if force_all_present and n_seq * 26 >= codebook_size:
        slots = rng.permutation(n_seq * 26)[:codebook_size]
        code[slots // 26, slots % 26] = np.arange(codebook_size)
    return code


def make_db(n_seq, seed, wavlm_dim=1024, with_body=False):
    """Dict of arrays for one split (train or test) of a speaker database."""
    rng = _rng(seed)
    d = {}
    d["mfcc"] = rng.standard_normal((n_seq, num_frames, 14)).astype(np.float32)
    d["energy"] = rng.random((n_seq, num_frames)).astype(np.float32)
    d["pitch"] = rng.random((n_seq, num_frames)).astype(np.float32)
    d["volume"] = rng.random((n_seq, num_frames)).astype(np.float32)
    d["context"] = rng.standard_normal((n_seq, num_frames_code, 1, 384)).astype(np.float32)
    # dense phase: (n, 240, 4, 8); index 0 = phase shift p, 2 = amplitude a (PAE.py:505-508)
    d["phase_dense"] = rng.standard_normal((n_seq, num_frames, 4, 8)).astype(np.float32)
    d["wavlm"] = rng.standard_normal((n_seq, 199, wavlm_dim)).astype(np.float32)
    d["wavvq"] = rng.integers(0, 320, size=(n_seq, 398, 2), dtype=np.int64)
    if with_body:
        d["body"] = rng.standard_normal((n_seq, num_frames, 135)).astype(np.float32)
    return d


def make_signature(seed):
    return _rng(seed).standard_normal((codebook_size, 135)).astype(np.float32)


def phase_dense_to_object(phase_dense):
    """(n,240,4,8) f32 -> object array (n,240,4) of torch tensors shaped (1,8,1).

    That is the layout PAE.py:505-508 / fix_device_bug.py:14-22 leave in the
    reference's *_txt_2.npz files.
    """
    import torch
    n, t, c, _ = phase_dense.shape
    out = np.empty((n, t, c), dtype=object)
    for i in range(n):
        for j in range(t):
            for k in range(c):
                out[i, j, k] = torch.from_numpy(phase_dense[i, j, k].copy()).reshape(1, 8, 1)
    return out


def apply_variant(tr, te, code, variant):
    """Planted structure on top of the seeded arrays (in place), shared by the golden generator and the tests.

    'neartie': audio near-ties below the rounding noise of any distance formula (SURVEY.md §7 hard part 2).  The two
        test windows are DB windows 5 and 9, so every query has a candidate at distance exactly 0; DB windows 20 / 30
        are copies of 5 / 9 with a few elements moved by one float32 ulp and the SAME code row (two candidates of one
        code within ~1e-19 of each other), 21 / 31 the same with their own code rows (minima of two different codes
        within ~1e-19), 22 an exact duplicate of 5 with the same codes (exact tie: first index wins).
    'texttie': repeated context rows, as real BEAT data has (every silent code frame carries the identical
        encode(['']) embedding, make_beat_dataset.py:556-565): many codes tie EXACTLY in txt_dist, often at distance 0,
        and the reference ranks them with NumPy's unstable argsort.
    'nearsilent': a near-silent stretch, as a speaker DB has between utterances: DB windows 8..37 and test window 0
        carry ONE quiet feature vector (amplitude 1e-3) in every frame, the copies differing by about one float32 ulp
        per element (8..12 are bit-identical), 22 of the 30 windows sharing six codes.  All 30 x 26 = 780 candidates of
        that stretch then sit within ~1e-14 of each other for a quiet query (about 130 per crowded code) - more than
        any fixed-size near-tie list holds - and the reference's strict `<` scan (GestureKNN.py:685-689) still has an
        answer for every one of them in its own arithmetic.
    'speechlike': feature statistics closer to real WavLM tracks than i.i.d. N(0,1): AR(1) in time (rho = 0.95) on
        rank-64 mixtures over the 1024 features plus a small full-rank residual, 10 % near-silent frames (amplitude
        1e-3 around a common quiet vector), and context rows that repeat over a few code frames (words spanning
        frames) with 20 % silence embeddings."""
    if variant is None:
        return
    if variant == "neartie":
        w = tr["wavlm"]
        te["wavlm"][0], te["wavlm"][1] = w[5], w[9]

        def ulp(src, stride):
            x = src.copy()
            flat = x.reshape(x.shape[0], -1)
            flat[:, ::stride] = np.nextafter(flat[:, ::stride], np.float32(np.inf))
            return x
        w[20], w[21] = ulp(w[5], 97), ulp(w[5], 89)
        w[30], w[31] = ulp(w[9], 97), ulp(w[9], 89)
        w[22] = w[5]
        code[20], code[30], code[22] = code[5], code[9], code[5]
    elif variant == "texttie":
        rng = _rng(777)
        sil = rng.standard_normal((384,)).astype(np.float32)           # the "silence" embedding
        ctx = tr["context"]
        mask = rng.random(ctx.shape[:2]) < 0.35
        ctx[mask] = sil
        for j in range(0, ctx.shape[0], 3):                            # words spanning several code frames
            ctx[j, 10:14] = ctx[j, 10]
        tq = te["context"]
        tq[:, ::4] = sil                                               # silent query frames: distance exactly 0
    elif variant == "nearsilent":
        rng = _rng(778)
        F = tr["wavlm"].shape[2]
        quiet = (rng.standard_normal((F,)) * 1e-3).astype(np.float32)

        def stretch(n):
            x = np.broadcast_to(quiet, (n, 199, F)).astype(np.float32)
            flip = rng.integers(-1, 2, size=x.shape)                    # -1 / 0 / +1 ulp per element
            up = np.nextafter(x, np.float32(np.inf))
            dn = np.nextafter(x, np.float32(-np.inf))
            return np.where(flip > 0, up, np.where(flip < 0, dn, x)).astype(np.float32)
        w = tr["wavlm"]
        w[8:38] = stretch(30)
        w[8:13] = quiet                                                 # bit-identical windows: exact ties, first wins
        code[8:30] = 100 + rng.integers(0, 6, size=(22, code.shape[1]))  # six crowded codes; 30..37 keep their own
        te["wavlm"][0] = stretch(1)[0]
    elif variant == "speechlike":
        speechlike_transform(tr, 779)
        speechlike_transform(te, 780)
    else:
        raise ValueError(variant)


def speechlike_transform(d, seed):
    """In place: give one split of a synthetic database (make_db) feature statistics closer to real WavLM / sentence-
    embedding tracks (see apply_variant 'speechlike').  `seed` drives the temporal processes; the mixing matrix, the
    quiet vector and the silence embedding are common to all splits and chunks (one speaker, one room)."""
    rng = _rng(seed)
    n, T, F = d["wavlm"].shape
    mix = (_rng(783).standard_normal((64, F)) / 8.0).astype(np.float32) if seed not in (779, 780) else \
        (rng.standard_normal((64, F)) / 8.0).astype(np.float32)
    z = np.empty((n, T, 64), np.float32)
    z[:, 0] = rng.standard_normal((n, 64))
    innov = (rng.standard_normal((n, T, 64)) * np.sqrt(1 - 0.95 ** 2)).astype(np.float32)
    for t in range(1, T):
        z[:, t] = 0.95 * z[:, t - 1] + innov[:, t]
    x = z @ mix + 0.05 * d["wavlm"]                              # low-rank, slowly varying + a little of the rest
    quiet = (_rng(781).standard_normal((F,)) * 1e-3).astype(np.float32)
    sil = rng.random((n, T)) < 0.10
    x[sil] = quiet * (1.0 + 1e-3 * rng.standard_normal((int(sil.sum()), F)).astype(np.float32))
    d["wavlm"][...] = x.astype(np.float32)
    ctx = d["context"]
    silence = _rng(782).standard_normal((384,)).astype(np.float32)
    for j in range(n):
        r = 0
        while r < ctx.shape[1]:
            span = int(rng.integers(1, 5))
            ctx[j, r:r + span] = silence if rng.random() < 0.2 else ctx[j, r]
            r += span


def write_npz_set(outdir, n_train, n_test, seed_train=0, seed_test=1, seed_code=2, seed_sig=3,
                  wavlm_dim=1024, variant=None):
    """Write the 8 npz files GestureKNN.py's CLI takes; returns the path dict (flag -> path)."""
    os.makedirs(outdir, exist_ok=True)
    tr = make_db(n_train, seed_train, wavlm_dim)
    te = make_db(n_test, seed_test, wavlm_dim)
    code = make_codes(n_train, seed_code)
    sig = make_signature(seed_sig)
    apply_variant(tr, te, code, variant)
    p = {k: os.path.join(outdir, v) for k, v in dict(
        train_database="train_240_txt_2.npz", test_data="test_240_txt_2.npz",
        train_codebook="train_240_code.npz", codebook_signature="code.npz",
        train_wavlm="train_240_WavLM.npz", test_wavlm="test_240_WavLM.npz",
        train_wavvq="train_240_WavVQ.npz", test_wavvq="test_wavvq_240.npz").items()}
    for d, path in ((tr, p["train_database"]), (te, p["test_data"])):
        np.savez(path, mfcc=d["mfcc"], energy=d["energy"], pitch=d["pitch"], volume=d["volume"],
                 context=d["context"], phase=phase_dense_to_object(d["phase_dense"]))
    np.savez(p["train_codebook"], code=code)
    np.savez(p["codebook_signature"], signature=sig)
    np.savez(p["train_wavlm"], wavlm=tr["wavlm"])
    np.savez(p["test_wavlm"], wavlm=te["wavlm"])
    np.savez(p["train_wavvq"], wavvq=tr["wavvq"])
    np.savez(p["test_wavvq"], wavvq=te["wavvq"])
    return p


def write_npz_set_dense(outdir, n_train, n_test, chunked_db=None, seed_code=2, seed_sig=3):
    """The 8 npz files of the CLI at bench size (n_train in the thousands), generated in 64-window chunks, with the phase
    track written as a DENSE f32 (n,240,4,8) array (data_processing.densify_phase accepts it; the reference's object
    array of pickled torch tensors takes ~0.2 s per DB window to write and to read back).  Returns the path dict."""
    os.makedirs(outdir, exist_ok=True)

    def chunks(n, seed0):
        parts = [make_db(min(64, n - c0), seed0 * 100003 + c0 // 64) for c0 in range(0, n, 64)]
        return {k: np.concatenate([p_[k] for p_ in parts]) for k in parts[0]}
    tr, te = chunks(n_train, 0), chunks(n_test, 1)
    p = {k: os.path.join(outdir, v) for k, v in dict(
        train_database="train_240_txt_2.npz", test_data="test_240_txt_2.npz",
        train_codebook="train_240_code.npz", codebook_signature="code.npz",
        train_wavlm="train_240_WavLM.npz", test_wavlm="test_240_WavLM.npz",
        train_wavvq="train_240_WavVQ.npz", test_wavvq="test_wavvq_240.npz").items()}
    for d, path in ((tr, p["train_database"]), (te, p["test_data"])):
        np.savez(path, mfcc=d["mfcc"], energy=d["energy"], pitch=d["pitch"], volume=d["volume"],
                 context=d["context"], phase=np.ascontiguousarray(d["phase_dense"], np.float32))
    np.savez(p["train_codebook"], code=make_codes(n_train, seed_code))
    np.savez(p["codebook_signature"], signature=make_signature(seed_sig))
    np.savez(p["train_wavlm"], wavlm=tr["wavlm"])
    np.savez(p["test_wavlm"], wavlm=te["wavlm"])
    np.savez(p["train_wavvq"], wavvq=tr["wavvq"])
    np.savez(p["test_wavvq"], wavvq=te["wavvq"])
    return p


# ----------------------------------------------------------------------------------------------
# gesture VQ-VAE: seeded weights with the reference's checkpoint key names
# ----------------------------------------------------------------------------------------------
VQVAE_HPS = dict(input_dim=135, width=512, emb_width=512, l_bins=512, down_t=3, stride_t=2, depth=3,
                 dilation_growth_rate=3, reverse_decoder_dilation=True)


def make_vqvae_state_dict(seed, hps=None, prefix="module."):
    """No pretrained checkpoint ships with the reference (pretrained_model/ is an empty placeholder),
    so parity runs use seeded weights with the checkpoint's exact key names and shapes
    (train.py:114-116 saves a DataParallel state_dict: keys carry `module.`)."""
    h = dict(VQVAE_HPS, **(hps or {}))
    rng = _rng(seed)
    W, E, Cin, K = h["width"], h["emb_width"], h["input_dim"], h["l_bins"]
    sd = {}

    def conv(name, cout, cin, k, gain=1.0):
        sd[prefix + name + ".weight"] = (rng.standard_normal((cout, cin, k)) * gain / np.sqrt(cin * k)).astype(np.float32)
        sd[prefix + name + ".bias"] = (rng.standard_normal((cout,)) * 0.05).astype(np.float32)

    def resnet(name):
        for d in range(h["depth"]):
            conv("%s.model.%d.model.1" % (name, d), W, W, 3, 1.4)
            conv("%s.model.%d.model.3" % (name, d), W, W, 1, 0.5)

    enc = "encoders.0.level_blocks.0.model"
    for i in range(h["down_t"]):
        conv("%s.%d.0" % (enc, i), W, Cin if i == 0 else W, 2 * h["stride_t"])
        resnet("%s.%d.1" % (enc, i))
    conv("%s.%d" % (enc, h["down_t"]), E, W, 3)
    dec = "decoders.0.level_blocks.0.model"
    conv(dec + ".0", W, E, 3)
    for i in range(h["down_t"]):
        resnet("%s.%d.0" % (dec, i + 1))
        # ConvTranspose1d weight is (C_in, C_out, k)
        sd[prefix + "%s.%d.1.weight" % (dec, i + 1)] = (rng.standard_normal((W, W, 2 * h["stride_t"]))
                                                        / np.sqrt(W * 2)).astype(np.float32)
        sd[prefix + "%s.%d.1.bias" % (dec, i + 1)] = (rng.standard_normal((W,)) * 0.05).astype(np.float32)
    conv("decoders.0.out", Cin, E, 3)
    sd[prefix + "bottleneck.level_blocks.0.k"] = rng.standard_normal((K, E)).astype(np.float32)
    return sd
