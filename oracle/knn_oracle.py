"""ORACLE (test infrastructure, not product): CPU restatement of the reference's
code-level motion matching (`CodeKNN`) in NumPy.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this file.  The product (qpgesture_amd/) never does.

Pinned against the reference itself: tests/golden/*.npz are captured by
tests/golden/make_golden.py, which imports /root/reference in the build
container; tests/test_oracle_golden.py checks this file against them bit-for-bit.

Citations are into /root/reference/codebook/Speech2GestureMatching/.

Third-party arithmetic restated here (the reference pins neither version;
the image has scikit-learn 1.7.2 / NumPy 2.2.6):
  * sklearn.metrics.pairwise.paired_distances(metric='cosine')
      = 0.5 * row_norms(normalize(X) - normalize(Y), squared=True)
    with row_norms = np.einsum('ij,ij->i', X, X).  NumPy's einsum inner loop is
    built for the SSE baseline only: 4 lanes (f32) / 2 lanes (f64), separate
    multiply and add (no FMA), 4x unrolled with the order a3,a2,a1,a0 inside each
    unrolled group, and a horizontal (l0+l1)+(l2+l3) at the end.  `einsum_sq`
    restates that order so that distances come out BIT-identical to sklearn's
    (checked in tests/test_oracle_golden.py::test_cosine_emulation_bitexact).
  * python-Levenshtein distance(): unit-cost edit distance (`lev`).
"""
import numpy as np

STEP_SZ = 4
NUM_TAPS = 6
N_CODE = 30
N_FRAMES = 240
K_CODES = 512
ABSENT = 1e+3


# ----------------------------------------------------------------------------
# third-party arithmetic
# ----------------------------------------------------------------------------
def einsum_sq(x):
    """sum(x*x) over the last axis in NumPy-einsum order.  x: (..., D) f32 or f64."""
    x = np.asarray(x)
    lanes = 4 if x.dtype == np.float32 else 2
    D = x.shape[-1]
    lead = x.shape[:-1]
    acc = np.zeros(lead + (lanes,), x.dtype)
    step4 = 4 * lanes
    nfull = D // step4
    if nfull:
        body = x[..., :nfull * step4].reshape(lead + (nfull, 4, lanes))
        for g in range(nfull):
            for u in (3, 2, 1, 0):
                seg = body[..., g, u, :]
                acc = seg * seg + acc
    i = nfull * step4
    while i < D:
        m = min(lanes, D - i)
        seg = np.zeros(lead + (lanes,), x.dtype)
        seg[..., :m] = x[..., i:i + m]
        acc = seg * seg + acc
        i += lanes
    if lanes == 4:
        return (acc[..., 0] + acc[..., 1]) + (acc[..., 2] + acc[..., 3])
    return acc[..., 0] + acc[..., 1]


def l2_normalize(x):
    """sklearn.preprocessing.normalize(x, 'l2') row-wise; rows with norm < 10*eps are left
    unscaled (sklearn _handle_zeros_in_scale)."""
    n = np.sqrt(einsum_sq(x))
    n = np.where(n < 10 * np.finfo(n.dtype).eps, np.ones_like(n), n)
    return x / n[..., None]


def cosine_rows(q, X):
    """paired cosine distance of one query against many rows, sklearn bit-exact.

    q: (D,), X: (C, D), same dtype (f32 stays f32, GestureKNN.py:716; f64 for
    wavlm_feat, data_processing.py:264).
    """
    qn = l2_normalize(q[None])[0]
    Xn = l2_normalize(X)
    return X.dtype.type(0.5) * einsum_sq(qn[None] - Xn)


def cosine_pair(a, b):
    return cosine_rows(np.asarray(a), np.asarray(b)[None])[0]


def lev(a, b):
    """Unit-cost edit distance between two integer sequences."""
    la, lb = len(a), len(b)
    prev = list(range(lb + 1))
    for i in range(1, la + 1):
        cur = [i] + [0] * lb
        for j in range(1, lb + 1):
            cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (a[i - 1] != b[j - 1]))
        prev = cur
    return prev[lb]


# ----------------------------------------------------------------------------
# feature windowing (data_processing.py:197-353)
# ----------------------------------------------------------------------------
def interp_wavlm(wavlm, n_code=N_CODE):
    """199 -> 180 frames, linear, align_corners=True, in f32 (data_processing.py:258-261).

    Uses torch's own kernel: it is the reference's dependency, not its code.
    """
    import torch
    import torch.nn.functional as F
    new_t = wavlm.shape[1] // n_code * n_code
    x = torch.from_numpy(np.ascontiguousarray(wavlm)).transpose(1, 2)
    return F.interpolate(x, size=new_t, align_corners=True, mode="linear").transpose(1, 2).numpy()


def wavlm_feat_rows(interp, j, ts):
    """Rows of the (never materialised) wavlm feature stack: for t in ts,
    concat_i interp[j, t+2i] (zeros past the end), promoted to f64
    (data_processing.py:264-268)."""
    T, F_ = interp.shape[1], interp.shape[2]
    out = np.zeros((len(ts), NUM_TAPS, F_), np.float64)
    for r, t in enumerate(ts):
        for i in range(NUM_TAPS):
            if t + 2 * i < T:
                out[r, i] = interp[j, t + 2 * i]
    return out.reshape(len(ts), -1)


def wavvq_feat(wavvq):
    """(n,398,2) -> (n,398,22): 6 backward taps then 5 forward taps, zero padded
    (data_processing.py:297-317).  Kept float64 like the reference's np.zeros."""
    n, T, G = wavvq.shape
    s = T / N_CODE
    out = np.zeros((n, T, 11, G), np.float64)
    for i in range(NUM_TAPS):                      # back taps: shift right by int((5-i)*s)
        sh = int((NUM_TAPS - i - 1) * s)
        out[:, sh:, i] = wavvq[:, :T - sh]
    for i in range(1, NUM_TAPS):                   # forward taps: shift left by int(i*s)
        sh = int(i * s)
        out[:, :T - sh, 5 + i] = wavvq[:, sh:]
    return out.reshape(n, T, 11 * G)


def densify_phase(phase_obj):
    """object (n,240,4)[tensor(1,8,1)] -> f32 (n,240,4,8)."""
    if phase_obj.dtype != object:
        return np.asarray(phase_obj, np.float32).reshape(phase_obj.shape[0], phase_obj.shape[1], 4, 8)
    n, t, c = phase_obj.shape
    out = np.empty((n, t, c, 8), np.float32)
    for i in range(n):
        for j in range(t):
            for k in range(c):
                out[i, j, k] = np.asarray(phase_obj[i, j, k].detach().cpu().numpy()).reshape(8)
    return out


def audio_grid(n_db_frm, step_sz):
    """The k-grid of search_audio_cands (GestureKNN.py:672-690), literal float ops.
    Returns (k_float list, int(k) list, int(k/step_sz) list)."""
    ks, kint, cidx = [], [], []
    k = 0
    while k < n_db_frm - STEP_SZ * step_sz:
        ks.append(k)
        kint.append(int(k))
        cidx.append(int(k / step_sz))
        k += step_sz
    return ks, kint, cidx


def phase_slot(k):
    """int(k/398*240): phase start frame of a candidate (GestureKNN.py:632), applied to
    k in whatever unit the scan used (wavlm 0..150, text 0..200, wavvq 0..344)."""
    return int(k / 398 * 240)


# ----------------------------------------------------------------------------
# CodeKNN restated
# ----------------------------------------------------------------------------
class CodeKNNOracle:
    """State of CodeKNN.__init__ (GestureKNN.py:423-499) for the wavlm or wavvq mode."""

    def __init__(self, code_train, signature, phase_dense, context_train,
                 wavlm_interp=None, wavvq_train_feat=None, mode="wavlm",
                 rng=None, rank_kind="numpy", scan="numpy"):
        self.mode = mode
        self.scan = scan                              # "numpy" (this file) or "c" (oracle/sweep_ref.c, same bits)
        self.code = np.asarray(code_train)
        self.sig = np.asarray(signature, np.float32)
        self.phase = phase_dense                      # (n,240,4,8) f32
        self.ctx = context_train                      # (n,30,384) f32
        self.interp = wavlm_interp                    # (n,180,F) f32
        self.vq = wavvq_train_feat                    # (n,398,22) f64
        if mode == "wavlm":
            self.step_sz = wavlm_interp.shape[1] // N_CODE        # :432
            self.n_db_frm = wavlm_interp.shape[1]
            self.n_db_seq = wavlm_interp.shape[0]
        else:
            self.step_sz = 398 / N_CODE                           # :436
            self.n_db_frm = 398
            self.n_db_seq = wavvq_train_feat.shape[0]
        self.rng = rng if rng is not None else np.random
        self.rank_kind = rank_kind
        # code_to_freq (:481-499): 1 - count/total, 1 for unseen codes
        cnt = np.bincount(self.code.reshape(-1), minlength=K_CODES)[:K_CODES]
        self.freq = np.where(cnt > 0, 1 - cnt / cnt.sum(), 1.0)
        self.freq_rank_override = None
        self.tied_decisions = 0

    # -- ranks ----------------------------------------------------------------
    def rank(self, x):
        x = np.asarray(x)
        if self.rank_kind == "numpy":                 # reference: unstable default sort
            return x.argsort().argsort()
        return x.argsort(kind="stable").argsort(kind="stable")

    def freq_rank(self):
        if self.freq_rank_override is not None:
            return np.asarray(self.freq_rank_override)
        return self.rank(self.freq)

    # -- init (:462-473) --------------------------------------------------------
    def init_code_phase(self):
        i = self.rng.randint(0, self.n_db_seq)
        j = self.rng.randint(0, self.n_db_frm - int(N_FRAMES / N_CODE))
        code = self.code[i, j // N_CODE]
        P = self.phase[i, j:j + 8]
        return code, np.concatenate((P[:, 0], P[:, 2]), axis=1)

    # -- candidate scans --------------------------------------------------------
    def search_audio_cands(self, q, faithful=False):
        """Per-code best audio candidate (GestureKNN.py:666-691).  q: (6144,) f64 [wavlm]
        or (22,) [wavvq].  Returns dist[512], payload[512,4] (-1 = empty), aux[512,2]."""
        ks, kint, cidx = audio_grid(self.n_db_frm, self.step_sz)
        if self.scan == "c" and self.mode == "wavlm":
            from . import cref
            d, ix = cref.audio_scan(self.interp, kint, self.code, cidx, np.asarray(q, np.float64)[None])
            return self._expand(d[0], ix[0], len(ks), kint, cidx)
        dist = np.full(K_CODES, ABSENT, np.float64)
        pay = np.full((K_CODES, 4), -1, np.int64)
        aux = np.full((K_CODES, 2), -1, np.int64)
        for j in range(self.n_db_seq):
            if self.mode == "wavlm":
                d = cosine_rows(q, wavlm_feat_rows(self.interp, j, kint))
            else:
                d = np.array([wavvq_distance(q, self.vq[j, t]) for t in kint], np.float64)
            for g in range(len(ks)):
                c = self.code[j, cidx[g]]
                if d[g] < dist[c]:
                    dist[c] = d[g]
                    p = self.code[j, cidx[g]:cidx[g] + STEP_SZ]
                    pay[c] = -1
                    pay[c, :len(p)] = p
                    aux[c] = (j, kint[g])
        return dist, pay, aux

    def search_text_cands(self, q):
        """Per-code best text candidate (GestureKNN.py:708-721).  q: (384,) f32."""
        dist = np.full(K_CODES, ABSENT, np.float64)     # list of python floats / np.float32 mix
        pay = np.full((K_CODES, 4), -1, np.int64)
        aux = np.full((K_CODES, 2), -1, np.int64)
        grid = list(range(0, N_FRAMES - STEP_SZ * 8, 8))
        rows = [k // 8 for k in grid]
        if self.scan == "c":
            from . import cref
            d, ix = cref.text_scan(self.ctx, rows, self.code, rows, np.asarray(q, np.float32)[None])
            dd, pay, aux = self._expand(d[0].astype(np.float64), ix[0], len(grid), grid, rows)
            return dd, pay, aux
        for j in range(self.n_db_seq):
            d = cosine_rows(q, self.ctx[j, rows])       # f32
            for g, k in enumerate(grid):
                c = self.code[j, k // 8]
                if d[g] < dist[c]:
                    dist[c] = d[g]
                    pay[c] = self.code[j, k // 8:k // 8 + STEP_SZ]
                    aux[c] = (j, k)
        return dist, pay, aux

    def _expand(self, d, ix, G, ks, cidx):
        """(dist, flat index) -> the reference's (dist, 4-code payload, [j,k]) triple."""
        pay = np.full((K_CODES, 4), -1, np.int64)
        aux = np.full((K_CODES, 2), -1, np.int64)
        for c in range(K_CODES):
            if ix[c] >= 0:
                j, g = divmod(int(ix[c]), G)
                p = self.code[j, cidx[g]:cidx[g] + STEP_SZ]
                pay[c, :len(p)] = p
                aux[c] = (j, ks[g])
        return np.asarray(d, np.float64), pay, aux

    # -- phase gate -------------------------------------------------------------
    def _cand_phase(self, aux_jk):
        j, k = int(aux_jk[0]), int(aux_jk[1])
        s = phase_slot(k)
        P = self.phase[j, s:s + 32]                    # (32,4,8)
        ph, am = P[:, 0], P[:, 2]
        head = np.concatenate((ph[:8], am[:8]), axis=1)        # (8,16)
        tail = np.concatenate((ph[-8:], am[-8:]), axis=1)
        return head, tail

    @staticmethod
    def _gate_score(prev, head):
        a = np.concatenate((prev[-5:], head[:3]), axis=0).reshape(-1)
        b = np.concatenate((prev[-3:], head[:5]), axis=0).reshape(-1)
        return cosine_pair(a, b)

    # -- one 4 s window (:501-664) ------------------------------------------------
    def search_code_knn(self, clip_test, clip_context, seed_code=None, seed_phase=None,
                        use_txt=True, use_aud=True, trace=None):
        """Shipped branch (use_phase & use_aud & use_txt, :627-657) and the
        audio-only phase branch (:593-608, top-2 audio candidates).
        clip_test: callable i -> query row (so the wavlm stack is never materialised)."""
        if seed_code is None:
            code0, phase0 = self.init_code_phase()
        else:
            code0, phase0 = seed_code, seed_phase
        result = [int(code0)]
        result_phase = [np.asarray(phase0, np.float32)]
        vote = []
        n_clip = self.n_db_frm if self.mode == "wavlm" else 398
        freq_rank = self.freq_rank()
        i = 0
        while i < n_clip:
            prev = result[-1]
            diff = self.sig[prev][None] - self.sig
            pos = np.array([np.linalg.norm(diff[c]) for c in range(K_CODES)], np.float64)  # :536 (f32 norm)
            pos[prev] = np.inf                                                            # :534
            pos_score = self.rank(pos) + freq_rank * 0.05                                 # :540-545
            if use_txt:
                row = int(i / self.interp.shape[1] * 30) if self.mode == "wavlm" else int(i / 398 * 30)
                txt_d, txt_pay, txt_aux = self.search_text_cands(clip_context[row])
                comb_t = pos_score + self.rank(txt_d)
                order_t = np.argsort(comb_t)
            if use_aud:
                aud_d, aud_pay, aud_aux = self.search_audio_cands(clip_test(int(i)))
                comb_a = pos_score + self.rank(aud_d)
                order_a = np.argsort(comb_a)
            if use_aud and use_txt:
                cands = [(order_a[0], aud_pay, aud_aux), (order_t[0], txt_pay, txt_aux)]
                self.tied_decisions += int((comb_a == comb_a.min()).sum() > 1)
                self.tied_decisions += int((comb_t == comb_t.min()).sum() > 1)
            elif use_aud:
                cands = [(order_a[0], aud_pay, aud_aux), (order_a[1], aud_pay, aud_aux)]
                self.tied_decisions += int((np.sort(comb_a)[:3] == np.sort(comb_a)[1]).sum() > 1)
            else:
                cands = [(order_t[0], txt_pay, txt_aux), (order_t[1], txt_pay, txt_aux)]
                self.tied_decisions += int((np.sort(comb_t)[:3] == np.sort(comb_t)[1]).sum() > 1)
            scores, tails = [], []
            for c, _, aux in cands:
                head, tail = self._cand_phase(aux[c])
                scores.append(self._gate_score(result_phase[-1], head))
                tails.append(tail)
            fi = scores.index(min(scores))
            if scores[0] == scores[1]:
                self.tied_decisions += 1
            c, pay, _ = cands[fi]
            result.extend(int(v) for v in pay[c] if v >= 0)
            result_phase.append(tails[fi])
            vote.append(fi)
            if trace is not None:
                trace.append(dict(pos_score=pos_score, aud_d=aud_d if use_aud else None,
                                  txt_d=txt_d if use_txt else None, final_index=fi))
            i += STEP_SZ * self.step_sz
        return (np.array(result)[1:1 + N_CODE], np.array(result_phase)[1:], np.array(vote))


def wavvq_distance(a, b):
    """wavvq_distances(mode='combine') (GestureKNN.py:57-67): 22 ints -> 11 symbols -> edit distance."""
    sa = np.asarray(a).reshape(-1, 2).T
    sb = np.asarray(b).reshape(-1, 2).T
    return lev(list((sa[0] * 320 + sa[1]).astype(np.int64)), list((sb[0] * 320 + sb[1]).astype(np.int64)))


def predict_code_from_audio(knn, test_interp=None, test_vq_feat=None, test_ctx=None, n_windows=None,
                            use_txt=True, use_aud=True, trace=None):
    """Window loop (GestureKNN.py:785-813): window i>0 is seeded by the previous window's
    30th code and last phase block."""
    motion, phases, votes = [], [], []
    for w in range(n_windows):
        if knn.mode == "wavlm":
            def clip(i, w=w):
                return wavlm_feat_rows(test_interp, w, [i])[0]
        else:
            def clip(i, w=w):
                return test_vq_feat[w, i]
        seed_c = motion[-1][-1] if w > 0 else None
        seed_p = phases[-1][-1] if w > 0 else None
        m, p, v = knn.search_code_knn(clip, test_ctx[w] if test_ctx is not None else None,
                                      seed_code=seed_c, seed_phase=seed_p,
                                      use_txt=use_txt, use_aud=use_aud, trace=trace)
        motion.append(m)
        phases.append(p)
        votes.append(v)
    return np.array(motion), np.array(phases), np.array(votes)


def load_and_match(paths, max_frames=0, seed=123456, rank_kind="numpy", trace=None, scan="numpy"):
    """main_codebook (GestureKNN.py:816-845), shipped flags (:842-843)."""
    tr = np.load(paths["train_database"], allow_pickle=True)
    te = np.load(paths["test_data"], allow_pickle=True)
    code = np.load(paths["train_codebook"])["code"]
    sig = np.load(paths["codebook_signature"])["signature"]
    tr_interp = interp_wavlm(np.load(paths["train_wavlm"])["wavlm"])
    te_interp = interp_wavlm(np.load(paths["test_wavlm"])["wavlm"])
    n_win = max_frames if max_frames != 0 else np.load(paths["test_wavvq"])["wavvq"].shape[0]   # :740
    rs = np.random.RandomState(seed)                                                          # :22
    knn = CodeKNNOracle(code, sig, densify_phase(tr["phase"]), tr["context"].squeeze(2),
                        wavlm_interp=tr_interp, mode="wavlm", rng=rs, rank_kind=rank_kind, scan=scan)
    out = predict_code_from_audio(knn, test_interp=te_interp, test_ctx=te["context"].squeeze(2),
                                  n_windows=n_win, trace=trace)
    return out, knn


def load_and_match_wavvq(paths, use_txt=True, max_frames=0, seed=2, rank_kind="numpy", trace=None):
    """predict_code_from_audio with the vq-wav2vec flags (use_wavvq, use_feature, use_phase, use_aud,
    use_txt optional): Levenshtein audio distance on 11-symbol strings (GestureKNN.py:57-67, 558-560)."""
    tr = np.load(paths["train_database"], allow_pickle=True)
    te = np.load(paths["test_data"], allow_pickle=True)
    code = np.load(paths["train_codebook"])["code"]
    sig = np.load(paths["codebook_signature"])["signature"]
    tr_vq = wavvq_feat(np.load(paths["train_wavvq"])["wavvq"])
    te_wavvq = np.load(paths["test_wavvq"])["wavvq"]
    te_vq = wavvq_feat(te_wavvq)
    n_win = max_frames if max_frames != 0 else te_wavvq.shape[0]
    rs = np.random.RandomState(seed)
    knn = CodeKNNOracle(code, sig, densify_phase(tr["phase"]), tr["context"].squeeze(2),
                        wavvq_train_feat=tr_vq, mode="wavvq", rng=rs, rank_kind=rank_kind)
    out = predict_code_from_audio(knn, test_vq_feat=te_vq, test_ctx=te["context"].squeeze(2), n_windows=n_win,
                                  use_txt=use_txt, use_aud=True, trace=trace)
    return out, knn
