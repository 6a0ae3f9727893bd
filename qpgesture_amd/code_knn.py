"""Host side of the code-level motion matcher: mirrors the reference's `CodeKNN` /
`predict_code_from_audio` interface (codebook/Speech2GestureMatching/GestureKNN.py:422-813)
on top of the C ABI of libqpg_hip.so.  Python here is plumbing: it owns the device tensors,
the index tables and the call order; every distance, minimum, rank and matching step runs in
the hand-written HIP kernels.  No CPU fallback exists.

Design difference from the reference (same results): the audio and text scans depend only on
the query position, never on the running (code, phase) state, so all Q = 8*M scans of a clip are
issued as two batched sweeps, and the sequential part walks (Q,512) tables on the device.
"""
import time

import numpy as np
import torch

from . import _lib
from .parallel import allreduce_max_, allreduce_min_index, exchange_bytes, shard_rows
from .sorted_rows import SortedRows
import ctypes

from .constant import (ABSENT_DIST, NUM_AUDIO_FEAT_FRAMES, STEP_SZ, WAVVQ_GROUP_SIZE, codebook_size, num_frames,
                       num_frames_code)

MODE_AUD_TXT, MODE_AUD, MODE_TXT = 0, 1, 2
# mixed-precision audio sweep: a-priori error bound of qpg_audio_cosine_mx (QPG_AUDIO_MX_ERR of include/qpg.h) and the
# band inside which qpg_percode_select_mixed_f64 re-evaluates (two values further apart than 2 x the bound are ordered
# like the exact distances; 5 % margin on top)
AUDIO_MX_ERR = 2.05e-6
AUDIO_MX_BAND = 2.1 * AUDIO_MX_ERR
# the split-operand f16 sweep (qpg_audio_cosine_hl, QPG_AUDIO_HL_ERR): a tighter bound, a narrower band.  (Round 4's
# 32-row kernel runs the cross products through the h h' chains, cross terms first: the same budget, csrc/qpg_audio_hl.hip.)
AUDIO_HL_ERR = 1.3e-6
AUDIO_HL_BAND = 2.1 * AUDIO_HL_ERR

# bits of the trouble word the sweeps / selects raise (stats[1] of include/qpg.h) and the walk carries out with the codes
FLAG_LIST_OVERFLOW, FLAG_SMALL_NORMS, FLAG_REQUEST_OVERFLOW, FLAG_CROSS_SHARD_TIE = 1, 2, 4, 8
# the text prefilter's band list overflowed (a zero-norm context query, thousands of repeated embeddings): only the TEXT side
# has to run again, on the exact-order sweep - the audio tables of the clip stand
FLAG_TEXT_OVERFLOW = 16


_PIN_SENTINEL = -1234567          # never a status word (flag bits are small non-negative integers)


def _wait_pinned(pin_np, stream, watch=None):
    """Host side of the zero-copy results: spin on the status word (the walk's last store, behind a system-scope fence),
    then make sure no other word still holds the sentinel - a store that the fabric delivered late is waited for, never
    copied as it is (codes / votes / flags can not take the sentinel's value) - and return a copy.  After ~2 ms without the
    word (a failed launch would never write it) the stream is synchronised the ordinary way.
    watch: the words to spin on (a view of pin_np: every clip's status word when one replay walks several clips - each
    chain's block writes its own pair last); default the buffer's last word."""
    if watch is None:
        watch = pin_np[-1:]
    for _ in range(40000):
        if watch[-1] != _PIN_SENTINEL and (watch.size == 1 or not (watch == _PIN_SENTINEL).any()):
            break
    else:
        stream.synchronize()
        if (watch == _PIN_SENTINEL).any():
            raise RuntimeError("the walk did not write its status word")
    out = pin_np.copy()
    if (out == _PIN_SENTINEL).any():
        stream.synchronize()
        out = pin_np.copy()
        if (out == _PIN_SENTINEL).any():
            raise RuntimeError("the walk left result words unwritten")
    return out


class GuardOverflow(RuntimeError):
    """The capped near-tie machinery of the fast audio paths could not guarantee the reference's candidates for this
    clip (a re-evaluation list overflowed, operand norms left the error bound's range, or shard minima tied across the
    exchange).  `flags` holds the FLAG_* bits.  CodeKNN.match_clip / ClipPipeline.collect / the CLI catch it and
    re-match the clip on the uncapped path (audio_precision "exact"); codes of a flagged clip are never returned."""

    def __init__(self, flags):
        super().__init__("near-tie guard raised flags 0x%x: results of the capped path are not guaranteed" % flags)
        self.flags = int(flags)


# ----------------------------------------------------------------------------------------------
# index grids — literal restatement of the reference's float loops; tiny, host-side, done once
# ----------------------------------------------------------------------------------------------
def audio_grid(n_db_frm, step_sz):
    """Grid of search_audio_cands (GestureKNN.py:672-690): k = 0, step, 2*step, ... while
    k < n_db_frm - 4*step, with step_sz possibly a float (398/30 in wavvq mode) accumulated
    sequentially.  Returns int(k) and int(k/step_sz) per grid position."""
    kint, cidx = [], []
    k = 0
    while k < n_db_frm - STEP_SZ * step_sz:
        kint.append(int(k))
        cidx.append(int(k / step_sz))
        k += step_sz
    return kint, cidx


def text_grid():
    """Grid of search_text_cands (GestureKNN.py:713-720): k = 0,8,..,200; row/code column k//8."""
    ks = list(range(0, num_frames - STEP_SZ * 8, 8))
    return ks, [k // 8 for k in ks]


def phase_slot(k):
    """Phase start frame of a candidate: int(k/398*240) whatever unit k is in (GestureKNN.py:632)."""
    return int(k / 398 * 240)


def wavvq_tap_offsets(T):
    """Frame offsets of the 11 taps of a wavvq feature row (data_processing.py:281, 297-317):
    6 backward taps shifted by int((5-i)*s), then 5 forward taps shifted by int(i*s), s = T/30."""
    s = T / num_frames_code
    return [-int((NUM_AUDIO_FEAT_FRAMES - i - 1) * s) for i in range(NUM_AUDIO_FEAT_FRAMES)] + \
           [int(i * s) for i in range(1, NUM_AUDIO_FEAT_FRAMES)]


def _i32(x, dev):
    if isinstance(x, torch.Tensor):
        return x
    return torch.as_tensor(np.asarray(x, np.int32), device=dev)


class ExchangeLayout:
    """Byte layout of the (minimum, index) tables a rank contributes to the cross-shard exchange: `nblk` blocks (one
    per destination rank for the owner-partitioned all-to-all, one in all for the all-gather) of `Qb` query rows,
    each block = [aud_d f64 | aud_i i32 | txt_d f32 | txt_i i32] (the modalities in use), every array Qb*K entries,
    then one 8-byte slot whose first i32 is the sender's TROUBLE WORD (qpg_flags_stamp: the bits travel with the tables
    instead of in their own all-reduce).
    The select kernels write straight into it (no packing pass); qpg_merge_select_* reads the received copy.
    A layout (and its send buffer) is built once per shape and kept by the matcher (CodeKNN._layout)."""

    def __init__(self, Qtot, K, nblk, parts, audio_f64, device):
        assert Qtot % nblk == 0, "query rows must split evenly over the ranks"
        self.Qb, self.K, self.nblk, self.parts = Qtot // nblk, K, nblk, list(parts)
        n = self.Qb * K
        self.off, o = {}, 0
        for p in self.parts:
            dsz = 8 if (p == "aud" and audio_f64) else 4
            self.off[p + "_d"], o = o, o + n * dsz
            self.off[p + "_i"], o = o, o + n * 4
            o = (o + 7) // 8 * 8
        self.off["flags"], o = o, o + 8
        self.block_bytes = o
        self.dtype = {p: (torch.float64 if (p == "aud" and audio_f64) else torch.float32) for p in self.parts}
        self.send = torch.zeros((nblk * self.block_bytes,), dtype=torch.uint8, device=device)

    def views(self, p):
        """(dist, idx) tensors aliasing block 0's arrays of modality p + the select kernel's layout arguments."""
        n = self.Qb * self.K
        dsz = 8 if self.dtype[p] == torch.float64 else 4
        d = self.send[self.off[p + "_d"]:self.off[p + "_d"] + n * dsz].view(self.dtype[p])
        i = self.send[self.off[p + "_i"]:self.off[p + "_i"] + n * 4].view(torch.int32)
        return d, i, (self.Qb if self.nblk > 1 else 0), (self.block_bytes if self.nblk > 1 else 0)


class GestureDB:
    """A speaker database resident in HBM (what load_db_codebook + CodeKNN.__init__ build).

    Layout:
      base   f32 [n_local][180][F]   interpolated WavLM frames of this rank's row shard
      cn2    f64 [n_local][26]       squared norm of each audio candidate (6 frames)
      ctxt   f32 [C/64][96][64][4]   text candidates (26 grid rows per window), sklearn-normalised, tiled
      code   i32 [N][30]             replicated (payloads of winners are looked up from it)
      phase  f32 [N][240][2][8]      replicated (phase shift, amplitude)
      pos_rank i16 [512][512], freq_rank i16 [512]
    Rows [lo, hi) of the N DB windows live on this rank; candidate indices are global.
    """

    def __init__(self, code, wavlm_interp, context, phase_dense, signature, device="cuda:0",
                 freq_rank=None, pos_rank=None, rank=0, world=1, wavvq=None, feature_dtype="f32", hl_image=True,
                 text_prefilter=True):
        dev = torch.device(device)
        if feature_dtype not in ("f32", "f16"):
            raise ValueError("feature_dtype must be 'f32' or 'f16'")
        self.feature_dtype = feature_dtype
        if dev.type != "cuda":
            raise RuntimeError("GestureDB needs a HIP device (got %s); there is no CPU path" % dev)
        _lib.load()
        self.device = dev
        self.rank, self.world = rank, world
        code = np.asarray(code)
        self.N = N = code.shape[0]
        self.lo, self.hi = shard_rows(N, rank, world)
        self.n_local = self.hi - self.lo
        self.K = codebook_size
        self.code_host = code.astype(np.int64)
        self.code = _i32(code, dev).contiguous()
        self.code_local = self.code[self.lo:self.hi].contiguous()

        self.T, self.F = wavlm_interp.shape[1], wavlm_interp.shape[2]
        if isinstance(wavlm_interp, torch.Tensor):        # already on the device (interp_wavlm_device)
            self.base = wavlm_interp[self.lo:self.hi].to(dev, torch.float32).contiguous()
        else:
            self.base = torch.from_numpy(np.ascontiguousarray(wavlm_interp[self.lo:self.hi], np.float32)).to(dev)
        self.step_sz = self.T // num_frames_code                      # GestureKNN.py:432
        kint, cidx = audio_grid(self.T, self.step_sz)
        self.aud_k, self.aud_cidx_host = kint, cidx
        self.Ga = len(kint)
        self.aud_t = _i32(kint, dev)
        self.aud_cidx = _i32(cidx, dev)
        self.aud_pslot = _i32([phase_slot(k) for k in kint], dev)
        self.tap_stride = 2                                           # FRAME_INTERVAL-2, data_processing.py:266
        # code of every local candidate c = j*G + g in scan order (i16): what the one-launch select kernels index
        local = code[self.lo:self.hi]
        self.aud_cand_code = self._cand_code(local, cidx, dev)

        # vq-wav2vec track (optional; the mode the paper describes): symbols g1*320+g2, float grid of 398/30
        self.has_wavvq = wavvq is not None
        if self.has_wavvq:
            vq = np.asarray(wavvq[self.lo:self.hi])
            self.Tv = wavvq.shape[1]
            self.vq_sym = _i32(vq[..., 0].astype(np.int64) * WAVVQ_GROUP_SIZE + vq[..., 1], dev).contiguous()
            self.vq_step = self.Tv / num_frames_code                              # GestureKNN.py:436
            vk, vc = audio_grid(self.Tv, self.vq_step)
            self.vq_k, self.vq_cidx_host = vk, vc
            self.Gv = len(vk)
            self.vq_t = _i32(vk, dev)
            self.vq_cidx = _i32(vc, dev)
            self.vq_pslot = _i32([phase_slot(k) for k in vk], dev)
            self.vq_taps = wavvq_tap_offsets(self.Tv)
            self.vq_cand_code = self._cand_code(local, vc, dev)

        ks, rows = text_grid()
        self.txt_k, self.txt_rows_host = ks, rows
        self.Gt = len(ks)
        self.txt_r = _i32(rows, dev)
        self.txt_cidx = self.txt_r
        self.txt_pslot = _i32([phase_slot(k) for k in ks], dev)
        self.txt_cand_code = self._cand_code(local, rows, dev)

        # per-candidate squared norms (f64) without materialising the 6144-d windows
        fn2 = torch.empty((self.n_local, self.T), dtype=torch.float64, device=dev)
        self.cn2 = torch.empty((self.n_local, self.Ga), dtype=torch.float64, device=dev)
        if feature_dtype == "f16":
            # f16 storage of the dominant array (half the HBM bytes); norms are those of the ROUNDED values, which the
            # sweep widens in registers (qpg_audio_cosine_f64_h): the distances are the reference's on the rounded track
            self.base = self.base.to(torch.float16).contiguous()
        if self.n_local:
            src = self.base if feature_dtype == "f32" else self.base.float()
            _lib.call("qpg_frame_norm2_f64", dev, src, self.n_local * self.T, self.F, fn2)
            del src
            _lib.call("qpg_audio_cand_norm2", dev, fn2, self.n_local, self.T, self.aud_t, self.Ga,
                      NUM_AUDIO_FEAT_FRAMES, self.tap_stride, self.cn2)

        # split-operand f16 image of the track for the HBM-bound sweep (qpg_audio_cosine_hl): every frame once, in MFMA
        # fragment order; built when the grid has the reference's shape (6 taps 2 frames apart, 26 positions 6 apart)
        self.hl_image = None
        lib = _lib.load()
        # the bounded (split-f16) paths rest on one measured property of the matrix core: re-measured once per process
        # and device (selfcheck.mfma_bound_ok); a device that fails it gets the f64 sweep and the exact-order text sweep
        self.hl_bound_ok, self.hl_bound_report = (True, {"skipped": True})
        if (hl_image or text_prefilter) and self.n_local:
            from .selfcheck import mfma_bound_ok
            self.hl_bound_ok, self.hl_bound_report = mfma_bound_ok(dev)
            if not self.hl_bound_ok:
                hl_image = text_prefilter = False
        # (feature_dtype "f16", round 5: an f16 value is its own h plane - a ONE-plane image, half the bytes, two products
        # per element instead of three: qpg_audio_cosine_hl1)
        self.hl_planes = 2 if feature_dtype == "f32" else 1
        supported = lib.qpg_audio_hl_supported if feature_dtype == "f32" else lib.qpg_audio_hl1_supported
        if (hl_image and self.n_local and len(kint) > 1 and
                kint == [i * (kint[1] - kint[0]) for i in range(len(kint))] and
                supported(self.T, self.F, self.Ga, NUM_AUDIO_FEAT_FRAMES, self.tap_stride, kint[1] - kint[0])):
            if feature_dtype == "f32":
                nb = int(lib.qpg_audio_hl_db_bytes(self.n_local, self.F))
                self.hl_image = torch.empty((nb,), dtype=torch.uint8, device=dev)
                _lib.call("qpg_audio_hl_pack_db", dev, self.base, self.n_local, self.T, self.F, self.Ga,
                          NUM_AUDIO_FEAT_FRAMES, self.tap_stride, kint[1] - kint[0], self.hl_image, nb)
            else:
                nb = int(lib.qpg_audio_hl1_db_bytes(self.n_local, self.F))
                self.hl_image = torch.empty((nb,), dtype=torch.uint8, device=dev)
                _lib.call("qpg_audio_hl1_pack_db", dev, self.base, self.n_local, self.T, self.F, self.Ga,
                          NUM_AUDIO_FEAT_FRAMES, self.tap_stride, kint[1] - kint[0], self.hl_image, nb)

        ctx = np.ascontiguousarray(context[self.lo:self.hi], np.float32)
        self.R, self.Dt = context.shape[1], context.shape[2]
        ctx_d = torch.from_numpy(ctx).to(dev)
        self.Ct = self.n_local * self.Gt
        # normalised grid rows, tiled [C/64][Dt/4][64][4] for lane-per-candidate access
        self.ctxt = torch.zeros((((self.Ct + 63) // 64) * 64 * self.Dt,), dtype=torch.float32, device=dev)
        if self.n_local:
            _lib.call("qpg_text_pack_candidates_f32", dev, ctx_d, self.n_local, self.R, self.Dt, self.txt_r,
                      self.Gt, self.ctxt)
        # text candidates sorted by code + their split-f16 image for the bounded prefilter (sorted_rows.SortedRows,
        # csrc/qpg_sorted.hip): rows normalised by the kernel the exact sweep's candidates are normalised by
        self.txt_sorted = None
        if text_prefilter and self.n_local and self.Dt % 128 == 0 and self.K < 0x2000:
            rows_f = ctx_d[:, torch.as_tensor(np.asarray(rows, np.int64), device=dev), :].reshape(self.Ct, self.Dt)
            rows_n = torch.empty_like(rows_f)
            _lib.call("qpg_l2_normalize_rows_f32", dev, rows_f.contiguous(), self.Ct, self.Dt, rows_n)
            self.txt_sorted = SortedRows(rows_n, self.txt_cand_code[:self.Ct], self.K, dev)
            # batches of >= 256 text queries (six clips or more per sweep) take the h-plane prefilter + the by-code select
            # (round 5: every row is then wanted by several queries); a single clip's 48 queries stay on the by-query path
            self.txt_sorted.by_code = True
            del rows_f, rows_n
        del ctx_d

        ph = np.ascontiguousarray(np.asarray(phase_dense, np.float32)[:, :, [0, 2], :])
        self.Tp = ph.shape[1]
        self.phase = torch.from_numpy(ph).to(dev)
        self.phase_host = ph

        sig = torch.from_numpy(np.ascontiguousarray(signature, np.float32)).to(dev)
        self.signature = sig
        if pos_rank is None:
            pd = torch.empty((self.K, self.K), dtype=torch.float32, device=dev)
            _lib.call("qpg_l2_table_f32", dev, sig, self.K, sig.shape[1], pd)
            self.pos_dist = pd
            self.pos_rank = torch.empty((self.K, self.K), dtype=torch.int16, device=dev)
            _lib.call("qpg_rank_rows_f32", dev, pd, self.K, self.K, self.pos_rank)
        else:
            self.pos_rank = torch.as_tensor(np.asarray(pos_rank, np.int16), device=dev).contiguous()

        # code_to_freq (GestureKNN.py:481-499): 1 - count/total, 1 for unseen codes; rank of it (:544)
        cnt = np.bincount(code.reshape(-1), minlength=self.K)[:self.K]
        self.freq_dist = np.where(cnt > 0, 1 - cnt / cnt.sum(), 1.0)
        # (the walk-relevance cut of the select reads a code's column: qpg_percode_select_mixed_f64_cut)
        self.pos_rank_t = self.pos_rank.t().contiguous()
        if freq_rank is None:
            fd = torch.from_numpy(self.freq_dist[None].copy()).to(dev)
            self.freq_rank = torch.empty((1, self.K), dtype=torch.int16, device=dev)
            _lib.call("qpg_rank_rows_f64", dev, fd, 1, self.K, self.freq_rank)
            self.freq_rank = self.freq_rank[0].contiguous()
        else:
            self.freq_rank = torch.as_tensor(np.asarray(freq_rank, np.int16), device=dev).contiguous()

    # -- prepared-database cache (round 5; db_cache.py): what the constructor built, written once, restored without a
    #    single build launch - the drop-in CLI's second and later invocations (GestureKNN.py:816-845 reloads everything)
    def save(self, path, key="", sources=None):
        from . import db_cache
        return db_cache.save(self, path, key, sources=sources)

    @classmethod
    def load(cls, path, device="cuda:0", key=None):
        """The GestureDB saved at `path`, or None (missing / foreign / keyed differently).  The bounded paths' one
        measured hardware property is re-measured on THIS device (selfcheck.mfma_bound_ok): a device that fails it keeps
        the restored object but not its split-f16 images."""
        from . import db_cache
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("GestureDB needs a HIP device (got %s); there is no CPU path" % dev)
        _lib.load()
        db = db_cache.load(path, dev, key)
        if db is None:
            return None
        if db.hl_image is not None or db.txt_sorted is not None:
            from .selfcheck import mfma_bound_ok
            db.hl_bound_ok, db.hl_bound_report = mfma_bound_ok(dev)
            if not db.hl_bound_ok:
                db.hl_image = db.txt_sorted = None
        return db

    @staticmethod
    def _cand_code(code_local, cidx, dev):
        cc = np.ascontiguousarray(code_local[:, np.asarray(cidx, np.int64)].reshape(-1))
        cc = np.where((cc >= 0) & (cc < 32767), cc, -1).astype(np.int16)        # out-of-range ids are skipped
        return torch.from_numpy(cc if cc.size else np.zeros((1,), np.int16)).to(dev)

    @property
    def idx_base(self):
        return self.lo


class CodeKNN:
    """Mirror of the reference's CodeKNN (GestureKNN.py:422-721) over a GestureDB."""

    def __init__(self, db, use_wavlm=True, use_wavvq=False, use_phase=True, use_txt=True, rng=None):
        if use_wavlm == use_wavvq:
            raise ValueError("exactly one of use_wavlm / use_wavvq (GestureKNN.py:431-438)")
        if use_wavvq and not db.has_wavvq:
            raise ValueError("GestureDB was built without a wavvq track")
        self.db = db
        self.use_wavvq = use_wavvq
        if use_wavvq:                                                       # GestureKNN.py:435-438
            self.step_sz, self.n_db_frm = db.vq_step, db.Tv
        else:                                                               # :431-434
            self.step_sz, self.n_db_frm = db.step_sz, db.T
        self.n_db_seq = db.N
        self.use_phase, self.use_txt = use_phase, use_txt
        self.rng = rng if rng is not None else np.random
        self.overlap_sweeps = True          # text sweep on a second HIP stream underneath the audio sweep
        self.text_after_sweep = True        # ... started when the audio sweep ends, i.e. underneath the audio SELECT ...
        self.audio_first = None             # no ordering between the two streams; None: auto (sweep_tables)
        self.serial_walk = False            # True: force the one-wave sequential walk (tests compare the two)
        # Near-tie guard of the audio select (qpg_percode_select_guarded_f64): candidates / code minima closer than
        # tie_eps are re-evaluated in the reference's own arithmetic inside the select launch.  0 disables it.
        self.tie_eps = 1e-12
        self._guard_stats = torch.zeros((4,), dtype=torch.int32, device=db.device)
        # audio_precision "mixed" (default): the sweep runs on the f32 matrix cores with an a-priori error bound
        # (qpg_audio_cosine_mx, |error| <= AUDIO_MX_ERR) and the select re-evaluates every comparison the bound leaves
        # open with f64 dot products, then the near-tie guard (qpg_percode_select_mixed_f64): same candidates and ranks
        # as "f64", the sweep at twice the matrix rate.  Taken only where the select sees every comparison that
        # follows (single-GPU DB, ranks fused, guard on; f32 or f16 base); everything else runs the f64 sweep.
        # "exact" (round 3): the f64 sweep + the UNCAPPED guard (qpg_percode_select_exact_f64; across shards a
        # reference-arithmetic request / response round): no list that can overflow, whatever the data.  It is the path
        # a clip is re-matched on when the faster ones raise their trouble word (GuardOverflow), and can be selected
        # outright.  fallbacks counts the clips that took it that way.
        self.audio_precision = "mixed"
        # kernel of the mixed-precision sweep: "hl" = split-operand f16 matrix cores on the frame-major image (HBM-bound;
        # needs GestureDB.hl_image), "mx" = the f32 matrix cores on the f32 / f16 base.  Same bound, same select.
        self.audio_kernel = "hl"
        # text_kernel "mfma" (round 3, default where the DB holds the sorted image: one GPU): bounded split-f16 prefilter
        # over the candidates sorted by code + exact sklearn-order evaluation of every code's band (bit-identical tables;
        # an overflowing band list raises the trouble word and the clip is re-matched on the exact sweep); "valu": the
        # exact-order sweep of every pair (qpg_text_cosine_f32).
        self.text_kernel = "mfma"
        # column image / prefilter matrix / tile minima of the text prefilter: per matcher (= per stream), never on the
        # shared GestureDB.txt_sorted - the lanes of a ClipPipeline run their text sides concurrently
        self._txt_scratch = {}
        self.fallbacks = 0
        # a clip's audio AND text query packs in one launch (qpg_clip_pack_hl; False / QPG_FUSED_PACK=0: round 3's separate
        # launches - tests and measurements compare the two)
        import os as _os
        self.fused_pack = _os.environ.get("QPG_FUSED_PACK", "1") != "0"
        # Walk-relevance cut (round 4, last hours): when the tables go straight into the walk (match_clip without
        # return_tables, ClipGraph, ClipPipeline, bench.py's step: sweep_tables(for_walk=True)) the select settles in f64
        # only what the walk can read - codes whose rank is certainly above every step's winning fused score keep their
        # sweep values (include/qpg.h: qpg_percode_select_mixed_f64_cut).  The codes the walk returns are the same; the
        # tables sweep_tables() hands to anyone else are exact everywhere, as before.  QPG_RANK_CUT=0: off.
        self.rank_cut = _os.environ.get("QPG_RANK_CUT", "1") != "0"
        self.rank_cut_probe = 64
        self.mixed_single_launch = False    # True: the select's tier-1 work stays inside one launch (tests compare both)
        self.split_fuse = True              # the rank fusion per modality behind its select (False: one launch in the walk)
        self.split_fuse_max_steps = 256     # ... for sweeps of at most this many matching steps
        # Row-sharded DB: the shards sweep in mixed precision too and the cross-shard merge re-evaluates what their
        # bounded tables leave open through a request / response exchange (sweep_tables; qpg_merge_mixed_*).
        # Two more (small) exchanges buy a sweep at ~1.6x the rate, so it is used where the shard's sweep is long enough
        # to pay for them: at least sharded_mixed_min_gflop per rank and step (24 s clip x 2048 windows = 31 GFLOP).
        self.sharded_mixed = True
        # force_sharded: take the row-shard code path (exchange layout, collectives, merge kernels) even with world == 1,
        # so that a one-GPU box can execute it over RCCL (backend nccl, world_size 1): tests / bench only
        self.force_sharded = False
        self.sharded_mixed_min_gflop = 20.0
        self.mixed_requests = None          # request slots per (owner, shard) pair and step; None: 16384 / world
        # host_ranks (the CLI's --tie_rule numpy): rank the (Q,512) audio / text minima with the reference's own
        # `np.array(x).argsort().argsort()` on the host, so that EXACT ties between codes (structural in real text
        # embeddings: silent frames share one vector) get NumPy's unstable-sort order like the reference's.
        self.host_ranks = False

    def _audio_grid(self):
        db = self.db
        if self.use_wavvq:
            return db.vq_cidx, db.vq_pslot, db.Gv
        return db.aud_cidx, db.aud_pslot, db.Ga

    def query_positions(self):
        """Matching-step positions of a window: i = 0, 4*step, ... while i < n (GestureKNN.py:528,659);
        with the float wavvq step the literal accumulation is kept."""
        key = (self.n_db_frm, self.step_sz)
        hit = self.__dict__.get("_qpos")
        if hit is not None and hit[0] == key:          # (twice per clip, in front of its first launch)
            return hit[1]
        pos, i = [], 0
        while i < self.n_db_frm:
            pos.append(i)
            i += STEP_SZ * self.step_sz
        self.__dict__["_qpos"] = (key, pos)
        return pos

    # -- init (GestureKNN.py:462-473): same two draws from the (seeded) NumPy global stream ------
    def init_code_phase(self):
        db = self.db
        i = self.rng.randint(0, self.n_db_seq)
        j = self.rng.randint(0, self.n_db_frm - int(num_frames / num_frames_code))
        code = int(db.code_host[i, j // num_frames_code])
        P = db.phase_host[i, j:j + 8]                              # (8,2,8)
        if P.shape[0] != 8:
            # wavvq mode draws j up to 389 on a 240-frame phase track (GestureKNN.py:464-469): the reference's
            # np.array(result_phase) then raises "inhomogeneous shape" on NumPy >= 1.24
            raise ValueError("init_code_phase drew frame %d: phase slice has %d rows, not 8 "
                             "(the reference fails on this draw too); use another seed" % (j, P.shape[0]))
        return code, np.concatenate((P[:, 0], P[:, 1]), axis=1).astype(np.float32)

    # -- batched sweeps ------------------------------------------------------------------------
    def _hl_plan(self, sharded=False, Q=0):
        """Will sweep_audio take the split-f16 (hl) mixed-precision path for a whole-clip sweep of Q queries on this DB
        (the same conditions as sweep_audio's own; `sharded`: as a row shard inside sweep_tables)?"""
        db = self.db
        base = (self.audio_precision == "mixed" and self.tie_eps > 0 and db.n_local > 0 and db.K <= 512 and
                db.hl_bound_ok and self.audio_kernel == "hl" and db.hl_image is not None)
        if not sharded:
            return base and db.world == 1
        gflop = 2e-9 * Q * (-(-db.N // db.world) * db.Ga) * NUM_AUDIO_FEAT_FRAMES * db.F
        return base and self.sharded_mixed and gflop >= self.sharded_mixed_min_gflop

    def sweep_audio(self, qbase, q_win, q_t, tap_stride=None, want_rank=False, reduce=True, out=None, prepacked=None,
                    cut_top_n=None):
        """Per-code best audio candidate for every query: returns (dist f64 [Q,512], idx i32 [Q,512])
        with global candidate indices j*26+g (-1 = code absent), min-reduced across ranks; with
        want_rank also the stable ranks i16 [Q,512]."""
        db, dev = self.db, self.db.device
        Q = int(q_win.shape[0]) if isinstance(q_win, torch.Tensor) else len(q_win)
        qbase = qbase.contiguous()
        M, T, F = qbase.shape
        ts = db.tap_stride if tap_stride is None else tap_stride
        if prepacked is not None:           # (sweep_tables packed the clip's whole query side in one launch)
            q32, qn2 = prepacked[0], prepacked[1]
        else:
            q32 = torch.empty((Q, NUM_AUDIO_FEAT_FRAMES * F), dtype=torch.float32, device=dev)
            qn2 = torch.empty((Q,), dtype=torch.float64, device=dev)
        C = db.n_local * db.Ga
        fused_rank = want_rank and db.world == 1
        half = db.feature_dtype == "f16"
        local_final = fused_rank and reduce and out is None               # one GPU: this select decides everything
        shard_part = (db.world > 1 or self.force_sharded) and not reduce and out is not None   # row shard: sweep_tables merges
        # (the same number on every rank — the largest shard's — so that all ranks take the same path: the mixed merge
        # has two more collectives than the f64 one)
        gflop = 2e-9 * Q * (-(-db.N // db.world) * db.Ga) * NUM_AUDIO_FEAT_FRAMES * db.F
        mixed = (self.audio_precision == "mixed" and self.tie_eps > 0 and C > 0 and db.K <= 512 and db.hl_bound_ok and
                 (local_final or (shard_part and self.sharded_mixed and gflop >= self.sharded_mixed_min_gflop)))
        exact = self.audio_precision == "exact" and self.tie_eps > 0 and C > 0
        self._last_audio_mixed, self._last_audio_exact = mixed, exact
        self._last_rank_cut = False
        # the mixed-precision sweep stores its matrix in f32: it only feeds the select's two streaming passes
        D = torch.empty((Q, max(C, 1)), dtype=torch.float32 if mixed else torch.float64, device=dev)
        ev = getattr(self, "kernel_events", None)       # bench.py: HIP events around the dominant kernel
        every = getattr(self, "kernel_events_every", 1)  # ... of every n-th call
        if ev is not None and every > 1:
            self._ev_calls = getattr(self, "_ev_calls", 0) + 1
            if self._ev_calls % every:
                ev = None
        if ev is not None:
            pool = getattr(self, "kernel_event_pool", None)      # events created ahead of the timed region
            e0, e1 = pool.pop() if pool else (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        use_hl = mixed and self.audio_kernel == "hl" and db.hl_image is not None
        hl_fn = "qpg_audio_cosine_hl" if db.hl_planes == 2 else "qpg_audio_cosine_hl1"
        self._last_audio_hl = use_hl
        sweep_launch = None
        if use_hl:                    # gather + norms + split-f16 image in ONE launch
            nbq = int(_lib.load().qpg_audio_hl_query_bytes(Q, db.F))
            qi = self.__dict__.get("_hl_qimage")
            if qi is None or qi.numel() < nbq:
                qi = self._hl_qimage = torch.empty((nbq,), dtype=torch.uint8, device=dev)
            # the sweep's arguments are converted BEFORE the pack goes out: its launch follows the pack's at once
            sweep_launch = _lib.prepare(hl_fn, dev, db.hl_image, db.n_local, db.F, db.Ga, db.cn2, qi,
                                        qn2, Q, D, 1, D.stride(0), self._guard_stats)
            if prepacked is None:
                _lib.call("qpg_audio_pack_queries_hl", dev, qbase, M, T, F, _i32(q_win, dev), _i32(q_t, dev), Q,
                          NUM_AUDIO_FEAT_FRAMES, ts, q32, qn2, qi, qi.numel())
        else:
            assert prepacked is None, "the one-launch clip pack feeds the split-f16 sweep only"
            _lib.call("qpg_audio_pack_queries", dev, qbase, M, T, F, _i32(q_win, dev), _i32(q_t, dev), Q,
                      NUM_AUDIO_FEAT_FRAMES, ts, q32, qn2)
        if ev is not None:
            e0.record(torch.cuda.current_stream(dev))          # (the events bracket the sweep kernel alone)
        if sweep_launch is not None:
            sweep_launch()
        elif mixed:
            _lib.call("qpg_audio_cosine_mx_h" if half else "qpg_audio_cosine_mx", dev, db.base, db.n_local, db.T, db.F, db.aud_t, db.Ga,
                      NUM_AUDIO_FEAT_FRAMES, db.tap_stride, db.cn2, q32, qn2, Q, D, 1, D.stride(0), self._guard_stats)
        else:
            _lib.call("qpg_audio_cosine_f64" if db.feature_dtype == "f32" else "qpg_audio_cosine_f64_h", dev, db.base,
                      db.n_local, db.T, db.F, db.aud_t, db.Ga, NUM_AUDIO_FEAT_FRAMES, db.tap_stride, db.cn2, q32, qn2,
                      Q, D, D.stride(0))
        if ev is not None:
            e1.record(torch.cuda.current_stream(dev))
            ev.append((e0, e1))
        hook = self.__dict__.get("after_sweep")          # (ClipGraph: the encode leg forks here, behind the sweep kernel)
        if hook is not None:
            hook()
        if getattr(self, "_want_sweep_event", False):    # sweep_tables: the text side starts behind the sweep
            self._record_sweep_event(dev)
        if out is not None:          # exchange layout of the sharded path: written in place, merged after the collective
            dist, idx, qb, bs = out
        else:
            dist = torch.empty((Q, db.K), dtype=torch.float64, device=dev)
            idx = torch.empty((Q, db.K), dtype=torch.int32, device=dev)
            qb = bs = 0
        rank = torch.empty((Q, db.K), dtype=torch.int16, device=dev) if fused_rank else None
        if mixed:
            need = int(_lib.load().qpg_percode_select_mixed_ws_bytes(Q, db.K))
            ws = getattr(self, "_mix_ws", None)
            if ws is None or ws.numel() < need:
                # (zero-filled ONCE: the select's streamed state is all-zero between launches, qpg.h)
                ws = self._mix_ws = torch.zeros((need,), dtype=torch.uint8, device=dev)
            self._last_mix_Q = Q
            sel_args = (D, 1, D.stride(0), Q, db.aud_cand_code, C, db.K,
                        float(ABSENT_DIST), db.idx_base * db.Ga, dist, idx, rank, qb, bs, db.base, db.T, db.F, db.aud_t,
                        db.Ga, NUM_AUDIO_FEAT_FRAMES, db.tap_stride, q32, qn2, db.cn2,
                        AUDIO_HL_BAND if use_hl else AUDIO_MX_BAND, float(self.tie_eps),
                        self._guard_stats, None if self.mixed_single_launch else ws,
                        0 if self.mixed_single_launch else ws.numel(), int(half))
            use_cut = cut_top_n in (1, 2) and local_final and rank is not None and not self.mixed_single_launch
            self._last_rank_cut = use_cut
            try:
                if use_cut:
                    _lib.call("qpg_percode_select_mixed_f64_cut", dev, *sel_args, db.pos_rank_t, db.freq_rank,
                              int(cut_top_n), int(self.rank_cut_probe))
                else:
                    _lib.call("qpg_percode_select_mixed_f64", dev, *sel_args)
            except Exception:
                # a failed launch between the streaming pass and the list pass would leave streamed state behind that
                # later clips consume silently (the kernels only restore the all-zero state when all of them ran)
                self._mix_ws = None
                raise
        elif exact:
            need = int(_lib.load().qpg_percode_select_exact_ws_bytes(Q, C, db.K))
            ws = getattr(self, "_exact_ws", None)
            if ws is None or ws.numel() < need:
                ws = self._exact_ws = torch.empty((need,), dtype=torch.uint8, device=dev)
            _lib.call("qpg_percode_select_exact_f64", dev, D, D.stride(0), Q, db.aud_cand_code, C, db.K,
                      float(ABSENT_DIST), db.idx_base * db.Ga, dist, idx, rank, qb, bs, db.base, db.T, db.F, db.aud_t,
                      db.Ga, NUM_AUDIO_FEAT_FRAMES, db.tap_stride, q32, float(self.tie_eps), self._guard_stats, int(half),
                      ws, ws.numel())
        elif self.tie_eps > 0 and C > 0:
            _lib.call("qpg_percode_select_guarded_f64", dev, D, D.stride(0), Q, db.aud_cand_code, C, db.K,
                      float(ABSENT_DIST), db.idx_base * db.Ga, dist, idx, rank, qb, bs, db.base, db.T, db.F, db.aud_t,
                      db.Ga, NUM_AUDIO_FEAT_FRAMES, db.tap_stride, q32, float(self.tie_eps), self._guard_stats, int(half))
        else:
            _lib.call("qpg_percode_select_f64", dev, D, D.stride(0), Q, db.aud_cand_code, C, db.K, float(ABSENT_DIST),
                      db.idx_base * db.Ga, dist, idx, rank, qb, bs)
        self._last_D_aud = D
        self._last_q32, self._last_qn2 = q32, qn2          # the sharded mixed merge re-evaluates requested pairs from these
        if not reduce:              # sharded caller combines several tables in one exchange (sweep_tables)
            return dist, idx
        dist, idx = self._reduce_min(dist, idx)
        if want_rank:
            return dist, idx, (rank if fused_rank else self.rank_rows(dist))
        return dist, idx

    def _record_sweep_event(self, dev):
        if self.__dict__.get("_sweep_event") is None:
            self._sweep_event = torch.cuda.Event()                      # one event, re-recorded by every clip
        self._sweep_done = self._sweep_event
        self._sweep_done.record(torch.cuda.current_stream(dev))

    def sweep_text(self, queries, want_rank=False, reduce=True, normalised=False, out=None, cols_packed=False):
        """queries: f32 [Q,384] on the device (already sklearn-normalised if `normalised`).
        Returns (dist f32 [Q,512], idx i32 [Q,512][, rank])."""
        db, dev = self.db, self.db.device
        Q = queries.shape[0]
        if normalised:
            qn = queries
        else:
            qn = torch.empty_like(queries)
            _lib.call("qpg_l2_normalize_rows_f32", dev, queries, Q, db.Dt, qn)
        self._last_text_mfma = False
        if (self.text_kernel == "mfma" and db.txt_sorted is not None and Q > 0 and self.audio_precision != "exact" and
                ((out is None and reduce and db.world == 1) or (out is not None and not reduce))):
            self._last_text_mfma = True
            if out is not None:           # row shard: straight into the exchange buffer, global indices, merged later
                dist, idx, qb, bs = out
                db.txt_sorted.select(qn, float(ABSENT_DIST), self._guard_stats, dist=dist, idx=idx,
                                     idx_base=db.idx_base * db.Gt, q_block=qb, block_stride=bs, scratch=self._txt_scratch,
                                     cols_packed=cols_packed)
                return dist, idx
            rank = torch.empty((Q, db.K), dtype=torch.int16, device=dev) if want_rank else None
            dist, idx, _ = db.txt_sorted.select(qn, float(ABSENT_DIST), self._guard_stats, rank=rank,
                                                scratch=self._txt_scratch, cols_packed=cols_packed)
            if want_rank:
                return dist, idx, rank
            return dist, idx
        D = torch.empty((Q, max(db.Ct, 1)), dtype=torch.float32, device=dev)
        _lib.call("qpg_text_cosine_f32", dev, db.ctxt, db.Ct, db.Dt, qn, Q, D, D.stride(0))
        if out is not None:
            dist, idx, qb, bs = out
        else:
            dist = torch.empty((Q, db.K), dtype=torch.float32, device=dev)
            idx = torch.empty((Q, db.K), dtype=torch.int32, device=dev)
            qb = bs = 0
        fused_rank = want_rank and db.world == 1
        rank = torch.empty((Q, db.K), dtype=torch.int16, device=dev) if fused_rank else None
        _lib.call("qpg_percode_select_f32", dev, D, D.stride(0), Q, db.txt_cand_code, db.Ct, db.K, float(ABSENT_DIST),
                  db.idx_base * db.Gt, dist, idx, rank, qb, bs)
        if not reduce:              # sharded caller combines several tables in one exchange (sweep_tables)
            return dist, idx
        dist, idx = self._reduce_min(dist, idx)
        if want_rank:
            return dist, idx, (rank if fused_rank else self.rank_rows(dist))
        return dist, idx

    def sweep_audio_wavvq(self, test_wavvq, q_win, q_t, want_rank=False, reduce=True, out=None):
        """vq-wav2vec audio sweep: test_wavvq (M,398,2) ints (device tensor or array).  Distances are exact
        small integers (Levenshtein), returned as f32 [Q,512] with the winners' global candidate indices."""
        db, dev = self.db, self.db.device
        tw = torch.as_tensor(test_wavvq).to(dev)
        sym_q = (tw[..., 0].to(torch.int64) * WAVVQ_GROUP_SIZE + tw[..., 1].to(torch.int64)).to(torch.int32).contiguous()
        Q = int(q_win.shape[0]) if isinstance(q_win, torch.Tensor) else len(q_win)
        C = db.n_local * db.Gv
        D = torch.empty((Q, max(C, 1)), dtype=torch.float32, device=dev)
        taps = (ctypes.c_int32 * len(db.vq_taps))(*db.vq_taps)
        _lib.call("qpg_wavvq_lev_f32", dev, db.vq_sym, db.n_local, db.Tv, db.vq_t, db.Gv, taps, len(db.vq_taps),
                  sym_q, sym_q.shape[0], sym_q.shape[1], _i32(q_win, dev), _i32(q_t, dev), Q, D, D.stride(0))
        if out is not None:
            dist, idx, qb, bs = out
        else:
            dist = torch.empty((Q, db.K), dtype=torch.float32, device=dev)
            idx = torch.empty((Q, db.K), dtype=torch.int32, device=dev)
            qb = bs = 0
        fused_rank = want_rank and db.world == 1
        rank = torch.empty((Q, db.K), dtype=torch.int16, device=dev) if fused_rank else None
        _lib.call("qpg_percode_select_f32", dev, D, D.stride(0), Q, db.vq_cand_code, C, db.K, float(ABSENT_DIST),
                  db.idx_base * db.Gv, dist, idx, rank, qb, bs)
        self._last_D_aud = D
        if not reduce:              # sharded caller combines several tables in one exchange (sweep_tables)
            return dist, idx
        dist, idx = self._reduce_min(dist, idx)
        if want_rank:
            return dist, idx, (rank if fused_rank else self.rank_rows(dist))
        return dist, idx

    def _reduce_min(self, dist, idx):
        """Cross-shard min + index: all-reduce(MIN) on the distances, then
        all-reduce(MIN) on the indices of the ranks that hold that minimum.  Shards are contiguous
        row blocks, so the lowest index == the reference's first-wins scan order."""
        if self.db.world == 1:
            return dist, idx
        return allreduce_min_index(dist, idx)

    def guard_stats(self):
        """(pairs re-evaluated in the reference's arithmetic so far, trouble flag) of the near-tie guard; the flag is
        set by a list overflow or, on the mixed-precision path, by operand norms small enough to void its error bound."""
        v = self._guard_stats.cpu().numpy()
        return int(v[0]), bool(v[1])

    def mixed_stats(self):
        """Mixed-precision audio path: pairs re-evaluated with an f64 dot product so far (tier 1), pairs re-evaluated
        in the reference's arithmetic (tier 2), raw flag word (1 = list overflow, 2 = norms below the bound's range)."""
        v = self._guard_stats.cpu().numpy()
        return {"tier1_pairs": int(v[2]), "tier2_pairs": int(v[0]), "flags": int(v[1])}

    def tier1_list_lengths(self):
        """Entries of every query's tier-1 re-evaluation list in the last mixed-precision select (capacity 2048 each):
        read back from the select's workspace.  Diagnostics (bench.py --data speechlike, tests): what the caps see."""
        ws = getattr(self, "_mix_ws", None)
        Q = getattr(self, "_last_mix_Q", 0)
        if ws is None or not Q:
            return np.zeros((0,), np.int64)
        K = self.db.K
        stride = int(_lib.load().qpg_percode_select_mixed_ws_stride(K))
        w = ws[:Q * stride].view(Q, stride)[:, 24 * K:24 * K + 4].contiguous().view(torch.int32)
        return w.cpu().numpy().reshape(-1).astype(np.int64)

    def clear_flags(self):
        self._guard_stats[1:2].zero_()

    @staticmethod
    def numpy_ranks(dist, idx=None, integer=False):
        """The reference's rank expression (GestureKNN.py:553, 574) on the host: np.array(list).argsort().argsort().
        The dtype of that array is part of the tie behaviour (NumPy's sort kernels differ per dtype) and follows from
        what the list holds: the `1e+3` placeholders are Python floats, the distances NumPy scalars of the metric's
        dtype (np.float64 audio, np.float32 text, Python ints for the Levenshtein audio).  So a row in which EVERY code
        has a candidate is sorted in the distances' own dtype (float32 text, int64 wavvq), and a row with an absent
        code as float64 - reproduced here row by row (`idx` < 0 marks absent codes; `integer`: Levenshtein row)."""
        d = dist.detach().cpu().numpy()
        present = None if idx is None else (idx.detach().cpu().numpy() >= 0)
        out = np.empty(d.shape, np.int16)
        for r_, row in enumerate(d):
            full = present is not None and bool(present[r_].all())
            if full and integer:
                arr = np.array([int(x) for x in row])                    # list of Python ints -> int64
            elif full:
                arr = np.array(list(row))                                # list of np.float32 / np.float64 scalars
            else:
                arr = np.array(list(row.astype(np.float64)))             # a Python float in the list: float64
            out[r_] = arr.argsort().argsort()
        return torch.from_numpy(out).to(dist.device)

    def rank_rows(self, dist):
        out = torch.empty(dist.shape, dtype=torch.int16, device=dist.device)
        name = "qpg_rank_rows_f64" if dist.dtype == torch.float64 else "qpg_rank_rows_f32"
        _lib.call(name, self.db.device, dist.contiguous(), dist.shape[0], dist.shape[1], out)
        return out

    # -- reference-shaped single-query API (GestureKNN.py:666-691, 708-721) --------------------------
    def _unpack(self, dist, idx, G, ks, cidx):
        d = dist[0].cpu().numpy()
        ix = idx[0].cpu().numpy()
        code = self.db.code_host
        dists, pays, aux = [], [], []
        for c in range(self.db.K):
            if ix[c] < 0:
                dists.append(ABSENT_DIST)
                pays.append([])
                aux.append([])
            else:
                j, g = divmod(int(ix[c]), G)
                dists.append(d[c])
                pays.append(code[j, cidx[g]:cidx[g] + STEP_SZ])
                aux.append([j, ks[g]])
        return dists, pays, aux

    def search_audio_cands(self, clip_input, mode="wavlm_feat"):
        """clip_input: one 6144-d WavLM feature row (6 taps x 1024).  Same return triple as the
        reference: per-code distance list, per-code 4-code payload (or []), per-code [j, k] (or [])."""
        db = self.db
        if mode == "wavvq_feat":
            # clip_input: 22 numbers = 11 taps x (g1,g2) (data_processing.py:317); hand them to the kernel as a
            # 1-window track whose tap gather reproduces exactly those 11 symbols
            v = np.asarray(clip_input).reshape(11, 2).astype(np.int64)
            track = np.zeros((1, db.Tv, 2), np.int64)
            t0 = -min(db.vq_taps)
            for i, off in enumerate(db.vq_taps):
                track[0, t0 + off] = v[i]
            dist, idx = self.sweep_audio_wavvq(track, [0], [t0])
            return self._unpack(dist, idx, db.Gv, db.vq_k, db.vq_cidx_host)
        if mode != "wavlm_feat":
            raise NotImplementedError(mode)
        q = torch.as_tensor(np.asarray(clip_input, np.float32).reshape(1, NUM_AUDIO_FEAT_FRAMES, db.F),
                            device=db.device).contiguous()
        dist, idx = self.sweep_audio(q, [0], [0], tap_stride=1)
        return self._unpack(dist, idx, db.Ga, db.aud_k, db.aud_cidx_host)

    def search_text_cands(self, clip_input, mode="wavvq_feat"):
        db = self.db
        q = torch.as_tensor(np.asarray(clip_input, np.float32).reshape(1, db.Dt), device=db.device).contiguous()
        dist, idx = self.sweep_text(q)
        return self._unpack(dist, idx, db.Gt, db.txt_k, db.txt_rows_host)

    # -- whole clip ---------------------------------------------------------------------------------
    def n_steps(self):
        return len(self.query_positions())

    def sweep_tables(self, test_interp, test_context, n_windows, mode=MODE_AUD_TXT, owner_blocks=False, for_walk=False):
        """Both batched sweeps + ranks for all Q = n_windows*steps query positions (the windows may
        belong to several clips).  test_interp: f32 [M,180,F]; test_context: f32 [M,30,384] (device).
        Returns a dict of device tensors: aud_d/aud_idx/aud_rank, txt_d/txt_idx/txt_rank.
        owner_blocks (sharded DB only): the windows are `world` equal blocks and this rank only needs the final
        tables of block `rank` — one all-to-all instead of two all-reduces; the returned tables then hold only that
        block's Q/world rows.
        for_walk: the tables go straight into walk() and nowhere else - the audio select may then leave unsettled what the
        walk can not read (CodeKNN.rank_cut; the returned tables are exact only where the walk reads them)."""
        db, dev = self.db, self.db.device
        M, steps = n_windows, self.n_steps()
        if test_interp.shape[0] < M or (mode != MODE_AUD and test_context.shape[0] < M):
            # the reference indexes test_wavlm_feat[i] / test_context[i] for i < n_test_seq (GestureKNN.py:788-800)
            raise IndexError("n_windows=%d but the test arrays hold %d audio / %d context windows"
                             % (M, test_interp.shape[0], test_context.shape[0]))
        if self.use_wavvq:
            ok = tuple(test_interp.shape[1:]) == (db.Tv, 2)
        else:
            ok = tuple(test_interp.shape[1:]) == (db.T, db.F)
        if not ok:
            raise ValueError("test audio windows have shape %s, database expects %s"
                             % (tuple(test_interp.shape[1:]), (db.Tv, 2) if self.use_wavvq else (db.T, db.F)))
        pos = self.query_positions()
        cache = self.__dict__.setdefault("_qcache", {})
        if M not in cache:          # index tensors are built once per clip length (also keeps H2D copies out of graphs)
            qw = np.repeat(np.arange(M), steps)
            qt = np.tile(np.array([int(i) for i in pos]), M)                  # clip_test[int(i)]  (:559, :565)
            rows_ = [int(i / self.n_db_frm * 30) for i in pos] * M           # GestureKNN.py:549, 551
            cache[M] = (_i32(qw, dev), _i32(qt, dev), _i32(np.asarray(rows_), dev))
        q_win, q_t, q_row = cache[M]
        T = dict(aud_d=None, aud_idx=None, aud_rank=None, txt_d=None, txt_idx=None, txt_rank=None)
        sharded = db.world > 1 or self.force_sharded
        # Round 5: the rank fusion in front of the walk is two independent argmins (audio order, text order:
        # GestureKNN.py:574-576, :553-555) - each side's half is launched behind its own select on its own stream
        # (qpg_fuse_best_ranked) into the walk's gate tables, which the walk then takes as they are (QPG_MODE_PREFUSED):
        # the join of the two streams has half a rank fusion less behind it.  split_fuse = False: one launch in the walk.
        # (a clip or a few: the fusion is one latency-bound round of blocks and the critical path loses 3-4 us; 16 clips'
        # worth of steps are throughput-bound and two launches buy nothing: experiments/round_scripts/r05_ab_split.sh)
        split = (for_walk and mode == MODE_AUD_TXT and not sharded and not self.host_ranks and self.split_fuse and
                 db.K % 16 == 0 and db.K <= 4096 and M * steps <= self.split_fuse_max_steps)
        gtab = torch.empty((3, max(M, 1) * steps, db.K), dtype=torch.int32, device=dev) if split else None
        lay = None
        if sharded:
            # per-shard tables go straight into the exchange buffer (ExchangeLayout); merged after ONE collective
            parts = [p_ for p_, on in (("aud", mode in (MODE_AUD_TXT, MODE_AUD)), ("txt", mode in (MODE_AUD_TXT, MODE_TXT)))
                     if on]
            lkey = (M * steps, db.world if owner_blocks else 1, tuple(parts), not self.use_wavvq)
            lays = self.__dict__.setdefault("_layouts", {})
            lay = lays.get(lkey)
            if lay is None:             # one send buffer per shape, reused by every clip (stream-ordered: the previous
                lay = lays[lkey] = ExchangeLayout(lkey[0], db.K, lkey[1], parts, lkey[3], dev)   # clip's exchange read it)
        # The two sweeps are independent until the walk and lean on different pipes (f64 matrix cores vs f32
        # VALU): with both modalities on, the text side runs on a second HIP stream underneath the audio sweep.
        overlap = mode == MODE_AUD_TXT and self.overlap_sweeps
        if overlap:
            side = self.__dict__.get("_side_stream")
            if side is None:
                side = self.__dict__["_side_stream"] = torch.cuda.Stream(dev)
            # (wait_stream() makes a new event per call; two cached events do the same for ~5 us less host time per clip,
            # most of it in front of the step's first launch)
            gate = self.__dict__.get("_side_gate")
            if gate is None:
                gate = self.__dict__["_side_gate"] = torch.cuda.Event()
                self.__dict__["_side_done"] = torch.cuda.Event()
            if dev.index is None or dev.index == torch.cuda.current_device():
                gate.record()                   # (the current stream: no Stream object in front of the step's launches)
            else:
                gate.record(torch.cuda.current_stream(dev))
            side.wait_event(gate)

        # Round 4: the clip's WHOLE query side in one launch (qpg_clip_pack_hl: the audio gather / norms / split-f16 image
        # AND the text queries' gather / sklearn normalisation / column image) when both sides take their matrix-core
        # paths (one GPU or a row shard).  Behind the 32-row sweep, which holds every register of every CU, the text side's two
        # tiny pack launches did not get a wave slot before the sweep was over.
        mfma_text_ = (self.text_kernel == "mfma" and db.txt_sorted is not None and self.audio_precision != "exact")
        packed = None
        if (overlap and not self.use_wavvq and mfma_text_ and self._hl_plan(sharded, M * steps) and db.Dt % 128 == 0 and
                self.fused_pack):
            Qn = M * steps
            q32_ = torch.empty((Qn, NUM_AUDIO_FEAT_FRAMES * db.F), dtype=torch.float32, device=dev)
            qn2_ = torch.empty((Qn,), dtype=torch.float64, device=dev)
            nbq = int(_lib.load().qpg_audio_hl_query_bytes(Qn, db.F))
            qi_ = self.__dict__.get("_hl_qimage")
            if qi_ is None or qi_.numel() < nbq:
                qi_ = self._hl_qimage = torch.empty((nbq,), dtype=torch.uint8, device=dev)
            qn_ = torch.empty((Qn, db.Dt), dtype=torch.float32, device=dev)
            cols_ = db.txt_sorted.cols_buffer(Qn, self._txt_scratch)
            tc_ = test_context.contiguous()
            ti_ = test_interp.contiguous()
            _lib.call("qpg_clip_pack_hl", dev, ti_, ti_.shape[0], db.T, db.F, q_win, q_t, Qn,
                      NUM_AUDIO_FEAT_FRAMES, db.tap_stride, q32_, qn2_, qi_, qi_.numel(), tc_, tc_.shape[0], tc_.shape[1],
                      db.Dt, q_win, q_row, Qn, qn_, cols_, cols_.numel())
            packed = (q32_, qn2_, qn_)
            # the side stream starts BEHIND the pack
            gate = self.__dict__["_side_gate"]
            gate.record(torch.cuda.current_stream(dev))
            side.wait_event(gate)

        def text_pack():
            # gather clip_context[int(i/n*30)] of every step + sklearn normalisation in one launch
            tc = test_context.contiguous()
            qn = torch.empty((M * steps, db.Dt), dtype=torch.float32, device=dev)
            _lib.call("qpg_text_pack_queries_f32", dev, tc, tc.shape[0], tc.shape[1], db.Dt, q_win, q_row, M * steps, qn)
            return qn

        def text_side(qn=None):
            if packed is not None:
                qn = packed[2]
            elif qn is None:
                qn = text_pack()
            r = self.sweep_text(qn, want_rank=not sharded, reduce=not sharded, normalised=True,
                                out=lay.views("txt") if sharded else None, cols_packed=packed is not None)
            T["txt_d"], T["txt_idx"] = r[0], r[1]
            if not sharded:
                T["txt_rank"] = r[2]
            if gtab is not None:            # (on the stream the text select ran on)
                _lib.call("qpg_fuse_best_ranked", dev, T["txt_rank"], T["txt_idx"], db.pos_rank, db.freq_rank, M * steps,
                          db.K, gtab[1])
        # Order of the two sides (both modalities on).  Round 1 enqueued the text side first: its kernels ran while the
        # host was still enqueueing the audio side, but its sweep (all CUs, ~56 us) then delayed the audio sweep by as
        # much; with the audio side first and no ordering between the streams the two sweeps contend for the same CUs
        # (`audio_first`, kept for measurements: no faster).  Now (`text_after_sweep`): audio side first, and the text
        # sweep waits on its own stream for the END of the audio sweep, so that it fills the CUs the audio select leaves
        # idle (one block per query, ~90 us): 0.58 -> 0.555 ms per clip.
        # text_after_sweep: the audio side is enqueued first and the text side waits (on its own stream) for the END of
        # the audio sweep: the text sweep then fills the CUs the audio select leaves idle (one block per query) instead
        # of delaying the audio sweep by its own duration at the start of the clip.
        # Round 3: the text side on the matrix-core prefilter is ~60 us of small launches and the audio select now runs on
        # every CU, so behind the sweep the text side became the longer of the two chains.  audio_first (None = auto: on
        # with the matrix-core text side): the text side is enqueued behind the audio side's launches on its own stream
        # with NO ordering - its GEMM trickles through under the sweep and the tables are ready before the audio select
        # is.  bench.py, alternating in one run (tools/try_orders.sh), ms per clip: behind the sweep 0.400-0.404,
        # text first 0.384-0.408, audio_first 0.360-0.374.
        # (Enqueueing the text side even earlier - between the sweep's launch and the select's - starts its GEMM 50 us
        # sooner and costs the sweep 25 us: 197 instead of 172.)
        mfma_text = self.text_kernel == "mfma" and db.txt_sorted is not None and self.audio_precision != "exact"
        audio_first = getattr(self, "audio_first", None)
        audio_first = mfma_text if audio_first is None else bool(audio_first)
        after = overlap and self.text_after_sweep and not audio_first and not self.use_wavvq
        qn_early = None
        if overlap and not audio_first and not after:
            with torch.cuda.stream(side):
                text_side()
        if mode in (MODE_AUD_TXT, MODE_AUD):
            fn = self.sweep_audio_wavvq if self.use_wavvq else self.sweep_audio
            self._want_sweep_event, self._sweep_done = after, None
            kw = {"prepacked": packed} if (packed is not None and not self.use_wavvq) else {}
            if (for_walk and self.rank_cut and not sharded and not self.use_wavvq and not self.host_ranks and
                    mode in (MODE_AUD_TXT, MODE_AUD)):
                kw["cut_top_n"] = 1 if mode == MODE_AUD_TXT else 2
            r = fn(test_interp, q_win, q_t, want_rank=not sharded, reduce=not sharded,
                   out=lay.views("aud") if sharded else None, **kw)
            self._want_sweep_event = False
            T["aud_d"], T["aud_idx"] = r[0], r[1]
            if not sharded:
                T["aud_rank"] = r[2]
            if gtab is not None:
                _lib.call("qpg_fuse_best_ranked", dev, T["aud_rank"], T["aud_idx"], db.pos_rank, db.freq_rank, M * steps,
                          db.K, gtab[0])
        if after:
            with torch.cuda.stream(side):
                qn_early = None if packed is not None else text_pack()    # one small block, next to the audio sweep
            if self._sweep_done is not None:
                side.wait_event(self._sweep_done)
            with torch.cuda.stream(side):
                text_side(qn_early)
        elif overlap and audio_first:
            with torch.cuda.stream(side):
                text_side()
        if overlap:
            done = self.__dict__["_side_done"]
            done.record(side)
            torch.cuda.current_stream(dev).wait_event(done)
            # The text tables are allocated on `side` and consumed on `main`.  No record_stream() (measured +15 us per
            # clip for the allocator's events): a freed block can only be reused by a later `side` allocation, and every
            # use of `side` starts by waiting for `main` above, i.e. after main's consumers of the block.
        elif mode in (MODE_AUD_TXT, MODE_TXT):
            text_side()
        if sharded:
            # ONE collective for both modalities (all-to-all when every rank only needs its own clip's rows, all-gather
            # otherwise), then one merge launch per modality: min distance, lowest global index among equals, ranks.
            # The trouble word rides in the blocks (ExchangeLayout "flags"): stamped here, ORed in by the receivers.
            _lib.call("qpg_flags_stamp", dev, lay.send, lay.nblk, lay.block_bytes, lay.off["flags"], self._guard_stats)
            recv = exchange_bytes(lay.send, db.world, owner_blocks)
            src_stride = lay.block_bytes if owner_blocks else lay.send.numel()
            gathered = False
            for p_ in lay.parts:
                f64 = lay.dtype[p_] == torch.float64
                d = torch.empty((lay.Qb, db.K), dtype=lay.dtype[p_], device=dev)
                ix = torch.empty((lay.Qb, db.K), dtype=torch.int32, device=dev)
                rk = torch.empty((lay.Qb, db.K), dtype=torch.int16, device=dev)
                wavlm_aud = p_ == "aud" and not self.use_wavvq
                if wavlm_aud and getattr(self, "_last_audio_mixed", False):
                    self._merge_mixed(recv, src_stride, lay, owner_blocks, d, ix, rk)
                    gathered = True
                elif wavlm_aud and getattr(self, "_last_audio_exact", False):
                    self._merge_mixed(recv, src_stride, lay, owner_blocks, d, ix, rk, exact=True)
                    gathered = True
                elif f64:
                    # (f64 sweep + capped guard per shard: near-ties ACROSS shards / codes are detected here and
                    # re-matched on the exact path; exact integer distances of the wavvq mode need no guard)
                    guard = wavlm_aud and self.tie_eps > 0
                    _lib.call("qpg_merge_select_f64", dev, recv, db.world, src_stride, lay.off[p_ + "_d"],
                              lay.off[p_ + "_i"], lay.Qb, db.K, float(ABSENT_DIST), d, ix, rk,
                              float(self.tie_eps) if guard else 0.0, self._guard_stats if guard else None)
                else:
                    _lib.call("qpg_merge_select_f32", dev, recv, db.world, src_stride,
                              lay.off[p_ + "_d"], lay.off[p_ + "_i"], lay.Qb, db.K, float(ABSENT_DIST), d, ix, rk)
                T[p_ + "_d"], T[p_ + "_idx"], T[p_ + "_rank"] = d, ix, rk
            if not gathered:            # (the mixed merge's prologue ORs the received words in itself)
                _lib.call("qpg_flags_gather", dev, recv, db.world, src_stride, lay.off["flags"], self._guard_stats)
            # Every rank must take the same decision about a re-match (it is a collective path).  Bits raised BEFORE an
            # exchange reach every rank with it.  In the all-gather form every rank then runs the same merge on the same
            # bytes, so the bits the merge itself raises (cross-shard near-ties) are the same everywhere: no collective.
            # In the all-to-all form each owner merges its own query block: those last bits still take a 4-byte MAX
            # all-reduce, on the device, stream-ordered - the walk carries the agreed value out with the codes.
            if owner_blocks:
                allreduce_max_(self._guard_stats[1:2], force=self.force_sharded)
        if self.host_ranks:
            for p_ in ("aud", "txt"):
                if T[p_ + "_d"] is not None:
                    T[p_ + "_rank"] = self.numpy_ranks(T[p_ + "_d"], T[p_ + "_idx"],
                                                       integer=(p_ == "aud" and self.use_wavvq))
        if gtab is not None:
            T["gate_tables"] = gtab         # [0], [1]: both modalities' candidates for every (step, previous code)
        return T

    def _merge_mixed(self, recv, src_stride, lay, owner_blocks, d, ix, rk, exact=False):
        """Cross-shard merge of audio tables whose comparisons are not all decided by their values:
        approximate merge + requests, re-evaluation of the requested pairs where the rows live, final merge + ranks.
        Request slots are deterministic (qpg_merge_mixed_phase1_f64), so:
          all-gather form (every rank holds every shard's tables and runs the same merge): a shard refines ITS block of
            its OWN phase-1 run - no request exchange; ONE all-gather of the responses.  Two collectives per clip in all.
          all-to-all form (every rank owns one query block): the owner's requests travel to the shards and the responses
            back: two more all-to-alls.
        Mixed-precision tables: band = 2.1 x the sweep's bound, responses = f64 dot-product distances; what those leave
        within tie_eps raises FLAG_CROSS_SHARD_TIE.  exact=True (f64 tables of the uncapped select): band = tie_eps,
        responses in the reference's own arithmetic, request and flag lists sized for the worst case - the cross-shard
        tier 2, which cannot overflow and flags nothing."""
        db, dev = self.db, self.db.device
        W, Qb, K = db.world, lay.Qb, db.K
        if exact:
            Rq, fl_cap, band = K, K * W, float(self.tie_eps)
        else:
            # slots per (query, shard): ~70 / W requests per query are usual with the split-f16 band
            Rq = int(self.mixed_requests) if self.mixed_requests else max(64, 512 // W)
            fl_cap, band = 1024, (AUDIO_HL_BAND if getattr(self, "_last_audio_hl", False) else AUDIO_MX_BAND)
        R = Qb * Rq
        req_stride = resp_stride = 8 + 8 * R
        key = ("_mm_bufs_exact" if exact else "_mm_bufs", W, R)
        cache = self.__dict__.setdefault("_mm_cache", {})
        bufs = cache.get(key)
        need_ws = int(_lib.load().qpg_merge_mixed_ws_bytes(Qb, K, fl_cap))
        if bufs is None or bufs[2].numel() < need_ws:
            bufs = cache[key] = (torch.empty((W * req_stride,), dtype=torch.uint8, device=dev),
                                 torch.empty((W * resp_stride,), dtype=torch.uint8, device=dev),
                                 torch.empty((need_ws,), dtype=torch.uint8, device=dev))
        req, resp, ws = bufs
        _lib.call("qpg_merge_mixed_phase1_f64", dev, recv, W, src_stride, lay.off["aud_d"], lay.off["aud_i"], Qb, K,
                  float(ABSENT_DIST), band, R, req, req_stride, ws, ws.numel(), self._guard_stats, fl_cap,
                  lay.off["flags"])
        half = int(db.feature_dtype == "f16")
        if owner_blocks:
            req_recv = exchange_bytes(req, W, True)
            # block o of req_recv comes from owner o: its queries are rows o*Qb .. of this rank's packed query set
            _lib.call("qpg_shard_refine_f64", dev, req_recv, W, req_stride, R, Qb, db.idx_base * db.Ga, db.base, half,
                      db.T, db.F, db.aud_t, db.Ga, NUM_AUDIO_FEAT_FRAMES, db.tap_stride, self._last_q32, self._last_qn2,
                      db.cn2, resp, resp_stride, int(exact), self._guard_stats, Rq)
            resp_recv = exchange_bytes(resp, W, True)
        else:
            # every rank ran the same phase 1: block `rank` of MY request buffer is what owner(s) would have sent me
            mine = req[db.rank * req_stride:(db.rank + 1) * req_stride]
            _lib.call("qpg_shard_refine_f64", dev, mine, 1, req_stride, R, 0, db.idx_base * db.Ga, db.base, half,
                      db.T, db.F, db.aud_t, db.Ga, NUM_AUDIO_FEAT_FRAMES, db.tap_stride, self._last_q32, self._last_qn2,
                      db.cn2, resp, resp_stride, int(exact), self._guard_stats, Rq)
            resp_recv = exchange_bytes(resp[:resp_stride], W, False)
        _lib.call("qpg_merge_mixed_phase2_f64", dev, recv, W, src_stride, lay.off["aud_i"], Qb, K, float(ABSENT_DIST), ws,
                  ws.numel(), resp_recv, resp_stride, d, ix, rk, self._guard_stats, fl_cap,
                  0.0 if exact else float(self.tie_eps))

    def walk(self, T, n_windows, window_offset=0, mode=MODE_AUD_TXT, seed_code=None, seed_phase=None, sync=True,
             seed_ptrs=None, out_pin=None, n_chains=1):
        """Device-side walk of windows [window_offset, window_offset+n_windows) of the tables.
        sync=True: (codes, phases, votes) as NumPy arrays; sync=False: device tensors (+ the status pair), nothing waited
        for, `_last_ints` = codes | votes | status on the device; sync="ints": the integer results only, as ONE host array
        codes | votes | status - the walk's last kernel writes them straight into pinned host memory (zero-copy) and the
        stream is synchronised: no D2H copy launch behind the walk (bench.py's step; ~5 us of a 0.33 ms clip).
        seed_ptrs = (address of an i32 seed code, address of its f32 [8][16] phase block), both readable by the device
        (ClipGraph: pinned host memory the host rewrites before every replay - the seed is then DATA, not a kernel
        argument, and one captured graph serves every clip); out_pin: a pinned int32 tensor [M*30 + M*steps + 2] the
        integer results go to (with sync=False: nothing is waited for).
        n_chains > 1 (with seed_ptrs and out_pin: ClipGraph over several clips): that many INDEPENDENT clips of n_windows
        windows whose steps sit back to back in the tables from window_offset on; seed_ptrs then address i32 [n_chains]
        seed codes and f32 [n_chains][8][16] phase blocks, out_pin holds codes [n_chains][M*30] | votes [n_chains][M*steps]
        | status [n_chains][2]."""
        db, dev = self.db, self.db.device
        M, steps = n_windows, self.n_steps()
        CL = int(n_chains)
        assert CL == 1 or (seed_ptrs is not None and out_pin is not None), "several chains: the graph path only"
        if seed_ptrs is not None:
            seed_code, sp = 0, int(seed_ptrs[1])
        elif seed_code is None:
            seed_code, seed_phase = self.init_code_phase()
        if seed_ptrs is not None:
            pass
        elif isinstance(seed_phase, torch.Tensor):
            sp = seed_phase.to(dev, torch.float32).contiguous()
        else:
            sp = torch.as_tensor(np.asarray(seed_phase, np.float32), device=dev).contiguous()
        # codes | votes | status (2) in ONE buffer: the integer results leave in a single D2H copy, no gather kernel
        # before it.  status[0] = an absent code won a rank fusion, status[1] = the sweeps' / selects' trouble word
        # (copied by the walk's last kernel from _guard_stats[1]): a clip whose word is not 0 is never returned.
        n_c, n_v = M * num_frames_code, M * steps
        host = sync is True or sync == "ints"
        if out_pin is not None:
            assert not host and out_pin.numel() >= CL * (n_c + n_v + 2)
            base = out_pin.data_ptr()
            out_codes, out_vote, status = base, base + 4 * CL * n_c, base + 4 * CL * (n_c + n_v)
        elif host:
            # pinned (device-visible) host memory, one buffer per clip length: safe to reuse because this call does
            # not return before the stream has drained and the values have been copied out of it
            pins = self.__dict__.setdefault("_pinned_ints", {})
            pin = pins.get(M)
            if pin is None:
                pin = pins[M] = torch.empty((n_c + n_v + 2,), dtype=torch.int32).pin_memory()
            base = pin.data_ptr()
            out_codes, out_vote, status = base, base + 4 * n_c, base + 4 * (n_c + n_v)
            pin_np = pin.numpy()
            # every word is a sentinel until the walk has written it; the status word is the walk's LAST store (behind a
            # system-scope fence), the others are checked as well before the buffer is copied (_wait_pinned)
            pin_np.fill(_PIN_SENTINEL)
        else:
            ints_d = torch.empty((n_c + n_v + 2,), dtype=torch.int32, device=dev)
            out_codes = ints_d[:n_c].view(M, num_frames_code)
            out_vote = ints_d[n_c:n_c + n_v].view(M, steps)
            status = ints_d[n_c + n_v:]                                  # always written by the walk kernels
        out_phase = torch.empty((CL * M, steps, 8, 16), dtype=torch.float32, device=dev)
        q0 = window_offset * steps
        gate = T.get("gate_tables")
        prefused = (gate is not None and mode == MODE_AUD_TXT and q0 == 0 and gate.shape[1] == CL * M * steps and
                    M > 0 and not self.serial_walk)
        if not prefused:
            gate = torch.empty((3, max(CL * M, 1) * steps, db.K), dtype=torch.int32, device=dev)
        mode_w = mode | (0x200 if prefused else 0)             # QPG_MODE_PREFUSED

        def sl(t):
            return None if t is None else t[q0:q0 + CL * M * steps]
        a_cidx, a_pslot, a_G = self._audio_grid()
        if seed_ptrs is not None:
            # one chain through the batch entry: its seed code is read from memory by the kernels
            _lib.call("qpg_match_steps_batch", dev, sl(T["aud_rank"]), sl(T["aud_idx"]), sl(T["txt_rank"]),
                      sl(T["txt_idx"]), db.pos_rank, db.freq_rank, db.code, db.code.shape[1], a_cidx, a_pslot, a_G,
                      db.txt_cidx, db.txt_pslot, db.Gt, db.phase, db.Tp, mode_w, M, steps, db.K, CL, int(seed_ptrs[0]), sp,
                      gate, out_codes, out_phase, out_vote, status, 2, self._guard_stats[1:2])
        else:
            _lib.call("qpg_match_steps", dev, sl(T["aud_rank"]), sl(T["aud_idx"]), sl(T["txt_rank"]), sl(T["txt_idx"]),
                      db.pos_rank, db.freq_rank, db.code, db.code.shape[1], a_cidx, a_pslot, a_G,
                      db.txt_cidx, db.txt_pslot, db.Gt, db.phase, db.Tp, mode_w | (0x100 if self.serial_walk else 0), M,
                      steps, db.K, int(seed_code), sp,
                      gate, out_codes, out_phase, out_vote, status, self._guard_stats[1:2])
        if out_pin is not None:
            return out_codes, out_phase, out_vote, status
        if not host:
            self._last_ints = ints_d
            return out_codes, out_phase, out_vote, status
        if sync == "ints":
            # the host watches the last word instead of sleeping in hipStreamSynchronize (~3.5 us sooner per clip); after
            # ~2 ms without it (a failed launch would never write it) the stream is synchronised the ordinary way
            return _wait_pinned(pin_np, torch.cuda.current_stream(dev))
        phases = out_phase.cpu().numpy()                    # (synchronises the stream: the pinned integers are complete)
        ints = pin.numpy().copy()
        self.check_status(ints[n_c + n_v:])
        codes = ints[:n_c].reshape(M, num_frames_code).astype(np.int64)
        votes = ints[n_c:n_c + n_v].reshape(M, steps).copy()
        return codes, phases, votes

    def walk_batch(self, T, n_windows, n_clips, seed_codes, seed_phases, mode=MODE_AUD_TXT):
        """Device-side walk of n_clips INDEPENDENT clips of n_windows windows each, whose steps sit back to back in the
        tables (one batched sweep: bench.py --clips 16, BASELINE configs[4]), in one set of launches
        (qpg_match_steps_batch).  seed_codes: ints [n_clips]; seed_phases: f32 [n_clips][8][16] (array or device tensor).
        Returns the device tensors (codes i32 [n_clips][M][30], phases f32 [n_clips][M][steps][8][16], votes i32
        [n_clips][M][steps]) and leaves `_last_ints` = i32 [n_clips][M*30 + M*steps + 2] (codes | votes | status per clip:
        ONE D2H copy)."""
        db, dev = self.db, self.db.device
        M, steps, CL = n_windows, self.n_steps(), int(n_clips)
        seeds = np.asarray(seed_codes, np.int64).reshape(-1)
        if seeds.shape[0] != CL or (seeds < 0).any() or (seeds >= db.K).any():
            raise ValueError("walk_batch: one seed code in [0, %d) per clip" % db.K)
        sc = torch.as_tensor(seeds.astype(np.int32), device=dev)
        if isinstance(seed_phases, torch.Tensor):
            sp = seed_phases.to(dev, torch.float32).contiguous()
        else:
            sp = torch.as_tensor(np.asarray(seed_phases, np.float32), device=dev).contiguous()
        if sp.numel() != CL * 128:
            raise ValueError("walk_batch: seed_phases must hold [n_clips][8][16] floats")
        n_c, n_v = M * num_frames_code, M * steps
        status_d = torch.empty((CL, 2), dtype=torch.int32, device=dev)
        codes_d = torch.empty((CL, M, num_frames_code), dtype=torch.int32, device=dev)
        votes_d = torch.empty((CL, M, steps), dtype=torch.int32, device=dev)
        out_phase = torch.empty((CL, M, steps, 8, 16), dtype=torch.float32, device=dev)
        Qt = CL * M * steps
        gate = T.get("gate_tables")
        prefused = gate is not None and mode == MODE_AUD_TXT and gate.shape[1] == Qt
        if not prefused:
            gate = torch.empty((3, Qt, db.K), dtype=torch.int32, device=dev)
        a_cidx, a_pslot, a_G = self._audio_grid()

        def sl(t):
            return None if t is None else t[:Qt]
        _lib.call("qpg_match_steps_batch", dev, sl(T["aud_rank"]), sl(T["aud_idx"]), sl(T["txt_rank"]), sl(T["txt_idx"]),
                  db.pos_rank, db.freq_rank, db.code, db.code.shape[1], a_cidx, a_pslot, a_G,
                  db.txt_cidx, db.txt_pslot, db.Gt, db.phase, db.Tp, mode | (0x200 if prefused else 0), M, steps, db.K, CL,
                  sc, sp, gate, codes_d, out_phase, votes_d, status_d, 2, self._guard_stats[1:2])
        self._last_ints = torch.cat((codes_d.view(CL, n_c), votes_d.view(CL, n_v), status_d), dim=1)
        self._last_gate_tables = gate                       # (tests compare the candidate tables of the two fusion paths)
        return codes_d, out_phase, votes_d

    @staticmethod
    def check_status(status):
        """status: the walk's two status ints on the host.  Raises what must never be ignored."""
        if int(status[1]) != 0:
            raise GuardOverflow(int(status[1]))
        if int(status[0]) != 0:
            raise IndexError("a code that never occurs in the database won a rank fusion "
                             "(the reference raises IndexError at GestureKNN.py:631-632)")

    def capture_clip_graph(self, n_windows, mode=MODE_AUD_TXT, n_sweep_windows=None, window_offset=0, audio=None,
                           context=None, owner_blocks=False, n_clips=1, encoder=None, encode_input=None,
                           encode_precision="f32", sweep_signal=False, doorbell=False):
        """Capture the whole per-clip launch sequence (pack, both sweeps, per-code argmin passes, ranks,
        rank-fusion tables, walk) into one HIP graph for a fixed clip shape.  Returns a ClipGraph whose
        run(test_audio, test_context, seed_code, seed_phase) replays it; results are device tensors.
        n_sweep_windows > n_windows sweeps more windows than it walks (several clips per sweep: bench.py N>1).
        audio / context: bind the graph to the caller's resident input tensors instead of static copies.
        n_clips > 1 (round 5; BASELINE configs[4]): that many independent clips of n_windows windows per replay, ONE
        batched sweep and one set of walk launches for all of them (their steps back to back in the tables).
        encoder / encode_input: a VQVAE and a resident pose batch f32 [B][T][C] whose encode (make_beat_dataset.py:314-316)
        runs INSIDE the capture on a branch of its own beside the match - one replay = the fused encode + match step."""
        from .replay import ClipGraph
        return ClipGraph(self, n_windows, mode, n_sweep_windows or n_windows * n_clips, window_offset, audio, context,
                         owner_blocks, n_clips, encoder, encode_input, encode_precision, sweep_signal, doorbell)

    def match_clip(self, test_interp, test_context, n_windows, mode=MODE_AUD_TXT, seed_code=None,
                   seed_phase=None, return_tables=False):
        """All windows of one clip: two batched sweeps + rank kernels + one device-side tail walk.
        Returns (codes int64 [M,30], phases f32 [M,8,8,16], votes [M,8]) as NumPy arrays.
        A clip for which the capped near-tie machinery raised its trouble word (GuardOverflow) is matched again on
        the uncapped path before anything is returned (on a sharded DB every rank sees the same word and re-matches)."""
        if seed_code is None:                       # drawn ONCE: a re-match starts from the same state
            seed_code, seed_phase = self.init_code_phase()
        if n_windows == 0:                          # an empty clip (the reference's loop body never runs, :785)
            return (np.zeros((0, num_frames_code), np.int64), np.zeros((0, self.n_steps(), 8, 16), np.float32),
                    np.zeros((0, self.n_steps()), np.int32))
        test_interp = test_interp.contiguous()
        try:
            T = self.sweep_tables(test_interp, test_context, n_windows, mode, for_walk=not return_tables)
            if return_tables:
                self.tables = T
            return self.walk(T, n_windows, 0, mode, seed_code, seed_phase)
        except GuardOverflow as e:
            if self.audio_precision == "exact":
                self.clear_flags()          # (the sticky word must not poison the clips after this one)
                raise RuntimeError("the uncapped path raised flags 0x%x: this is a bug" % e.flags)
            return self.rematch(e.flags, test_interp, test_context, n_windows, mode, seed_code, seed_phase, return_tables)

    def rematch(self, flags, test_interp, test_context, n_windows, mode, seed_code, seed_phase, return_tables=False):
        """The clip again on a path that cannot raise `flags`: only the text prefilter overflowed (FLAG_TEXT_OVERFLOW alone)
        -> the same audio path with the text side on the exact-order sweep; anything else -> audio_precision "exact"
        (f64 sweep + uncapped guard, which also takes the exact-order text sweep).  Clears the trouble word."""
        if flags == FLAG_TEXT_OVERFLOW and self.text_kernel == "mfma":
            self.clear_flags()
            self.fallbacks += 1
            self.text_fallbacks = getattr(self, "text_fallbacks", 0) + 1
            self.text_kernel = "valu"
            try:
                T = self.sweep_tables(test_interp, test_context, n_windows, mode)
                if return_tables:
                    self.tables = T
                return self.walk(T, n_windows, 0, mode, seed_code, seed_phase)
            except GuardOverflow as e2:          # the audio side of this clip is in trouble as well
                return self.rematch_exact(test_interp, test_context, n_windows, mode, seed_code, seed_phase, return_tables)
            finally:
                self.text_kernel = "mfma"
        return self.rematch_exact(test_interp, test_context, n_windows, mode, seed_code, seed_phase, return_tables)

    def rematch_exact(self, test_interp, test_context, n_windows, mode, seed_code, seed_phase, return_tables=False):
        """The clip again with audio_precision "exact" (f64 sweep + uncapped guard); clears the trouble word."""
        prev = self.audio_precision
        self.clear_flags()
        self.audio_precision = "exact"
        self.fallbacks += 1
        try:
            T = self.sweep_tables(test_interp, test_context, n_windows, mode)
            if return_tables:
                self.tables = T
            return self.walk(T, n_windows, 0, mode, seed_code, seed_phase)
        finally:
            self.audio_precision = prev


def predict_code_from_audio(db, test_interp, test_context, n_windows, mode=MODE_AUD_TXT, rng=None):
    """predict_code_from_audio (GestureKNN.py:724-813) for the shipped flags; returns (M,30) int64."""
    knn = CodeKNN(db, rng=rng)
    codes, _, _ = knn.match_clip(test_interp, test_context, n_windows, mode=mode)
    return codes


# The captured / pipelined replays live in replay.py since round 6 (this module was 2 000 lines).  Every caller imports them
# from here, so the names resolve lazily (PEP 562; replay.py imports THIS module, whichever of the two is imported first).
_REPLAY_NAMES = ("ClipGraph", "ClipPipeline", "GraphPipeline", "SerialReplayer")


def __getattr__(name):
    if name in _REPLAY_NAMES:
        from . import replay
        return getattr(replay, name)
    raise AttributeError("module %r has no attribute %r" % (__name__, name))
