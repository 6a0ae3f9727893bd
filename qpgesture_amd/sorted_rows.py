"""Rows sorted by code for the bounded matrix-core prefilter of the exact-f32 cosine family (csrc/qpg_sorted.hip).

The reference's text distance (GestureKNN.py:708-721 -> sklearn paired_distances(metric='cosine') on float32) is defined by
its arithmetic, but only the per-code minimum and its first-wins candidate are wanted: a split-f16 GEMM with an a-priori
error bound (qpg_hl_gemm_distance) leaves a band of rows per (query, code) that can be the minimum, and only those are
evaluated in sklearn's exact order (qpg_percode_select_sorted_f32).  Used by BASELINE configs[2] (cfg3.CosineIndex) and by
the matcher's text side (code_knn.CodeKNN.sweep_text)."""
import torch

from . import _lib

U32 = 2.0 ** -24
HL_GEMM_ERR = 1.3e-6          # QPG_AUDIO_HL_ERR: the split-f16 GEMM on unit-norm operands, f64 block sums (round 3's kernel)


def gemm32_err(d):
    """A-priori bound of the round-4 prefilter GEMM (hl_gemm32_kernel: the h h' products stay in the MFMA's f32
    accumulator over the whole K = d): a chain of d/32 full-size instructions is within kappa_chain 2^-24 of the exact sum
    relative to sum |products| (<= 1 for unit vectors).  kappa_chain: 40 for chains of up to 16 (selfcheck.KAPPA16_ASSUMED;
    measured at load time - 21-25 on MI355X - and the prefilter is not used when the measurement exceeds its limit;
    until round 5 this read 12 + d/32, a model the 16-instruction chain was never measured against: ADVICE r4), beyond
    that 40 + 2 per further instruction (every instruction: at most ~1.5 units for re-rounding the running sum);
    + cross-term chains 1.0e-7, split representation 3.0e-7, f32 result 1.2e-7."""
    n = d / 32
    kappa = 40.0 + 2.0 * max(0.0, n - 16)
    return kappa * U32 + 5.2e-7


def gemm_h_err(d):
    """A-priori bound of the round-5 prefilter GEMM on the h planes alone (hl_gemm64h_kernel): h = fl16(x 2^e) is within
    2^-11 |x| of x in the scaled normal range (below it: 2^-25 absolute against scaled norms >= 2^14 - 4e-11), so the dropped
    terms h l' + l h' + l l' are at most (2 x 2^-11 + 2^-22) sum |x_i||q_i| <= (2^-10 + 2^-22) |x||q| (Cauchy-Schwarz) =
    9.8e-4 for unit vectors; + the chain of d/32 instructions in the f32 accumulator (gemm32_err's kappa_chain 2^-24);
    + the f32 epilogue 1.2e-7."""
    return 2.0 ** -10 + 2.0 ** -22 + (gemm32_err(d) - 5.2e-7) + 1.2e-7 + 1e-9


def prefilter_band_h(d):
    """Band of the by-code select behind the h-plane prefilter: 2.1 x (gemm_h_err + 2 eps1 + E_sk) - 2.1e-3 at d = 512
    (prefilter_band: 8.8e-5; on BASELINE configs[2]'s rows ~12 % more pairs are listed)."""
    eps1 = ((d / 4 + 2) / 2 + 2) * U32
    e_sk = 0.5 * (8 * eps1 + 4 * (d / 4 + 3) * U32)
    return 2.1 * (gemm_h_err(d) + 2 * eps1 + e_sk)


def prefilter_band(d):
    """Band of the bounded prefilter for the exact-f32 cosine (derivation: csrc/qpg_sorted.hip): 2.1 x (E_pre + E_sk).
    eps1: relative error of an f32 sklearn-normalised element (norm^2 by 4 lane chains of d/4 squares, sqrt, divide);
    E_pre: the GEMM's bound + the two operands being off the true unit vectors by eps1 each;
    E_sk: sklearn's own rounding against the real-number distance (normalisation errors through the difference,
    Cauchy-Schwarz with |delta| <= 2, then the chains of d/4 squares)."""
    eps1 = ((d / 4 + 2) / 2 + 2) * U32
    e_pre = max(HL_GEMM_ERR, gemm32_err(d)) + 2 * eps1
    e_sk = 0.5 * (8 * eps1 + 4 * (d / 4 + 3) * U32)
    return 2.1 * (e_pre + e_sk)


class SortedRows:
    """xn: f32 [n][d] sklearn-normalised rows on the device; codes: int tensor [n] (outside [0, K): the row can never win).
    Builds: the non-zero rows - without exact duplicates of an earlier row of the same code - sorted by code (stable:
    original order inside a code = first-wins), every code's segment
    padded to 16 rows with copies of its first row (a 16-row tile of the GEMM then lies inside ONE code and a padding row
    never lowers its minimum), the total padded to 64; `row_index` i32 [R] (original index, -1 padding), `row_code` i16 [R]
    (padding rows: bit 14 set; the tail beyond the last segment: code 0x1fff), `xs` f32 [R + 1][d] (row R: zeros), the
    split-f16 fragment image of rows [0, R), `code_tile` i32 [K + 1] (first 16-row tile of every code) and `zero_row` i32 [K]: per code the lowest original index among the rows
    the normalisation left at zero (all-zero embeddings: all at the same distance from any query, first one wins), -1."""

    def __init__(self, xn, codes, K, device):
        dev, d = torch.device(device), xn.shape[1]
        cmt = codes.to(dev).to(torch.int64)
        valid = (cmt >= 0) & (cmt < K)
        zero = xn.double().square().sum(1) < 0.5                              # not a unit vector: sklearn left it at zero
        zr = torch.full((K,), 0x7fffffff, dtype=torch.int64, device=dev)
        zi = torch.nonzero(valid & zero).reshape(-1)
        if zi.numel():
            zr.scatter_reduce_(0, cmt[zi], zi, "amin")
        self.zero_row = torch.where(zr == 0x7fffffff, torch.full_like(zr, -1), zr).to(torch.int32).contiguous()
        self.n_zero_rows = int(zi.numel())
        keep = torch.nonzero(valid & ~zero).reshape(-1)
        keep = keep[self._first_of_duplicates(xn, cmt, keep)]
        self.n_rows_kept = int(keep.numel())
        order = keep[torch.sort(cmt[keep], stable=True).indices]              # original indices, by (code, index)
        cd = cmt[order]
        cnt = torch.bincount(cd, minlength=K)
        pad_cnt = (cnt + 15) // 16 * 16
        start = torch.cumsum(pad_cnt, 0) - pad_cnt
        Rs = int(pad_cnt.sum().item())
        R = max((Rs + 63) // 64 * 64, 64)                                     # (64: hl_gemm64h_kernel's wave tiles)
        first = (torch.cumsum(cnt, 0) - cnt)                                  # position in `order` of a code's first row
        within = torch.arange(order.numel(), device=dev) - first[cd]
        pos = start[cd] + within
        row_index = torch.full((R,), -1, dtype=torch.int32, device=dev)
        row_index[pos] = order.to(torch.int32)
        seg_code = torch.repeat_interleave(torch.arange(K, device=dev), pad_cnt)
        seg_code = torch.cat((seg_code, torch.full((R - Rs,), 0x1fff, device=dev, dtype=seg_code.dtype)))
        self.row_code = torch.where(row_index >= 0, seg_code, seg_code | 0x4000).to(torch.int16).contiguous()
        self.xs = torch.zeros((R + 1, d), dtype=torch.float32, device=dev)
        if order.numel():
            src = order[first.clamp(max=order.numel() - 1)[seg_code[:Rs]]]    # every slot: its segment's first row ...
            self.xs[:Rs] = xn[src]
            self.xs[pos] = xn[order]                                          # ... real rows: themselves
        lib = _lib.load()
        self.image = torch.empty((int(lib.qpg_hl_rows_bytes(R, d)),), dtype=torch.uint8, device=dev)
        _lib.call("qpg_hl_pack_rows", dev, self.xs, R, d, self.image, self.image.numel())
        self.R, self.d, self.K, self.row_index, self.device = R, d, K, row_index, dev
        self.code_tile = torch.cat((torch.zeros((1,), dtype=torch.int64, device=dev),
                                    torch.cumsum(pad_cnt, 0) // 16)).to(torch.int32).contiguous()      # [K + 1] tile prefix
        self.band = float(prefilter_band(d))
        # round 4: the prefilter hands the select tile minima + row masks instead of the Q x R matrix (False: the matrix,
        # as in round 3 - tests compare the two)
        self.use_masks = True
        # round 5: batches of >= by_code_min_q queries take the h-plane prefilter + the by-code select (cfg3.CosineIndex
        # turns it on: every row is wanted by several queries there; a clip's 48 text queries stay on the by-query select)
        self.by_code = False
        self.by_code_min_q = 256
        self.band_h = float(prefilter_band_h(d))
        self._xs_perm = None

    def xs_perm(self):
        """The rows chain-permuted for the four-lane exact evaluation (qpg_perm32_rows_f32), built on first use."""
        if self._xs_perm is None:
            self._xs_perm = torch.empty_like(self.xs)
            _lib.call("qpg_perm32_rows_f32", self.device, self.xs, self.R + 1, self.d, self._xs_perm)
        return self._xs_perm

    def uses_by_code(self, Q, q_block=0):
        return bool(self.by_code and Q >= self.by_code_min_q and self.d % 128 == 0 and q_block == 0 and self.use_masks)

    def _select_by_code(self, qn, absent, stats, dist, idx, nn, rank, idx_base, sc, cols_packed, raw=None):
        dev = self.device
        Q = (raw if qn is None else qn).shape[0]
        nt, ldq = self.R // 16, (Q + 15) // 16 * 16
        if sc.get("tmin_t") is None or sc["tmin_t"].shape != (nt, ldq):
            sc["tmin_t"] = torch.empty((nt, ldq), dtype=torch.float32, device=dev)
            sc["tmask_t"] = torch.empty((nt, ldq), dtype=torch.int16, device=dev)
        if sc.get("qperm") is None or tuple(sc["qperm"].shape) != (Q, self.d):
            sc["qperm"] = torch.empty((Q, self.d), dtype=torch.float32, device=dev)
        if raw is not None:
            # round 6: sklearn's normalisation, the column image and the chain-permuted copy in ONE launch on the raw queries
            _lib.call("qpg_hl_prepare_queries", dev, raw, Q, self.d, None, sc["cols"], sc["cols"].numel(), sc["qperm"])
        else:
            if not cols_packed:
                _lib.call("qpg_hl_pack_cols", dev, qn, Q, self.d, sc["cols"], sc["cols"].numel())
            _lib.call("qpg_perm32_rows_f32", dev, qn, Q, self.d, sc["qperm"])
        _lib.call("qpg_hl_gemm_tilemin_h", dev, self.image, self.R, self.d, sc["cols"], Q, self.band_h, sc["tmin_t"],
                  sc["tmask_t"], ldq)
        _lib.call("qpg_percode_select_bycode_f32", dev, sc["tmin_t"], sc["tmask_t"], ldq, Q, self.R, self.row_code,
                  self.row_index, self.zero_row, self.code_tile, self.K, self.band_h, sc["qperm"], self.xs_perm(), self.d,
                  absent, dist, idx, rank, nn, stats, int(idx_base))
        return dist, idx, nn

    @staticmethod
    def _first_of_duplicates(xn, codes, keep):
        """Mask over `keep` (ascending original indices): False for a row that equals an EARLIER row of the same code.
        Identical rows are at the identical distance from any query, so only the first can win (first-wins) - and real
        per-frame text embeddings repeat (one sentence embedding over its frames, one embedding for silence:
        /root/reference/process/make_beat_dataset.py:556-565), which would otherwise put thousands of exact ties into
        every band.  Rows are grouped by (code, 64-bit hash of the row); inside a group every row is compared with the
        group's first row, elementwise: a hash collision keeps a row, it never drops one."""
        if keep.numel() == 0:
            return torch.ones((0,), dtype=torch.bool, device=keep.device)
        rows = xn[keep]
        bits = rows.contiguous().view(torch.int32).to(torch.int64)
        g = torch.Generator(device="cpu").manual_seed(0x5eed)
        w = (torch.randint(1, 2 ** 31 - 1, (rows.shape[1],), generator=g, dtype=torch.int64) * 2 + 1).to(rows.device)
        h = (bits * w).sum(1)                                                  # wraps mod 2^64
        key_order = torch.argsort(h, stable=True)                              # (ascending index inside equal hashes)
        key_order = key_order[torch.argsort(codes[keep][key_order], stable=True)]
        ck, hk = codes[keep][key_order], h[key_order]
        new_group = torch.ones_like(ck, dtype=torch.bool)
        new_group[1:] = (ck[1:] != ck[:-1]) | (hk[1:] != hk[:-1])
        first = torch.cummax(torch.where(new_group, torch.arange(ck.numel(), device=ck.device),
                                         torch.zeros_like(ck)), 0).values      # position of the group's first row
        same = (rows[key_order] == rows[key_order[first]]).all(1)
        drop_sorted = same & ~new_group
        mask = torch.ones((keep.numel(),), dtype=torch.bool, device=keep.device)
        mask[key_order[drop_sorted]] = False
        return mask

    def cols_buffer(self, Q, scratch):
        """The column-image buffer of `scratch` sized for Q queries (qpg_clip_pack_hl writes it ahead of select)."""
        nb = int(_lib.load().qpg_hl_cols_bytes(Q, self.d))
        if scratch.get("cols") is None or scratch["cols"].numel() < nb:
            scratch["cols"] = torch.empty((nb,), dtype=torch.uint8, device=self.device)
        return scratch["cols"]

    def select_raw(self, q, absent, stats, dist=None, idx=None, nn=None, rank=None, idx_base=0, scratch=None):
        """select() on RAW (not yet normalised) queries q f32 [Q][d], for batches that take the by-code path
        (uses_by_code(Q)): the whole query side - sklearn's normalisation included - is one launch (qpg_hl_prepare_queries).
        Same tables, bit for bit, as select(normalise(q))."""
        dev, Q = self.device, q.shape[0]
        if not self.uses_by_code(Q):
            raise ValueError("select_raw: this batch does not take the by-code path (normalise and call select)")
        sc = scratch if scratch is not None else {}
        nb = int(_lib.load().qpg_hl_cols_bytes(Q, self.d))
        if sc.get("cols") is None or sc["cols"].numel() < nb:
            sc["cols"] = torch.empty((nb,), dtype=torch.uint8, device=dev)
        if dist is None:
            dist = torch.empty((Q, self.K), dtype=torch.float32, device=dev)
            idx = torch.empty((Q, self.K), dtype=torch.int32, device=dev)
        return self._select_by_code(None, absent, stats, dist, idx, nn, rank, idx_base, sc, False, raw=q.contiguous())

    def select(self, qn, absent, stats, dist=None, idx=None, nn=None, rank=None, idx_base=0, q_block=0, block_stride=0,
               scratch=None, cols_packed=False):
        """qn: f32 [Q][d] sklearn-normalised queries.  Prefilter GEMM + banded exact select; per-code tables (dist f32
        [Q][K], idx i32 [Q][K] original row indices), optionally the nearest neighbours nn i32 [Q] and the ranks of the
        table rows (rank i16 [Q][K]).  An overflowing band list ORs 16 (FLAG_TEXT_OVERFLOW) into stats[1] (the caller
        re-evaluates on the exact sweep).  idx_base is added to the indices; q_block / block_stride: the row shards' exchange layout (dist / idx are
        then views of the exchange buffer; no ranks).
        scratch: a dict OWNED BY THE CALLER that keeps the column image, the prefilter matrix and the tile minima between
        calls.  This object is shared (GestureDB.txt_sorted is used by every lane of a ClipPipeline, each on its own
        stream), so it holds no per-call state itself: one caller = one stream = one scratch dict; None allocates per
        call (stream-ordered by torch's caching allocator)."""
        dev, Q = self.device, qn.shape[0]
        lib = _lib.load()
        nb = int(lib.qpg_hl_cols_bytes(Q, self.d))
        sc = scratch if scratch is not None else {}
        if sc.get("cols") is None or sc["cols"].numel() < nb:
            sc["cols"] = torch.empty((nb,), dtype=torch.uint8, device=dev)
        nt = self.R // 16
        if self.uses_by_code(Q, q_block):
            if dist is None:
                dist = torch.empty((Q, self.K), dtype=torch.float32, device=dev)
                idx = torch.empty((Q, self.K), dtype=torch.int32, device=dev)
            return self._select_by_code(qn, absent, stats, dist, idx, nn, rank, idx_base, sc, cols_packed)
        if sc.get("tmin") is None or sc["tmin"].shape[0] < Q or sc["tmin"].shape[1] != nt:
            sc["tmin"] = torch.empty((Q, nt), dtype=torch.float32, device=dev)
            sc["tmask"] = torch.empty((Q, nt), dtype=torch.int16, device=dev)
            sc["Dm"] = None
        cols, tmin, tmask = sc["cols"], sc["tmin"], sc["tmask"]
        if dist is None:
            dist = torch.empty((Q, self.K), dtype=torch.float32, device=dev)
            idx = torch.empty((Q, self.K), dtype=torch.int32, device=dev)
        if not cols_packed:           # (cols_packed: qpg_clip_pack_hl wrote scratch["cols"] for exactly these queries)
            _lib.call("qpg_hl_pack_cols", dev, qn, Q, self.d, cols, cols.numel())
        if self.use_masks:
            # tile minima + 16-bit masks of the rows within the band of their tile's minimum: the matrix never exists
            _lib.call("qpg_hl_gemm_tilemin", dev, self.image, self.R, self.d, cols, Q, self.band, tmin, tmask, nt)
            Dm = None
        else:
            if sc.get("Dm") is None or sc["Dm"].shape[0] < Q:
                sc["Dm"] = torch.empty((Q, self.R), dtype=torch.float32, device=dev)
            Dm = sc["Dm"]
            _lib.call("qpg_hl_gemm_distance", dev, self.image, self.R, self.d, cols, Q, Dm, self.R, tmin, nt)
        _lib.call("qpg_percode_select_sorted_f32", dev, Dm, self.R, tmin, tmask if self.use_masks else None, nt, Q, self.R,
                  self.row_code, self.row_index, self.zero_row, self.code_tile, self.K, self.band, qn, self.xs, self.d,
                  absent, dist, idx, rank, nn, stats, int(idx_base), int(q_block), int(block_stride))
        return dist, idx, nn
