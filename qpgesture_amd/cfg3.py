"""BASELINE.json configs[2] / SURVEY.md §8d cfg-3: the generic "distance matrix + per-code minimum" kernel at scale.

Synthetic DB of 100 000 code vectors x 512-d (f32, ~N(0,1), seed 0), code ids uniform in [0,512) (seed 1), validity
mask Bernoulli(0.9) (seed 2), 1 000 queries (seed 3).  Output per query: the per-code minimum cosine distance and its
candidate (512 of each) and the global nearest neighbour.  The arithmetic is the reference's for this distance
(GestureKNN.py:716 -> sklearn paired_distances(metric='cosine') on float32: normalise, einsum-order sum of squares),
bit for bit, which rules out FMA and the matrix cores; the kernel is VALU-bound, not HBM-bound, and says so.

Round 3: the tables no longer need the exact arithmetic for EVERY pair.  `CosineIndex(method="mfma")` (the default for f32
rows) sorts the valid rows by code, runs a bounded prefilter on the f16 matrix cores (qpg_hl_gemm_distance) and evaluates
in sklearn's exact order only the rows inside each code's band (qpg_percode_select_sorted_f32, csrc/qpg_sorted.hip):
bit-identical tables; a query whose band list overflows (massive exact ties) falls back to the exact sweep.

`CosineIndex` is the host-side mirror: build once (normalise + tile the candidates), `query()` per batch."""
import ctypes
import time

import numpy as np
import torch

from . import _lib

N_DB, DIM, N_Q, K_CODES = 100_000, 512, 1000, 512
ABSENT = 1000.0
from .sorted_rows import HL_GEMM_ERR, U32, SortedRows, prefilter_band  # noqa: F401  (re-exported)


def make_inputs(n=N_DB, d=DIM, nq=N_Q, k=K_CODES):
    X = np.random.Generator(np.random.PCG64(0)).standard_normal((n, d), dtype=np.float32)
    code = np.random.Generator(np.random.PCG64(1)).integers(0, k, size=n).astype(np.int32)
    valid = np.random.Generator(np.random.PCG64(2)).random(n) < 0.9
    q = np.random.Generator(np.random.PCG64(3)).standard_normal((nq, d), dtype=np.float32)
    return X, code, valid, q


class CosineIndex:
    """Candidates resident in HBM, sklearn-normalised and tiled for lane-per-candidate access
    (qpg_text_pack_candidates_f32); masked rows carry code -1 and can never win."""

    def __init__(self, X, code, valid=None, n_codes=K_CODES, device="cuda:0", tiles_per_chunk=1, feature_dtype="f32",
                 method="mfma"):
        """feature_dtype "f16": the rows are stored ROUNDED to f16 (half the bytes) with their f32 norms; the tables are
        then the reference's on the f16-rounded database (qpg_text_percode_f16)."""
        dev = torch.device(device)
        n, d = X.shape
        self.device, self.n, self.d, self.K = dev, n, d, n_codes
        self.tiles_per_chunk = tiles_per_chunk
        self.feature_dtype = feature_dtype
        xd = torch.as_tensor(X, dtype=torch.float32).to(dev).contiguous().view(n, 1, d)
        cand_r = torch.zeros((1,), dtype=torch.int32, device=dev)
        if feature_dtype == "f16":
            if tiles_per_chunk != 1:
                raise ValueError("the f16 image is swept by the LDS-free organisation only (tiles_per_chunk=1)")
            self.xt = torch.zeros((((n + 63) // 64) * 64 * d,), dtype=torch.float16, device=dev)
            self.nrm = torch.empty((n,), dtype=torch.float32, device=dev)
            _lib.call("qpg_text_pack_candidates_f16", dev, xd, n, 1, d, cand_r, 1, self.xt, self.nrm)
        elif feature_dtype == "f32":
            self.xt = torch.zeros((((n + 63) // 64) * 64 * d,), dtype=torch.float32, device=dev)
            _lib.call("qpg_text_pack_candidates_f32", dev, xd, n, 1, d, cand_r, 1, self.xt)
        else:
            raise ValueError("feature_dtype must be 'f32' or 'f16'")
        cm = np.asarray(code, np.int64)
        if valid is not None:
            cm = np.where(np.asarray(valid), cm, -1)
        self.cand_code = torch.from_numpy(cm.astype(np.int16)).to(dev)
        self._ws = None
        self.fallbacks = 0
        self.method = method if (feature_dtype == "f32" and d % 128 == 0 and n_codes < 0x2000) else "valu"
        if self.method == "mfma":
            from .selfcheck import mfma_bound_ok          # the prefilter's bound rests on a measured hardware constant
            if not mfma_bound_ok(dev)[0]:
                self.method = "valu"
        if self.method == "mfma":
            self._build_sorted(xd.view(n, d), cm)

    def _build_sorted(self, xd, cm):
        """Rows for the bounded prefilter: sklearn-normalised (the exact kernel, so the refine reads what the exact sweep
        would), masked rows dropped, sorted by code (sorted_rows.SortedRows)."""
        dev = self.device
        xn = torch.empty_like(xd)
        _lib.call("qpg_l2_normalize_rows_f32", dev, xd.contiguous(), self.n, self.d, xn)
        self.sorted = SortedRows(xn, torch.from_numpy(cm), self.K, dev)
        self.sorted.by_code = True                # round 5: h-plane prefilter + by-code exact select for batches >= 256 queries
        self.R, self.band = self.sorted.R, self.sorted.band
        self._stats = torch.zeros((4,), dtype=torch.int32, device=dev)
        self._scratch = {}            # column image / prefilter matrix / tile minima (one index = one stream)

    def query(self, q, want_nn=True):
        """q: f32 [Q][d] device tensor.  Returns (dist f32 [Q][K], idx i32 [Q][K], nn i32 [Q])."""
        dev = self.device
        Q = q.shape[0]
        qn = None
        if self.method == "mfma" and not getattr(self, "_force_valu", False):
            self.sorted.band = self.band
            nn = torch.empty((Q,), dtype=torch.int32, device=dev) if want_nn else None
            if self.sorted.uses_by_code(Q) and getattr(self, "fused_prepare", True):
                # (round 6) normalise + column image + permuted copy: one launch on the raw queries
                dist, idx, nn = self.sorted.select_raw(q, ABSENT, self._stats, nn=nn, scratch=self._scratch)
            else:
                qn = torch.empty_like(q)
                _lib.call("qpg_l2_normalize_rows_f32", dev, q, Q, self.d, qn)
                dist, idx, nn = self.sorted.select(qn, ABSENT, self._stats, nn=nn, scratch=self._scratch)
            if not getattr(self, "check_flags", True):
                return dist, idx, nn                   # (timing loops: the flag is read once, after the loop)
            if int(self._stats[1].item()) == 0:
                return dist, idx, nn
            # a band list overflowed (massive exact ties): the exact sweep decides this batch
            self._stats.zero_()
            self.fallbacks += 1
        if qn is None:
            qn = torch.empty_like(q)
            _lib.call("qpg_l2_normalize_rows_f32", dev, q, Q, self.d, qn)
        need = int(_lib.load().qpg_text_percode_ws_bytes(self.n, Q, self.K, self.tiles_per_chunk))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty((need,), dtype=torch.uint8, device=dev)
        dist = torch.empty((Q, self.K), dtype=torch.float32, device=dev)
        idx = torch.empty((Q, self.K), dtype=torch.int32, device=dev)
        nn = torch.empty((Q,), dtype=torch.int32, device=dev) if want_nn else None
        if self.feature_dtype == "f16":
            _lib.call("qpg_text_percode_f16", dev, self.xt, self.nrm, self.n, self.d, self.cand_code, self.K, qn, Q,
                      0, ABSENT, self._ws, self._ws.numel(), dist, idx, None, nn)
        else:
            _lib.call("qpg_text_percode_f32", dev, self.xt, self.n, self.d, self.cand_code, self.K, qn, Q,
                      self.tiles_per_chunk, 0, ABSENT, self._ws, self._ws.numel(), dist, idx, None, nn)
        return dist, idx, nn


def bench(a, dev, world, rank, hbm_peak_gbs):
    """bench.py --workload cfg3: one step = normalise the 1 000 queries + the fused sweep / per-code minimum + merge.
    N > 1: the DB is row-sharded (replicas of the query set), no exchange is timed (each rank reports its shard's
    tables; the min+index merge across shards is the same qpg_merge_select_f32 as the matcher's)."""
    import torch
    import torch.distributed as dist
    X, code, valid, q = make_inputs()
    per = (N_DB + world - 1) // world
    lo, hi = min(rank * per, N_DB), min((rank + 1) * per, N_DB)
    import os
    f16 = getattr(a, "feature_dtype", "f32") == "f16"
    method = getattr(a, "cfg3_method", "mfma")
    index = CosineIndex(X[lo:hi], code[lo:hi], valid[lo:hi], device=dev,
                        tiles_per_chunk=1 if f16 else int(os.environ.get("QPG_CFG3_TPC", "1")),
                        feature_dtype="f16" if f16 else "f32", method=method)
    index.check_flags = False                      # (the timed loop reads the overflow flag once, at the end)
    qd = torch.from_numpy(q).to(dev)
    steps = min(a.steps, 50)
    for _ in range(max(a.warmup, 3)):
        index.query(qd)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    import gc
    gc.collect()
    gc.disable()                       # (a full collection inside the timed region is ~40 ms)
    t0 = time.perf_counter()
    for e0, e1 in ev:
        e0.record()
        out = index.query(qd)
        e1.record()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if index.method == "mfma":
        assert int(index._stats[1].item()) == 0, "a band list overflowed during the timed loop"
    ms = sorted(e0.elapsed_time(e1) for e0, e1 in ev)
    k_ms = ms[len(ms) // 2]
    mfma = index.method == "mfma"
    n_loc = hi - lo
    alg_bytes = n_loc * DIM * (2 if f16 else 4) + n_loc * (8 if f16 else 4) + N_Q * DIM * 4 + N_Q * K_CODES * 8
    lane_ops = 3.0 * N_Q * n_loc * DIM                                                   # sub, mul, add per element pair
    valu_peak = 1024 * 2.4e9 * 32                                                        # packed f32 non-FMA lane-ops/s
    by_code = mfma and index.sorted.uses_by_code(N_Q)
    rec = {"metric": "per-code min cosine sweep, query-candidate pairs/sec (cfg-3)", "value": round(N_Q * N_DB * steps / dt, 1),
           "unit": "pairs/s", "n_gpus": world, "steps": steps, "warmup": max(a.warmup, 3),
           "ms_per_step": round(dt / steps * 1e3, 4), "higher_is_better": True, "scaling": "strong",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "cfg-3: DB 100000 x 512 %s, 512 codes, Bernoulli(0.9) validity mask, 1000 queries; "
                                  "sklearn-exact f32 cosine, per-code min + argmin + global nearest neighbour%s"
                                  % ("stored f16 (rounded), widened + normalised in registers" if f16 else "f32",
                                     " (bounded f16 matrix-core prefilter + exact-order refine of the band: "
                                     "bit-identical tables)" if mfma else ""),
                      "method": index.method + ("/h-plane prefilter + by-code select" if by_code else ""),
                      "feature_dtype": "f16" if f16 else "f32",
                      "n_db": N_DB, "dim": DIM, "queries": N_Q, "parallelism": "db rows / %d" % world}}
    hbm_view = {"achieved": round(alg_bytes / (k_ms * 1e-3) / 1e9, 1), "peak": hbm_peak_gbs, "unit": "GB/s",
                "frac": round(alg_bytes / (k_ms * 1e-3) / 1e9 / hbm_peak_gbs, 4), "algorithmic_bytes": int(alg_bytes)}
    valu_view = {"lane_ops": lane_ops, "peak_lane_ops_per_s": valu_peak, "floor_ms": round(lane_ops / valu_peak * 1e3, 3),
                 "frac": round(lane_ops / valu_peak / (k_ms * 1e-3), 4)}
    if mfma:
        # Q = 1000 is beyond the f16 ridge (Q ~ 300): the matrix cores bound this shape.  `achieved` = the
        # step's ALGORITHMIC flops (2 Q N D = 102.4 GFLOP, SURVEY's figure) over the step time; `issued` = what the
        # prefilter really issues on the padded problem (one f16 MFMA per 16 x 16 x 32 block on the by-code path, three
        # on the by-query path)
        per_block = 1 if by_code else 3
        qpad = (N_Q + 95) // 96 * 96
        alg_flop = 2.0 * N_Q * n_loc * DIM
        issued = per_block * 2.0 * index.R * qpad * DIM
        traffic = _pmc_step_traffic("cfg3_step_bycode|100000x512 Q=1000" if by_code else "cfg3_step|100000x512 Q=1000") \
            if world == 1 else None
        rec["roofline"] = {
            "bound": "mfma", "achieved": round(alg_flop / (k_ms * 1e-3) / 1e12, 1), "peak": 2500.0, "unit": "TFLOP/s",
            "frac": round(alg_flop / (k_ms * 1e-3) / 1e12 / 2500.0, 4),
            "algorithmic_gflop": round(alg_flop / 1e9, 1),
            "issued_tflops_f16": round(issued / (k_ms * 1e-3) / 1e12, 1),
            "issued_frac": round(issued / (k_ms * 1e-3) / 1e12 / 2500.0, 4),
            "traffic": traffic,
            "traffic_source": ("profiles/pmc_traffic.json: FETCH_SIZE x 2 + WRITE_SIZE summed over the step's kernels, rocprofv3 "
                               "--pmc passes (tools/pmc_cfg3.sh); not re-measured per run") if traffic else None,
            "kernel": ("hl_gemm64h_kernel (prefilter on the h planes alone: one f16 MFMA per block, 64-row wave tiles, tile "
                       "minima + row masks tile-major) + percode_select_bycode_kernel (a code's rows through LDS once, four "
                       "lanes per exact-order pair) (+ query normalise / pack / permute, ranks + nearest neighbours)" if by_code
                       else "hl_gemm32_kernel (prefilter GEMM, 32-row wave tiles: tile minima + row masks, no matrix) "
                            "+ percode_select_sorted_kernel (+ query normalise / pack)"),
            "kernel_ms": round(k_ms, 4),
            "hbm": hbm_view,
            "note": "the tables are sklearn's separately rounded f32 arithmetic (bit-exact indices are the bar): the matrix "
                    "cores run a PREFILTER with an a-priori bound, only the rows inside each (query, code) band get the exact "
                    "order.  The whole step (GEMM + select + packs) is priced against the dense f16 matrix peak; the GEMM "
                    "alone is ~45 % of the step (profiles/r05_cfg3_*.md)"}
    else:
        rec["roofline"] = dict(hbm_view, bound="valu" if not f16 else "valu", traffic=CFG3_TRAFFIC_BYTES if world == 1 and not f16 else None,
                               traffic_source=None, kernel="text_cosine_gmin_f32_kernel (+ fill, merge)",
                               kernel_ms=round(k_ms, 4), valu=valu_view,
                               note="the distance is sklearn's separately rounded f32 arithmetic (bit-exact indices are the "
                                    "bar), so neither FMA nor the matrix cores are admissible: the kernel is VALU-bound")
        rec["roofline"]["bound"] = "hbm"          # (the contract's vocabulary; the binding unit is the VALU: see `valu`)
    if mfma and world == 1:
        # the timed path's tables against the exact-order sweep (every pair in sklearn's f32 arithmetic, no prefilter) on the
        # same inputs, outside the timed region: distances, first-wins indices and nearest neighbours bit for bit
        index._force_valu = True
        try:
            de, ie, ne = index.query(qd)
        finally:
            index._force_valu = False
        rec["tables_equal_exact_sweep"] = bool(torch.equal(out[0], de) and torch.equal(out[1], ie) and torch.equal(out[2], ne))
    return rec


# Fabric-side bytes per step from rocprofv3 PMC (FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024, separate passes):
# profiles/r02_cfg3_pmc.md.  ~0.2 GB of it is the candidate array (once per XCD-local group of query blocks); the rest
# is the 8-byte look-before-atomicMin reads of the [Q][K] table, which must bypass the per-XCD (non-coherent) L2s.
CFG3_TRAFFIC_BYTES = 6_440_000_000


def _pmc_step_traffic(key="cfg3_step|100000x512 Q=1000"):
    """HBM-side bytes of one cfg-3 step on the prefilter path, from the committed PMC summary (profiles/pmc_traffic.json)."""
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
    try:
        return int(json.load(open(path))[key]["hbm_bytes_per_step"])
    except (OSError, KeyError, ValueError):
        return None
