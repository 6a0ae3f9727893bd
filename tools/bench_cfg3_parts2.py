"""cfg-3 (h-plane prefilter + by-code select): HIP-event time of every launch of one batch, 20 repetitions."""
# needs a -DQPG_DEBUG_HOOKS variant of the library (the product exports no qpg_debug_* setters since round 6):
#   tools/build_variant.sh qpg_audio_hl hooks "-DQPG_DEBUG_HOOKS" && QPG_LIB_PATH=experiments/variants/libqpg_hooks.so python tools/bench_cfg3_parts2.py
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from qpgesture_amd import _lib
from qpgesture_amd.cfg3 import CosineIndex, make_inputs, ABSENT
X, code, valid, q = make_inputs()
dev = torch.device("cuda:0")
ix = CosineIndex(X, code, valid, device=dev)
sr = ix.sorted
qd = torch.from_numpy(q).to(dev)
Q, d = qd.shape
ix.check_flags = False
for _ in range(3): ix.query(qd)
sc = ix._scratch
qn = torch.empty_like(qd)
ldq = sc["tmin_t"].shape[1]
dist = torch.empty((Q, sr.K), dtype=torch.float32, device=dev); idx = torch.empty((Q, sr.K), dtype=torch.int32, device=dev)
nn = torch.empty((Q,), dtype=torch.int32, device=dev)
steps = [
 ("normalise", lambda: _lib.call("qpg_l2_normalize_rows_f32", dev, qd, Q, d, qn)),
 ("pack_cols", lambda: _lib.call("qpg_hl_pack_cols", dev, qn, Q, d, sc["cols"], sc["cols"].numel())),
 ("perm16", lambda: _lib.call("qpg_perm32_rows_f32", dev, qn, Q, d, sc["qperm"])),
 ("gemm64h", lambda: _lib.call("qpg_hl_gemm_tilemin_h", dev, sr.image, sr.R, d, sc["cols"], Q, sr.band_h, sc["tmin_t"], sc["tmask_t"], ldq)),
 ("select+finish", lambda: _lib.call("qpg_percode_select_bycode_f32", dev, sc["tmin_t"], sc["tmask_t"], ldq, Q, sr.R, sr.row_code, sr.row_index, sr.zero_row, sr.code_tile, sr.K, sr.band_h, sc["qperm"], sr.xs_perm(), d, ABSENT, dist, idx, None, nn, ix._stats, 0)),
]
lib = _lib.load()
for nw in (8, 4, 8, 4):
    lib.qpg_debug_gemm64_waves(nw)
    fn = steps[3][1]
    for _ in range(3): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    print("gemm64h, %d waves per block: min %.1f med %.1f us" % (nw, ts[0] * 1e3, ts[10] * 1e3))
lib.qpg_debug_gemm64_waves(4)
for name, fn in steps:
    for _ in range(3): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    print("%-14s min %.1f med %.1f us" % (name, ts[0] * 1e3, ts[10] * 1e3))
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
for a, b in ev:
    a.record(); ix.query(qd); b.record()
torch.cuda.synchronize()
ts = sorted(a.elapsed_time(b) for a, b in ev)
print("whole query()  min %.1f med %.1f us; R=%d pairs flag=%d" % (ts[0] * 1e3, ts[10] * 1e3, sr.R, int(ix._stats[1].item())))
