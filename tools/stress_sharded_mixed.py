"""Randomised stress of the SHARDED mixed-precision merge (qpg_merge_mixed_phase1 / qpg_shard_refine /
qpg_merge_mixed_phase2) on one GPU: W = 2..8 row shards, the byte exchanges done by hand (all-gather form, owner 0),
near-ties planted across shard boundaries; winners and ranks must equal the unsharded f64 tables.
    python tools/stress_sharded_mixed.py [trials]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from qpgesture_amd import _lib, synth
from qpgesture_amd.code_knn import ABSENT_DIST, AUDIO_MX_BAND, AUDIO_MX_ERR, CodeKNN, ExchangeLayout, GestureDB
from qpgesture_amd.data_processing import interp_wavlm

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rs = np.random.RandomState(2026)
dev = torch.device("cuda:0")
bad = 0
for t in range(trials):
    W = int(rs.randint(2, 9)); N = int(rs.randint(W * 8, 700)); M = int(rs.randint(1, 5))
    half = bool(rs.rand() < 0.3)
    tr = synth.make_db(N, int(rs.randint(0, 10000)))
    x = interp_wavlm(tr["wavlm"]); code = synth.make_codes(N, int(rs.randint(0, 10000)))
    for _ in range(int(rs.randint(4, 40))):
        j, k = rs.choice(N, 2, replace=False)
        eps = 0.0 if rs.rand() < 0.25 else 10.0 ** rs.uniform(-7.3, -4.0)
        x[k] = (x[j] * (1.0 + eps * rs.standard_normal(x[j].shape))).astype(np.float32)
        if rs.rand() < 0.5:
            code[k] = code[j]
    ctx = np.ascontiguousarray(tr["context"].squeeze(2)); sig = synth.make_signature(3)
    te = synth.make_db(M, int(rs.randint(0, 10000)))
    ti = torch.from_numpy(interp_wavlm(te["wavlm"])).to(dev)
    tc = torch.from_numpy(np.ascontiguousarray(te["context"].squeeze(2))).to(dev)
    fd = "f16" if half else "f32"
    full = GestureDB(code, x, ctx, tr["phase_dense"], sig, device=dev, feature_dtype=fd)
    kf = CodeKNN(full, rng=np.random.RandomState(1)); kf.audio_precision = "f64"
    T = kf.sweep_tables(ti, tc, M)
    steps = kf.n_steps()
    q_win, q_t = np.repeat(np.arange(M), steps), np.tile(np.arange(steps) * 24, M)
    Q, K = M * steps, full.K
    shards, lays = [], []
    for r in range(W):
        db = GestureDB(code, x, ctx, tr["phase_dense"], sig, device=dev, rank=r, world=W, feature_dtype=fd)
        knn = CodeKNN(db, rng=np.random.RandomState(1))
        knn.sharded_mixed_min_gflop = 0.0                        # (these shards are far below the default work threshold)
        lay = ExchangeLayout(Q, K, 1, ["aud"], True, dev)
        knn.sweep_audio(ti, q_win, q_t, reduce=False, out=lay.views("aud"))
        shards.append(knn); lays.append(lay)
    recv = torch.cat([l.send for l in lays]); src_stride = lays[0].send.numel()
    Rq = 512; R = Q * Rq; req_stride = resp_stride = 8 + 8 * R      # deterministic slots: Rq = K per (query, shard) can not overflow (the product sizes them smaller and re-matches on the flag); 8-byte headers
    req = torch.zeros((W * req_stride,), dtype=torch.uint8, device=dev)
    ws = torch.empty((int(_lib.load().qpg_merge_mixed_ws_bytes(Q, K, 1024)),), dtype=torch.uint8, device=dev)
    stats = torch.zeros((4,), dtype=torch.int32, device=dev)
    _lib.call("qpg_merge_mixed_phase1_f64", dev, recv, W, src_stride, lays[0].off["aud_d"], lays[0].off["aud_i"], Q, K,
              float(ABSENT_DIST), AUDIO_MX_BAND, R, req, req_stride, ws, ws.numel(), stats, 1024, -1)
    resp_recv = torch.zeros((W * resp_stride,), dtype=torch.uint8, device=dev)
    for w in range(W):
        req_recv = torch.full((W * req_stride,), 255, dtype=torch.uint8, device=dev)      # (unused slots / blocks: ~0)
        for b in range(W):
            req_recv[b * req_stride:b * req_stride + 8] = 0                               # headers: count | flags
        req_recv[:req_stride] = req[w * req_stride:(w + 1) * req_stride]
        resp = torch.zeros((W * resp_stride,), dtype=torch.uint8, device=dev)
        k, db = shards[w], shards[w].db
        _lib.call("qpg_shard_refine_f64", dev, req_recv, W, req_stride, R, 0, db.idx_base * db.Ga, db.base, int(half), db.T,
                  db.F, db.aud_t, db.Ga, 6, db.tap_stride, k._last_q32, k._last_qn2, db.cn2, resp, resp_stride, 0, None, Rq)
        resp_recv[w * resp_stride:(w + 1) * resp_stride] = resp[:resp_stride]
    d = torch.empty((Q, K), dtype=torch.float64, device=dev); ix = torch.empty((Q, K), dtype=torch.int32, device=dev)
    rk = torch.empty((Q, K), dtype=torch.int16, device=dev)
    _lib.call("qpg_merge_mixed_phase2_f64", dev, recv, W, src_stride, lays[0].off["aud_i"], Q, K, float(ABSENT_DIST), ws,
              ws.numel(), resp_recv, resp_stride, d, ix, rk, stats, 1024, 1e-12)
    st = stats.cpu().numpy()
    ok = (torch.equal(ix, T["aud_idx"]) and torch.equal(rk, T["aud_rank"]) and (st[1] & ~8) == 0
          and float((d - T["aud_d"]).abs().max()) <= AUDIO_MX_ERR)
    bad += not ok
    print("trial %2d W=%d N=%3d M=%d %s  cross-shard re-evaluations %d  %s" % (t, W, N, M, fd, int(st[3]),
          "ok" if ok else "MISMATCH idx=%d rank=%d flags=%d" % (int((ix != T["aud_idx"]).sum()), int((rk != T["aud_rank"]).sum()), int(st[1]))), flush=True)
print("%d trials, %d mismatches" % (trials, bad))
sys.exit(1 if bad else 0)
