#!/usr/bin/env python
"""bench.py — matched gesture frames/s of the GestureKNN hot path on MI355X.

Workload (BASELINE.json configs[1]): one 24 s query clip (M = 6 windows -> Q = 48 matching steps
-> 1440 output frames at 60 fps) matched against a full speaker-10-class database of
N_db = 2048 windows (synthetic, real schema: SURVEY.md §8d cfg-2), shipped flags
(WavLM cosine f64 + text cosine f32 + phase gate).  A "step" is one complete pass of the hot
path for one clip per GPU: query packing, both candidate sweeps, per-code argmin, ranks and the
device-side matching walk, ending with the (M,30) code indices on the host.  The database is
already resident in HBM when the timed region starts.

N > 1 (torch.distributed.run, one rank per GPU): the DB is row-sharded across the ranks, every
rank sweeps ALL ranks' clips (one clip per rank) against its shard, the per-(query,code) minima
are combined with an RCCL all-reduce(MIN) + index all-reduce, and each rank walks its own clip.
Per-GPU work is constant in N (N clips x N_db/N rows)  ->  "scaling": "weak";
value = frames of all N clips / max-over-ranks time.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s
F64_MFMA_PEAK_TFLOPS = 78.6  # v_mfma_f64_16x16x4_f64: 2048 flop / 64 cycles / SIMD x 1024 SIMDs x 2.4 GHz


def chunked_db(n_db, lo, hi, seed, F=1024):
    """Rows [lo,hi) of the synthetic DB, generated in 64-window chunks with per-chunk seeds so that
    every rank can build just its shard (and the CPU baseline just its sample)."""
    from qpgesture_amd import synth
    from qpgesture_amd.data_processing import interp_wavlm
    parts = []
    c0 = (lo // 64) * 64
    while c0 < hi:
        n = min(64, n_db - c0)
        d = synth.make_db(n, seed * 100003 + c0 // 64, F)
        a, b = max(lo, c0) - c0, min(hi, c0 + n) - c0
        parts.append(dict(interp=interp_wavlm(d["wavlm"][a:b]), ctx=d["context"][a:b].squeeze(2)))
        c0 += 64
    return (np.concatenate([p["interp"] for p in parts]), np.concatenate([p["ctx"] for p in parts]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--n-db", type=int, default=2048)
    ap.add_argument("--windows", type=int, default=6)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=2048, help="DB windows in the CPU baseline sample")
    ap.add_argument("--no-overlap", action="store_true", help="run the text sweep after the audio sweep (one stream)")
    ap.add_argument("--check", action="store_true", help="verify the matched codes against a 1-rank run")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    from qpgesture_amd import synth
    from qpgesture_amd.code_knn import CodeKNN, GestureDB
    from qpgesture_amd.data_processing import interp_wavlm

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus, "launch with torch.distributed.run --nproc-per-node %d" % a.gpus
    # QPG_BENCH_ONE_GPU=1 (testing only): all ranks share cuda:0 and exchange through gloo, so that the
    # N>1 code path can be exercised on a 1-GPU box; the driver's runs use one GPU per rank over RCCL.
    one_gpu = os.environ.get("QPG_BENCH_ONE_GPU") == "1"
    dev = torch.device("cuda", 0 if one_gpu else local)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    N, M = a.n_db, a.windows
    per = (N + world - 1) // world
    lo, hi = min(rank * per, N), min((rank + 1) * per, N)
    code = synth.make_codes(N, 2)
    sig = synth.make_signature(3)
    phase = np.random.Generator(np.random.PCG64(5)).standard_normal((N, 240, 4, 8)).astype(np.float32)
    interp_shard, ctx_shard = chunked_db(N, lo, hi, seed=0)
    # GestureDB slices rows [lo,hi) of what it is given: hand it full-height views without the copies
    interp_full = _ShardView(interp_shard, lo, hi, N)
    ctx_full = _ShardView(ctx_shard, lo, hi, N)
    db = GestureDB(code, interp_full, ctx_full, phase, sig, device=dev, rank=rank, world=world)
    knn = CodeKNN(db, rng=np.random.RandomState(123456))
    knn.overlap_sweeps = not a.no_overlap

    # one clip per rank; every rank holds all clips' windows (they are small: M*180*1024 f32 = 4.4 MB)
    clips = [synth.make_db(M, 1000 + r) for r in range(world)]
    te_interp = torch.from_numpy(np.concatenate([interp_wavlm(c["wavlm"]) for c in clips])).to(dev)
    te_ctx = torch.from_numpy(np.concatenate([c["context"].squeeze(2) for c in clips])).to(dev)
    seed_code, seed_phase = knn.init_code_phase()
    seed_phase_d = torch.from_numpy(seed_phase).to(dev)

    def step():
        # N > 1: every rank sweeps all clips' queries against its DB shard; the per-(query, code) minima are exchanged
        # with ONE all-to-all that leaves each rank with the final tables of its own clip only
        T = knn.sweep_tables(te_interp, te_ctx, M * world, owner_blocks=world > 1)
        out_codes, _, _, status = knn.walk(T, M, window_offset=0, seed_code=seed_code,
                                           seed_phase=seed_phase_d, sync=False)
        return out_codes.cpu()          # the step ends with the indices on the host (drop-in: np.savez)

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(a.warmup):
        step()
    knn.kernel_events = []
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        codes = step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms = [e0.elapsed_time(e1) for e0, e1 in knn.kernel_events]
    knn.kernel_events = None
    k_ms = float(np.mean(ms))

    frames_per_step = 240 * M * world
    value = frames_per_step * a.steps / dt

    # ---- roofline of the dominant kernel (audio_cosine_f64_kernel), per launch on this rank --------------
    Q = M * world * 8
    C = db.n_local * db.Ga
    flops = 2.0 * Q * C * 6 * db.F                                  # 2*Q*N*26*6144
    achieved = flops / (k_ms * 1e-3) / 1e12
    alg_bytes = db.n_local * 81 * db.F * 4 + C * 8 + Q * 6 * db.F * 8 + Q * C * 8
    roofline = {"bound": "mfma", "achieved": round(achieved, 3), "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / F64_MFMA_PEAK_TFLOPS, 4),
                # HBM bytes per launch from rocprofv3 PMC (FETCH_SIZE x2 per the gfx950 correction + WRITE_SIZE),
                # measured for exactly this launch shape only: profiles/r01_pmc_audio.md
                "traffic": 925_000_000 if (world == 1 and N == 2048 and M == 6) else None,
                "kernel": "audio_cosine_f64_kernel", "kernel_ms": round(k_ms, 4),
                "algorithmic_gflop": round(flops / 1e9, 3),
                "algorithmic_bytes": int(alg_bytes),
                "hbm_gbs_algorithmic": round(alg_bytes / (k_ms * 1e-3) / 1e9, 1),
                "hbm_frac": round(alg_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}

    out = {"metric": "matched gesture frames/sec (GestureKNN)", "value": round(value, 1), "unit": "frames/s",
           "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 4),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
           "data": "synthetic",
           "config": {"workload": "24 s clip (M=%d windows, Q=%d steps, %d frames) per GPU vs speaker-10-class DB "
                                  "N_db=%d windows (%d candidates), shipped mode wavlm_feat(f64)+text(f32)+phase"
                                  % (M, M * 8, 240 * M, N, N * 26),
                      "n_db": N, "windows_per_clip": M, "clips": world,
                      "parallelism": ("db-row-shard x%d + all-to-all(min,index)" % world) if world > 1
                      else "single GPU, unsharded DB"},
           "roofline": roofline,
           "realtime_factor": round(value / 60.0 / world, 1)}

    out.update(vqvae_bench(dev, a, world, rank))

    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(a, code, clips[0], M, N)
        out["vs_cpu_baseline"] = round(value / out["cpu_baseline"]["value"], 1)
    if a.check:
        # every rank re-matches ITS clip against the WHOLE database on its own (world=1 semantics) and must
        # get the same codes as the sharded + all-reduced run above
        full_i, full_c = chunked_db(N, 0, N, seed=0)
        db1 = GestureDB(code, full_i, full_c, phase, sig, device=dev)
        k1 = CodeKNN(db1, rng=np.random.RandomState(123456))
        T1 = k1.sweep_tables(te_interp[rank * M:(rank + 1) * M], te_ctx[rank * M:(rank + 1) * M], M)
        want, _, _ = k1.walk(T1, M, 0, seed_code=seed_code, seed_phase=seed_phase_d)
        ok = bool(np.array_equal(want, codes.numpy().astype(np.int64)))
        out["check"] = ok
        assert ok, "rank %d: sharded result differs from the single-rank result" % rank
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def vqvae_bench(dev, a, world, rank):
    """Second half of BASELINE.json's metric: gesture VQ-VAE encode (and decode) frames/s.  Each rank encodes
    its own batch of 256 pose windows (240 frames x 135: dataset_to_code over a speaker DB runs in such
    batches) — pure data parallel, no collective — with the
    full-size codebook.yml architecture and seeded weights; the decode leg decodes one 24 s clip's worth of
    codes per rank in one pass (1440 frames)."""
    import torch
    import torch.distributed as dist
    from qpgesture_amd import synth
    from qpgesture_amd.vqvae import VQVAE
    model = VQVAE(None, 135, device=dev).load_state_dict(synth.make_vqvae_state_dict(7))
    Bw = 256
    x = torch.randn((Bw, 240, 135), device=dev)
    ids = torch.randint(0, 512, (1, 180), device=dev)

    def timed(fn, iters):
        for _ in range(2):
            fn()
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt / iters
    te = timed(lambda: model.encode(x), 10)
    td = timed(lambda: model.decode([ids]), 10)
    enc_flop = 1.639e9 * Bw                                        # 1.639 GFLOP per 240-frame window
    # training step (codebook/train.py:120-131) at the reference's batch size of 256 windows per rank: forward with
    # EMA codebook update, backward, flat-gradient all-reduce (N > 1), Adam
    from qpgesture_amd import parallel
    from qpgesture_amd.optim import Adam
    tm = VQVAE(dict(vel=1, acc=1), 135, device=dev).load_state_dict(synth.make_vqvae_state_dict(7)).train()
    opt = Adam(tm.parameters(), lr=3e-5, betas=(0.5, 0.999))

    def train_step():
        tm(x)
        tm.backward(sync_grads=True)
        opt.step()
    tt = timed(train_step, 5)
    train_flop = 3 * (1.639e9 + 1.908e9) * Bw                      # forward + data-gradient + weight-gradient GEMMs
    return {"vqvae_train_windows_per_s": round(Bw * world / tt, 1),
            "vqvae_train_ms_per_step_b256": round(tt * 1e3, 3),
            "vqvae_train_tflops_f32": round(train_flop / tt / 1e12, 2),
            "vqvae_encode_frames_per_s": round(240 * Bw * world / te, 1),
            "vqvae_encode_ms_per_batch256": round(te * 1e3, 3),
            "vqvae_encode_tflops_f32": round(enc_flop / te / 1e12, 2),
            "vqvae_decode_frames_per_s": round(1440 * world / td, 1),
            "vqvae_decode_ms_per_24s_clip": round(td * 1e3, 3)}


class _ShardView:
    """Full-height facade over one rank's row shard: supports [lo:hi] slicing and .shape only."""

    def __init__(self, shard, lo, hi, n):
        self.shard, self.lo, self.hi = shard, lo, hi
        self.shape = (n,) + shard.shape[1:]

    def __getitem__(self, sl):
        assert isinstance(sl, slice) and sl.start == self.lo and sl.stop == self.hi
        return self.shard


def cpu_baseline(a, code, clip, M, N):
    """The oracle's C port of the two reference scans (bit-identical results, OpenMP over DB windows)
    on a bounded sample: the same 48 queries against the first `cpu_sample` DB windows; both scans
    are linear in DB windows, so frames/s at full N_db = 1440 / (t * N_db/sample)."""
    from oracle import cref, knn_oracle as O
    from qpgesture_amd.data_processing import interp_wavlm
    cref.build()
    ns = min(a.cpu_sample, N)
    interp, ctx = chunked_db(N, 0, ns, seed=0)
    te = interp_wavlm(clip["wavlm"])
    q = np.stack([O.wavlm_feat_rows(te, w, [24 * s])[0] for w in range(M) for s in range(8)])
    qt = np.stack([clip["context"].squeeze(2)[w][int(24 * s / 180 * 30)] for w in range(M) for s in range(8)])
    cores = os.cpu_count() or 1
    t0 = time.perf_counter()
    cref.audio_scan(interp, np.arange(26) * 6, code[:ns], np.arange(26), q, n_threads=cores)
    cref.text_scan(ctx, np.arange(26), code[:ns], np.arange(26), qt, n_threads=cores)
    t = time.perf_counter() - t0
    full = t * N / ns
    return {"value": round(240 * M / full, 2), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "audio+text scans of the same %d queries vs the first %d of %d DB windows "
                      "(%.2f s measured, scaled linearly to N_db); C port of the reference arithmetic "
                      "(oracle/sweep_ref.c), OpenMP; matching walk excluded (<1%% of CPU time)" % (8 * M, ns, N, t),
            "sample_seconds": round(t, 3)}


if __name__ == "__main__":
    main()
