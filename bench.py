#!/usr/bin/env python
"""bench.py — matched gesture frames/s of the GestureKNN hot path on MI355X.

Default workload (BASELINE.json configs[1]): one 24 s query clip (M = 6 windows -> Q = 48 matching steps
-> 1440 output frames at 60 fps) matched against a full speaker-10-class database of
N_db = 2048 windows (synthetic, real schema: SURVEY.md §8d cfg-2), shipped flags
(WavLM cosine f64 + text cosine f32 + phase gate).  A "step" is one complete pass of the hot
path for one clip per GPU: query packing, both candidate sweeps, per-code argmin, ranks and the
device-side matching walk, ending with the (M,30) code indices on the host.  The database is
already resident in HBM when the timed region starts.  --clips-in-flight N (one GPU) issues the K steps (K independent
clips) with up to N in flight (code_knn.ClipPipeline: each clip goes through exactly the launches above on its own
stream, every clip's indices are on the host before the timed region ends); the same clips one at a time are then
timed right after and reported as `one_clip_at_a_time`.

Modes
  --scaling weak   (default) N ranks: N clips (one per rank) vs the DB row-sharded N ways; every rank sweeps all N clips
                   against its shard, ONE all-to-all(min,index) leaves each rank with its own clip's tables.
                   Per-GPU work is constant in N; value = frames of all N clips / max-over-ranks time.
  --scaling strong (BASELINE.json configs[3]) ONE clip vs a speaker-1-class DB (--n-db 8192) row-sharded N ways, ONE
                   all-gather(min,index) + local merge (the all-reduce(min+index) of the north star), replicated walk;
                   value = 1440 frames / latency of that one clip; realtime_factor = 24 s / latency.
  --clips C        (BASELINE.json configs[4]) C concurrent clips per GPU in one batched sweep (Q = 48 C), optionally
                   with --encode-batch B pose windows encoded (VQ-VAE) inside the same timed step and
                   --feature-dtype f16 (WavLM base stored in f16, widened in registers, same f64 arithmetic on the
                   f16-rounded values).
  --scaling replicated  (SURVEY.md 8e "shard Q instead of N"; the throughput axis of BASELINE.json configs[4]) the WHOLE
                   DB on every GPU (2.2 GB of 288), clips split across the ranks, NO collective in the step.  With
                   --scaling weak and N > 1 the same leg is measured after the row-sharded figure (`replicated` object).
  --workload cfg3  (BASELINE.json configs[2]) synthetic 100 000 codes x 512-d, 1 000 queries, per-code min + argmin.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")      # (qpgesture_amd/__init__.py: before the HIP runtime initialises)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s
F64_MFMA_PEAK_TFLOPS = 78.6  # v_mfma_f64_16x16x4_f64: 2048 flop / 64 cycles / SIMD x 1024 SIMDs x 2.4 GHz
F32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2 / 16x16x4, 64 flop/clk/SIMD
F16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense f16 / bf16 matrix peak


JSON_OUT = sys.stdout
DATA = "gaussian"      # --data: "gaussian" (i.i.d. N(0,1): SURVEY.md §8d) or "speechlike" (synth.speechlike_transform)


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
        if DATA == "speechlike":
            synth.speechlike_transform(d, 5000011 + seed * 100003 + c0 // 64)
        a, b = max(lo, c0) - c0, min(hi, c0 + n) - c0
        parts.append(dict(interp=interp_wavlm(d["wavlm"][a:b]), ctx=d["context"][a:b].squeeze(2)))
        c0 += 64
    return (np.concatenate([p["interp"] for p in parts]), np.concatenate([p["ctx"] for p in parts]))


def self_launch(n, argv=None, extra_env=None):
    """Re-exec this script as `python -m torch.distributed.run --nnodes=1 --nproc-per-node n ... bench.py <same flags>`
    (127.0.0.1 rendezvous on a free port) and pass its output through.  Returns the exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)]
    cmd += list(sys.argv[1:] if argv is None else argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
    env.update(extra_env or {})
    return subprocess.call(cmd, env=env)


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--n-db", type=int, default=None, help="DB windows (default 2048; 8192 with --scaling strong)")
    ap.add_argument("--windows", type=int, default=6)
    ap.add_argument("--scaling", choices=["weak", "strong", "replicated"], default="weak")
    ap.add_argument("--no-replicated", action="store_true",
                    help="N > 1, --scaling weak: skip the extra `replicated` leg (whole DB on every GPU, one clip per rank, no "
                         "collective) that is otherwise measured after the row-sharded figure, in the same run")
    ap.add_argument("--step-mode", choices=["auto", "graph", "eager"], default="auto",
                    help="how the timed steps are issued: graph = ONE hipGraphLaunch per clip (code_knn.ClipGraph: the same "
                         "kernels, seed and results through pinned memory; one capture serves every clip), eager = one "
                         "launch per kernel from Python; auto (default) = graph for one clip per step and rank (row "
                         "shards: a PROGRAM of hipGraph segments with the collectives issued eagerly between them, "
                         "step_mode graph-segments), eager otherwise.  In graph mode the eager figure and the HIP-event timing of the sweep kernel come from "
                         "an eager leg of the same K steps right after the timed region (`eager` object)")
    ap.add_argument("--no-graph", action="store_true", help="same as --step-mode eager")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end CLI leg (`e2e_cli` object)")
    ap.add_argument("--sharded-mixed-min-gflop", type=float, default=None,
                    help="row shards sweep in mixed precision when their sweep is at least this long (default: "
                         "CodeKNN.sharded_mixed_min_gflop = 20; 0 = always)")
    ap.add_argument("--audio-precision", choices=["mixed", "f64", "exact"], default="mixed",
                    help="mixed (default, the product default): f32 matrix-core sweep with an a-priori error bound + f64 / "
                         "reference-arithmetic re-evaluation of every undecided comparison; f64: the f64 matrix-core sweep; "
                         "exact: f64 sweep + the uncapped near-tie guard (the path flagged clips are re-matched on)")
    ap.add_argument("--audio-kernel", choices=["hl", "mx"], default="hl",
                    help="kernel of the mixed-precision sweep: hl = split-operand f16 matrix cores on the frame-major image "
                         "(HBM-bound, the default), mx = round 2's f32-matrix-core kernel")
    ap.add_argument("--clips", type=int, default=1, help="concurrent clips per GPU in one batched sweep")
    ap.add_argument("--clips-in-flight", type=int, default=1,
                    help="lanes of independent clips in flight (code_knn.ClipPipeline; single GPU, one clip per step): the "
                         "next clips' sweeps run under the previous clip's select / walk / D2H - 0.51 / 0.46 ms per clip "
                         "with 2 / 3 lanes against 0.54 one at a time.  The default line stays one clip at a time: its "
                         "`roofline` is then the sweep kernel ALONE on the GPU (with clips in flight the kernel's "
                         "duration includes what the other clips' kernels take from it)")
    ap.add_argument("--encode-batch", type=int, default=0, help="pose windows VQ-VAE-encoded inside the timed step")
    ap.add_argument("--encode-precision", choices=["f32", "f16x3"], default="f32",
                    help="encode leg: f32 = the f32 matrix-core kernels (exact f32 FMA chains); f16x3 = the split-operand f16 "
                         "kernels under their margin check, flagged windows re-encoded in f32 (VQVAE.encode_f16x3)")
    ap.add_argument("--feature-dtype", choices=["f32", "f16"], default="f32")
    ap.add_argument("--workload", choices=["match", "cfg3"], default="match")
    ap.add_argument("--cfg3-method", choices=["mfma", "valu"], default="mfma",
                    help="cfg3: bounded matrix-core prefilter + exact refine (default) or round 2's exact VALU sweep")
    ap.add_argument("--data", choices=["gaussian", "speechlike"], default="gaussian",
                    help="feature statistics of the synthetic DB and clips: i.i.d. N(0,1) (SURVEY.md §8d, the default line) "
                         "or speech-like (AR(1) rho 0.95 on rank-64 mixtures, 10 %% near-silent frames, repeating context "
                         "rows): prints the re-evaluation list populations the capped selects see (`band_lists`)")
    ap.add_argument("--no-sub-records", action="store_true",
                    help="skip the compact records of BASELINE configs[2], [3] (one GPU) and [4] that the default one-GPU run "
                         "appends to its line (`sub_records`: each is this script run again with that configuration's flags)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-vqvae", action="store_true", help="skip the VQ-VAE legs")
    ap.add_argument("--no-cold", action="store_true", help="skip the cold (H2D-inclusive) measurements")
    ap.add_argument("--no-prewarm", action="store_true", help="skip the time-based pre-warm (clock steady state)")
    ap.add_argument("--no-f64-line", action="store_true", help="skip the extra f64-sweep steps (`f64_sweep` object)")
    ap.add_argument("--cpu-sample", type=int, default=2048, help="DB windows in the CPU baseline sample")
    ap.add_argument("--no-overlap", action="store_true", help="run the text sweep after the audio sweep (one stream)")
    ap.add_argument("--text-first", action="store_true",
                    help="enqueue the text side before the audio side (round-1 order) instead of after the audio sweep")
    ap.add_argument("--check", action="store_true", help="verify the matched codes against a 1-rank run")
    ap.add_argument("--mixed-requests", type=int, default=None,
                    help="(testing) request slots per (owner, shard) pair of the sharded mixed-precision merge: a tiny "
                         "value forces the overflow -> agreed re-match path")
    return ap


def main():
    a = build_parser().parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` as the driver types it: become the launcher - one rank per GPU under
        # torch.distributed.run (RCCL), rank 0 prints the ONE JSON line, which passes through
        sys.exit(self_launch(a.gpus))

    # ONE JSON line on stdout, whatever the libraries print: RCCL writes its version banner to the C stdout when a
    # communicator comes up (it surfaced BEHIND the JSON line of a forced-sharded run).  File descriptor 1 is pointed at
    # stderr for the rest of the process and the line goes to a duplicate of the original stdout.
    global JSON_OUT
    sys.stdout.flush()
    JSON_OUT = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    global DATA
    DATA = a.data
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
    # QPG_BENCH_ONE_GPU=nccl: the same, but over backend nccl (RCCL) if it accepts two ranks on one device.
    one_gpu = os.environ.get("QPG_BENCH_ONE_GPU", "")
    dev = torch.device("cuda", 0 if one_gpu else local)
    # QPG_BENCH_FORCE_SHARDED=1 (testing / measurement on a 1-GPU box): ONE rank takes the row-shard code path - exchange
    # layout, the collectives over backend nccl (RCCL) with world_size 1, the merge kernels with W = 1
    force_sharded = os.environ.get("QPG_BENCH_FORCE_SHARDED", "") == "1"
    if os.environ.get("QPG_BENCH_LAUNCH_CHECK", "") == "1":
        # launcher self-test (CPU, gloo): prove that every rank of `--gpus N` came up and can talk, print ONE line
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.tensor([rank + 1], dtype=torch.int64)
        dist.all_reduce(t)
        if rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": world, "rank_sum": int(t.item())}), file=JSON_OUT, flush=True)
        dist.destroy_process_group()
        return
    torch.cuda.set_device(dev)
    if world > 1 or force_sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29534")
        if one_gpu == "1":
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    # Row shards: the exchanges go through the LIBRARY's own RCCL communicator (round 5, csrc/qpg_comm.hip: no
    # torch.distributed call - ~25 us of host time each - in a step, and the whole sharded clip is ONE hipGraph) unless it
    # cannot be brought up (no RCCL, gloo test runs) or QPG_LIB_COLLECTIVES=0; every rank takes the same transport.
    lib_coll, lib_coll_why = False, None
    if (world > 1 or force_sharded) and one_gpu != "1" and os.environ.get("QPG_LIB_COLLECTIVES", "1") != "0" and \
            a.scaling != "replicated":
        from qpgesture_amd import parallel as _par
        lib_coll, lib_coll_why = _par.negotiate_lib_collectives(dev)      # (every rank ends up on the same transport)

    if a.workload == "cfg3":
        out = cfg3_bench(a, dev, world, rank)
        if rank == 0:
            print(json.dumps(out), file=JSON_OUT, flush=True)
        if world > 1 or force_sharded:
            dist.destroy_process_group()
        return

    strong = a.scaling == "strong"
    replicated = a.scaling == "replicated"
    N = a.n_db if a.n_db is not None else (8192 if strong else 2048)
    M, CL = a.windows, a.clips
    db_world, db_rank = (1, 0) if replicated else (world, rank)       # replicated: every rank holds all N windows
    per = (N + db_world - 1) // db_world
    lo, hi = min(db_rank * per, N), min((db_rank + 1) * per, N)
    code = synth.make_codes(N, 2)
    sig = synth.make_signature(3)
    phase = np.random.Generator(np.random.PCG64(5)).standard_normal((N, 240, 4, 8)).astype(np.float32)
    interp_shard, ctx_shard = chunked_db(N, lo, hi, seed=0)
    # GestureDB slices rows [lo,hi) of what it is given: hand it full-height views without the copies
    interp_full = _ShardView(interp_shard, lo, hi, N)
    ctx_full = _ShardView(ctx_shard, lo, hi, N)
    db = GestureDB(code, interp_full, ctx_full, phase, sig, device=dev, rank=db_rank, world=db_world,
                   feature_dtype=a.feature_dtype)
    knn = CodeKNN(db, rng=np.random.RandomState(123456))
    knn.overlap_sweeps = not a.no_overlap
    if a.text_first:
        knn.text_after_sweep, knn.audio_first = False, False
    if "QPG_BENCH_AUDIO_FIRST" in os.environ:
        knn.audio_first = os.environ["QPG_BENCH_AUDIO_FIRST"] == "1"
    knn.audio_precision = a.audio_precision
    knn.audio_kernel = a.audio_kernel
    if a.sharded_mixed_min_gflop is not None:
        knn.sharded_mixed_min_gflop = a.sharded_mixed_min_gflop
    if a.mixed_requests is not None:
        knn.mixed_requests = a.mixed_requests
    knn.force_sharded = force_sharded
    # the mixed-precision sweep: one GPU, or row shards whose merge re-evaluates through a request / response exchange
    # (taken when the shard's sweep is long enough to pay for the two extra exchanges: CodeKNN.sharded_mixed_min_gflop)
    shard_gflop = 2e-9 * (M * 8 * (CL if strong else CL * world)) * (per * 26) * 6 * 1024
    sharded_run = (world > 1 and not replicated) or force_sharded
    mixed = a.audio_precision == "mixed" and (not sharded_run or shard_gflop >= knn.sharded_mixed_min_gflop)

    # clips: weak = CL per rank (every rank holds all of them: M*180*1024 f32 = 4.4 MB each); strong = ONE clip in all
    n_clips = CL if strong else CL * world
    clips = [synth.make_db(M, 1000 + r) for r in range(n_clips)]
    if a.data == "speechlike":
        for r, c_ in enumerate(clips):
            synth.speechlike_transform(c_, 7000003 + r)
    te_interp = torch.from_numpy(np.concatenate([interp_wavlm(c["wavlm"]) for c in clips])).to(dev)
    te_ctx = torch.from_numpy(np.concatenate([c["context"].squeeze(2) for c in clips])).to(dev)
    seed_code, seed_phase = knn.init_code_phase()
    seed_phase_d = torch.from_numpy(seed_phase).to(dev)
    enc = None
    if a.encode_batch:
        from qpgesture_amd.vqvae import VQVAE
        enc = VQVAE(None, 135, device=dev).load_state_dict(synth.make_vqvae_state_dict(7))
        enc_x = torch.randn((a.encode_batch, 240, 135), device=dev)
        enc.encode(enc_x)

    my_clips = CL                       # clips whose tables this rank ends up with (and walks)

    n_codes = M * 30
    rematched = [0]
    batch_walk = os.environ.get("QPG_BENCH_SERIAL_WALKS", "") != "1"          # (measurements: one walk per clip)
    seed_phases_d = seed_phase_d.reshape(1, -1).repeat(max(my_clips, 1), 1).contiguous()

    te_interp_all, te_ctx_all = te_interp, te_ctx
    if replicated and world > 1:
        # this rank's own clips only: the sweep sees CL clips, nothing is exchanged
        te_interp = te_interp[rank * CL * M:(rank + 1) * CL * M].contiguous()
        te_ctx = te_ctx[rank * CL * M:(rank + 1) * CL * M].contiguous()
    n_sweep_clips = CL if replicated else n_clips

    def step_eager():
        # weak, N > 1: every rank sweeps all clips' queries against its DB shard; ONE all-to-all leaves each rank with
        # the final tables of its own clips.  strong: ONE all-gather + merge, every rank holds the clip's tables.
        # replicated: this rank's clips against the whole DB, no exchange.
        if enc is not None:
            ids = enc.encode_f16x3(enc_x) if a.encode_precision == "f16x3" else enc.encode(enc_x)[0]
        T = knn.sweep_tables(te_interp, te_ctx, M * n_sweep_clips, owner_blocks=sharded_run and not strong, for_walk=True)
        if my_clips > 1 and batch_walk:
            # the clips are independent chains (their own seeds / window chaining): ONE set of walk launches for all
            knn.walk_batch(T, M, my_clips, [seed_code] * my_clips, seed_phases_d)
            res = knn._last_ints.cpu()                         # codes | votes | status (2) per clip: ONE D2H per step
        elif my_clips == 1:
            # codes | votes | status written by the walk's last kernel straight into pinned host memory: the step ends
            # on the host with a stream synchronise, no copy launch
            # (the harness stays in NumPy here: three torch CPU operations per step were ~10 us of a 0.34 ms clip)
            arr = knn.walk(T, M, seed_code=seed_code, seed_phase=seed_phase_d, sync="ints")
            if arr[-1] != 0 and knn.audio_precision != "exact":
                rematched[0] += 1
                prev, knn.audio_precision = knn.audio_precision, "exact"
                knn.clear_flags()
                try:
                    return step_eager()
                finally:
                    knn.audio_precision = prev
            knn.check_status(arr[-2:])
            return arr[:n_codes].reshape(M, 30)
        else:
            outs = []
            for c in range(my_clips):
                knn.walk(T, M, window_offset=c * M, seed_code=seed_code, seed_phase=seed_phase_d, sync=False)
                outs.append(knn._last_ints)
            res = torch.cat(outs).cpu().view(my_clips, -1)                                    # the step ends on the host
        if enc is not None:
            ids.cpu()
        if int(res[:, -1].max()) != 0 and knn.audio_precision != "exact":
            # the near-tie guard's trouble word came out with the codes (on a sharded DB every rank sees the same,
            # MAX-reduced word): this step again on the uncapped path - unguarded codes are never used
            rematched[0] += 1
            prev, knn.audio_precision = knn.audio_precision, "exact"
            knn.clear_flags()
            try:
                return step_eager()
            finally:
                knn.audio_precision = prev
        for c in range(my_clips):
            knn.check_status(res[c, -2:].tolist())
        return res[:, :n_codes].reshape(my_clips * M, 30)

    # Step mode.  graph: the whole per-clip launch sequence captured ONCE (nothing about a clip is baked in: the seed code /
    # phase block are data in pinned host memory, the inputs are the resident tensors) and replayed with one hipGraphLaunch
    # per clip; the integer results land in pinned host memory and the host watches the status word.  A step still ends
    # with the clip's codes on the host, a flagged clip is still re-matched (eagerly) before anything is returned.
    # Row shards (N > 1, or forced on one rank): the clip is recorded in SEGMENTS - one hipGraph per run of kernels between
    # two collectives, the collectives issued eagerly between the graph launches (parallel.SegmentRecorder); no RCCL kernel
    # is a graph node.  (ONE graph with the collectives inside works replay-only - tools/step_loop.py with
    # QPG_FORCE_SHARDED=1 QPG_EXPERIMENTAL_SHARDED_GRAPH=1 - but replays followed by eager collectives on the same
    # communicator hung this ROCm / torch build, and this script needs both.)  QPG_BENCH_SHARDED_EAGER=1: shards eager.
    shards_eager = sharded_run and os.environ.get("QPG_BENCH_SHARDED_EAGER", "") == "1" and a.step_mode != "graph"
    # (round 5: several clips per replay and the VQ-VAE encode leg inside the capture - BASELINE configs[4] - on one GPU or a
    # replicated database; row shards are captured one clip per rank)
    multi_ok = (CL == 1 and enc is None) or (not sharded_run)
    graph_mode = (a.step_mode == "graph" or (a.step_mode == "auto" and not a.no_graph)) and multi_ok and \
        a.clips_in_flight == 1 and not shards_eager and (world == 1 or sharded_run or replicated)
    if a.step_mode == "graph" and not graph_mode:
        raise SystemExit("--step-mode graph: no clips in flight; row shards: one clip per step and rank, no encode leg")
    cg, graph_fallback = None, None
    if graph_mode:
        knn_g = CodeKNN(db, rng=np.random.RandomState(123456))      # (its own workspaces / side stream: the capture's)
        knn_g.overlap_sweeps, knn_g.audio_precision, knn_g.audio_kernel = knn.overlap_sweeps, knn.audio_precision, knn.audio_kernel
        # (QPG_BENCH_GRAPH_TEXT_FIRST=1: the text side's nodes captured first.  A replay submits a second branch's first
        # node ~45 us behind the first branch's, whichever comes first; measured with tools/graph_orders.sh: text first 0.2985
        # ms per clip / GPU span 319 us, audio first 0.3048 / 307 - no order wins, the default keeps the eager order)
        if os.environ.get("QPG_BENCH_GRAPH_TEXT_FIRST", "0") == "1":
            knn_g.text_after_sweep, knn_g.audio_first = False, False
        knn_g.force_sharded, knn_g.sharded_mixed_min_gflop, knn_g.mixed_requests = (
            knn.force_sharded, knn.sharded_mixed_min_gflop, knn.mixed_requests)
        cg = knn_g.capture_clip_graph(M, n_sweep_windows=M * n_sweep_clips, audio=te_interp, context=te_ctx,
                                      owner_blocks=sharded_run and not strong, n_clips=my_clips,
                                      encoder=enc, encode_input=enc_x if enc is not None else None,
                                      encode_precision=a.encode_precision,
                                      # (not with the encode leg: its margin check may enqueue an eager f32 encode between
                                      # two replays, and nothing may sit on this stream in front of a pre-launched replay)
                                      # EXPERIMENTAL, off by default: -2.4 % per step, but a graph launched behind a running
                                      # graph launch occasionally loses its kernels on this ROCm build (code_knn.SerialReplayer)
                                      doorbell=(not sharded_run and enc is None and
                                                os.environ.get("QPG_BENCH_DOORBELL", "0") == "1"))
        if sharded_run:
            # the segments are recorded NOW, and every rank ends up in the same step mode: a capture that failed on any
            # rank sends all of them to the eager step (MIN over the ranks of "captured")
            ok_ = 1
            try:
                cg._set_seed(seed_code, seed_phase)
                cg._capture()
            except Exception as e_:                       # noqa: BLE001 (reported in the record)
                ok_, graph_fallback = 0, repr(e_)[:300]
            if world > 1:
                f_ = torch.tensor([ok_], dtype=torch.int32, device=dev)
                dist.all_reduce(f_, op=dist.ReduceOp.MIN)
                ok_ = int(f_.item())
            if not ok_:
                graph_mode, cg = False, None
                graph_fallback = graph_fallback or "the capture failed on another rank"

    graph_fallbacks = [0]

    # Round 6: the replay of the NEXT step is enqueued while this one still runs (ClipGraph.prelaunch: hipGraphLaunch costs
    # ~17 us of host time and the command processor's start-up) and waits at its first node for the host's doorbell; the
    # host rings it in launch() AFTER this step's codes are on the host and the next seed is written.  GPU work of a step
    # still starts only when its inputs are final - one clip at a time - only the launch overhead moved out of the gap
    # between two steps.  `more` = another step of the same loop follows (nothing is left pre-launched behind a loop).
    door = bool(cg is not None and getattr(cg, "_doorbell", False))
    sr = None
    if door:
        # (two captures taking turns on one stream: a graph exec is never re-launched while its own replay executes)
        from qpgesture_amd.code_knn import SerialReplayer
        knn_g2 = CodeKNN(db, rng=np.random.RandomState(123456))
        knn_g2.overlap_sweeps, knn_g2.audio_precision, knn_g2.audio_kernel = knn.overlap_sweeps, knn.audio_precision, knn.audio_kernel
        if "audio_first" in knn_g.__dict__:
            knn_g2.text_after_sweep, knn_g2.audio_first = knn_g.text_after_sweep, knn_g.audio_first
        cg2 = knn_g2.capture_clip_graph(M, n_sweep_windows=M * n_sweep_clips, audio=te_interp, context=te_ctx, n_clips=my_clips,
                                        doorbell=True)
        sr = SerialReplayer([cg, cg2])

    def step_graph(more=False):
        if sr is not None:
            arr, cgx = sr.step(seed_code, seed_phase, more)
        else:
            arr, cgx = cg.run_ints(seed_code, seed_phase), cg
        if enc is not None:
            cg.encoded_ids(arr)                    # (f16x3: windows the margin check flagged are re-encoded in f32 HERE)
        st_ = cgx.statuses(arr)
        if (st_[:, 1] != 0).any():                 # a trouble word came out with the codes: this step again, eagerly
            graph_fallbacks[0] += 1                # (ClipGraph.wait_ints cleared the capture's matcher's sticky word)
            if sr is not None:
                sr.drain()
            return step_eager()
        for c in range(my_clips):
            knn.check_status(st_[c])
        return arr[:my_clips * n_codes].reshape(my_clips * M, 30)

    step = step_graph if graph_mode else step_eager

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    pipe = None
    can_pipe = (world == 1 or strong) and CL == 1 and enc is None and not a.no_overlap and not force_sharded
    if a.clips_in_flight > 1:
        # throughput mode: the SAME per-clip launches, issued on `clips_in_flight` lanes; a step still ends with its
        # clip's indices on the host (collected one lane later)
        assert can_pipe, "--clips-in-flight: one GPU (or --scaling strong: every rank matches every clip), one clip per step"
        from qpgesture_amd.code_knn import ClipPipeline
        pipe = ClipPipeline(db, depth=a.clips_in_flight, rng=np.random.RandomState(123456))
        for ln in pipe.lanes:
            ln["knn"].overlap_sweeps = knn.overlap_sweeps
            ln["knn"].audio_precision = knn.audio_precision
            ln["knn"].audio_kernel = knn.audio_kernel

        def run_steps(n):
            pending, res = [], None
            for _ in range(n):
                if len(pending) == pipe.depth:
                    res = pipe.collect(pending.pop(0))[0]
                pending.append(pipe.submit(te_interp, te_ctx, M, seed_code=seed_code, seed_phase=seed_phase_d))
            while pending:
                res = pipe.collect(pending.pop(0))[0]
            return torch.from_numpy(res.astype(np.int32))
    else:
        step_times = []

        def run_steps(n):
            res = None
            rec = os.environ.get("QPG_BENCH_STEP_TIMES", "") == "1"
            for i_ in range(n):
                t_ = time.perf_counter()
                res = step(more=i_ + 1 < n) if graph_mode else step()
                if rec:
                    step_times.append(time.perf_counter() - t_)
            return torch.from_numpy(res) if isinstance(res, np.ndarray) else res

    # Clock / cache steady state regardless of --warmup (the driver's 20-step run was 13 % slower than the
    # 200-step profile, the GPU having idled for ~20 s of host-side data generation before a 12 ms timed region):
    # untimed steps until at least 1 s of back-to-back work has run AND three consecutive 10-step means agree within
    # 2 % (bounded at 6 s).  Not part of --warmup / --steps, which keep their contract meaning.
    # Everything that idles the GPU (the event pool, a 40 ms garbage collection) happens BEFORE the pre-warm: between the
    # pre-warm and the timed region there is nothing but the W warm-up steps.  (With the collection behind the pre-warm
    # the first ~15 timed steps ran 5 % slow - 0.37 falling to 0.35 ms - which a 20-step run averages in.)
    ev_list = [] if os.environ.get("QPG_BENCH_NO_EVENTS", "") != "1" else None     # (diagnostics)
    knn.kernel_events_every = int(os.environ.get("QPG_BENCH_EVENTS_EVERY", "1"))
    # the HIP events that bracket the sweep exist before the timed region (torch creates an event at its first record)
    knn.kernel_event_pool = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                             for _ in range(a.steps * max(1, CL if strong else 1) + 8)]
    for e0, e1 in knn.kernel_event_pool:
        e0.record()
        e1.record()
    # Python's cyclic collector off during the timed region: a full (generation-2) collection of this process takes
    # ~40 ms, and whether one falls into the 110 ms of 200 steps depends on the allocation count so far - measured as
    # 0.73 instead of 0.54 ms per step in most runs with --steps 200 and in none with --steps 400 / 1000
    import gc
    gc.collect()
    gc.disable()
    prewarm = {"seconds": 0.0, "steps": 0, "settled": False}
    if not a.no_prewarm:
        tp0 = time.perf_counter()
        means = []
        while True:
            t_ = time.perf_counter()
            run_steps(10)
            torch.cuda.synchronize(dev)
            means.append((time.perf_counter() - t_) / 10)
            prewarm["steps"] += 10
            el = time.perf_counter() - tp0
            ok3 = len(means) >= 3 and max(means[-3:]) <= 1.02 * min(means[-3:])
            stop = (el >= 1.0 and ok3) or el >= 6.0
            if world > 1:                                   # all ranks leave the loop together
                f = torch.tensor([1 if (el >= 1.0 and ok3) else 0, 0 if el >= 6.0 else 1], dtype=torch.int32, device=dev)
                dist.all_reduce(f, op=dist.ReduceOp.MIN)
                stop = bool(f[0].item() == 1 or f[1].item() == 0)
            if stop:
                prewarm.update(seconds=round(el, 3), settled=bool(ok3),
                               last_10step_means_ms=[round(x * 1e3, 4) for x in means[-3:]])
                break
    # the W warm-up steps come LAST, right in front of the fence: a 40 ms garbage collection (or anything else that
    # idles the GPU) between them and the timed region costs the first timed step 0.15 ms of clock ramp - 7 us per step
    # of a 20-step run
    knn.kernel_events = None
    run_steps(a.warmup)
    knn.kernel_events = ev_list
    if pipe is not None:
        for ln in pipe.lanes:
            ln["knn"].kernel_events = knn.kernel_events
            ln["knn"].kernel_event_pool = knn.kernel_event_pool
    fence()
    prof = None
    if os.environ.get("QPG_BENCH_CPROFILE", "") == "1":       # diagnostics: where the host spends the timed region
        import cProfile
        prof = cProfile.Profile()
        prof.enable()
    t0 = time.perf_counter()
    codes = run_steps(a.steps)
    fence()
    dt = time.perf_counter() - t0
    cut_used = bool(getattr(knn_g if graph_mode else knn, "_last_rank_cut", False))     # (of the timed steps' last sweep)
    gc.enable()
    if os.environ.get("QPG_BENCH_STEP_TIMES", "") == "1" and pipe is None:
        print("per-step ms (timed region):", ["%.3f" % (x * 1e3) for x in step_times[-a.steps:]], file=sys.stderr)
    if prof is not None:
        import pstats
        prof.disable()
        pstats.Stats(prof, stream=sys.stderr).sort_stats("tottime").print_stats(8)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    eager_leg = None
    if graph_mode:
        # the eager figure and the sweep kernel's HIP-event timing: the same K steps, one launch per kernel, right behind
        # the timed region (events cannot bracket a node of a replayed graph); the pool of events exists already
        for _ in range(max(a.warmup, 10)):
            step_eager()
        knn.kernel_events = []
        gc.collect()
        gc.disable()
        st0_ = knn.mixed_stats()
        fence()
        te0 = time.perf_counter()
        for _ in range(a.steps):
            ce = step_eager()
        fence()
        de = time.perf_counter() - te0
        gc.enable()
        st1_ = knn.mixed_stats()
        eager_leg = {"ms_per_step": round(de / a.steps * 1e3, 4), "steps": a.steps,
                     "tier1_pairs_per_step": (st1_["tier1_pairs"] - st0_["tier1_pairs"]) / a.steps,
                     "tier2_pairs_per_step": (st1_["tier2_pairs"] - st0_["tier2_pairs"]) / a.steps,
                     "frames_per_s": round(240 * M * n_clips * a.steps / de, 1),
                     "codes_equal_graph_steps": bool(np.array_equal(np.asarray(ce).reshape(-1), codes.numpy().reshape(-1)))}
    ms = [e0.elapsed_time(e1) for e0, e1 in knn.kernel_events] if knn.kernel_events else [float("nan")]
    if os.environ.get("QPG_BENCH_DUMP_KERNEL_MS"):          # (measurement aid: the eager leg's per-step sweep times, in order)
        print("kernel_ms per eager step:", " ".join("%.0f" % (1e3 * v) for v in ms), file=sys.stderr)
    knn.kernel_events = None
    if pipe is not None:
        for ln in pipe.lanes:
            ln["knn"].kernel_events = None
    k_ms = float(np.mean(ms))

    frames_per_step = 240 * M * n_clips
    value = frames_per_step * a.steps / dt

    serial = None
    if pipe is not None:
        # the same clip, one at a time (a step = submit + wait): the latency figure, and the sweep kernel alone on the GPU
        n1 = min(a.steps, 100)
        for _ in range(10):
            step()
        knn.kernel_events = []
        knn.kernel_event_pool = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                                 for _ in range(n1 + 8)]
        for e0, e1 in knn.kernel_event_pool:
            e0.record()
            e1.record()
        gc.collect()
        gc.disable()
        fence()
        t1 = time.perf_counter()
        for _ in range(n1):
            c1 = step()
        fence()
        d1 = time.perf_counter() - t1
        gc.enable()
        ms1 = [e0.elapsed_time(e1) for e0, e1 in knn.kernel_events]
        knn.kernel_events = None
        c1 = torch.from_numpy(c1) if isinstance(c1, np.ndarray) else c1
        assert torch.equal(c1.reshape(-1).to(torch.int32), codes.reshape(-1)), "clips in flight changed the result"
        serial = {"latency_ms_per_clip": round(d1 / n1 * 1e3, 4), "frames_per_s": round(frames_per_step * n1 / d1, 1),
                  "steps": n1, "kernel_ms": round(float(np.mean(ms1)), 4)}

    # ---- roofline of the dominant kernel (audio_cosine_f64_kernel), per launch on this rank --------------
    Q = M * n_sweep_clips * 8
    C = db.n_local * db.Ga
    fb = 2 if a.feature_dtype == "f16" else 4
    flops = 2.0 * Q * C * 6 * db.F                                  # 2*Q*N*26*6144
    achieved = flops / (k_ms * 1e-3) / 1e12
    alg_bytes = db.n_local * 81 * db.F * fb + C * 8 + Q * 6 * db.F * 4 + Q * C * 8
    default_shape = world == 1 and N == 2048 and M == 6 and CL == 1 and fb == 4
    peak = F32_MFMA_PEAK_TFLOPS if mixed else F64_MFMA_PEAK_TFLOPS
    hl = bool(mixed and getattr(knn, "_last_audio_hl", False))
    if hl:
        # split-operand f16 sweep: every database frame is read once from the frame-major image (27 super-rows x 3 frames
        # x F x 4 B per window = the 81 even frames), the matrix leaves in f32; three f16 MFMAs per 32 k-steps of a
        # 32-row x 96-column tile (27 of 32 rows live)
        planes = int(getattr(db, "hl_planes", 2))      # 1: the f16-stored track is its own h plane (qpg_audio_cosine_hl1)
        alg_bytes = db.n_local * 81 * db.F * (4 if planes == 2 else 2) + C * 8 + Q * 6 * db.F * 4 + Q * C * 4
        chunks = (Q + 47) // 48
        mfma_issued = (planes + 1) * 2.0 * (db.n_local * 32) * (chunks * 96) * (3 * db.F)
        gbs = alg_bytes / (k_ms * 1e-3) / 1e9
        tfs = mfma_issued / (k_ms * 1e-3) / 1e12
        # which roof: the launch's arithmetic intensity (issued matrix flops per algorithmic byte) against the ridge of the
        # two peaks - one clip (Q = 48) sits below it (HBM-bound), 16 clips per sweep far above (matrix-bound)
        ridge = F16_MFMA_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)
        intensity = mfma_issued / alg_bytes
        on_hbm = intensity < ridge
        # (third template argument: non-temporal fragment loads - the launcher's choice for a single query chunk)
        kname = "audio_cosine_hl2_kernel<%d, %d, %s>" % (planes, 2 if planes == 2 else 3, "true" if chunks == 1 else "false")
        pkey = None
        if world == 1 and N == 2048 and M == 6:
            pkey = "%s|N_db=2048 Q=%d" % ("audio_cosine_hl2_kernel" if planes == 2 else "audio_cosine_hl1", Q)
        roofline = {"bound": "hbm" if on_hbm else "mfma",
                    "achieved": round(gbs, 1) if on_hbm else round(tfs, 1),
                    "peak": HBM_PEAK_GBS if on_hbm else F16_MFMA_PEAK_TFLOPS, "unit": "GB/s" if on_hbm else "TFLOP/s",
                    "frac": round(gbs / HBM_PEAK_GBS, 4) if on_hbm else round(tfs / F16_MFMA_PEAK_TFLOPS, 4),
                    "arithmetic_intensity_flop_per_byte": round(intensity, 1), "ridge_flop_per_byte": round(ridge, 1),
                    **pmc_traffic(pkey),
                    **rocprof_kernel_ms(pkey),
                    "kernel": ("%s (split-operand f16 matrix cores on a frame-major image: every database frame read once, "
                               "32-row wave tiles; %s; chain sums added in f64, error bounded a priori, f64 re-evaluation "
                               "in the select)" % (kname, "two planes h | l of the f32 track, three products per element"
                                                   if planes == 2 else
                                                   "ONE plane: the f16-stored track is its own h, two products per element")),
                    "kernel_ms_source": ("HIP events around the kernel on its launch stream in the %d eager steps run right "
                                         "behind the timed region (a replayed graph's nodes cannot be bracketed)" % a.steps)
                    if graph_mode else "HIP events around the kernel on its launch stream, every timed step",
                    "precision": "mixed", "kernel_ms": round(k_ms, 4),
                    "kernel_ms_min": round(float(np.min(ms)), 4), "kernel_ms_median": round(float(np.median(ms)), 4),
                    "kernel_ms_max": round(float(np.max(ms)), 4), "kernel_launches_timed": len(ms),
                    "algorithmic_bytes": int(alg_bytes), "algorithmic_gflop": round(flops / 1e9, 3),
                    "mfma": {"issued_tflops_f16": round(mfma_issued / (k_ms * 1e-3) / 1e12, 1), "peak": F16_MFMA_PEAK_TFLOPS,
                             "frac": round(mfma_issued / (k_ms * 1e-3) / 1e12 / F16_MFMA_PEAK_TFLOPS, 4),
                             "equivalent_tflops_of_the_f32_formulation": round(achieved, 1),
                             "f32_matrix_peak": F32_MFMA_PEAK_TFLOPS},
                    # what a read-only stream over a buffer of this size reaches on these boxes with non-temporal loads
                    # (experiments/hbm_read/read_bw.hip, profiles/r05_read_bw.txt: 6.57-6.63 TB/s; plain loads 6.1)
                    "hbm_frac_of_6_6_tbs_read_stream_ceiling": round(gbs / 6600.0, 4)}
    else:
        roofline = {"bound": "mfma", "achieved": round(achieved, 3), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(achieved / peak, 4),
                    # HBM bytes per launch from rocprofv3 PMC (FETCH_SIZE x2 per the gfx950 correction + WRITE_SIZE),
                    # measured for exactly this launch shape only: profiles/r02_pmc_audio.md
                    "traffic": (AUDIO_MX_TRAFFIC_BYTES if mixed else AUDIO_TRAFFIC_BYTES) if default_shape else None,
                    "traffic_source": TRAFFIC_SOURCE if default_shape else None,
                    "kernel": ("audio_cosine_mx2_kernel (one launch: LDS-shared-query blocks + split-K remainder blocks; "
                               "f32 matrix cores, error bounded a priori, f64 re-evaluation in the select)")
                    if mixed else "audio_cosine_f64_kernel",
                    "precision": "mixed" if mixed else "f64",
                    **({"note": "kernel_ms is measured with %d clips in flight: other clips' kernels share the CUs during "
                                "the launch (alone: one_clip_at_a_time.kernel_ms)" % a.clips_in_flight}
                       if a.clips_in_flight > 1 else {}),
                    "kernel_ms": round(k_ms, 4),
                    "kernel_ms_min": round(float(np.min(ms)), 4), "kernel_ms_median": round(float(np.median(ms)), 4),
                    "kernel_ms_max": round(float(np.max(ms)), 4), "kernel_launches_timed": len(ms),
                    "algorithmic_gflop": round(flops / 1e9, 3),
                    "algorithmic_bytes": int(alg_bytes),
                    "hbm_gbs_algorithmic": round(alg_bytes / (k_ms * 1e-3) / 1e9, 1),
                    "hbm_frac": round(alg_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}

    if strong:
        par = ("db-row-shard x%d + all-gather(min,index) + local merge, replicated walk" % world) if world > 1 \
            else "single GPU, unsharded DB"
    elif replicated:
        par = "replicated DB x%d, clip-parallel, no collective" % world
    else:
        par = ("db-row-shard x%d + all-to-all(min,index)" % world) if world > 1 else "single GPU, unsharded DB"
    # collectives per step: all-gather form (strong) = tables + responses; all-to-all form (weak) = tables + requests +
    # responses + the 4-byte MAX of the merge's own trouble bits; f64 shards: the tables' exchange (+ the word in the
    # all-to-all form); replicated / one GPU: none
    if not sharded_run:
        n_coll = 0
    elif mixed or a.audio_precision == "exact":
        n_coll = 2 if strong else 4
    else:
        n_coll = 1 if strong else 2
    out = {"metric": "matched gesture frames/sec (GestureKNN)", "value": round(value, 1), "unit": "frames/s",
           "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 4),
           "higher_is_better": True, "scaling": "weak" if replicated else a.scaling, "vs_baseline": None,
           "dtype": ("split-f16 products / f64 block sums + f64 re-evaluation" if hl else "f32 sweep + f64 re-evaluation")
           if mixed else "f64",
           "data": "synthetic",
           "config": {"workload": "%d x 24 s clip%s (M=%d windows, Q=%d steps, %d frames each) %s vs speaker-%s-class DB "
                                  "N_db=%d windows (%d candidates), shipped mode wavlm_feat(%s)+text(f32)+phase%s%s"
                                  % (n_clips, "s" if n_clips > 1 else "", M, M * 8, 240 * M,
                                     "in all" if strong else "per job", "1" if N >= 8192 else "10", N, N * 26,
                                     "f32-bounded/f64-exact" if mixed else "f64", ", WavLM base stored f16" if fb == 2 else "",
                                     (", + VQ-VAE encode of %d pose windows in the step (%s)" % (a.encode_batch, a.encode_precision))
                                     if a.encode_batch else ""),
                      "n_db": N, "windows_per_clip": M, "clips": n_clips, "clips_per_gpu": CL,
                      "feature_dtype": a.feature_dtype, "audio_precision": "mixed" if mixed else "f64",
                      "encode_batch": a.encode_batch, "clips_in_flight": a.clips_in_flight, "parallelism": par,
                      "collectives_per_step": n_coll},
           "roofline": roofline,
           "realtime_factor": round(24.0 * M / 6 / (dt / a.steps), 1) if strong
           else round(value / 60.0 / world, 1)}

    out["data"] = "synthetic" if a.data == "gaussian" else "synthetic, speech-like statistics (synth.speechlike_transform)"
    if mixed and not sharded_run and pipe is None:
        ln = knn.tier1_list_lengths()
        if ln.size:
            out["band_lists"] = {"tier1_entries_per_query": {"min": int(ln.min()), "median": int(np.median(ln)),
                                                             "p99": int(np.percentile(ln, 99)), "max": int(ln.max())},
                                 "tier1_capacity": 2048, "tier2_capacity": 256,
                                 "band": 2.1 * (1.3e-6 if hl else 2.05e-6)}
    out["prewarm"] = prewarm
    out["rematched_steps"] = rematched[0] + (pipe.fallbacks if pipe is not None else 0)
    if world > 1 or force_sharded:
        out["ranks_seen"] = ranks_seen(dev, world, rank, one_gpu == "1")
    if mixed and not sharded_run and world == 1 and CL == 1 and pipe is None and enc is None and not a.no_f64_line:
        # the reference-precision figure beside the mixed one, in the same record: 20 more steps with the f64 sweep
        knn.audio_precision = "f64"
        pool6 = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(28)]
        for e0, e1 in pool6:
            e0.record()
            e1.record()
        gc.collect()
        gc.disable()
        for _ in range(10):             # (after the collection: nothing idles the GPU between these and the timed steps)
            step_eager()
        knn.kernel_events, knn.kernel_event_pool = [], pool6
        fence()
        t6 = time.perf_counter()
        for _ in range(20):
            c64 = step_eager()
        fence()
        d6 = time.perf_counter() - t6
        gc.enable()
        c64 = torch.from_numpy(c64) if isinstance(c64, np.ndarray) else c64
        ms6 = [e0.elapsed_time(e1) for e0, e1 in knn.kernel_events]
        knn.kernel_events = None
        knn.audio_precision = a.audio_precision
        k6 = float(np.mean(ms6))
        # the like-for-like figure (the reference's f64 cosine, GestureKNN.py:685) as top-level keys beside `value`
        out["value_f64"] = round(frames_per_step * 20 / d6, 1)
        out["ms_per_step_f64"] = round(d6 / 20 * 1e3, 4)
        out["f64_sweep"] = {"ms_per_step": round(d6 / 20 * 1e3, 4), "steps": 20, "kernel": "audio_cosine_f64_kernel",
                            "kernel_ms": round(k6, 4), "kernel_ms_min": round(float(np.min(ms6)), 4),
                            "achieved": round(flops / (k6 * 1e-3) / 1e12, 3), "peak": F64_MFMA_PEAK_TFLOPS,
                            "unit": "TFLOP/s", "frac": round(flops / (k6 * 1e-3) / 1e12 / F64_MFMA_PEAK_TFLOPS, 4),
                            "codes_equal_default_path": bool(torch.equal(c64, codes))}
    out["step_mode"] = ("graph-segments" if (sharded_run and not lib_coll) else "graph") if graph_mode else "eager"
    if sharded_run:
        out["collectives"] = {"transport": "libqpg_hip.so -> RCCL on the step's stream (qpg_comm_*; captured in the clip's "
                                           "hipGraph)" if lib_coll else "torch.distributed (%s)" % dist.get_backend(),
                              **({"library_transport_unavailable": lib_coll_why} if lib_coll_why else {})}
    if graph_fallback:
        out["step_mode_fallback"] = graph_fallback
    if graph_mode:
        out["eager"] = eager_leg
        # another seed through the SAME capture must equal the eager path started from that seed
        sc2 = (seed_code + 101) % 512
        sp2 = np.roll(seed_phase, 3, axis=0)
        g2 = cg.run_ints(sc2, sp2)
        T2 = knn.sweep_tables(te_interp, te_ctx, M * n_sweep_clips, owner_blocks=sharded_run and not strong)
        if my_clips == 1:
            e2 = knn.walk(T2, M, seed_code=sc2, seed_phase=sp2, sync="ints")
            same2 = bool(np.array_equal(g2[:e2.size], e2))
        else:
            knn.walk_batch(T2, M, my_clips, [sc2] * my_clips, np.tile(sp2.reshape(1, -1), (my_clips, 1)))
            e2 = knn._last_ints.cpu().numpy()                 # [clip][codes | votes | status]
            same2 = bool(np.array_equal(cg.codes(g2).reshape(my_clips, -1), e2[:, :n_codes]) and
                         np.array_equal(cg.statuses(g2), e2[:, -2:]))
        enc_same = None
        if enc is not None:
            enc_same = bool(np.array_equal(cg.encoded_ids(g2), enc.encode(enc_x)[0].cpu().numpy()))
        out["graph_replay"] = {"ms_per_step": out["ms_per_step"], "steps": a.steps, "captures": cg.captures,
                               "is_the_timed_region": True, "clips_per_replay": my_clips,
                               # the next step's replay is enqueued while the current one runs and waits at its first node
                               # for the host's doorbell, rung when this step's codes are on the host and the next seed is
                               # written (ClipGraph.prelaunch; QPG_BENCH_DOORBELL=0: launched after the results, as round 5)
                               "next_replay_prelaunched_behind_a_doorbell": door,
                               **({"doorbell_steps_recovered_after_a_lost_launch": sr.recovered} if sr is not None else {}),
                               "fallbacks_to_eager_in_the_timed_region": graph_fallbacks[0],
                               **({"encode_leg_in_the_capture": True, "encoded_ids_equal_eager_encode": enc_same,
                                   "encode_precision": a.encode_precision,
                                   "encode_windows_re_encoded_in_f32": cg.enc_redone}
                                  if enc is not None else {}),
                               "text_side_captured_first": bool(not knn_g.audio_first and knn_g.audio_first is not None),
                               "other_seed_equals_eager": same2,
                               **({"segments": list(cg.segment_kinds)} if cg.segmented else {})}
    if (mixed and not sharded_run and world == 1 and can_pipe and pipe is None and not a.no_f64_line and
            os.environ.get("QPG_BENCH_NO_PIPELINED", "") != "1"):
        # Clips in flight (round 6: GraphPipeline).  Round 5's figure here came from ClipPipeline's eager lanes: host-bound
        # (~0.25 ms of Python per clip) and without overlap - a one-chunk sweep holds every register of every CU, so another
        # clip's tail kernels advance by one launch per sweep.  Now: `clips_per_replay` independent clips per captured graph
        # (one batched sweep - the database image out of HBM once, out of the XCD's L2 for the other chunks - batched selects,
        # ONE batched walk: the ~100 us post-sweep chain is paid once per replay) on `lanes` streams (a multi-chunk sweep
        # retires its blocks in rounds, the other lane's tail gets CUs meanwhile).  Every clip's integers are on the host
        # inside the timed region; the clips of a group are DIFFERENT clips and each is compared with the one-clip path.
        # A throughput figure beside the one-clip-at-a-time `value`, not a replacement for it (a clip's latency goes up).
        from qpgesture_amd.code_knn import GraphPipeline
        out["pipelined"] = {}
        # (staggered lanes: a replay is launched when the other lane's SWEEP is over, so the sweeps follow one another and a
        # lane's tail runs under the other lane's sweep - profiles/r06_pipeline_timeline_*.md; it pays at 4 clips per replay,
        # not at 8, where a lane's sweep is long against its tail either way: tools/bench_graph_pipeline.py)
        for key_, G_, L_, S_ in (("", int(os.environ.get("QPG_BENCH_PIPE_CLIPS", "4")),
                                  int(os.environ.get("QPG_BENCH_PIPE_LANES", "2")), True), ("deeper", 8, 2, False)):
            gp = GraphPipeline(db, M, clips_per_replay=G_, depth=L_, rng=np.random.RandomState(123456), stagger=S_)
            for ln in gp.lanes:
                ln["knn"].overlap_sweeps = knn.overlap_sweeps
                ln["knn"].audio_precision = knn.audio_precision
                ln["knn"].audio_kernel = knn.audio_kernel
            gclips = [clips[0]] + [synth.make_db(M, 2000 + r) for r in range(1, G_)]
            gi = torch.from_numpy(np.concatenate([interp_wavlm(c["wavlm"]) for c in gclips])).to(dev)
            gc_ = torch.from_numpy(np.concatenate([c["context"].squeeze(2) for c in gclips])).to(dev)
            want_g = [codes.numpy().reshape(-1).astype(np.int64)] + [
                knn.match_clip(gi[c * M:(c + 1) * M], gc_[c * M:(c + 1) * M], M, seed_code=seed_code,
                               seed_phase=seed_phase)[0].reshape(-1) for c in range(1, G_)]
            for l_ in range(L_):
                ba_, bc_ = gp.buffers(l_)
                ba_.copy_(gi)
                bc_.copy_(gc_)

            def run_g(n, gp=gp, L_=L_):
                pending, res = [], None
                for _ in range(n):
                    if len(pending) == L_:
                        res = gp.collect(pending.pop(0))
                    pending.append(gp.submit(None, None, seed_code, seed_phase))
                while pending:
                    res = gp.collect(pending.pop(0))
                return res
            n_g = max(10, 120 // G_)
            tw_ = time.perf_counter()
            while time.perf_counter() - tw_ < 0.3:      # (clock / power steady state: the first of the three runs was 10 % slow)
                run_g(max(5, n_g // 3))
            gc.collect()
            gc.disable()
            d7s = []
            for _ in range(3):              # ~20 ms each: one hiccup of the host shows - best of three, all three reported
                fence()
                t7 = time.perf_counter()
                rg = run_g(n_g)
                fence()
                d7s.append(time.perf_counter() - t7)
            fence()
            t8 = time.perf_counter()
            for _ in range(10):
                gp.collect(gp.submit(None, None, seed_code, seed_phase))     # one group at a time: a group's latency
            fence()
            lat = (time.perf_counter() - t8) / 10
            gc.enable()
            d7 = min(d7s)
            rec_ = {"clips_in_flight": G_ * L_, "clips_per_replay": G_, "lanes": L_, "staggered_lanes": bool(gp.stagger),
                    "steps": n_g * G_,
                    "ms_per_step": round(d7 / (n_g * G_) * 1e3, 4),
                    "ms_per_step_all_three_runs": [round(x / (n_g * G_) * 1e3, 4) for x in d7s],
                    "frames_per_s": round(frames_per_step * n_g * G_ / d7, 1),
                    "latency_ms_per_group_one_at_a_time": round(lat * 1e3, 4),
                    "codes_equal_default_path": bool(all(np.array_equal(rg[c][0].reshape(-1), want_g[c]) for c in range(G_))),
                    "distinct_clips_per_group": len({tuple(w) for w in want_g}),
                    "rematched_steps": gp.rematched,
                    "step": "one clip of a group; a replay = one hipGraph of %d clips (ClipGraph(n_clips)), %d replays in "
                            "flight on their own streams (code_knn.GraphPipeline)" % (G_, L_)}
            if key_:
                out["pipelined"][key_] = rec_
            else:
                out["pipelined"].update(rec_)
            del gp
    if serial is not None:
        serial["roofline_frac"] = (round(alg_bytes / (serial["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if hl else
                                   round(flops / (serial["kernel_ms"] * 1e-3) / 1e12 / peak, 4))
        out["one_clip_at_a_time"] = serial
    if mixed and sharded_run:
        st = knn.mixed_stats()
        fl = knn._guard_stats.cpu().numpy()
        n_run = a.steps + a.warmup + prewarm["steps"]
        out["mixed_precision"] = {"shard_f64_dot_pairs_per_step": round(st["tier1_pairs"] / n_run, 1),
                                  "cross_shard_reevaluations_per_step": round(int(fl[3]) / n_run, 1),
                                  "flags": st["flags"], "error_bound": 1.3e-6 if hl else 2.05e-6}
    if mixed and not sharded_run:
        # re-evaluation activity of the timed steps (+ warm-up) and the same clip through the f64 sweep: the mixed path
        # must return the same codes (its tables differ only inside the sweep's error bound)
        st = knn.mixed_stats()
        if pipe is not None:                      # the timed clips ran on the lanes' matchers
            ls = [ln["knn"].mixed_stats() for ln in pipe.lanes]
            st = {"tier1_pairs": sum(x["tier1_pairs"] for x in ls), "tier2_pairs": sum(x["tier2_pairs"] for x in ls),
                  "flags": int(np.bitwise_or.reduce([x["flags"] for x in ls]))}
        n_run = a.steps + a.warmup + prewarm["steps"]
        if graph_mode:                             # the timed steps ran on the capture's matcher
            sg = knn_g.mixed_stats()
            st = {"tier1_pairs": sg["tier1_pairs"], "tier2_pairs": sg["tier2_pairs"], "flags": sg["flags"] | st["flags"]}
            n_run += 2                                 # (the capture's two warm-up passes)
            if sr is not None:                         # (two captures took turns: the steps are split between their matchers)
                s2 = knn_g2.mixed_stats()
                st = {"tier1_pairs": st["tier1_pairs"] + s2["tier1_pairs"], "tier2_pairs": st["tier2_pairs"] + s2["tier2_pairs"],
                      "flags": st["flags"] | s2["flags"]}
                n_run += 2 + 1                         # (+ its warm-up passes, + the other-seed check's replay on the first)
        k64 = CodeKNN(db, rng=np.random.RandomState(123456))
        k64.audio_precision = "f64"
        T64 = k64.sweep_tables(te_interp, te_ctx, M * n_sweep_clips)
        same = True
        for c in range(my_clips):
            w64 = k64.walk(T64, M, window_offset=c * M, seed_code=seed_code, seed_phase=seed_phase_d)[0]
            same = same and bool(np.array_equal(np.asarray(w64).reshape(-1),
                                                codes.numpy().reshape(my_clips, -1)[c].astype(np.int64)))
        Tm = knn.sweep_tables(te_interp, te_ctx, M * n_sweep_clips)
        out["mixed_precision"] = {
            "f64_dot_pairs_per_step": round(st["tier1_pairs"] / n_run, 1),
            # the timed steps' select settles in f64 only what the walk can read (QPG_RANK_CUT=0: everything);
            # the tables compared with the f64 sweep below are the fully settled ones, the CODES compared are the timed steps'
            "walk_relevance_cut": cut_used,
            "reference_arithmetic_pairs_per_step": round(st["tier2_pairs"] / n_run, 2),
            "flags": st["flags"], "error_bound": 1.3e-6 if hl else 2.05e-6,
            # the one measured constant under that bound, re-measured at load time on THIS device (selfcheck.py)
            "mfma_selfcheck": {k_: db.hl_bound_report.get(k_) for k_ in ("kappa", "kappa2", "kappa2_assumed", "kappa2_limit",
                                                                         "kappa6", "kappa4", "kappa6_assumed", "kappa16", "kappa16_assumed",
                                                                         "subnormals_exact",
                                                                         "skipped")},
            "max_table_difference_vs_f64_sweep": float((Tm["aud_d"] - T64["aud_d"]).abs().max()),
            "winners_equal_f64_sweep": bool(torch.equal(Tm["aud_idx"], T64["aud_idx"])),
            "ranks_equal_f64_sweep": bool(torch.equal(Tm["aud_rank"], T64["aud_rank"])),
            "codes_equal_f64_sweep": same}
    if world > 1 and not replicated and not strong and not a.no_replicated and not force_sharded and CL == 1:
        out["replicated"] = replicated_leg(a, dev, world, rank, N, M, code, phase, sig, te_interp_all, te_ctx_all,
                                           seed_code, seed_phase_d, codes if a.check else None)
    if not a.no_vqvae:
        out.update(vqvae_bench(dev, a, world, rank))
    if rank == 0 and world == 1 and CL == 1 and not a.no_cold and fb == 4:
        out["cold"] = cold_bench(dev, N, M)
    if rank == 0 and world == 1 and CL == 1 and not a.no_e2e and not a.no_cold and fb == 4 and a.data == "gaussian":
        out["e2e_cli"] = e2e_cli_bench(dev, N, M)
    if rank == 0 and world == 1 and not a.no_cpu_baseline and fb == 4:
        out["cpu_baseline"] = cpu_baseline(a, code, clips[0], M, N)
        out["vs_cpu_baseline"] = round(value / out["cpu_baseline"]["value"], 1)
    if a.check:
        # every rank re-matches ITS clips against the WHOLE database on its own (world=1 semantics) and must
        # get the same codes as the sharded + exchanged run above
        full_i, full_c = chunked_db(N, 0, N, seed=0)
        db1 = GestureDB(code, full_i, full_c, phase, sig, device=dev, feature_dtype=a.feature_dtype)
        k1 = CodeKNN(db1, rng=np.random.RandomState(123456))
        first = 0 if strong else rank * CL
        want = []
        for c in range(my_clips):
            w0 = (first + c) * M
            T1 = k1.sweep_tables(te_interp_all[w0:w0 + M], te_ctx_all[w0:w0 + M], M)
            want.append(k1.walk(T1, M, 0, seed_code=seed_code, seed_phase=seed_phase_d)[0])
        ok = bool(np.array_equal(np.concatenate(want), codes.numpy().astype(np.int64)))
        out["check"] = ok
        assert ok, "rank %d: sharded result differs from the single-rank result" % rank
    if (rank == 0 and world == 1 and default_shape and a.workload == "match" and a.data == "gaussian" and not force_sharded
            and a.clips_in_flight == 1 and not a.encode_batch and not a.no_sub_records and a.audio_precision == "mixed"):
        out["sub_records"] = sub_records(dev)
    if rank == 0:
        print(json.dumps(out), file=JSON_OUT, flush=True)
    if world > 1 or force_sharded:
        dist.destroy_process_group()


# The other BASELINE.json configurations in the driver's line (cfg-3, configs[3] on one GPU and
# configs[4] existed only as builder-run lines under profiles/).  Each entry is THIS script run again as a child process with
# that configuration's flags (so the numbers are exactly what `python bench.py <flags>` prints) and a short timed region;
# the record keeps the fields a reader needs.  The children run after the parent's own timed regions, one at a time.
SUB_RUNS = [
    ("cfg3", "BASELINE configs[2]: 100 000 x 512-d rows, 512 codes, 1 000 queries - per-code min + argmin",
     ["--workload", "cfg3", "--steps", "30", "--warmup", "5"]),
    ("strong8192", "BASELINE configs[3] on ONE GPU: one 24 s clip vs N_db = 8192 windows (the whole speaker-1-class DB on one "
                   "device; the 8-GPU row-sharded form needs a node)",
     ["--scaling", "strong", "--steps", "40", "--warmup", "5"]),
    ("clips16_f16_enc96", "BASELINE configs[4]: 16 clips per replay, fp16 features, 96 pose windows VQ-VAE-encoded in the step",
     ["--clips", "16", "--feature-dtype", "f16", "--encode-batch", "96", "--steps", "20", "--warmup", "3"]),
    ("clips16_f16_enc96_f16x3", "the same with the encoder on the split-f16 convolutions under their margin check (windows the "
                                "bound cannot vouch for are re-encoded in f32 inside the step: ids identical)",
     ["--clips", "16", "--feature-dtype", "f16", "--encode-batch", "96", "--encode-precision", "f16x3", "--steps", "20",
      "--warmup", "3"]),
]
SUB_COMMON = ["--gpus", "1", "--no-sub-records", "--no-cpu-baseline", "--no-vqvae", "--no-cold", "--no-e2e", "--no-f64-line"]


def sub_records(dev):
    import subprocess
    out = {}
    env = dict(os.environ)
    for name, what, flags in SUB_RUNS:
        if os.environ.get("QPG_BENCH_SUB", "") and name not in os.environ["QPG_BENCH_SUB"].split(","):
            continue
        cmd = [sys.executable, os.path.abspath(__file__)] + flags + SUB_COMMON
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
            line = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not line:
                out[name] = {"what": what, "error": "rc %d: %s" % (r.returncode, r.stderr.decode()[-300:])}
                continue
            d = json.loads(line[-1])
        except (subprocess.TimeoutExpired, ValueError) as e:
            out[name] = {"what": what, "error": repr(e)[:300]}
            continue
        rf = d.get("roofline", {})
        rec = {"what": what, "command": "python bench.py " + " ".join(flags),
               "metric": d.get("metric"), "value": d.get("value"), "unit": d.get("unit"), "ms_per_step": d.get("ms_per_step"),
               "steps": d.get("steps"), "step_mode": d.get("step_mode"), "dtype": d.get("dtype"),
               "roofline": {k: rf.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes",
                                                    "algorithmic_gflop", "kernel_ms", "issued_tflops_f16", "issued_frac")
                            if k in rf},
               "wall_s": round(time.perf_counter() - t0, 1)}
        if isinstance(rf.get("hbm"), dict):
            rec["roofline"]["hbm"] = {k: rf["hbm"].get(k) for k in ("achieved", "frac", "algorithmic_bytes")}
        if isinstance(rf.get("mfma"), dict):
            rec["roofline"]["mfma_issued_frac_of_f16_peak"] = rf["mfma"].get("frac")
        if "tables_equal_exact_sweep" in d:
            rec["tables_equal_exact_sweep"] = d["tables_equal_exact_sweep"]
        mp_ = d.get("mixed_precision") or {}
        for k in ("codes_equal_f64_sweep", "winners_equal_f64_sweep", "ranks_equal_f64_sweep"):
            if k in mp_:
                rec[k] = mp_[k]
        gr = d.get("graph_replay") or {}
        for k in ("other_seed_equals_eager", "encoded_ids_equal_eager_encode", "clips_per_replay", "encode_precision",
                  "encode_windows_re_encoded_in_f32"):
            if k in gr:
                rec[k] = gr[k]
        if "eager" in d and d["eager"]:
            rec["eager_ms_per_step"] = d["eager"].get("ms_per_step")
            rec["codes_equal_eager_steps"] = d["eager"].get("codes_equal_graph_steps")
        if "realtime_factor" in d:
            rec["realtime_factor"] = d["realtime_factor"]
        out[name] = rec
    return out


def rocprof_kernel_ms(key):
    """The dominant kernel's average duration over ALL launches of a profiled run of this command - graph replays included,
    which HIP events cannot bracket - from the committed rocprofv3 --kernel-trace --stats summary (profiles/
    kernel_replay.json, written by tools/kernel_replay.py from the last evidence pass: experiments/round_scripts/r06_gpu_pass.sh); null for shapes that were not profiled."""
    path = os.path.join(ROOT, "profiles", "kernel_replay.json")
    if key is None or not os.path.exists(path):
        return {"kernel_ms_rocprof": None}
    rec = json.load(open(path)).get(key)
    if rec is None:
        return {"kernel_ms_rocprof": None}
    return {"kernel_ms_rocprof": round(rec["avg_us"] / 1e3, 4),
            "kernel_ms_rocprof_source": "profiles/kernel_replay.json [%s]: rocprofv3 --kernel-trace --stats of `%s`, %d "
                                        "launches (not re-measured per run)" % (key, rec.get("command", "bench.py"), rec["calls"])}


def pmc_traffic(key):
    """`roofline.traffic` = HBM bytes per launch of the dominant kernel from the rocprofv3 PMC passes (FETCH_SIZE x 2 per the
    gfx950 correction + WRITE_SIZE, separate passes), read at run time from the committed summary profiles/pmc_traffic.json
    (written by tools/pmc_traffic.py from the passes of experiments/round_scripts/r04_gpu_pass.sh); null for shapes that were not measured."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if key is None or not os.path.exists(path):
        return {"traffic": None, "traffic_source": None}
    rec = json.load(open(path)).get(key)
    if rec is None:
        return {"traffic": None, "traffic_source": None}
    return {"traffic": int(rec["hbm_bytes_per_launch"]),
            "traffic_source": "profiles/pmc_traffic.json [%s]: FETCH_SIZE %.0f KB x 1024 x 2 + WRITE_SIZE %.0f KB x 1024, "
                              "rocprofv3 --pmc passes of this launch shape (not re-measured per run)"
                              % (key, rec["fetch_size_kb"], rec["write_size_kb"])}


# HBM bytes per audio_cosine_f64_kernel launch at the default shape (N_db=2048, Q=48), rocprofv3 PMC, separate
# --pmc FETCH_SIZE / WRITE_SIZE passes with the gfx950 x2 correction on FETCH_SIZE: profiles/r02_pmc_audio.md
TRAFFIC_SOURCE = "profiles/r02_pmc_audio.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this launch shape; not re-measured per run)"
HL_TRAFFIC_SOURCE = "profiles/r03_pmc_audio_hl.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this launch shape; not re-measured per run)"
AUDIO_HL_TRAFFIC_BYTES = 709_000_000      # split-operand f16 sweep, dense image: FETCH_SIZE 341.4e3 KB x 1024 x 2 + WRITE_SIZE 9 984 KB x 1024
AUDIO_TRAFFIC_BYTES = 923_000_000
AUDIO_MX_TRAFFIC_BYTES = 1_023_000_000    # mixed-precision sweep (one launch: mx2 blocks + split-K remainder), same file


def ranks_seen(dev, world, rank, host_staged):
    """Proof that N ranks on N devices took part: an all-gather of every rank's device UUID (16 bytes) over the job's
    process group (RCCL for the driver's runs)."""
    import hashlib
    import torch
    import torch.distributed as dist
    props = torch.cuda.get_device_properties(dev)
    raw = getattr(props, "uuid", None)
    try:
        b = raw.bytes if raw is not None else None
    except AttributeError:
        b = None
    if b is None:
        b = hashlib.md5(("%s|%s|%d" % (props.name, getattr(props, "pci_bus_id", "?"), dev.index or 0)).encode()).digest()
    mine = torch.tensor(list(b[:16]), dtype=torch.uint8, device="cpu" if host_staged else dev)
    allb = torch.empty((world * 16,), dtype=torch.uint8, device=mine.device)
    if host_staged or world == 1:
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        allb = torch.cat(parts)
    else:
        dist.all_gather_into_tensor(allb, mine)
    ids = [bytes(allb[i * 16:(i + 1) * 16].cpu().tolist()).hex() for i in range(world)]
    return {"ranks": world, "distinct_devices": len(set(ids)), "device_uuids": ids, "backend": dist.get_backend()}


def replicated_leg(a, dev, world, rank, N, M, code, phase, sig, te_interp, te_ctx, seed_code, seed_phase_d, want):
    """The clip-parallel mode beside the row-sharded figure, in the same run: the WHOLE database on every GPU (0.68 GB
    image + 1.5 GB base of 288 GB), rank r matches clip r, no collective in the step (BASELINE.json
    configs[4]'s throughput axis).  Same timing contract: barrier + synchronise on both sides, MAX over ranks."""
    import gc
    import torch
    import torch.distributed as dist
    from qpgesture_amd.code_knn import CodeKNN, GestureDB
    full_i, full_c = chunked_db(N, 0, N, seed=0)
    dbr = GestureDB(code, full_i, full_c, phase, sig, device=dev, feature_dtype=a.feature_dtype)
    del full_i, full_c
    kr = CodeKNN(dbr, rng=np.random.RandomState(123456))
    kr.audio_precision = a.audio_precision
    ti = te_interp[rank * M:(rank + 1) * M].contiguous()
    tc = te_ctx[rank * M:(rank + 1) * M].contiguous()

    # the same step as N = 1: one hipGraph replay per clip (--step-mode eager / --no-graph: one launch per kernel)
    use_graph = a.step_mode != "eager" and not a.no_graph
    sp_np = seed_phase_d.cpu().numpy()
    cg = kr.capture_clip_graph(M, audio=ti, context=tc) if use_graph else None

    def one():
        if cg is not None:
            arr = cg.run_ints(seed_code, sp_np)
            if arr[-1] == 0:
                kr.check_status(arr[-2:])
                return arr[:M * 30]
            # the trouble word came out with the codes: this clip again on the uncapped path
            prev, kr.audio_precision = kr.audio_precision, "exact"
            kr.clear_flags()
            try:
                T = kr.sweep_tables(ti, tc, M, for_walk=True)
                arr = kr.walk(T, M, seed_code=seed_code, seed_phase=seed_phase_d, sync="ints")
            finally:
                kr.audio_precision = prev
            kr.check_status(arr[-2:])
            return arr[:M * 30]
        T = kr.sweep_tables(ti, tc, M, for_walk=True)
        arr = kr.walk(T, M, seed_code=seed_code, seed_phase=seed_phase_d, sync="ints")
        kr.check_status(arr[-2:])
        return arr[:M * 30]
    for _ in range(30):
        got = one()
    gc.collect()
    gc.disable()
    torch.cuda.synchronize(dev)
    dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        got = one()
    torch.cuda.synchronize(dev)
    dist.barrier()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    gc.enable()
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if dist.get_backend() == "gloo":
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.MAX)
        dt = float(h.item())
    else:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    res = {"value": round(240 * M * world * a.steps / dt, 1), "unit": "frames/s", "ms_per_step": round(dt / a.steps * 1e3, 4),
           "steps": a.steps, "scaling": "weak", "collectives_per_step": 0, "step_mode": "graph" if cg is not None else "eager",
           "parallelism": "replicated DB x%d, clip-parallel, no collective" % world}
    if want is not None:
        res["codes_equal_row_sharded"] = bool(np.array_equal(got.reshape(-1), want.numpy().reshape(-1)))
    return res


def e2e_cli_bench(dev, N, M):
    """BASELINE.json configs[0] / [1] end to end: the drop-in command line (qpgesture_amd/GestureKNN.py main, the
    reference's flags) from eight .npz files on disk to result.npz - np.load, H2D, device-side resample, DB build, match,
    np.savez_compressed - for both --tie_rule values (`numpy`, the default: both (Q,512) tables are ranked by the
    reference's own NumPy call on the host in the middle of the step).  The phase track is written DENSE (f32
    (N,240,4,8), accepted by the loader): the reference's object array of 245 760 pickled torch tensors per 256 windows
    costs ~65 ms per DB window to unpickle on either side and says nothing about this path.  PCIe- and disk-inclusive:
    never `value`."""
    import contextlib
    import io
    import shutil
    import tempfile
    from qpgesture_amd import synth
    from qpgesture_amd import GestureKNN as cli
    td = tempfile.mkdtemp(prefix="qpg_e2e_")
    try:
        t0 = time.perf_counter()
        paths = synth.write_npz_set_dense(td, N, M, chunked_db)
        t_write = time.perf_counter() - t0
        res = {"n_db": N, "windows": M, "npz_bytes": int(sum(os.path.getsize(v) for v in paths.values())),
               "write_s": round(t_write, 2)}
        ref = None
        cdir = os.path.join(td, "cache")
        for rule in ("numpy", "stable"):
            # three invocations per rule: the first finds no prepared database (builds from the .npz files, then writes the
            # cache BEHIND its result: `seconds_first_run` excludes nothing), the second and third restore it
            ts, cached = [], []
            for rep in range(3):
                outp = os.path.join(td, "result_%s.npz" % rule)
                argv = []
                for k, v in paths.items():
                    argv += ["--" + k, v]
                argv += ["--out_knn_filename", outp, "--tie_rule", rule, "--device", str(dev), "--db_cache_dir", cdir]
                t1 = time.perf_counter()
                buf = io.StringIO()
                with contextlib.redirect_stdout(buf):
                    cli.main(argv)
                ts.append(time.perf_counter() - t1)
                cached.append("prepared-database cache)" in buf.getvalue())
            pred = np.load(outp)["knn_pred"]
            assert pred.shape == (M, 30) and pred.dtype == np.int64
            res["tie_rule_" + rule] = {"seconds": round(min(ts[1:]), 3), "seconds_first_run": round(ts[0], 3),
                                       "restored_from_cache": cached,
                                       "frames_per_s": round(240 * M / min(ts[1:]), 1)}
            if ref is None:
                ref = pred
            else:
                res["same_codes_both_rules"] = bool(np.array_equal(ref, pred))
        res["cache_bytes"] = int(sum(os.path.getsize(os.path.join(cdir, f)) for f in os.listdir(cdir)))
        # the no-cache figure (round 4's `seconds`): every invocation re-reads and re-packs the database
        t1 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            cli.main(argv[:-2] + ["--db_cache", "off"])
        res["seconds_without_cache"] = round(time.perf_counter() - t1, 3)
        return res
    finally:
        shutil.rmtree(td, ignore_errors=True)


def vqvae_bench(dev, a, world, rank):
    """Second half of BASELINE.json's metric: gesture VQ-VAE encode (and decode) frames/s.  Each rank encodes
    its own batch of 256 pose windows (240 frames x 135: dataset_to_code over a speaker DB runs in such
    batches) — pure data parallel, no collective — with the full-size codebook.yml architecture and seeded
    weights; the decode leg decodes one 24 s clip's worth of codes per rank in one pass (1440 frames).
    Timing: workspaces are allocated by the warm-up calls; 50 encode iterations, each bracketed by HIP events on the
    launch stream (min / median reported) inside one wall-clocked loop (the frames/s figure)."""
    import torch
    import torch.distributed as dist
    from qpgesture_amd import synth
    from qpgesture_amd.vqvae import VQVAE
    model = VQVAE(None, 135, device=dev).load_state_dict(synth.make_vqvae_state_dict(7))
    Bw = 256
    x = torch.randn((Bw, 240, 135), device=dev)
    ids = torch.randint(0, 512, (1, 180), device=dev)

    def timed(fn, iters, warm):
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
        import gc
        gc.collect()                     # (before the warm-up: the GPU does not idle between it and the timed iterations)
        gc.disable()
        for _ in range(warm):
            fn()
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for e0, e1 in ev:
            e0.record()
            fn()
            e1.record()
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        gc.enable()
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        per = sorted(e0.elapsed_time(e1) for e0, e1 in ev)
        return dt / iters, per[0], per[len(per) // 2]
    te, te_min, te_med = timed(lambda: model.encode(x), 50, 8)
    td, td_min, td_med = timed(lambda: model.decode([ids]), 50, 30)
    enc_flop = 1.639e9 * Bw                                        # 1.639 GFLOP per 240-frame window
    dec_flop = 1.908e9 * 6                                         # 1.908 GFLOP per 240 output frames
    res = {"vqvae_encode_frames_per_s": round(240 * Bw * world / te, 1),
           "vqvae_encode_ms_per_batch256": round(te * 1e3, 3),
           "vqvae_encode_tflops_f32": round(enc_flop / te / 1e12, 2),
           # roofline of the encode leg: flops of SURVEY §8d over the HIP-event time of the whole launch sequence
           # (19 kernels: csrc/qpg_convt.hip + the quantiser argmin) against the f32 matrix peak
           "vqvae_roofline": {"bound": "mfma", "achieved": round(enc_flop / (te_med * 1e-3) / 1e12, 2),
                              "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": round(enc_flop / (te_med * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS, 4),
                              "kernel": "resblock_fused_f32_kernel + convt_f32_kernel (encode, B=256)",
                              "ms_median": round(te_med, 3), "ms_min": round(te_min, 3), "iters": 50,
                              "algorithmic_gflop": round(enc_flop / 1e9, 1)},
           "vqvae_decode_frames_per_s": round(1440 * world / td, 1),
           "vqvae_decode_ms_per_24s_clip": round(td * 1e3, 3),
           "vqvae_decode_ms_median": round(td_med, 3),
           "vqvae_decode_tflops_f32": round(dec_flop / td / 1e12, 2)}
    # beside the f32 encode: the split-operand f16 encoder under its margin check (ids identical to the f32 path's by
    # construction: windows the bound cannot vouch for are re-encoded in f32 inside the timed call)
    try:
        t16e, _, t16e_med = timed(lambda: model.encode_f16x3(x), 20, 4)
        ids16, st16 = model.encode_f16x3(x, return_stats=True)
        res["vqvae_encode_f16x3"] = {"ms_per_batch256": round(t16e * 1e3, 3), "frames_per_s": round(240 * Bw * world / t16e, 1),
                                     "windows_re_encoded_in_f32": int(st16["windows_re_encoded_in_f32"]),
                                     "ids_equal_f32_path": bool(torch.equal(ids16, model.encode(x)[0]))}
    except Exception as e_:                                   # noqa: BLE001
        res["vqvae_encode_f16x3"] = {"error": repr(e_)[:200]}
    # beside the f32 decode: the split-operand f16 decoder on the signature pass's shape (VisualizeCodebook.py:93-116: every
    # code's 30-code window, 512 decodes batched as B = 512) - and, honestly, on the 24 s clip, where its general kernel's
    # ~22 us per launch loses to the short-sequence f32 kernels
    try:
        ids512 = torch.arange(512, device=dev).view(512, 1).repeat(1, 30).contiguous()
        t32b, _, _ = timed(lambda: model.decode([ids512]), 10, 3)
        t16b, _, _ = timed(lambda: model.decode_f16x3([ids512]), 10, 3)
        t16c, _, _ = timed(lambda: model.decode_f16x3([ids], force=True), 10, 3)
        o32, (o16, st16d) = model.decode([ids512]), model.decode_f16x3([ids512], return_stats=True)
        res["vqvae_decode_f16x3"] = {"signature_pass_512x30_ms": round(t16b * 1e3, 3), "signature_pass_512x30_ms_f32": round(t32b * 1e3, 3),
                                     # (the split-f16 kernels forced onto ONE clip: slower than decode() - which is why
                                     # decode_f16x3 routes fewer than F16X3_MIN_POSITIONS code positions to decode() since round 6)
                                     "clip_24s_ms_forced_split_f16": round(t16c * 1e3, 3), "clip_24s_ms_f32": round(td * 1e3, 3),
                                     "clip_24s_routed_to_f32_kernels": True,
                                     "max_abs_pose_difference_vs_f32": float((o32 - o16).abs().max()),
                                     "redone_in_f32": bool(st16d["activation_outside_f16_range"])}
    except Exception as e_:                                   # noqa: BLE001
        res["vqvae_decode_f16x3"] = {"error": repr(e_)[:200]}
    # training step (codebook/train.py:120-131) at the reference's batch size of 256 windows per rank: forward with
    # EMA codebook update, backward, flat-gradient all-reduce (N > 1), Adam
    from qpgesture_amd.optim import Adam
    tm = VQVAE(dict(vel=1, acc=1), 135, device=dev).load_state_dict(synth.make_vqvae_state_dict(7)).train()
    opt = Adam(tm.parameters(), lr=3e-5, betas=(0.5, 0.999))

    def train_step():
        tm(x)
        tm.backward(sync_grads=True)
        opt.step()
    tt, _, _ = timed(train_step, 8, 3)
    train_flop = 3 * (1.639e9 + 1.908e9) * Bw                      # forward + data-gradient + weight-gradient GEMMs
    res.update({"vqvae_train_windows_per_s": round(Bw * world / tt, 1),
                "vqvae_train_ms_per_step_b256": round(tt * 1e3, 3),
                "vqvae_train_tflops_f32": round(train_flop / tt / 1e12, 2)})
    # BESIDE the f32 figure, never instead of it: the same step with its FORWARD convolutions on the split-operand f16
    # kernels (VQVAE.train_precision = "f16x3", train.py --train_precision f16x3; backward and optimiser unchanged, the weight
    # images re-packed from the updated weights every step inside the timed region)
    tm.train_precision = "f16x3"
    try:
        t16, _, _ = timed(train_step, 8, 3)
        res["vqvae_train_f16x3_forward"] = {"ms_per_step_b256": round(t16 * 1e3, 3),
                                            "windows_per_s": round(Bw * world / t16, 1),
                                            "note": "opt-in; x_out within 3e-5 and gradients within 1e-4 (relative, in norm) "
                                                    "of the f32 step's: tests/test_gpu_vqvae_train.py"}
    except Exception as e_:                                   # noqa: BLE001 (recorded, the f32 figures stand)
        res["vqvae_train_f16x3_forward"] = {"error": repr(e_)[:200]}
    return res


def cold_bench(dev, N, M):
    """SURVEY §8d 'report both cold and hot': (a) host arrays as load_db_codebook leaves them -> resident GestureDB
    (chunked H2D of the raw (N,199,1024) WavLM track, device-side 199->180 resample, norms / packing / rank tables);
    (b) one clip handed over as HOST arrays (raw WavLM + context, pageable memory) -> codes back on the host.
    PCIe-inclusive figures: never `value`."""
    import torch
    from qpgesture_amd import synth
    from qpgesture_amd.code_knn import CodeKNN, GestureDB
    from qpgesture_amd.data_processing import interp_wavlm_device
    rng = np.random.default_rng(0)
    wavlm = rng.standard_normal((N, 199, 1024), dtype=np.float32)
    ctx = rng.standard_normal((N, 30, 384), dtype=np.float32)
    phase = rng.standard_normal((N, 240, 4, 8), dtype=np.float32)
    code, sig = synth.make_codes(N, 2), synth.make_signature(3)
    best = None
    for _ in range(2):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        interp = interp_wavlm_device(wavlm, dev)
        db = GestureDB(code, interp, ctx, phase, sig, device=dev)
        torch.cuda.synchronize(dev)
        t = (time.perf_counter() - t0) * 1e3
        best = t if best is None else min(best, t)
    knn = CodeKNN(db, rng=np.random.RandomState(1))
    clip_w = rng.standard_normal((M, 199, 1024), dtype=np.float32)
    clip_c = rng.standard_normal((M, 30, 384), dtype=np.float32)
    sc, sp = knn.init_code_phase()
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        ti = interp_wavlm_device(clip_w, dev)
        tc = torch.from_numpy(clip_c).to(dev)
        knn.match_clip(ti, tc, M, seed_code=sc, seed_phase=sp)
        ts.append((time.perf_counter() - t0) * 1e3)
    del db, interp
    return {"db_build_ms": round(best, 1), "db_build_gbs_h2d_inclusive": round(wavlm.nbytes / best / 1e6, 1),
            "clip_pcie_inclusive_ms": round(min(ts[1:]), 3),
            "clip_pcie_inclusive_frames_per_s": round(240 * M / (min(ts[1:]) * 1e-3), 1),
            "note": "host arrays (pageable) -> device: DB = raw (N,199,1024) WavLM track + context + phase; clip = raw "
                    "WavLM + context in, (M,30) codes out"}


def cfg3_bench(a, dev, world, rank):
    """BASELINE.json configs[2] / SURVEY §8d cfg-3: DB X f32 (100 000, 512) ~N(0,1) seed 0, code ids uniform [0,512)
    seed 1, validity mask Bernoulli(0.9) seed 2, queries (1000, 512) seed 3; per (query, code) min cosine distance +
    argmin.  Runs the exact-f32 (sklearn-order) sweep with the per-code min fused into its epilogue: the Q x C
    distance matrix never reaches HBM."""
    import torch
    from qpgesture_amd import cfg3
    return cfg3.bench(a, dev, world, rank, HBM_PEAK_GBS)


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
    are linear in DB windows, so frames/s at full N_db = 1440 / (t * N_db/sample).
    Five repeats; `value` is the MEDIAN pass (round 6, VERDICT r5 weak #13: best-of-three of a 0.8 s OpenMP sample on 256 cores
    wandered 1 724 -> 2 144 -> 1 735 frames/s across rounds on the same code), the fastest pass rides beside it; `cores` is
    the number of OpenMP threads the scans were given (= the host's logical cores)."""
    from oracle import cref, knn_oracle as O
    from qpgesture_amd.data_processing import interp_wavlm
    cref.build()
    ns = min(a.cpu_sample, N)
    interp, ctx = chunked_db(N, 0, ns, seed=0)
    te = interp_wavlm(clip["wavlm"])
    q = np.stack([O.wavlm_feat_rows(te, w, [24 * s])[0] for w in range(M) for s in range(8)])
    qt = np.stack([clip["context"].squeeze(2)[w][int(24 * s / 180 * 30)] for w in range(M) for s in range(8)])
    cores = os.cpu_count() or 1
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        cref.audio_scan(interp, np.arange(26) * 6, code[:ns], np.arange(26), q, n_threads=cores)
        cref.text_scan(ctx, np.arange(26), code[:ns], np.arange(26), qt, n_threads=cores)
        ts.append(time.perf_counter() - t0)
    t = float(np.median(ts))
    full = t * N / ns
    faithful = faithful_loop(interp, ctx, code[:ns], te, clip["context"].squeeze(2), q, qt)
    return {"value": round(240 * M / full, 2), "unit": "frames/s", "cores": cores, "kind": "port",
            "value_fastest_pass": round(240 * M / (min(ts) * N / ns), 2), "threads": cores,
            "faithful_loop": faithful,
            "sample": "audio+text scans of the same %d queries vs the first %d of %d DB windows "
                      "(median of 5 passes: %.2f s, all five %s; scaled linearly to N_db); C port of the reference arithmetic "
                      "(oracle/sweep_ref.c), OpenMP with %d threads; matching walk excluded (<1%% of CPU time)"
                      % (8 * M, ns, N, t, ["%.2f" % x for x in ts], cores),
            "sample_seconds": round(t, 3)}


def faithful_loop(interp, ctx, code, te_interp, te_ctx, q, qt, n_win=4):
    """SURVEY §8d CPU baseline (2): the reference's OWN cost structure on this host - a Python loop over candidates that
    calls sklearn.metrics.pairwise.paired_distances(metric='cosine') once per (query step, candidate), as
    CodeKNN.search_audio_cands / search_text_cands do (GestureKNN.py:672-690, 713-720) - on a reduced DB of `n_win`
    windows for the 8 steps of ONE query window.  Ties the C port back to reference-as-is semantics: its distances must
    equal the port's bit for bit, and its time per (DB window x query window) is what the extrapolation in BASELINE.md
    §2 is built on."""
    from sklearn.metrics.pairwise import paired_distances
    from oracle import cref, knn_oracle as O
    n = min(n_win, interp.shape[0])
    g = np.arange(26)
    t0 = time.perf_counter()
    da = np.full((8, n * 26), np.nan)
    dt_ = np.full((8, n * 26), np.nan, np.float32)
    feat = [O.wavlm_feat_rows(interp, j, list(g * 6)) for j in range(n)]            # (26, 6144) f64 per window
    for s in range(8):
        qa, qx = q[s].astype(np.float64), qt[s]
        for j in range(n):
            for k in range(26):
                da[s, j * 26 + k] = paired_distances(qa[None], feat[j][k][None], metric="cosine")[0]
                dt_[s, j * 26 + k] = paired_distances(qx[None], ctx[j][k][None], metric="cosine")[0]
    sec = time.perf_counter() - t0
    # the C port on the same pairs: per-code minima of both must agree exactly
    d_c, i_c = cref.audio_scan(interp[:n], g * 6, code[:n], g, q[:8], n_threads=1)
    t_c, j_c = cref.text_scan(ctx[:n], g, code[:n], g, qt[:8], n_threads=1)
    ok = True
    for s in range(8):
        for c_ in range(512):
            if i_c[s, c_] >= 0:
                ok &= bool(da[s, i_c[s, c_]] == d_c[s, c_]) and bool(dt_[s, j_c[s, c_]] == t_c[s, c_])
    per = sec / n                                                                     # s per (DB window x query window)
    return {"seconds_per_dbwindow_x_querywindow": round(per, 4), "db_windows": n, "sklearn_calls": 8 * n * 26 * 2,
            "matches_c_port_bit_for_bit": ok,
            "note": "reference-as-is throughput at N_db windows = 240 frames / (this x N_db) per query window"}


if __name__ == "__main__":
    main()
