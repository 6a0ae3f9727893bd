"""The N>1 paths of bench.py (row-sharded DB + one min/index exchange + the HIP merge kernel) on a single GPU:
the ranks share cuda:0 and exchange through gloo (QPG_BENCH_ONE_GPU=1); every rank checks its matched codes
against an unsharded match of the same clip(s) (--check).  weak = one clip per rank, owner-partitioned all-to-all;
strong = ONE clip, all-gather + merge on every rank (BASELINE.json configs[3]'s shape); 2 clips per rank = the batched
multi-clip sweep of configs[4]."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 3])
def test_bench_sharded_matches_unsharded(world):
    env = dict(os.environ, QPG_BENCH_ONE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", str(world), "--steps", "2", "--warmup", "1", "--n-db", "200", "--windows", "2", "--check",
           "--no-cpu-baseline", "--no-vqvae", "--no-prewarm", "--sharded-mixed-min-gflop", "0"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == world and out["check"] is True and out["scaling"] == "weak"
    assert out["config"]["clips"] == world and out["value"] > 0
    # one clip per rank: the timed steps replay the clip's segments (hipGraph, collective, hipGraph, ...), the eager leg
    # behind them returned the same codes, and another seed through the same capture equals the eager path
    assert out["step_mode"] == "graph-segments" and out["eager"]["codes_equal_graph_steps"] is True
    gr = out["graph_replay"]
    assert gr["other_seed_equals_eager"] is True and gr["captures"] == 1
    assert gr["segments"].count("collective") == out["config"]["collectives_per_step"] == 4
    assert gr["segments"][0] == "graph" and gr["segments"][-1] == "graph"
    # the clip-parallel leg beside it (whole DB on every rank, no collective): same codes, one graph replay per clip
    rep = out["replicated"]
    assert rep["codes_equal_row_sharded"] is True and rep["step_mode"] == "graph" and rep["collectives_per_step"] == 0
    assert "roofline" in out and out["roofline"]["bound"] == "hbm"       # the shards sweep with the split-f16 kernel
    # the shards sweep in mixed precision; the merge re-evaluated something across shards and nothing overflowed
    assert out["roofline"]["precision"] == "mixed" and out["mixed_precision"]["flags"] == 0
    assert out["mixed_precision"]["cross_shard_reevaluations_per_step"] > 0


@pytest.mark.parametrize("extra,scaling,clips", [(["--scaling", "strong", "--n-db", "300"], "strong", 1),
                                                 (["--clips", "2", "--n-db", "200"], "weak", 4)])
def test_bench_strong_and_multiclip(extra, scaling, clips):
    env = dict(os.environ, QPG_BENCH_ONE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "2", "--warmup", "1", "--windows", "2", "--check", "--no-cpu-baseline",
           "--no-vqvae", "--no-prewarm", "--sharded-mixed-min-gflop", "0"] + extra
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["check"] is True and out["scaling"] == scaling and out["config"]["clips"] == clips
    assert out["roofline"]["precision"] == "mixed" and out["mixed_precision"]["flags"] == 0


def test_bench_sharded_small_shards_keep_the_f64_sweep():
    """Below CodeKNN.sharded_mixed_min_gflop per rank the two extra exchanges would cost more than the faster sweep saves:
    the shards run the f64 sweep and the one-exchange merge (default threshold, tiny DB)."""
    env = dict(os.environ, QPG_BENCH_ONE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "2", "--warmup", "1", "--n-db", "200", "--windows", "2", "--check", "--no-cpu-baseline",
           "--no-vqvae", "--no-prewarm"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["check"] is True and out["roofline"]["precision"] == "f64" and "mixed_precision" not in out


def test_bench_self_launches_and_rematches_on_request_overflow():
    """`python bench.py --gpus 2` WITHOUT torchrun (how the driver types it) launches its own ranks and prints one line;
    with ONE request slot per (query, shard) the cross-shard request lists overflow in every step: the trouble word
    is MAX-reduced over the ranks, comes out with the codes, and every rank re-matches the step on the uncapped sharded
    path (cross-shard tier 2) - the codes still equal the unsharded match (--check)."""
    env = dict(os.environ, QPG_BENCH_ONE_GPU="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--n-db", "200",
           "--windows", "2", "--check", "--no-cpu-baseline", "--no-vqvae", "--no-prewarm", "--sharded-mixed-min-gflop", "0",
           "--mixed-requests", "1"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["check"] is True
    assert out["rematched_steps"] >= 2


@pytest.mark.parametrize("world,scaling", [(4, "weak"), (4, "strong"), (8, "strong"), (8, "weak")])
def test_bench_four_and_eight_ranks_on_one_gpu(world, scaling):
    """The round-4 protocol (deterministic request slots; all-gather form = two collectives and no request exchange;
    all-to-all form = three + the 4-byte word; trouble bits riding in the block headers) with 4 and 8 ranks sharing the
    one GPU over gloo: codes == the unsharded match on every rank (--check), nothing re-matched; `ranks_seen` counts
    the ranks and reports that they share one device."""
    env = dict(os.environ, QPG_BENCH_ONE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", str(world), "--steps", "2", "--warmup", "1", "--n-db", "264", "--windows", "2", "--check",
           "--no-cpu-baseline", "--no-vqvae", "--no-prewarm", "--sharded-mixed-min-gflop", "0", "--scaling", scaling,
           "--no-replicated"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["check"] is True and out["n_gpus"] == world and out["rematched_steps"] == 0
    assert out["ranks_seen"]["ranks"] == world and out["ranks_seen"]["distinct_devices"] == 1
    assert out["config"]["collectives_per_step"] == (2 if scaling == "strong" else 4)


def test_clips_in_flight_on_a_row_sharded_db():
    """ClipPipeline over a row-sharded DB (round 4): two ranks, two lanes, the all-gather form of the exchange on every
    lane's stream; the codes equal the unsharded match (--check)."""
    env = dict(os.environ, QPG_BENCH_ONE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "6", "--warmup", "2", "--n-db", "264", "--windows", "2", "--check", "--no-cpu-baseline",
           "--no-vqvae", "--no-prewarm", "--scaling", "strong", "--sharded-mixed-min-gflop", "0", "--clips-in-flight", "2"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["check"] is True and out["config"]["clips_in_flight"] == 2 and out["rematched_steps"] == 0


def test_bench_replicated_mode_two_ranks():
    """--scaling replicated (shard the QUERIES, not the database): every rank holds the whole DB and matches
    its own clip, no collective in the step; the codes equal the single-rank match (--check)."""
    env = dict(os.environ, QPG_BENCH_ONE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "3", "--warmup", "1", "--n-db", "200", "--windows", "2", "--check", "--no-cpu-baseline",
           "--no-vqvae", "--no-prewarm", "--scaling", "replicated"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["check"] is True and out["n_gpus"] == 2 and out["scaling"] == "weak"
    assert out["config"]["collectives_per_step"] == 0 and "replicated" in out["config"]["parallelism"]


def test_sharded_path_over_rccl_with_one_rank():
    """VERDICT r2 next #3b: the row-shard code path (exchange layout, all_gather_into_tensor / all_to_all_single on uint8
    device buffers, the three mixed-merge kernels, the trouble word's all-reduce) executed over backend nccl = RCCL with
    world_size 1, weak (all-to-all) and strong (all-gather) forms, mixed-precision and f64 shards; codes == the plain
    single-GPU match (--check)."""
    cases = [(e, "0") for e in (["--sharded-mixed-min-gflop", "0"], ["--scaling", "strong", "--sharded-mixed-min-gflop", "0"],
                                ["--sharded-mixed-min-gflop", "1e9"], ["--audio-precision", "exact"])]
    # round 5: the same forms with the LIBRARY's own RCCL communicator as the transport (qpg_comm_*: the default when it
    # comes up) - no torch.distributed call in a step, the whole sharded clip ONE hipGraph, eager re-matches behind replays
    cases += [(e, "1") for e in (["--sharded-mixed-min-gflop", "0"], ["--scaling", "strong", "--sharded-mixed-min-gflop", "0"],
                                 ["--sharded-mixed-min-gflop", "1e9"])]
    for extra, libc in cases:
        env = dict(os.environ, QPG_BENCH_FORCE_SHARDED="1", MASTER_PORT=str(_free_port()), QPG_LIB_COLLECTIVES=libc)
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
            env.pop(k, None)
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--n-db", "200",
               "--windows", "2", "--check", "--no-cpu-baseline", "--no-vqvae", "--no-prewarm", "--no-cold"] + extra
        r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, str(extra) + r.stdout[-2000:] + r.stderr[-3000:]
        out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        assert out["check"] is True and out["rematched_steps"] == 0, extra
        if libc == "1":
            assert out["step_mode"] == "graph" and out["collectives"]["transport"].startswith("libqpg_hip.so"), (extra, out["collectives"])
            assert "segments" not in out["graph_replay"]
        else:
            # the timed region replayed the clip's segments; the collectives between them are the eager path's own calls
            assert out["step_mode"] == "graph-segments" and out["collectives"]["transport"].startswith("torch.distributed"), extra
            assert out["graph_replay"]["segments"].count("collective") == out["config"]["collectives_per_step"], extra
        assert out["graph_replay"]["other_seed_equals_eager"] and out["eager"]["codes_equal_graph_steps"], extra
    # QPG_BENCH_SHARDED_EAGER=1: one Python launch per kernel, as before
    env = dict(os.environ, QPG_BENCH_FORCE_SHARDED="1", QPG_BENCH_SHARDED_EAGER="1", MASTER_PORT=str(_free_port()))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--n-db", "200",
           "--windows", "2", "--check", "--no-cpu-baseline", "--no-vqvae", "--no-prewarm", "--no-cold",
           "--sharded-mixed-min-gflop", "0"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["check"] is True and out["step_mode"] == "eager" and "graph_replay" not in out


def test_merge_kernel_vs_reference():
    """qpg_merge_select_f64 / _f32 == parallel.merge_reference on random tables with planted ties and absent codes."""
    import numpy as np
    import torch
    from qpgesture_amd import _lib
    from qpgesture_amd.parallel import merge_reference
    g = torch.Generator().manual_seed(3)
    W, Q, K = 5, 7, 512
    for dt, name in ((torch.float64, "qpg_merge_select_f64"), (torch.float32, "qpg_merge_select_f32")):
        d = torch.rand((W, Q, K), generator=g).to(dt)
        i = torch.randint(0, 10 ** 6, (W, Q, K), generator=g, dtype=torch.int32)
        d[1] = d[3]                                             # exact ties across shards: lowest index wins
        i[torch.rand((W, Q, K), generator=g) < 0.2] = -1        # code absent in that shard
        i[:, :, 17] = -1                                        # absent everywhere
        want_d, want_i = merge_reference(d, i, 1e3)
        dev = torch.device("cuda:0")
        buf = torch.cat((d.reshape(W, -1).view(torch.uint8).reshape(W, -1), i.reshape(W, -1).view(torch.uint8).reshape(W, -1)),
                        dim=1).contiguous().to(dev)
        n = Q * K
        dsz = d.element_size()
        od = torch.empty((Q, K), dtype=dt, device=dev)
        oi = torch.empty((Q, K), dtype=torch.int32, device=dev)
        ork = torch.empty((Q, K), dtype=torch.int16, device=dev)
        extra = (0.0, None) if dt == torch.float64 else ()       # (near-tie detection off: random tables)
        _lib.call(name, dev, buf, W, buf.shape[1], 0, n * dsz, Q, K, 1e3, od, oi, ork, *extra)
        assert torch.equal(od.cpu(), want_d) and torch.equal(oi.cpu(), want_i)
        assert (oi.cpu()[:, 17] == -1).all() and (od.cpu()[:, 17] == 1e3).all()
        rk = np.argsort(np.argsort(want_d.numpy(), axis=1, kind="stable"), axis=1, kind="stable")
        assert np.array_equal(ork.cpu().numpy(), rk)


def test_library_owned_collectives_one_rank():
    """csrc/qpg_comm.hip through parallel.LibComm on a one-rank RCCL communicator: byte all-gather / all-to-all return the
    send buffer, the MAX / packed-MIN all-reduces leave their operands, pack / unpack of (distance, index) keys round-trip
    (order-preserving for negative values, -1 = absent <-> all ones), and the calls are capturable in a hipGraph."""
    import numpy as np
    code = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
dev = torch.device("cuda:0")
dist.init_process_group("gloo", rank=0, world_size=1)
from qpgesture_amd import parallel as P, _lib
lc = P.enable_lib_collectives(dev)
send = torch.arange(4096, dtype=torch.uint8, device=dev)
assert torch.equal(P.exchange_bytes(send, 1, False), send) and torch.equal(P.exchange_bytes(send, 1, True), send)
t = torch.tensor([5, -3, 7], dtype=torch.int32, device=dev)
assert torch.equal(P.allreduce_max_(t.clone(), force=True), t)
rng = np.random.default_rng(3)
d = torch.from_numpy(rng.standard_normal(5000).astype(np.float32)).to(dev)
i = torch.from_numpy(rng.integers(-1, 1 << 30, size=5000).astype(np.int32)).to(dev)
d2, i2 = lc.allreduce_min_packed(d, i, 1000.0)
absent = i < 0
assert torch.equal(i2, i) and torch.equal(d2[~absent], d[~absent]) and bool((d2[absent] == 1000.0).all())
packed = torch.empty((5000,), dtype=torch.int64, device=dev)
_lib.call("qpg_pack_min_u64", dev, d, i, 5000, packed)
pk = packed.cpu().numpy().view(np.uint64)
dd, ii = d.cpu().numpy(), i.cpu().numpy()
ok = ii >= 0
order = np.lexsort((ii[ok], dd[ok]))
assert np.array_equal(np.argsort(pk[ok], kind="stable"), order)       # u64 order == (distance, index) order
assert (pk[~ok] == np.uint64(0xffffffffffffffff)).all()
g = torch.cuda.CUDAGraph()
out = torch.empty((4096,), dtype=torch.uint8, device=dev)
s = torch.cuda.Stream(dev)
with torch.cuda.stream(s):
    lc.exchange(send, out, False)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        lc.exchange(send, out, False)
        lc.allreduce_max_i32(t)
out.zero_()
for _ in range(3):
    g.replay()
lc.exchange(send, out, True)          # an eager collective BEHIND graph replays on the same communicator
torch.cuda.synchronize()
assert torch.equal(out, send)
P.disable_lib_collectives()
print("LIBCOMM_OK", lc.calls)
''' % ROOT
    env = dict(os.environ, MASTER_PORT=str(_free_port()))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "LIBCOMM_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
