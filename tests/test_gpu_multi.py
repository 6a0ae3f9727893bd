"""Tests that need SEVERAL GPUs (one process per GPU over RCCL / xGMI): collected everywhere, skipped cleanly where
torch.cuda.device_count() is smaller than the world they need (the 1-GPU boxes behind gpurun; VERDICT r5 next #8a).  On a
multi-GPU node they are what first executes the W > 1 RCCL path: the library-owned communicator (qpg_comm_*) against host
reductions, and bench.py's row-sharded / strong / replicated modes with --check."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _n_gpus():
    import torch
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _need(world):
    n = _n_gpus()
    if n < world:
        pytest.skip("needs %d GPUs, this box has %d" % (world, n))


def _libcomm_worker(rank, world, port, out):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", rank)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from qpgesture_amd import parallel as par
    ok, why = par.negotiate_lib_collectives(dev)
    res = {"enabled": bool(ok), "why": why}
    if ok:
        lc = par._libcomm
        g = torch.Generator(device="cpu")

        def block(r, n):                       # what rank r sends: deterministic bytes every rank can recompute
            g.manual_seed(1000 + r)
            return torch.randint(0, 256, (n,), generator=g, dtype=torch.uint8)
        nb = 4096 * world
        send = block(rank, nb).to(dev)
        # all-gather: chunk w of the result = rank w's whole buffer
        got = par.exchange_bytes(send, world, False)
        want = torch.cat([block(w, nb) for w in range(world)])
        res["allgather"] = bool(torch.equal(got.cpu(), want))
        # all-to-all: chunk w of the result = block `rank` of rank w's buffer
        got = par.exchange_bytes(send, world, True)
        per = nb // world
        want = torch.cat([block(w, nb)[rank * per:(rank + 1) * per] for w in range(world)])
        res["alltoall"] = bool(torch.equal(got.cpu(), want))
        # MAX of the trouble word
        t = torch.tensor([1 << rank, rank], dtype=torch.int32, device=dev)
        par.allreduce_max_(t)
        res["max"] = t.cpu().tolist() == [1 << (world - 1), world - 1]
        # min + index in ONE packed MIN all-reduce: global minimum, lowest index among equals (first-wins across shards)
        rng = np.random.Generator(np.random.PCG64(7))
        d_all = rng.random((world, 3000)).astype(np.float32)
        d_all[:, :50] = d_all[0, :50]                               # exact ties across ALL ranks: the lowest index wins
        i_all = rng.integers(0, 1 << 20, size=(world, 3000)).astype(np.int32)
        i_all[:, 100:120] = -1                                       # absent everywhere
        i_all[1:, 120:140] = -1                                      # present on rank 0 only
        d, i = lc.allreduce_min_packed(torch.from_numpy(d_all[rank]).to(dev), torch.from_numpy(i_all[rank]).to(dev), 1000.0)
        dm = np.where(i_all >= 0, d_all, np.inf)
        best = dm.min(axis=0)
        cand = np.where((dm == best) & (i_all >= 0), i_all, np.iinfo(np.int32).max).min(axis=0)
        have = cand != np.iinfo(np.int32).max
        res["min_index"] = bool(np.array_equal(i.cpu().numpy(), np.where(have, cand, -1)) and
                                np.array_equal(d.cpu().numpy()[have], best[have].astype(np.float32)))
        res["calls"] = lc.calls
        # the same collectives captured in a hipGraph and replayed (the sharded clip is ONE graph on this transport)
        gsend, gout = send.clone(), torch.empty((nb * world,), dtype=torch.uint8, device=dev)
        s = torch.cuda.Stream(dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            lc.exchange(gsend, gout, False)
        torch.cuda.current_stream(dev).wait_stream(s)
        torch.cuda.synchronize(dev)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            lc.exchange(gsend, gout, False)
        gsend.copy_(block(rank + 100, nb).to(dev))
        gr.replay()
        torch.cuda.synchronize(dev)
        want = torch.cat([block(w + 100, nb) for w in range(world)])
        res["graph_replay"] = bool(torch.equal(gout.cpu(), want))
    json.dump(res, open(out % rank, "w"))
    dist.barrier()
    par.disable_lib_collectives()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_library_communicator_over_rccl_vs_host_reductions(world, tmp_path):
    _need(world)
    import time
    import torch.multiprocessing as mp
    out = str(tmp_path / "lc_%d.json")
    # (never leave a hung rendezvous behind on a GPU box: the workers get 300 s, then they are killed and the test fails)
    ctx = mp.spawn(_libcomm_worker, args=(world, _free_port(), out), nprocs=world, join=False)
    t0 = time.time()
    while not ctx.join(timeout=5):
        if time.time() - t0 > 300:
            for p_ in ctx.processes:
                if p_.is_alive():
                    p_.kill()
            pytest.fail("the %d-rank RCCL workers did not finish within 300 s" % world)
    for r in range(world):
        res = json.load(open(out % r))
        assert res["enabled"], res
        assert res["allgather"] and res["alltoall"] and res["max"] and res["min_index"] and res["graph_replay"], (r, res)


def _bench(world, extra, timeout=600):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", str(world), "--steps", "5", "--warmup", "2", "--check", "--no-cpu-baseline", "--no-vqvae"] + extra
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


@pytest.mark.parametrize("world", [2, 8])
def test_bench_weak_row_shards_over_rccl(world):
    """bench.py --gpus N as the driver runs it (one rank per GPU, RCCL): the sharded + exchanged codes equal a one-rank match
    of the same clips (--check), the exchanges ran on the library's communicator inside ONE hipGraph, every rank was seen."""
    _need(world)
    out = _bench(world, ["--n-db", str(256 * world), "--sharded-mixed-min-gflop", "0"])
    assert out["n_gpus"] == world and out["check"] is True and out["scaling"] == "weak" and out["value"] > 0
    assert out["ranks_seen"]["ranks"] == world and out["ranks_seen"]["distinct_devices"] == world
    assert out["ranks_seen"]["backend"] == "nccl"
    assert out["collectives"]["transport"].startswith("libqpg_hip.so") and out["step_mode"] == "graph"
    assert out["replicated"]["codes_equal_row_sharded"] is True


@pytest.mark.parametrize("world", [2, 8])
def test_bench_strong_8192_over_rccl(world):
    """BASELINE configs[3]: ONE clip vs N_db = 8192 row-sharded over the ranks, all-gather(min, index) + merge."""
    _need(world)
    out = _bench(world, ["--scaling", "strong"])
    assert out["check"] is True and out["scaling"] == "strong" and out["config"]["n_db"] == 8192
    assert out["mixed_precision"]["flags"] == 0


def test_bench_replicated_clip_parallel():
    _need(2)
    out = _bench(2, ["--scaling", "replicated", "--n-db", "512"])
    assert out["check"] is True and out["config"]["collectives_per_step"] == 0
