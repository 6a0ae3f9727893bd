"""The N>1 path of bench.py (row-sharded DB, min+index all-reduce, one clip per rank) on a single GPU:
2 ranks share cuda:0 and exchange through gloo (QPG_BENCH_ONE_GPU=1); every rank checks its matched codes
against an unsharded match of the same clip (--check)."""
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
           "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == world and out["check"] is True and out["scaling"] == "weak"
    assert out["config"]["clips"] == world and out["value"] > 0
    assert "roofline" in out and out["roofline"]["bound"] == "mfma"
