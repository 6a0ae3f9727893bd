"""Multi-GPU decomposition of the matcher: the database's windows are split into
contiguous row blocks, one per rank (one process per GPU, torch.distributed over RCCL/xGMI); each
rank sweeps its block and the per-(query, code) minima are combined with a min + index exchange.

The message is tiny (Q*512*(8+4) B = 295 KB for a 24 s clip), so the exchange is latency- not
link-bound: two all-reduces on the packed tables, no per-candidate traffic."""
import torch

INT_MAX = 2 ** 31 - 1


def shard_rows(n, rank, world):
    """Contiguous row block [lo, hi) of rank `rank` (ceil split; trailing ranks may be empty)."""
    per = (n + world - 1) // world
    return min(rank * per, n), min((rank + 1) * per, n)


def allreduce_min_index(dist, idx, group=None):
    """all-reduce(min + index): `dist` [Q,K] per-shard minima, `idx` [Q,K] GLOBAL candidate indices
    (-1 = absent).  Returns the global minima and, among ranks that hold the minimum, the lowest
    index — i.e. the reference's first-wins scan order (GestureKNN.py:686), because shards are
    contiguous ascending row blocks.  Works on any backend (nccl == RCCL on ROCm, gloo on CPU)."""
    import torch.distributed as dist_
    best = dist.clone()
    dist_.all_reduce(best, op=dist_.ReduceOp.MIN, group=group)
    cand = torch.where((dist == best) & (idx >= 0), idx, torch.full_like(idx, INT_MAX))
    dist_.all_reduce(cand, op=dist_.ReduceOp.MIN, group=group)
    cand = torch.where(cand == INT_MAX, torch.full_like(cand, -1), cand)
    return best, cand


def alltoall_min_index(dist, idx, world, group=None):
    """Owner-partitioned form of the same exchange, for when the query rows are `world` equal blocks and rank r only
    needs the global result for block r (bench.py: one clip per rank).  ONE all-to-all instead of two all-reduces:
    every rank sends block r of its per-shard (minimum, index) tables to rank r, which takes the minimum and, among
    the shards that hold it, the lowest global index (== first-wins, shards being ascending row blocks).
    dist [world*Qc, K] (f32/f64), idx [world*Qc, K] -> (best f64 [Qc, K], idx i32 [Qc, K]) of this rank's block.
    Message per rank: (world-1)/world of world*Qc*K*16 B, received: the same — half the bytes of the all-reduce pair
    on the wire and a single collective's latency."""
    import torch.distributed as dist_
    Qt, K = dist.shape
    assert Qt % world == 0, "query rows must split evenly over the ranks"
    Qc = Qt // world
    send = torch.stack((dist.to(torch.float64), idx.to(torch.float64)), dim=-1).contiguous()    # indices < 2^53: exact
    host = dist_.get_backend(group) == "gloo" and send.is_cuda
    s_ = send.cpu() if host else send
    r_ = torch.empty_like(s_)
    dist_.all_to_all_single(r_, s_, group=group)
    recv = (r_.to(send.device) if host else r_).view(world, Qc, K, 2)
    d, i = recv[..., 0], recv[..., 1]
    best = d.min(dim=0).values
    big = float(2 ** 53)
    cand = torch.where((d == best) & (i >= 0), i, torch.full_like(i, big))
    ibest = cand.min(dim=0).values
    ibest = torch.where(ibest == big, torch.full_like(ibest, -1.0), ibest)
    return best.contiguous(), ibest.to(torch.int32).contiguous()


# ---- data-parallel VQ-VAE training (codebook/train.py; bottleneck.py:44,73-75 collectives) ----------------------
def _active():
    import torch.distributed as dist_
    return dist_.is_available() and dist_.is_initialized() and dist_.get_world_size() > 1


def _host_staged():
    """gloo has no device-side transport here: stage through the host (the RCCL backend reduces in place)."""
    import torch.distributed as dist_
    return dist_.get_backend() == "gloo"


def allreduce_sum_(t, average=False):
    """In-place SUM (or mean) all-reduce of a device tensor; a no-op without an initialised process group."""
    if not _active():
        return t
    import torch.distributed as dist_
    if _host_staged():
        h = t.cpu()
        dist_.all_reduce(h)
        t.copy_(h)
    else:
        dist_.all_reduce(t)
    if average:
        t.div_(dist_.get_world_size())
    return t


def broadcast_(t, src=0):
    if not _active():
        return t
    import torch.distributed as dist_
    if _host_staged():
        h = t.cpu()
        dist_.broadcast(h, src)
        t.copy_(h)
    else:
        dist_.broadcast(t, src)
    return t


class GradBucket:
    """One in-flight SUM all-reduce of a contiguous slice of the flat gradient buffer (RCCL: asynchronous, ordered
    after the kernels already enqueued on the current stream, so the rest of backward overlaps it; gloo: host-staged
    and synchronous).  wait() makes the current stream wait for the result."""

    def __init__(self, t):
        import torch.distributed as dist_
        self.t, self.work = t, None
        if not _active():
            return
        if _host_staged():
            h = t.cpu()
            dist_.all_reduce(h)
            t.copy_(h)
        else:
            self.work = dist_.all_reduce(t, async_op=True)

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None


def world_size():
    import torch.distributed as dist_
    return dist_.get_world_size() if _active() else 1
