"""Multi-GPU decomposition of the matcher: the database's windows are split into
contiguous row blocks, one per rank (one process per GPU, torch.distributed over RCCL/xGMI); each
rank sweeps its block and the per-(query, code) minima are combined with a min + index exchange.

The message is tiny (Q*512*20 B = 492 KB per 24 s clip for both modalities), so the exchange is latency- not
link-bound: ONE collective on a byte buffer of packed tables (exchange_bytes) and one HIP merge kernel per modality
(qpg_merge_select_*), no per-candidate traffic.  allreduce_min_index / alltoall_min_index are the round-1 ATen forms,
kept as host-tensor references for the CPU (gloo) tests."""
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


# A ClipGraph that records a row-sharded clip in SEGMENTS sets this (SegmentRecorder): every collective below then ends the
# hipGraph being captured, runs eagerly - as torch.distributed issues it in an uncaptured step - and opens the next graph.
_recorder = None

# Library-owned collectives (round 5; csrc/qpg_comm.hip): when a LibComm is active the exchanges below are RCCL calls the
# LIBRARY makes on the current stream - no torch.distributed call, no host round trip (~25 us each), and capturable into
# the clip's hipGraph together with the kernels around them (one graph per sharded clip instead of segments).
_libcomm = None


class LibComm:
    """An RCCL communicator owned by libqpg_hip.so (qpg_comm_create), one per process and device.  The 128-byte unique
    id is generated on rank 0 and handed to the other ranks through the already-initialised torch.distributed process
    group (any backend: an object broadcast, once); after that torch.distributed is not involved in the data path."""

    def __init__(self, device, group=None, uid=None, deadline_s=None):
        """uid: the 128-byte id if the caller has already exchanged it (exchange_unique_id).  deadline_s: run the RCCL
        rendezvous (qpg_comm_create = ncclCommInitRank, which blocks until every rank has arrived) in a helper thread and
        raise TimeoutError when it has not returned by then.  ONLY the rendezvous runs there: the id's broadcast is a
        torch.distributed collective and stays on the caller's thread (an abandoned helper still inside
        dist.broadcast_object_list would race the caller's next collective on the same process group)."""
        import ctypes
        import threading
        from . import _lib
        lib = _lib.load()
        self.device = torch.device(device)
        self.handle = None
        self.calls = 0
        if uid is None:
            uid = LibComm.exchange_unique_id(group)
        import torch.distributed as dist_
        self.rank, self.world = dist_.get_rank(group), dist_.get_world_size(group)
        ctx = _lib.ctx(self.device)
        box = {"abandoned": False}
        lock = threading.Lock()

        def create():
            try:
                if self.device.type == "cuda":
                    torch.cuda.set_device(self.device)       # (the current device is per thread)
                h = ctypes.c_void_p()
                rc = lib.qpg_comm_create(ctx, uid, 128, self.rank, self.world, ctypes.byref(h))
                err = None if rc == 0 else RuntimeError("qpg_comm_create failed: %s" % _lib.last_error())
            except Exception as e:                           # noqa: BLE001 (re-raised on the caller's thread)
                h, err = None, e
            with lock:
                if box["abandoned"]:
                    if err is None and h is not None:        # finished after the deadline: nobody will use or free it
                        lib.qpg_comm_destroy(h)
                    return
                box["handle"], box["err"] = h, err

        if deadline_s is None:
            create()
        else:
            th = threading.Thread(target=create, name="qpg-libcomm-init", daemon=True)
            th.start()
            th.join(deadline_s)
            with lock:
                if "err" not in box:
                    box["abandoned"] = True
                    raise TimeoutError("LibComm: the RCCL rendezvous did not complete within %.0f s" % deadline_s)
        if box["err"] is not None:
            raise box["err"]
        self.handle = box["handle"]

    @staticmethod
    def exchange_unique_id(group=None):
        """The 128-byte RCCL unique id: generated on rank 0, handed to the other ranks through the already-initialised
        torch.distributed process group (an object broadcast, once) - on the CALLER's thread."""
        import ctypes
        import torch.distributed as dist_
        from . import _lib
        if not (dist_.is_available() and dist_.is_initialized()):
            raise RuntimeError("LibComm: initialise torch.distributed first (it carries the unique id to the ranks)")
        lib = _lib.load()
        rank, world = dist_.get_rank(group), dist_.get_world_size(group)
        buf = ctypes.create_string_buffer(128)
        if rank == 0 and lib.qpg_comm_unique_id(buf, 128) != 0:
            raise RuntimeError("qpg_comm_unique_id failed: %s" % _lib.last_error())
        box = [bytes(buf.raw) if rank == 0 else None]
        if world > 1:
            dist_.broadcast_object_list(box, src=0, group=group)
        return box[0]

    def close(self):
        from . import _lib
        if self.handle is not None:
            if self.device.type == "cuda":
                torch.cuda.synchronize(self.device)
            _lib.load().qpg_comm_destroy(self.handle)
            self.handle = None

    def exchange(self, send, out, owner_blocks):
        from . import _lib
        # the library takes BYTE counts: numel() is one only for byte tensors
        assert send.dtype == torch.uint8 and out.dtype == torch.uint8 and send.is_contiguous() and out.is_contiguous(), \
            "LibComm.exchange moves contiguous uint8 buffers (code_knn.ExchangeLayout)"
        self.calls += 1
        if owner_blocks:
            _lib.call("qpg_comm_alltoall", self.device, self.handle, send, out, send.numel() // self.world)
        else:
            _lib.call("qpg_comm_allgather", self.device, self.handle, send, out, send.numel())
        return out

    def allreduce_max_i32(self, t):
        from . import _lib
        assert t.dtype == torch.int32 and t.is_contiguous()
        self.calls += 1
        _lib.call("qpg_comm_allreduce_max_i32", self.device, self.handle, t, t.numel())
        return t

    def allreduce_min_packed(self, dist, idx, absent):
        """Global per-entry (minimum f32 distance, lowest candidate index among equals) of `dist` / `idx` in ONE MIN
        all-reduce of packed u64 keys (qpg_allreduce_min_u64).  In place on copies; returns
        (dist, idx)."""
        from . import _lib
        n = dist.numel()
        packed = torch.empty((n,), dtype=torch.int64, device=dist.device)
        _lib.call("qpg_pack_min_u64", self.device, dist.contiguous(), idx.contiguous(), n, packed)
        self.calls += 1
        _lib.call("qpg_allreduce_min_u64", self.device, self.handle, packed, n)
        d, i = torch.empty_like(dist), torch.empty_like(idx)
        _lib.call("qpg_unpack_min_u64", self.device, packed, n, float(absent), d, i)
        return d, i


def enable_lib_collectives(device, group=None):
    """Make the library's own RCCL communicator the transport of exchange_bytes / allreduce_max_ (idempotent).  Returns the
    LibComm, or raises - the caller decides whether torch.distributed stays the transport (bench.py records the reason)."""
    global _libcomm
    if _libcomm is None:
        # The rendezvous (ncclCommInitRank: blocks until all ranks have arrived) runs under a deadline: a rank whose
        # rendezvous never completes reports a failure instead of hanging the job, and the caller's agreement step (bench.py:
        # a MIN all-reduce of the ranks' verdicts over torch.distributed) then puts EVERY rank on the torch.distributed
        # transport.  The id's broadcast - a torch.distributed collective - is made HERE, on the caller's thread, so an
        # abandoned helper thread is never inside a process-group collective when the caller issues the next one; a helper
        # that comes back late destroys its communicator itself (LibComm.__init__).  W > 1 over xGMI has never been run here
        # (no multi-GPU box behind gpurun): this is the seat belt; tests/test_distributed_gloo.py injects the failure.
        import os
        deadline = float(os.environ.get("QPG_LIB_COLLECTIVES_TIMEOUT_S", "120"))
        uid = LibComm.exchange_unique_id(group)
        _libcomm = LibComm(device, group, uid=uid, deadline_s=deadline)
    return _libcomm


def negotiate_lib_collectives(device, group=None):
    """enable_lib_collectives + the AGREEMENT every caller of it needs: a rank that could not bring the library's
    communicator up (no RCCL, a rendezvous that missed its deadline, a bad id) must not leave the others on it.  Each rank's
    verdict is MIN-reduced over torch.distributed (the process group that is up by definition); unless every rank
    succeeded, every rank closes what it has and stays on the torch.distributed transport.  Returns (enabled, why)."""
    import torch.distributed as dist_
    why = None
    try:
        enable_lib_collectives(device, group)
        ok = 1
    except Exception as e:                                        # noqa: BLE001 (reported to the caller)
        ok, why = 0, repr(e)[:200]
    if dist_.get_world_size(group) > 1:
        gloo = dist_.get_backend(group) == "gloo"
        f = torch.tensor([ok], dtype=torch.int32, device="cpu" if gloo else device)
        dist_.all_reduce(f, op=dist_.ReduceOp.MIN, group=group)
        ok = int(f.item())
    if not ok:
        disable_lib_collectives()
        why = why or "another rank could not create the library's communicator"
    return bool(ok), why


def disable_lib_collectives():
    global _libcomm
    if _libcomm is not None:
        _libcomm.close()
    _libcomm = None


def _exchange_into(out, send, world, owner_blocks, group=None):
    """exchange_bytes on a caller-owned receive buffer (`out` may be None: allocate).  RCCL (backend nccl) moves device
    buffers in place; gloo is staged through the host."""
    import torch.distributed as dist_
    gloo = dist_.get_backend(group) == "gloo"
    host = send.is_cuda and gloo
    s_ = send.cpu() if host else send
    if owner_blocks:
        r_ = torch.empty_like(s_) if (host or out is None) else out
        dist_.all_to_all_single(r_, s_, group=group)
    elif gloo:
        parts = [torch.empty_like(s_) for _ in range(world)]
        dist_.all_gather(parts, s_, group=group)
        r_ = torch.cat(parts)
    else:
        r_ = out if out is not None else torch.empty((world * s_.numel(),), dtype=s_.dtype, device=s_.device)
        dist_.all_gather_into_tensor(r_, s_, group=group)
    if out is None:
        return r_.to(send.device) if host else r_
    if r_ is not out:
        out.copy_(r_, non_blocking=False)
    return out


def exchange_bytes(send, world, owner_blocks, group=None):
    """The one collective of the sharded matcher.  `send` is this rank's byte buffer of per-shard (minimum, index)
    tables in exchange layout (code_knn.ExchangeLayout).  owner_blocks: the buffer is `world` equal blocks, block r
    goes to rank r (ONE all-to-all; every rank ends up with the `world` shards' tables of ITS query block) - else
    every rank receives every rank's whole buffer (ONE all-gather: the all-reduce(min, index) of SURVEY.md §8e with
    the reduction done locally by qpg_merge_select_*).  Returns the receive buffer: `world` source chunks, chunk w
    from rank w.  RCCL (backend nccl) moves device buffers in place; gloo is staged through the host."""
    lc = _libcomm
    if lc is not None and send.is_cuda:
        n = send.numel() if owner_blocks else world * send.numel()
        return lc.exchange(send, torch.empty((n,), dtype=send.dtype, device=send.device), owner_blocks)
    rec = _recorder
    if rec is not None:
        # a persistent receive buffer, allocated OUTSIDE the captures; the collective itself is replayed by calling it
        n = send.numel() if owner_blocks else world * send.numel()
        return rec.cut(lambda out: _exchange_into(out, send, world, owner_blocks, group),
                       lambda: torch.empty((n,), dtype=send.dtype, device=send.device))
    return _exchange_into(None, send, world, owner_blocks, group)


_hip_rt = None


def _graph_node_count(g):
    """Number of nodes of a captured torch.cuda.CUDAGraph(keep_graph=True), through the HIP runtime torch itself uses."""
    global _hip_rt
    import ctypes
    if _hip_rt is None:
        path = "libamdhip64.so"
        try:                                   # the runtime this process has ALREADY mapped (torch ships its own copy): a
            with open("/proc/self/maps") as f:  # second runtime would not know the graph handle
                for ln in f:
                    if "libamdhip64" in ln:
                        path = ln.split()[-1]
                        break
        except OSError:
            pass
        _hip_rt = ctypes.CDLL(path)
        _hip_rt.hipGraphGetNodes.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t)]
        _hip_rt.hipGraphGetNodes.restype = ctypes.c_int
    n = ctypes.c_size_t(0)
    rc = _hip_rt.hipGraphGetNodes(ctypes.c_void_p(g.raw_cuda_graph()), None, ctypes.byref(n))
    if rc != 0:
        raise RuntimeError("hipGraphGetNodes failed (%d)" % rc)
    return int(n.value)


class SegmentRecorder:
    """A row-sharded clip as a PROGRAM: hipGraph, collective, hipGraph, collective, ..., hipGraph.

    The kernels between two collectives are captured into one hipGraph each (one private memory pool for all of them:
    they are replayed in capture order); the collectives stay what they are in an uncaptured step - torch.distributed
    calls on persistent buffers, issued by the host between two graph launches.  A replay therefore costs the host one
    hipGraphLaunch per segment + one torch.distributed call per collective (~25 us each) instead of one Python launch
    per kernel, and nothing about RCCL differs from the eager path: no RCCL kernel is ever a graph node (a process that
    replays graphs WITH captured RCCL nodes and then issues eager collectives on the same communicator hung on this
    ROCm / torch build: DESIGN.md §5)."""

    def __init__(self):
        self.program = []           # callables, in replay order
        self.kinds = []             # "graph" | "collective" per entry
        self.pool = torch.cuda.graph_pool_handle()
        self._g = None
        self._n0 = 0
        self._keep = []             # receive buffers

    def begin(self):
        from . import _lib
        # keep_graph: the hipGraph_t stays readable after capture_end, so end() can COUNT its nodes
        self._g = torch.cuda.CUDAGraph(keep_graph=True)
        self._n0 = _lib.n_calls
        # thread_local: the process group's watchdog thread may query its events while this thread captures
        self._g.capture_begin(pool=self.pool, capture_error_mode="thread_local")

    def end(self):
        from . import _lib
        g, self._g = self._g, None
        g.capture_end()
        if _lib.n_calls == self._n0 and _graph_node_count(g) == 0:
            # No launch of this library's in the segment and no torch-native node either (a copy_, a zero_, an index op
            # between two collectives would be one, and a replay must not lose it: ADVICE r4): an empty capture, which torch
            # refuses to replay.  The nodes are COUNTED (hipGraphGetNodes on the kept graph), nothing is executed: round 5
            # told the two cases apart by a trial replay, which ran a torch-only segment's in-place ops a second time during
            # recording and hid real replay failures behind a bare except.
            return
        g.instantiate()
        self.program.append(g.replay)
        self.kinds.append("graph")

    def cut(self, issue, alloc=None):
        """End the open segment, run the collective `issue(out)` eagerly ONCE (every rank does: the call order on the
        communicator is the replay's), record it, open the next segment.  Returns the collective's result."""
        global _recorder
        self.end()
        out = alloc() if alloc is not None else None
        if out is not None:
            self._keep.append(out)
        fn = (lambda: issue(out))
        _recorder = None
        try:
            res = fn()
        finally:
            _recorder = self
        self.program.append(fn)
        self.kinds.append("collective")
        self.begin()
        return res


def merge_reference(dists, idxs, absent):
    """ATen statement of qpg_merge_select_* for HOST tensors (CPU tests of the exchange protocol; the product path
    on a GPU uses the HIP kernel): dists / idxs [W, Q, K]; winner = minimum distance, lowest index among equals,
    idx < 0 = absent in that shard."""
    big = torch.iinfo(torch.int32).max
    d = torch.where(idxs >= 0, dists, torch.full_like(dists, float("inf")))
    best = d.min(dim=0).values
    cand = torch.where((d == best) & (idxs >= 0), idxs, torch.full_like(idxs, big))
    ibest = cand.min(dim=0).values
    have = ibest != big
    return (torch.where(have, best, torch.full_like(best, absent)),
            torch.where(have, ibest, torch.full_like(ibest, -1)))


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


def allreduce_max_(t, force=False):
    """In-place MAX all-reduce of a small device tensor (the matcher's trouble word: every rank must take the same
    re-match decision); a no-op without an initialised process group of more than one rank (`force`: also with one)."""
    import torch.distributed as dist_
    if not (_active() or (force and dist_.is_available() and dist_.is_initialized())):
        return t
    if _libcomm is not None and t.is_cuda and t.dtype == torch.int32:
        return _libcomm.allreduce_max_i32(t)
    rec = _recorder
    if rec is not None:
        return rec.cut(lambda _out: allreduce_max_(t, force))
    if _host_staged():
        h = t.cpu()
        dist_.all_reduce(h, op=dist_.ReduceOp.MAX)
        t.copy_(h)
    else:
        dist_.all_reduce(t, op=dist_.ReduceOp.MAX)
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
