"""Captured and pipelined replays of the matching step (moved out of code_knn.py in round 6; that module re-exports every
name): ClipGraph - one hipGraph per clip shape, seeds and results as data in pinned memory; ClipPipeline - eager lanes
(rounds 2-5); SerialReplayer - two doorbell captures taking turns (experimental); GraphPipeline - several clips per captured
replay, several replays in flight (the serving-throughput mode).  The reference's loop over test clips
(codebook/Speech2GestureMatching/GestureKNN.py:785-813) carries no state from clip to clip: every arrangement here returns,
clip for clip, what CodeKNN.match_clip returns."""
import time

import numpy as np
import torch

from . import _lib
from .constant import num_frames_code
from .code_knn import (MODE_AUD, MODE_AUD_TXT, MODE_TXT, CodeKNN, GuardOverflow, _PIN_SENTINEL,          # noqa: F401
                       _wait_pinned)


class ClipGraph:
    """A captured clip: one hipGraph (torch.cuda.CUDAGraph is the HIP graph wrapper; every node is one of this library's
    kernels; no memset nodes, see fill_ff_kernel) that replays the whole per-clip launch sequence - query packs, both
    sweeps, selects, ranks, rank fusion, gate tables, walk - for a fixed clip shape.

    Nothing about a clip is baked into the capture (round 4): the seed code and the seed phase block live in pinned host
    memory the kernels read (the walk takes them through qpg_match_steps_batch's seed POINTERS), the inputs are either
    static buffers `run` copies into or the caller's own resident tensors (`bind`), and the integer results (codes |
    votes | status) land in pinned host memory behind a system-scope fence, so a replay is: write the seed, write the
    sentinel, hipGraphLaunch, watch the status word.  One capture serves every clip of that shape."""

    def __init__(self, knn, n_windows, mode, n_sweep_windows, window_offset, audio=None, context=None, owner_blocks=False,
                 n_clips=1, encoder=None, encode_input=None, encode_precision="f32", sweep_signal=False, doorbell=False):
        db, dev = knn.db, knn.db.device
        self.owner_blocks = owner_blocks
        # doorbell (round 6): the capture's FIRST node waits for the host's go (qpg_doorbell_wait), so the NEXT replay can be
        # enqueued while the current one still runs (prelaunch(): hipGraphLaunch is ~17 us of host time + the command
        # processor's start-up) and started by ONE store once the current results are read and the next seed is written
        # (launch()).  The GPU work of a step still begins only when its inputs are final; a pre-launched replay nobody wants
        # is run and discarded (drain()).  One-GPU, unsegmented captures only.
        self._doorbell = bool(doorbell)
        self._prelaunched = False
        if self._doorbell:
            self._db_cnt = torch.zeros((1,), dtype=torch.int32, device=dev)
            self._db_go = torch.zeros((1,), dtype=torch.int32).pin_memory()
            self._db_go_np = self._db_go.numpy()
            self._db_seq = 0
        # sweep_signal (GraphPipeline): a one-thread kernel behind the audio sweep stores 1 into pinned host memory
        # (qpg_signal_i32) - the host learns that the replay's sweep is over without waiting for its tail
        self._sweep_flag = torch.zeros((1,), dtype=torch.int32).pin_memory() if sweep_signal else None
        self._sweep_flag_np = self._sweep_flag.numpy() if sweep_signal else None
        self.CL = int(n_clips)
        if self.CL < 1 or n_sweep_windows < window_offset + self.CL * n_windows:
            raise ValueError("n_clips x n_windows windows must lie inside the swept windows")
        self.enc, self.enc_x = encoder, encode_input
        if (encoder is None) != (encode_input is None):
            raise ValueError("encoder and encode_input go together")
        if encode_precision not in ("f32", "f16x3"):
            raise ValueError("encode_precision must be 'f32' or 'f16x3'")
        # "f16x3": the split-f16 encoder under its margin check (VQVAE.encode_f16x3_device): the replay also brings back one
        # flag per window, and encoded_ids() re-encodes flagged windows on the f32 kernels before it returns anything
        self.enc_precision = encode_precision
        self.enc_redone = 0
        self.segmented = False
        if db.world != 1 or knn.force_sharded:
            # A row-sharded clip is recorded in SEGMENTS (parallel.SegmentRecorder): one hipGraph per run of kernels
            # between two collectives, the collectives themselves issued eagerly between the graph launches - the host
            # pays one hipGraphLaunch per segment instead of one Python launch per kernel (the eager sharded step is
            # host-bound: ~0.45-0.59 ms of host time against ~0.36 ms of GPU time on one rank), and RCCL sees exactly what
            # it sees in an uncaptured step.
            # QPG_EXPERIMENTAL_SHARDED_GRAPH=1: ONE graph with the collectives captured inside (RCCL supports capture
            # through torch.distributed's nccl backend).  With one rank over RCCL a process that only replays runs
            # (tools/step_loop.py: 0.36 ms per clip), but a process that replays AND issues eager collectives on the same
            # communicator afterwards (bench.py's eager leg, a flagged clip's re-match) hung on this ROCm / torch build.
            import os as _os
            import torch.distributed as dist_
            if not (dist_.is_available() and dist_.is_initialized()):
                raise NotImplementedError("graph capture of the sharded path needs an initialised process group")
            from . import parallel as _par0
            if _par0._libcomm is not None:
                # round 5: the collectives are the LIBRARY's own RCCL calls on the capturing stream (csrc/qpg_comm.hip) -
                # the whole sharded clip is ONE hipGraph, no torch.distributed call in a replay
                pass
            elif _os.environ.get("QPG_EXPERIMENTAL_SHARDED_GRAPH", "") == "1":
                if dist_.get_backend() != "nccl":
                    raise NotImplementedError("one-graph capture of the sharded path needs the nccl (RCCL) backend")
            else:
                self.segmented = True
        if knn.host_ranks:
            raise NotImplementedError("graph capture needs the device-side ranks (tie_rule 'stable')")
        if knn.serial_walk:
            raise NotImplementedError("graph capture replays the tabulated walk (serial_walk is an eager-path switch)")
        if self._doorbell and self.segmented:
            raise NotImplementedError("doorbell: one-GPU (unsegmented) captures only")
        self.knn, self.M, self.mode = knn, n_windows, mode
        Ms = n_sweep_windows
        if audio is not None:                     # the caller's resident tensors: no copy in front of a replay
            self.audio, self.context = audio, context
        else:
            if knn.use_wavvq:
                self.audio = torch.zeros((Ms, db.Tv, 2), dtype=torch.int64, device=dev)
            else:
                self.audio = torch.zeros((Ms, db.T, db.F), dtype=torch.float32, device=dev)
            self.context = torch.zeros((Ms, db.R, db.Dt), dtype=torch.float32, device=dev)
        # seed block in pinned host memory: [0 : 128 CL] the clips' f32 phase blocks, then their i32 seed codes
        CL = self.CL
        self._seed_pin = torch.zeros((128 * CL + CL + 3,), dtype=torch.float32).pin_memory()
        self._seed_np = self._seed_pin.numpy()
        self._seed_code_np = self._seed_np[128 * CL:128 * CL + CL].view(np.int32)
        steps = knn.n_steps()
        self._n_c, self._n_v = n_windows * num_frames_code, n_windows * steps
        # codes [CL][M*30] | votes [CL][M*steps] | status [CL][2]; then (encode leg) the code ids i32 [B][T/8]
        self._n_ints = CL * (self._n_c + self._n_v + 2)
        self._n_ids = 0
        if encoder is not None:
            if knn.db.world != 1 or knn.force_sharded:
                raise NotImplementedError("the encode leg is captured with the one-GPU step")
            B_, T_ = int(encode_input.shape[0]), int(encode_input.shape[1])
            self._ids_shape = (B_, T_ // encoder.hop)
            self._n_ids = B_ * (T_ // encoder.hop)
            self._n_flags = B_ if encode_precision == "f16x3" else 0
            if self._n_flags:
                encoder._hl_tolerance(T_)                  # (measured now: nothing inside the capture may synchronise)
        else:
            self._n_flags = 0
        self._pin = torch.empty((self._n_ints + self._n_ids + self._n_flags,), dtype=torch.int32).pin_memory()
        self._pin_np = self._pin.numpy()
        self._watch = self._pin_np[self._n_ints - 2 * CL + 1:self._n_ints:2]      # every clip's status[1]
        self._in_flight = False
        self._n_sweep, self._off = Ms, window_offset
        self.graph = None
        self.captures = 0

    def _capture(self):
        knn = self.knn
        dev = knn.db.device
        ptrs = (self._seed_pin.data_ptr() + 4 * 128 * self.CL, self._seed_pin.data_ptr())
        if self.enc is not None:
            self._enc_stream = torch.cuda.Stream(dev, priority=int(__import__("os").environ.get("QPG_ENC_PRIO", "0")))
            self._enc_gate, self._enc_done = torch.cuda.Event(), torch.cuda.Event()
            ids_pin = self._pin[self._n_ints:self._n_ints + self._n_ids]
            flags_pin = self._pin[self._n_ints + self._n_ids:]

        import os as _os
        # where the encode leg forks ("start": beside the whole match, the default; "sweep_end": behind the audio sweep;
        # "serial": no branch).  Measured with 16 clips + 96 windows per replay (experiments/round_scripts/r05_pass_e.sh, round 5's kernels):
        # start 3.80 / 3.19 ms (f32 / f16x3 encode), sweep_end 3.94 / 3.44, serial 4.27 / 3.60; stream priorities on either
        # branch only slow the step down.
        enc_at = _os.environ.get("QPG_ENCODE_AT", "start")

        def encode_leg():
            # the encode leg: a branch of its own (independent work: DB-side pose windows, make_beat_dataset.py:314-316),
            # its ids narrowed to i32 and copied into the replay's pinned result block; joined in front of the walk's
            # last kernel, whose final stores (the status words, behind a system-scope fence) are what the host waits
            # for.  (Forked behind the audio sweep - the first arrangement - its convolutions delay the selects' blocks on
            # the other queue by up to 0.6 ms: profiles/r05_step_timeline_c16_*.md.)
            main = torch.cuda.current_stream(dev)
            if enc_at == "serial":                 # (measurements) no branch: the encode between the sweep and the selects
                if self._n_flags:
                    ids, flags = self.enc.encode_f16x3_device(self.enc_x)
                    flags_pin.copy_(flags, non_blocking=True)
                else:
                    ids = self.enc.encode_fused(self.enc_x)
                ids_pin.copy_(ids.reshape(-1).to(torch.int32), non_blocking=True)
                self._enc_done.record(main)
                return
            self._enc_gate.record(main)
            self._enc_stream.wait_event(self._enc_gate)
            with torch.cuda.stream(self._enc_stream):
                if self._n_flags:
                    ids, flags = self.enc.encode_f16x3_device(self.enc_x)
                    flags_pin.copy_(flags, non_blocking=True)
                else:
                    ids = self.enc.encode_fused(self.enc_x)
                ids_pin.copy_(ids.reshape(-1).to(torch.int32), non_blocking=True)
                self._enc_done.record(self._enc_stream)

        def body():
            main = torch.cuda.current_stream(dev)
            if self._doorbell:                   # (first node: everything below is ordered behind it)
                _lib.call("qpg_doorbell_wait", dev, self._db_cnt, self._db_go.data_ptr(), 2000)
            if self.enc is not None:
                if enc_at == "start" or self.mode == MODE_TXT or knn.use_wavvq:
                    encode_leg()
                else:
                    knn.after_sweep = encode_leg
            elif self._sweep_flag is not None:
                knn.after_sweep = lambda: _lib.call("qpg_signal_i32", dev, self._sweep_flag.data_ptr(), 1)
            try:
                T = knn.sweep_tables(self.audio, self.context, self._n_sweep, self.mode, owner_blocks=self.owner_blocks,
                                     for_walk=True)
            finally:
                knn.after_sweep = None
            if self.enc is not None:
                main.wait_event(self._enc_done)
            return knn.walk(T, self.M, self._off, self.mode, sync=False, seed_ptrs=ptrs, out_pin=self._pin,
                            n_chains=self.CL)
        import os as _os
        if self._doorbell:
            self._db_go_np[0] = 0x7fffffff        # (the warm-up passes below run the doorbell kernel eagerly: open)
        s = torch.cuda.Stream(device=dev, priority=int(_os.environ.get("QPG_GRAPH_PRIO", "0")))   # (measurements)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            for _ in range(2):                       # warm-up: caches, lazy module loads, workspaces
                body()
        torch.cuda.current_stream(dev).wait_stream(s)
        torch.cuda.synchronize(dev)
        if self.segmented:
            from . import parallel as _par
            rec = _par.SegmentRecorder()
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            with torch.cuda.stream(s):
                _par._recorder = rec
                try:
                    rec.begin()
                    self.out = body()
                    rec.end()
                finally:
                    _par._recorder = None
                    if rec._g is not None:          # an exception inside a segment: close the capture before it propagates
                        try:
                            rec._g.capture_end()
                        except Exception:
                            pass
            torch.cuda.current_stream(dev).wait_stream(s)
            torch.cuda.synchronize(dev)
            self._program, self.segment_kinds, self._rec = rec.program, rec.kinds, rec
            self.graph = rec
            self.captures += 1
            return
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self.out = body()
        self.graph = g
        self.captures += 1
        if self._doorbell:
            torch.cuda.synchronize(dev)
            self._db_seq = int(self._db_cnt.item())        # (the warm-up passes took sequence numbers too)
            self._db_go_np[0] = self._db_seq               # closed: replay number _db_seq + 1 waits for launch()

    def _set_seed(self, seed_code, seed_phase):
        """One (code, phase block) for every clip, or one per clip (sequence of n_clips codes, [n_clips][8][16] blocks)."""
        CL, K = self.CL, self.knn.db.K
        if CL == 1 and type(seed_code) is int and type(seed_phase) is np.ndarray and seed_phase.dtype == np.float32 \
                and seed_phase.size == 128:
            # the one-clip step's usual call (a Python int and the previous clip's f32 phase block): no conversions - this
            # function is ~8 us of the ~23 us the host spends per replay otherwise
            if not 0 <= seed_code < K:
                raise ValueError("seed codes: %d values in [0, %d) wanted" % (CL, K))
            self._seed_np[:128] = seed_phase.reshape(-1)
            self._seed_code_np[0] = seed_code
            return
        sc = np.asarray(seed_code, np.int64).reshape(-1)
        if sc.size == 1:
            sc = np.repeat(sc, CL)
        if sc.size != CL or (sc < 0).any() or (sc >= K).any():
            raise ValueError("seed codes: %d values in [0, %d) wanted" % (CL, K))
        if isinstance(seed_phase, torch.Tensor):
            seed_phase = seed_phase.detach().cpu().numpy()
        sp = np.asarray(seed_phase, np.float32).reshape(-1)
        if sp.size == 128:
            sp = np.tile(sp, CL)
        if sp.size != 128 * CL:
            raise ValueError("seed phases: [n_clips][8][16] floats wanted")
        self._seed_np[:128 * CL] = sp
        self._seed_code_np[:] = sc.astype(np.int32)

    def launch(self, seed_code, seed_phase):
        """Replay on the current stream without waiting (inputs: whatever the bound / static buffers hold).  ONE replay in
        flight per graph: the seed block and the pinned result block are single buffers - the previous replay
        must have been collected with wait_ints()."""
        if self._in_flight:
            raise RuntimeError("ClipGraph.launch: the previous replay has not been collected (wait_ints)")
        self._set_seed(seed_code, seed_phase)
        if self.graph is None:
            self._capture()
        self._pin_np.fill(_PIN_SENTINEL)
        if self._sweep_flag_np is not None:
            self._sweep_flag_np[0] = 0
        self._in_flight = True
        if self.segmented:
            for f in self._program:                 # hipGraphLaunch, collective, hipGraphLaunch, ...
                f()
        elif self._doorbell:
            self._db_seq += 1
            self._db_go_np[0] = self._db_seq        # ring: the seed block and the sentinels above are final (x86 store order)
            if self._prelaunched:
                self._prelaunched = False           # (already enqueued: the store started it)
            else:
                self.graph.replay()
        else:
            self.graph.replay()

    def prelaunch(self):
        """doorbell graphs: enqueue the NEXT replay now (on the current stream, behind the one in flight); it waits at its
        first node until the next launch() rings.  At most one pre-launched replay; a pre-launched replay must be consumed
        by launch() or drain() before anything else is enqueued on this stream or the device is synchronised."""
        if not self._doorbell:
            raise RuntimeError("ClipGraph.prelaunch needs doorbell=True")
        if self._prelaunched:
            return
        if self._in_flight:
            # (re-launching a graph exec whose previous launch is still executing is legal HIP, and a one-branch graph
            # survives it, but this capture has cross-queue edges whose signals belong to the exec: measured, intermittently
            # a replay whose last kernel never ran.  Two captures taking turns - SerialReplayer - never do it.)
            raise RuntimeError("ClipGraph.prelaunch: this graph's own replay is still in flight; alternate two captures "
                               "(SerialReplayer)")
        if self.graph is None:
            self._capture()
        self.graph.replay()
        self._prelaunched = True

    def drain(self):
        """Run and discard a pre-launched replay nobody will use (the caller leaves the replay loop: an eager re-match, the
        end of a run).  Needs the replay in flight, if any, to have been collected."""
        if not getattr(self, "_prelaunched", False):
            return
        if self._in_flight:
            raise RuntimeError("ClipGraph.drain: collect the replay in flight first (wait_ints)")
        self._pin_np.fill(_PIN_SENTINEL)
        self._in_flight = True
        self._db_seq += 1
        self._db_go_np[0] = self._db_seq
        self._prelaunched = False
        self.wait_ints()

    def sweep_done(self):
        """sweep_signal graphs: has the replay in flight passed its audio sweep?  (True when nothing is in flight.)"""
        return (not self._in_flight) or self._sweep_flag_np is None or self._sweep_flag_np[0] != 0

    def wait_ints(self):
        """Host-side wait for the replay's last store (the status words, behind a system-scope fence); returns a copy of
        codes [n_clips][M*30] | votes [n_clips][M*steps] | status [n_clips][2] (| the encode leg's ids) as int32.  The
        caller hands every clip's status pair to CodeKNN.check_status().  A replay that raised the trouble word leaves the
        matcher's sticky word CLEARED (later replays of the same graph must not inherit it; the flagged clip is
        re-matched by the caller on a path that cannot raise it)."""
        try:
            out = _wait_pinned(self._pin_np, torch.cuda.current_stream(self.knn.db.device), self._watch)
        except RuntimeError as e:
            if self._doorbell:                  # (diagnostics: where host and device stand in the doorbell protocol)
                torch.cuda.synchronize(self.knn.db.device)
                raise RuntimeError("%s [doorbell: host seq %d, go %d, device counter %d, prelaunched %s]"
                                   % (e, self._db_seq, int(self._db_go_np[0]), int(self._db_cnt.item()), self._prelaunched))
            raise
        finally:
            self._in_flight = False
        st = out[self._n_ints - 2 * self.CL:self._n_ints].reshape(self.CL, 2)
        if (st[:, 1] != 0).any():
            self.knn.clear_flags()
        return out

    def statuses(self, ints):
        """[n_clips][2] status pairs of wait_ints()' array."""
        return ints[self._n_ints - 2 * self.CL:self._n_ints].reshape(self.CL, 2)

    def codes(self, ints):
        """[n_clips][M][30] codes of wait_ints()' array."""
        return ints[:self.CL * self._n_c].reshape(self.CL, self.M, num_frames_code)

    def encoded_ids(self, ints):
        """The encode leg's ids [B][T/8] (int32) of wait_ints()' array.  encode_precision "f16x3": windows the margin check
        flagged are encoded again on the f32 kernels first (VQVAE.resolve_f16x3; counted in `enc_redone`)."""
        ids = ints[self._n_ints:self._n_ints + self._n_ids].reshape(self._ids_shape)
        if self._n_flags:
            flags = ints[self._n_ints + self._n_ids:self._n_ints + self._n_ids + self._n_flags]
            if flags.any():
                ids, n = self.enc.resolve_f16x3(self.enc_x, ids, flags)
                self.enc_redone += n
        return ids

    def run_ints(self, seed_code, seed_phase):
        """One replay on the bound inputs, ending with the integer results on the host (bench.py's graph step)."""
        self.launch(seed_code, seed_phase)
        return self.wait_ints()

    def run(self, test_audio, test_context, seed_code, seed_phase):
        """Copies the clip into the static buffers (skipped for the tensors the graph is bound to) and replays.  Returns
        (codes i32 [M,30], phases f32 [M,steps,8,16] (device), votes i32 [M,steps], status i32 [2]); the integer results
        are host tensors (copies).  The caller must hand the two status ints to CodeKNN.check_status() before using the
        codes: status[1] != 0 (GuardOverflow) means this clip has to be matched with CodeKNN.match_clip / rematch_exact
        instead."""
        if test_audio.data_ptr() != self.audio.data_ptr():
            self.audio.copy_(test_audio, non_blocking=True)
        if test_context.data_ptr() != self.context.data_ptr():
            self.context.copy_(test_context, non_blocking=True)
        if self.CL != 1:
            raise NotImplementedError("run() returns one clip; several clips: run_ints() + codes() / statuses()")
        ints = torch.from_numpy(self.run_ints(seed_code, seed_phase))
        n_c, n_v = self._n_c, self._n_v
        return (ints[:n_c].view(self.M, num_frames_code), self.out[1], ints[n_c:n_c + n_v].view(self.M, -1),
                ints[n_c + n_v:n_c + n_v + 2])


class ClipPipeline:
    """Several independent clips in flight on one GPU (a matching service's throughput mode).

    A clip's select, rank fusion, walk and D2H copy occupy a fraction of the CUs for ~25 % of its latency; with `depth`
    lanes - one CodeKNN (its own workspaces and side stream), one HIP stream and one pinned result buffer each - the
    next clip's sweeps run underneath them.  Every clip goes through exactly the launches of CodeKNN.match_clip, so
    the results are the same arrays; only the host's wait moves from the end of a clip to `collect`.
    Measured (bench.py `pipelined`, 24 s clip vs 2048 windows, end of round 3): 0.33-0.37 ms per clip one at a time,
    0.25-0.30 with three lanes (round 2: 0.542 / 0.459).
    Row-sharded databases (round 4): every rank submits the SAME clips in the same order; a lane's collectives (the
    all-gather form: tables + responses) are issued in that order on every rank, torch.distributed serialises them on its
    communicator stream, and every rank walks every clip (replicated walk, as CodeKNN.match_clip on a sharded DB).  A flagged
    clip is re-matched collectively: the trouble word every rank reads is the same."""

    def __init__(self, db, depth=2, rng=None, **knn_flags):
        if depth < 1:
            raise ValueError("depth must be >= 1")
        self.db = db
        self.lanes = []
        for _ in range(depth):
            knn = CodeKNN(db, rng=rng, **knn_flags)
            self.lanes.append(dict(knn=knn, stream=torch.cuda.Stream(db.device), done=torch.cuda.Event(), ints=None,
                                   phase=None, shape=None, busy=False, inputs=None))
        self._next = 0

    @property
    def depth(self):
        return len(self.lanes)

    def submit(self, test_interp, test_context, n_windows, mode=MODE_AUD_TXT, seed_code=None, seed_phase=None):
        """Enqueue one clip on the next lane (which must have been collected) and return its ticket."""
        t = self._next
        ln = self.lanes[t]
        if ln["busy"]:
            raise RuntimeError("lane %d still holds an uncollected clip: collect() it first" % t)
        knn, dev = ln["knn"], self.db.device
        if seed_code is None:
            seed_code, seed_phase = knn.init_code_phase()
        ln["stream"].wait_stream(torch.cuda.current_stream(dev))       # the caller's inputs
        with torch.cuda.stream(ln["stream"]):
            T = knn.sweep_tables(test_interp.contiguous(), test_context, n_windows, mode, for_walk=True)
            oc, op, ov, st = knn.walk(T, n_windows, 0, mode, seed_code, seed_phase, sync=False)
            ints = knn._last_ints                                        # codes | votes | status, one buffer (walk)
            if ln["ints"] is None or ln["ints"].numel() < ints.numel():
                ln["ints"] = torch.empty((ints.numel(),), dtype=torch.int32).pin_memory()
                ln["phase"] = torch.empty((op.numel(),), dtype=torch.float32).pin_memory()
            ln["ints"][:ints.numel()].copy_(ints, non_blocking=True)
            ln["phase"][:op.numel()].copy_(op.reshape(-1), non_blocking=True)
            ln["done"].record(ln["stream"])
            # (the device results were allocated under this lane's stream: the caching allocator hands their blocks
            # back to this lane only, whose next submit is ordered after these copies)
        ln["shape"] = (tuple(oc.shape), tuple(ov.shape), tuple(op.shape))
        # the lane keeps the clip's inputs until collect(): the caller may drop or reuse its tensors while the lane's
        # stream is still reading them (no record_stream needed), and a flagged clip is re-matched from them
        ln["inputs"] = (test_interp, test_context, n_windows, mode, seed_code, seed_phase)
        ln["busy"] = True
        self._next = (t + 1) % len(self.lanes)
        return t

    def collect(self, ticket):
        """Wait for the clip of `ticket`; returns what CodeKNN.match_clip returns: (codes int64 [M,30], phases f32,
        votes) as NumPy arrays (copies: the lane's buffers are reused by its next clip)."""
        ln = self.lanes[ticket]
        if not ln["busy"]:
            raise RuntimeError("lane %d holds no clip" % ticket)
        ln["done"].synchronize()
        ln["busy"] = False
        inputs, ln["inputs"] = ln["inputs"], None
        sc, sv, sp = ln["shape"]
        n_c, n_v = int(np.prod(sc)), int(np.prod(sv))
        ints = ln["ints"].numpy()
        try:
            CodeKNN.check_status(ints[n_c + n_v:n_c + n_v + 2])
        except GuardOverflow as e:
            # never return codes the guard could not vouch for: this clip again, now, on a path that cannot raise the word
            ti, tc, m, mode, seed_code, seed_phase = inputs
            with torch.cuda.stream(ln["stream"]):
                return ln["knn"].rematch(e.flags, ti.contiguous(), tc, m, mode, seed_code, seed_phase)
        codes = ints[:n_c].reshape(sc).astype(np.int64)
        votes = ints[n_c:n_c + n_v].reshape(sv).copy()
        phases = ln["phase"].numpy()[:int(np.prod(sp))].reshape(sp).copy()
        return codes, phases, votes

    @property
    def fallbacks(self):
        return sum(ln["knn"].fallbacks for ln in self.lanes)

    def match_clips(self, clips, mode=MODE_AUD_TXT, seeds=None):
        """clips: iterable of (test_interp, test_context, n_windows); seeds: optional list of (seed_code, seed_phase).
        Returns the list of match_clip results, in order."""
        out, pending = [], []
        for i, (ti, tc, m) in enumerate(clips):
            if len(pending) == len(self.lanes):
                out.append(self.collect(pending.pop(0)))
            sc, sp = seeds[i] if seeds is not None else (None, None)
            pending.append(self.submit(ti, tc, m, mode, sc, sp))
        while pending:
            out.append(self.collect(pending.pop(0)))
        return out


class SerialReplayer:
    """EXPERIMENTAL (round 6; off by default in bench.py: QPG_BENCH_DOORBELL=1).  One clip at a time without the launch
    overhead between two steps: two captures of the same step (ClipGraph(doorbell=True), each with its own matcher and
    workspaces) take turns on ONE stream - while capture A's replay runs, capture B's next replay is enqueued behind it
    (hipGraphLaunch: ~17 us of host time + the command processor's start-up) and waits at its first node; when A's codes are
    on the host and the next seed is written, ONE store rings B.  The GPU work of a step still starts only after the previous
    step's results have been read.  Measured: 0.2272-0.2277 against 0.2314-0.2346 ms per step (-2.4 %, alternating runs).
    WHY IT IS NOT THE DEFAULT: a graph launched on a stream on which another graph launch is still executing occasionally
    comes out with its kernels elided on this ROCm build - the doorbell node runs (the device counter advances), nothing
    else does, no error is reported: the first launch of a fresh capture behind a running replay reproducibly
    (experiments/doorbell_stress.py), afterwards about once in 30 000 steps; re-launching ONE exec behind its own running
    replay loses kernels far more often (which is why two captures take turns).  The host notices (the status word never
    arrives), step() resynchronises both captures and runs the step again plainly (`recovered` counts) - but a runtime
    that can drop ALL kernels of a launch is not one to put under the default path of a matcher whose bar is bit-exact codes."""

    def __init__(self, graphs):
        assert len(graphs) == 2 and all(g._doorbell for g in graphs)
        self.graphs = list(graphs)
        self._i = 0                                    # the capture whose replay is launched next
        self.recovered = 0

    def _resync(self):
        dev = self.graphs[0].knn.db.device
        torch.cuda.synchronize(dev)                    # (a pending pre-launched replay times out by itself: <= 2 s)
        for g in self.graphs:
            g._prelaunched = False
            g._in_flight = False
            g._db_seq = int(g._db_cnt.item())
            g._db_go_np[0] = g._db_seq

    def step(self, seed_code, seed_phase, more):
        """launch (ring) -> pre-launch the other capture if `more` steps follow -> wait; returns (ints, the capture)."""
        cur, nxt = self.graphs[self._i], self.graphs[self._i ^ 1]
        self._i ^= 1
        cur.launch(seed_code, seed_phase)
        if more:
            nxt.prelaunch()
        try:
            return cur.wait_ints(), cur
        except RuntimeError as e:
            if "status word" not in str(e) and "result words" not in str(e):
                raise
            self.recovered += 1                        # the launch's kernels never ran (see the class comment): once more, plainly
            self._resync()
            cur.launch(seed_code, seed_phase)
            return cur.wait_ints(), cur

    def drain(self):
        for g in self.graphs:
            g.drain()


class GraphPipeline:
    """Throughput mode on captured graphs (round 6): `depth` lanes, each ONE hipGraph of `clips_per_replay` independent clips
    (ClipGraph(n_clips): one batched sweep over all their queries - the database image comes out of HBM once per 48-query
    chunk's first touch and out of the XCD's L2 after that - the batched selects and ONE batched walk), replayed on the
    lane's own stream.  What it buys over a clip at a time (the one-clip step is 2x its sweep, ~100 us of
    near-empty dependent launches behind it; ClipPipeline's eager lanes are host-bound at ~0.25 ms per clip and their tails
    never ran under another clip's sweep, because a one-chunk sweep holds every register of every CU for its whole
    duration): the post-sweep chain is paid once per REPLAY, not once per clip; a host call costs one hipGraphLaunch per
    `clips_per_replay` clips; and a multi-chunk sweep finishes its blocks in rounds, so the other lane's tail kernels get
    CUs while it runs.  The reference's loop over test clips (GestureKNN.py:785-813) has no state between clips, so any
    grouping returns the same codes; every clip's integers reach the host before its ticket is collected, and a clip whose
    trouble word is raised is matched again eagerly (CodeKNN.rematch) before anything is returned.
    Latency per clip goes UP (a clip waits for its group): this is the serving-throughput figure, bench.py reports it
    beside the one-clip `value`, never instead."""

    def __init__(self, db, n_windows, clips_per_replay=4, depth=2, mode=MODE_AUD_TXT, rng=None, stagger=True, **knn_flags):
        if depth < 1 or clips_per_replay < 1:
            raise ValueError("depth and clips_per_replay must be >= 1")
        if db.world != 1:
            raise NotImplementedError("GraphPipeline: one GPU holding the whole database (clip-parallel across GPUs: one "
                                      "pipeline per rank, bench.py --scaling replicated)")
        self.db, self.M, self.G, self.mode = db, int(n_windows), int(clips_per_replay), mode
        self.lanes = []
        for _ in range(depth):
            knn = CodeKNN(db, rng=rng, **knn_flags)
            self.lanes.append(dict(knn=knn, stream=torch.cuda.Stream(db.device), graph=None, busy=False, seeds=None))
        self._next = 0
        self.rematched = 0
        # stagger: a lane's replay is launched only when the replay launched BEFORE it (another lane's) has passed its audio
        # sweep.  Left to themselves two lanes fall into lockstep - their sweeps share the CUs (each takes twice as long),
        # end together, and both tails then run on an otherwise idle GPU (profiles/r06_pipeline_timeline_4x2_lockstep.md:
        # 330 of every 1 240 us with no sweep resident).  Staggered, the sweeps follow one another back to back and a lane's
        # tail runs underneath the other lane's sweep.  The host learns "sweep over" from a word the replay stores into
        # pinned memory behind its sweep kernel (ClipGraph(sweep_signal=True), qpg_signal_i32).
        self.stagger = bool(stagger) and depth > 1 and mode in (MODE_AUD_TXT, MODE_AUD)
        self._last = None                       # lane of the most recently launched replay
        self.stagger_timeouts = 0

    @property
    def depth(self):
        return len(self.lanes)

    def _graph(self, ln):
        if ln["graph"] is None:
            ln["graph"] = ln["knn"].capture_clip_graph(self.M, self.mode, n_clips=self.G, sweep_signal=self.stagger)
        return ln["graph"]

    def _wait_previous_sweep(self):
        if not self.stagger or self._last is None:
            return
        g = self.lanes[self._last]["graph"]
        if g is None or g.sweep_done():
            return
        t0 = time.perf_counter()
        while not g.sweep_done():
            if time.perf_counter() - t0 > 0.05:         # (a replay whose sweep never signals: do not hang the pipeline)
                self.stagger_timeouts += 1
                return

    def buffers(self, lane):
        """The lane's static input tensors (audio [G*M, T, F] or wavvq ids, context [G*M, 30, Dt]): a caller that produces
        its clips on the device can write them here directly and submit(None, None, ...)."""
        g = self._graph(self.lanes[lane])
        return g.audio, g.context

    def submit(self, audio, context, seed_codes, seed_phases):
        """Enqueue `clips_per_replay` clips on the next lane (which must have been collected); returns the ticket.
        audio / context: the clips' windows back to back ([G*M, ...] device tensors), copied into the lane's static
        buffers on the lane's stream - or None: the buffers already hold them (buffers()).  seed_codes: G ints (or one for
        all); seed_phases: [G][8][16] floats (or one block for all)."""
        t = self._next
        ln = self.lanes[t]
        if ln["busy"]:
            raise RuntimeError("lane %d still holds an uncollected group: collect() it first" % t)
        g = self._graph(ln)
        dev = self.db.device
        st = ln["stream"]
        caller = torch.cuda.current_stream(dev)                # (the stream the caller produced its inputs on)
        with torch.cuda.stream(st):
            if audio is not None:
                st.wait_stream(caller)
                if audio.data_ptr() != g.audio.data_ptr():
                    g.audio.copy_(audio, non_blocking=True)
                if context is not None and context.data_ptr() != g.context.data_ptr():
                    g.context.copy_(context, non_blocking=True)
            if g.graph is None:
                g._capture()                     # (the first replay of a lane: capture before waiting for the other lane)
            self._wait_previous_sweep()
            g.launch(seed_codes, seed_phases)
        self._last = t
        ln["busy"], ln["seeds"] = True, (seed_codes, seed_phases)
        self._next = (t + 1) % len(self.lanes)
        return t

    def collect(self, ticket):
        """Wait for the group of `ticket`; returns a list of (codes int64 [M,30], votes i32 [M,steps]) per clip (copies)."""
        ln = self.lanes[ticket]
        if not ln["busy"]:
            raise RuntimeError("lane %d holds no group" % ticket)
        g, knn = ln["graph"], ln["knn"]
        with torch.cuda.stream(ln["stream"]):
            ints = g.wait_ints()
        ln["busy"] = False
        st = g.statuses(ints)
        n_c, n_v = g._n_c, g._n_v
        codes = ints[:self.G * n_c].reshape(self.G, self.M, num_frames_code).astype(np.int64)
        votes = ints[self.G * n_c:self.G * (n_c + n_v)].reshape(self.G, self.M, -1).copy()
        out = []
        for c in range(self.G):
            try:
                CodeKNN.check_status(st[c])
                out.append((codes[c], votes[c]))
            except GuardOverflow as e:
                # never return codes the guard could not vouch for: this clip again, eagerly, on a path that cannot raise
                # the word (the sticky word was cleared by wait_ints)
                sc, sp = ln["seeds"]
                sc_c = int(np.asarray(sc).reshape(-1)[c if np.asarray(sc).size > 1 else 0])
                sp_a = np.asarray(sp.detach().cpu().numpy() if isinstance(sp, torch.Tensor) else sp, np.float32).reshape(-1, 128)
                sp_c = sp_a[c if sp_a.shape[0] > 1 else 0].reshape(8, 16)
                with torch.cuda.stream(ln["stream"]):
                    r = knn.rematch(e.flags, g.audio[c * self.M:(c + 1) * self.M].contiguous(),
                                    g.context[c * self.M:(c + 1) * self.M].contiguous(), self.M, self.mode, sc_c, sp_c)
                self.rematched += 1
                out.append((r[0], r[2]))
        return out

    def phases(self, ticket):
        """The last collected group's phase blocks of lane `ticket` (device tensor [G*M, steps, 8, 16]; valid until the lane's
        next submit)."""
        return self.lanes[ticket]["graph"].out[1]

    def match_groups(self, groups, seeds):
        """groups: iterable of (audio, context) with G clips each; seeds: list of (seed_codes, seed_phases) per group.
        Returns the per-clip results in order."""
        out, pending = [], []
        for i, (a_, c_) in enumerate(groups):
            if len(pending) == len(self.lanes):
                out.extend(self.collect(pending.pop(0)))
            pending.append(self.submit(a_, c_, *seeds[i]))
        while pending:
            out.extend(self.collect(pending.pop(0)))
        return out
