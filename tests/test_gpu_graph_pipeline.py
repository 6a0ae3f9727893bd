"""GraphPipeline (round 6): several clips per captured replay, several replays in flight - the serving-throughput mode.
The reference's loop over test clips (GestureKNN.py:785-813) carries no state from clip to clip, so every grouping must
return, clip for clip, what CodeKNN.match_clip returns for that clip alone with the same seed; a clip whose trouble word
is raised inside a group is never handed out unguarded."""
import numpy as np
import pytest

from tests.helpers import fixture_arrays, load_golden

pytestmark = pytest.mark.gpu


def _db(N, seed):
    from qpgesture_amd import synth
    from qpgesture_amd.data_processing import interp_wavlm
    tr = synth.make_db(N, seed, 1024)
    return dict(interp=interp_wavlm(tr["wavlm"]), ctx=np.ascontiguousarray(tr["context"].squeeze(2)),
                code=synth.make_codes(N, seed + 1), phase=tr["phase_dense"], sig=synth.make_signature(seed + 2))


@pytest.mark.parametrize("G,depth", [(3, 2), (1, 3), (5, 1)])
def test_groups_in_flight_equal_one_clip_at_a_time(G, depth):
    import torch
    from qpgesture_amd.code_knn import CodeKNN, GestureDB, GraphPipeline
    A = _db(160, 410)
    db = GestureDB(A["code"], A["interp"], A["ctx"], A["phase"], A["sig"], device="cuda:0")
    knn = CodeKNN(db, rng=np.random.RandomState(4))
    M, n_groups = 2, 5
    g = torch.Generator(device="cpu").manual_seed(31 + G)
    groups, seeds, want = [], [], []
    for i in range(n_groups):
        ti = torch.randn((G * M, 180, 1024), generator=g).cuda()
        tc = torch.randn((G * M, 30, 384), generator=g).cuda()
        sc, sp = [], []
        for c in range(G):
            c_, p_ = knn.init_code_phase()
            sc.append(c_)
            sp.append(p_)
            want.append(knn.match_clip(ti[c * M:(c + 1) * M], tc[c * M:(c + 1) * M], M, seed_code=c_, seed_phase=p_))
        groups.append((ti, tc))
        seeds.append((sc, np.stack(sp)))
    pipe = GraphPipeline(db, M, clips_per_replay=G, depth=depth, rng=np.random.RandomState(5))
    got = pipe.match_groups(groups, seeds)
    assert len(got) == n_groups * G and pipe.rematched == 0
    for (codes, votes), w in zip(got, want):
        assert codes.dtype == np.int64 and np.array_equal(codes, w[0]) and np.array_equal(votes, w[2])
    assert len({tuple(w[0].reshape(-1)) for w in want}) > 1                       # the clips really differ
    # one capture per lane, whatever the number of groups; a lane refuses a second group before the first is collected
    assert all(ln["graph"] is None or ln["graph"].captures == 1 for ln in pipe.lanes)
    t = pipe.submit(*groups[0], *seeds[0])
    if depth == 1:
        with pytest.raises(RuntimeError):
            pipe.submit(*groups[1], *seeds[1])
    # phase blocks of the group: those of the clips alone
    first = pipe.collect(t)
    ph = pipe.phases(t).cpu().numpy().reshape(G, M, -1, 8, 16)
    for c in range(G):
        assert np.array_equal(first[c][0], want[c][0]) and np.array_equal(ph[c], want[c][1])
    # inputs written straight into the lane's buffers
    a_, c_ = pipe.buffers(pipe._next)
    a_.copy_(groups[2][0])
    c_.copy_(groups[2][1])
    again = pipe.collect(pipe.submit(None, None, *seeds[2]))
    for c in range(G):
        assert np.array_equal(again[c][0], want[2 * G + c][0])


def test_a_flagged_clip_inside_a_group_is_rematched():
    """The near-silent golden stretch (780 candidates within 1e-14 of each other: the capped lists overflow) as one clip of a
    group of two: the group's trouble word is raised, both clips are matched again eagerly, both results are those of the
    clips alone (whose own match re-matches the quiet one)."""
    import torch
    from qpgesture_amd.code_knn import CodeKNN, GestureDB, GraphPipeline
    g = load_golden("shipped_nearsilent_n48_m2_s50")
    ntr, nte, s0, s1, s2, s3, mf = [int(v) for v in g["meta"]]
    variant = (str(g["variant"]) or None) if "variant" in g.files else None
    A = fixture_arrays(ntr, nte, s0, s1, s2, s3, variant=variant)
    db = GestureDB(A["code"], A["tr_interp"], A["tr_ctx"], A["tr_phase"], A["sig"], device="cuda:0",
                   freq_rank=g["step_freq_score"])
    knn = CodeKNN(db, rng=np.random.RandomState(123456))
    te_i = torch.from_numpy(A["te_interp"]).cuda()
    te_c = torch.from_numpy(np.ascontiguousarray(A["te_ctx"])).cuda()
    sc, sp = knn.init_code_phase()
    want = [knn.match_clip(te_i[c:c + 1], te_c[c:c + 1], 1, seed_code=sc, seed_phase=sp) for c in range(2)]
    assert knn.fallbacks >= 1                                                   # (the quiet window needs the uncapped path)
    pipe = GraphPipeline(db, 1, clips_per_replay=2, depth=2, rng=np.random.RandomState(1))
    for _ in range(2):                                                          # the sticky word must not leak into the next group
        got = pipe.collect(pipe.submit(te_i[:2], te_c[:2], sc, sp))
        for c in range(2):
            assert np.array_equal(got[c][0], want[c][0]) and np.array_equal(got[c][1], want[c][2])
    assert pipe.rematched >= 2


def test_prelaunched_replays_behind_the_doorbell_equal_plain_replays():
    """EXPERIMENTAL path (off by default): ClipGraph(doorbell=True) + SerialReplayer - the next step's replay (the OTHER of two
    captures) is enqueued while the current one runs and starts when launch() rings.  Codes, votes and status of every step
    equal the eager path's with a DIFFERENT seed per step (the seed is written after the pre-launch: the replay must read it
    when it runs, not when it was enqueued); a launch whose kernels the runtime dropped is noticed and repeated (`recovered`);
    a capture refuses to be pre-launched behind its own replay; drain() consumes a pre-launched replay nobody wants; a
    doorbell that is never rung times out instead of hanging the device."""
    import time
    import torch
    from qpgesture_amd.code_knn import CodeKNN, GestureDB, SerialReplayer
    A = _db(160, 510)
    db = GestureDB(A["code"], A["interp"], A["ctx"], A["phase"], A["sig"], device="cuda:0")
    knn = CodeKNN(db, rng=np.random.RandomState(4))
    M = 2
    g = torch.Generator(device="cpu").manual_seed(77)
    ti = torch.randn((M, 180, 1024), generator=g).cuda()
    tc = torch.randn((M, 30, 384), generator=g).cuda()
    seeds = [knn.init_code_phase() for _ in range(9)]
    want = [knn.match_clip(ti, tc, M, seed_code=c_, seed_phase=p_) for c_, p_ in seeds]
    cgs = [CodeKNN(db, rng=np.random.RandomState(5 + i)).capture_clip_graph(M, audio=ti, context=tc, doorbell=True)
           for i in range(2)]
    sr = SerialReplayer(cgs)
    n_c = M * 30
    for rounds in range(3):                                  # (the hand-over in both directions, many times)
        for i, (c_, p_) in enumerate(seeds):
            got, used = sr.step(c_, p_, more=i + 1 < len(seeds))
            assert np.array_equal(got[:n_c].reshape(M, 30), want[i][0]) and not got[-2:].any(), (rounds, i)
            assert np.array_equal(got[n_c:-2].reshape(M, -1), want[i][2])
    print("doorbell steps recovered after a lost launch: %d of %d" % (sr.recovered, 3 * len(seeds)))
    assert sr.recovered <= 2
    assert all(c.captures == 1 and not c._prelaunched and not c._in_flight for c in cgs)
    torch.cuda.synchronize()
    cg = cgs[0]
    # a capture is never pre-launched behind its own replay
    cg.launch(*seeds[0])
    with pytest.raises(RuntimeError):
        cg.prelaunch()
    assert np.array_equal(cg.wait_ints()[:n_c].reshape(M, 30), want[0][0])
    torch.cuda.synchronize()
    # a pre-launched replay nobody wants
    cg.prelaunch()
    cg.drain()
    assert not cg._prelaunched and not cg._in_flight
    torch.cuda.synchronize()                                 # nothing is left waiting on the device
    assert np.array_equal(cg.run_ints(*seeds[1])[:n_c].reshape(M, 30), want[1][0])
    torch.cuda.synchronize()
    # the wait is bounded: a replay whose doorbell is never rung starts by itself after the timeout (2 s) - the device
    # cannot hang on a host that went away
    cg.prelaunch()
    t0 = time.time()
    torch.cuda.synchronize()
    assert 1.0 < time.time() - t0 < 10.0
    sr._resync()                                             # (that replay took a sequence number)
    assert np.array_equal(cg.run_ints(*seeds[2])[:n_c].reshape(M, 30), want[2][0])
