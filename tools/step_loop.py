"""The bench.py matching step (N_db=2048, M=6) in a bare loop: 5 warm-up + <steps> iterations, eager or graph replay."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from qpgesture_amd import synth
from qpgesture_amd.code_knn import CodeKNN, GestureDB

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
graph = len(sys.argv) > 2 and sys.argv[2] == "graph"
N, M = 2048, 6
CL = int(os.environ.get("QPG_LOOP_CLIPS", "1"))            # clips per step (one batched sweep, one set of walk launches)
F16 = os.environ.get("QPG_LOOP_F16", "0") == "1"            # the track stored in f16 (one-plane sweep)
ENC = int(os.environ.get("QPG_LOOP_ENC", "0"))             # pose windows VQ-VAE-encoded in the step
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
interp = torch.randn((N, 180, 1024), device=dev)
ctx = rng.standard_normal((N, 30, 384)).astype(np.float32)
phase = rng.standard_normal((N, 240, 4, 8)).astype(np.float32)
db = GestureDB(synth.make_codes(N, 2), interp, ctx, phase, synth.make_signature(3), device=dev,
               feature_dtype="f16" if F16 else "f32")
knn = CodeKNN(db, rng=np.random.RandomState(123456))
if "QPG_AUDIO_FIRST" in os.environ:
    knn.audio_first = os.environ["QPG_AUDIO_FIRST"] == "1"
if "QPG_TEXT_AFTER" in os.environ:
    knn.text_after_sweep = os.environ["QPG_TEXT_AFTER"] == "1"
if os.environ.get("QPG_FORCE_SHARDED") == "1":          # the row-shard code path over RCCL with world_size 1
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    knn.force_sharded = True
if "QPG_GATE_DEDUP" in os.environ:                    # from how many chains the gate table is deduplicated (0: never)
    from qpgesture_amd import _lib as _l
    _l.set_option(dev, _l.QPG_OPT_GATE_DEDUP_FROM_CHAINS, int(os.environ["QPG_GATE_DEDUP"]))
knn.audio_kernel = os.environ.get("QPG_AUDIO_KERNEL", "hl")
if "QPG_LOOP_PROBE" in os.environ:                    # probe codes of the walk-relevance cut's bound (default 64)
    knn.rank_cut_probe = int(os.environ["QPG_LOOP_PROBE"])
if "QPG_SPLIT_FUSE" in os.environ:                    # 0: the rank fusion as ONE launch in the walk (rounds 3-5), 1: per modality
    knn.split_fuse = os.environ["QPG_SPLIT_FUSE"] == "1"
knn.tie_eps = float(os.environ.get("QPG_TIE_EPS", knn.tie_eps))
te_i = torch.randn((M * CL, 180, 1024), device=dev)
te_c = torch.randn((M * CL, 30, 384), device=dev)
enc = enc_x = None
if ENC:
    from qpgesture_amd.vqvae import VQVAE
    enc = VQVAE(None, 135, device=dev).load_state_dict(synth.make_vqvae_state_dict(7))
    enc_x = torch.randn((ENC, 240, 135), device=dev)
    enc.encode(enc_x)
sc, sp = knn.init_code_phase()
spd = torch.from_numpy(sp).to(dev)
from qpgesture_amd import code_knn as _ck
mode = getattr(_ck, os.environ.get("QPG_LOOP_MODE", "MODE_AUD_TXT"))        # MODE_AUD: audio side only (measurements)
g = knn.capture_clip_graph(M, mode=mode, audio=te_i, context=te_c, owner_blocks=knn.force_sharded, n_clips=CL,
                           encoder=enc, encode_input=enc_x,
                           encode_precision=os.environ.get("QPG_LOOP_ENC_PREC", "f32"),
                           doorbell=os.environ.get("QPG_LOOP_DOORBELL", "0") == "1") if graph else None
DOOR = graph and os.environ.get("QPG_LOOP_DOORBELL", "0") == "1"     # the next replay pre-launched behind the doorbell
if DOOR:                                                              # (two captures taking turns: code_knn.SerialReplayer)
    knn2 = CodeKNN(db, rng=np.random.RandomState(123456))
    g2 = knn2.capture_clip_graph(M, mode=mode, audio=te_i, context=te_c, n_clips=CL, doorbell=True)
    SR = _ck.SerialReplayer([g, g2])
spb = torch.from_numpy(np.tile(sp.reshape(1, -1), (CL, 1))).to(dev)


def step(more=False):
    if DOOR:
        return SR.step(sc, sp, more)[0]
    if graph:
        return g.run_ints(sc, sp)
    if enc is not None:
        ids = enc.encode(enc_x)[0]
    T = knn.sweep_tables(te_i, te_c, M * CL, mode=mode, owner_blocks=knn.force_sharded, for_walk=True)
    if CL > 1:
        knn.walk_batch(T, M, CL, [sc] * CL, spb, mode=mode)
        out = knn._last_ints.cpu()
        if enc is not None:
            ids.cpu()
        return out
    return knn.walk(T, M, 0, mode=mode, seed_code=sc, seed_phase=spd, sync="ints")      # (codes | votes | status, pinned host memory)


import time
prio = int(os.environ.get("QPG_LOOP_PRIO", "0"))      # -1: the loop's stream is a high-priority stream (the text side's is not)
if prio:
    torch.cuda.set_stream(torch.cuda.Stream(dev, priority=prio))
for _ in range(5):
    step()
torch.cuda.synchronize()
gap = float(os.environ.get("QPG_LOOP_GAP_MS", "0"))     # idle time between steps (what the sweep takes on a chip that rests)
t0 = time.perf_counter()
for i_ in range(steps):
    step(more=i_ + 1 < steps) if DOOR else step()
    if gap:
        time.sleep(gap * 1e-3)
torch.cuda.synchronize()
print("%s: %.4f ms/step" % ("graph" if graph else "eager", (time.perf_counter() - t0) / steps * 1e3))
