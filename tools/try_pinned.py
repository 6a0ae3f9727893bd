"""Experiment: the walk's integer results (codes | votes | status) written by the last kernel STRAIGHT into pinned host
memory (zero-copy) + a stream synchronize, against the device buffer + .cpu() of the product path."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["step_loop.py", "5"]
import runpy
g = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "step_loop.py"), run_name="__main__")
import numpy as np, torch
from qpgesture_amd import _lib
from qpgesture_amd.code_knn import MODE_AUD_TXT, num_frames_code
knn, te_i, te_c, M, sc, spd = g["knn"], g["te_i"], g["te_c"], g["M"], g["sc"], g["spd"]
db, dev = knn.db, knn.db.device
steps = knn.n_steps()
n_c, n_v = M * num_frames_code, M * steps
pin = torch.empty((n_c + n_v + 2,), dtype=torch.int32).pin_memory()
base = pin.data_ptr()
pin_np = pin.numpy()
out_phase = torch.empty((M, steps, 8, 16), dtype=torch.float32, device=dev)
gate = torch.empty((3, M * steps, db.K), dtype=torch.int32, device=dev)
a_cidx, a_pslot, a_G = knn._audio_grid()


def step_pinned():
    T = knn.sweep_tables(te_i, te_c, M)
    _lib.call("qpg_match_steps", dev, T["aud_rank"], T["aud_idx"], T["txt_rank"], T["txt_idx"], db.pos_rank, db.freq_rank,
              db.code, db.code.shape[1], a_cidx, a_pslot, a_G, db.txt_cidx, db.txt_pslot, db.Gt, db.phase, db.Tp,
              MODE_AUD_TXT, M, steps, db.K, int(sc), spd, gate, base, out_phase, base + 4 * n_c, base + 4 * (n_c + n_v),
              knn._guard_stats[1:2])
    torch.cuda.current_stream(dev).synchronize()
    return pin.numpy()[:n_c].copy()


def step_spin():
    T = knn.sweep_tables(te_i, te_c, M)
    pin_np[-1] = -1234567                                   # the kernel's LAST store overwrites it
    _lib.call("qpg_match_steps", dev, T["aud_rank"], T["aud_idx"], T["txt_rank"], T["txt_idx"], db.pos_rank, db.freq_rank,
              db.code, db.code.shape[1], a_cidx, a_pslot, a_G, db.txt_cidx, db.txt_pslot, db.Gt, db.phase, db.Tp,
              MODE_AUD_TXT, M, steps, db.K, int(sc), spd, gate, base, out_phase, base + 4 * n_c, base + 4 * (n_c + n_v),
              knn._guard_stats[1:2])
    while pin_np[-1] == -1234567:
        pass
    return pin_np[:n_c].copy()


def step_cpu():
    T = knn.sweep_tables(te_i, te_c, M)
    knn.walk(T, M, 0, seed_code=sc, seed_phase=spd, sync=False)
    return knn._last_ints.cpu().numpy()[:n_c]


a, b = step_pinned(), step_cpu()
print("equal:", np.array_equal(a, b))
print("spin equal:", np.array_equal(step_spin(), b))
for name, fn in (("cpu()", step_cpu), ("pinned", step_pinned), ("spin", step_spin), ("cpu()", step_cpu), ("pinned", step_pinned), ("spin", step_spin)):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(400):
        fn()
    torch.cuda.synchronize()
    print("%-7s %.4f ms/step" % (name, (time.perf_counter() - t0) / 400 * 1e3))
