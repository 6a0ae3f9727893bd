"""What the host does between the start of a matching step and its FIRST launch (sweep_tables -> sweep_audio ->
qpg_audio_pack_queries_hl): the first _lib.call is replaced by an exception, the truncated step is timed and profiled."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["step_loop.py", "5"]
import runpy
g = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "step_loop.py"), run_name="__main__")
from qpgesture_amd import _lib
import torch
knn, te_i, te_c, M = g["knn"], g["te_i"], g["te_c"], g["M"]


class Stop(Exception):
    pass


def stop(*a):
    raise Stop()


orig = _lib.call
_lib.call = stop


def prelude():
    try:
        knn.sweep_tables(te_i, te_c, M)
    except Stop:
        knn.__dict__.pop("_after_sweep_launch", None)


for _ in range(200):
    prelude()
t0 = time.perf_counter()
for _ in range(5000):
    prelude()
print("prelude: %.1f us" % ((time.perf_counter() - t0) / 5000 * 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(3000):
    prelude()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
