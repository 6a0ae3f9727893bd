"""Section times of the mixed select (block 0) on the bench workload: run tools/step_loop.py's steps, read the stamps.
usage: QPG_LIB_PATH=experiments/select_prof/libqpg_prof.so python experiments/select_prof/run.py"""
import ctypes, os, sys, runpy
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
sys.argv = ["step_loop.py", "10"]
runpy.run_path(os.path.join(root, "tools", "step_loop.py"), run_name="__main__")
from qpgesture_amd import _lib
lib = _lib.load()
buf = (ctypes.c_longlong * 96)()
lib.qpg_debug_select_prof.argtypes = [ctypes.c_void_p]
assert lib.qpg_debug_select_prof(buf) == 0
names = {0: "init", 1: "pass 1 (stream)", 2: "pass 2 (pot list)", 3: "list (a) + v", 4: "rank_pass(false)", 5: "list (b)",
         6: "park", 7: "phase 2 load", 8: "tier 1/2 merge + tables", 9: "rank_pass(true)", 10: "rank-level tie scan",
         11: "tier-2 refine", 12: "rank_pass again"}
snames = {1: "stream slice", 2: "merge minima (global atomics)", 3: "survivors -> workspace"}
for ph in (0, 1, 2):
    st = [buf[ph * 16 + i] for i in range(16)]
    ck = [buf[(3 + ph) * 16 + i] for i in range(16)]
    prev = pck = None
    print("phase", ph, "(mixed_stream_kernel, block (0, 0))" if ph == 0 else "")
    names.update({13: "cut: sort", 14: "cut: lo / hi", 15: "cut: probe + Umax"})
    for i in sorted(range(16), key=lambda j: st[j]):            # (in time order: the cut's stamps sit between 2 and 3)
        if st[i] == 0:
            continue
        if prev is not None:
            print("   %-26s %6.2f us  %7d shader clocks" % ((snames if ph == 0 else names).get(i, str(i)), (st[i] - prev) / 100.0, ck[i] - pck))
        prev, pck = st[i], ck[i]
