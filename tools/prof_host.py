import os, sys, cProfile, pstats
sys.path.insert(0, os.getcwd())
sys.argv = ["step_loop.py", "5"]
import runpy
g = runpy.run_path("tools/step_loop.py", run_name="__main__")
step = g["step"]
import torch
pr = cProfile.Profile()
pr.enable()
for _ in range(500):
    step()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
