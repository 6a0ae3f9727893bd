#!/usr/bin/env python
"""rocprofv3 --kernel-trace --stats output directory -> (a) copy of the kernel_stats CSV, (b) a markdown table.
    python tools/make_profile_summary.py <rocprof dir> <out prefix (profiles/r02_...)> "<title / command line>"
"""
import csv
import glob
import os
import shutil
import sys

d, prefix, title = sys.argv[1], sys.argv[2], sys.argv[3]
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True))[-1]
shutil.copy(f, prefix + "_kernel_stats.csv")
rows = list(csv.DictReader(open(f)))
with open(prefix + "_summary.md", "w") as o:
    o.write("# %s\n\n| kernel | calls | avg us | min us | max us | %% |\n|---|---|---|---|---|---|\n" % title)
    for r in rows:
        if float(r["Percentage"]) < 0.02:
            continue
        o.write("| %s | %s | %.1f | %.1f | %.1f | %s |\n" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                          float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
print(open(prefix + "_summary.md").read())
