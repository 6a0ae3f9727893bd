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
fs = sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True))
if fs:
    shutil.copy(fs[-1], prefix + "_kernel_stats.csv")
else:
    # rocprofv3 of ROCm 7.2 writes a rocpd SQLite database unless --output-format csv is given: fold its kernel
    # dispatch records into the same columns the CSV has
    import sqlite3
    import statistics
    db = sqlite3.connect(sorted(glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True))[-1])
    per = {}
    for name, a, b in db.execute("select name, start, end from kernels"):
        per.setdefault(name, []).append(b - a)
    tot = sum(sum(v) for v in per.values())
    with open(prefix + "_kernel_stats.csv", "w", newline="") as o:
        w = csv.writer(o, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for name, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([name, len(v), sum(v), sum(v) / len(v), round(100.0 * sum(v) / tot, 4), min(v), max(v),
                        statistics.pstdev(v)])
rows = list(csv.DictReader(open(prefix + "_kernel_stats.csv")))
with open(prefix + "_summary.md", "w") as o:
    o.write("# %s\n\n| kernel | calls | avg us | min us | max us | %% |\n|---|---|---|---|---|---|\n" % title)
    for r in rows:
        if float(r["Percentage"]) < 0.02:
            continue
        o.write("| %s | %s | %.1f | %.1f | %.1f | %s |\n" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                          float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
print(open(prefix + "_summary.md").read())
