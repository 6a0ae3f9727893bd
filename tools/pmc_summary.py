"""Summarise a rocprofv3 --pmc counter_collection.csv: mean per-dispatch counter value per (kernel, grid size),
with the mean dispatch duration.  Usage: python tools/pmc_summary.py <dir> [name-filter]"""
import collections
import csv
import glob
import sys

path = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float)
    t = {}
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"][:48], int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1))
        per[(r["Dispatch_Id"], k, r["Counter_Name"])] += float(r["Counter_Value"])
        t[(r["Dispatch_Id"], k)] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    for (d, k, c), v in per.items():
        acc[k][c].append(v)
    for (d, k), v in t.items():
        dur[k].append(v)
for f in glob.glob(path + "/**/*_results.db", recursive=True):      # rocpd SQLite output (no --output-format csv)
    import sqlite3
    per = collections.defaultdict(float)
    t = {}
    for did, name, grid, wg, cname, val, a, b in sqlite3.connect(f).execute(
            "select dispatch_id, kernel_name, grid_size, workgroup_size, counter_name, value, start, end "
            "from counters_collection"):
        k = (name[:48], int(grid) // max(int(wg), 1))
        per[(did, k, cname)] += float(val)
        t[(did, k)] = (b - a) / 1e3
    for (d, k, c), v in per.items():
        acc[k][c].append(v)
    for (d, k), v in t.items():
        dur[k].append(v)
for k, cs in sorted(acc.items()):
    if flt and flt not in k[0]:
        continue
    n = max(len(v) for v in cs.values())
    print("%-50s blocks=%-6d n=%-3d us=%.1f " % (k[0], k[1], n, sum(dur[k]) / len(dur[k])),
          " ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(cs.items())))
