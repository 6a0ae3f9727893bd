#!/usr/bin/env python
"""Fold one kernel's row of a rocprofv3 --kernel-trace --stats kernel_stats CSV into profiles/kernel_replay.json, the file
bench.py reads `roofline.kernel_ms_rocprof` from (the dominant kernel's average over ALL launches of a profiled run of the
same command - graph replays included, which the HIP events of the bench line cannot bracket).
    python tools/kernel_replay.py <kernel_stats.csv> "<kernel name substring>" "<key>" "<command>" [out.json]"""
import csv
import json
import os
import sys

src, sub, key, cmd = sys.argv[1:5]
out = sys.argv[5] if len(sys.argv) > 5 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                         "profiles", "kernel_replay.json")
rows = [r for r in csv.DictReader(open(src)) if sub in r["Name"]]
if not rows:
    sys.exit("no kernel matching %r in %s" % (sub, src))
r = max(rows, key=lambda r: float(r["TotalDurationNs"]))
rec = json.load(open(out)) if os.path.exists(out) else {}
rec[key] = {"kernel": r["Name"][:120], "calls": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 2),
            "min_us": round(float(r["MinNs"]) / 1e3, 2), "max_us": round(float(r["MaxNs"]) / 1e3, 2), "command": cmd,
            "source": os.path.basename(src)}
json.dump(rec, open(out, "w"), indent=1, sort_keys=True)
print(key, rec[key])
