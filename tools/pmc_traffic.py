#!/usr/bin/env python
"""FETCH_SIZE / WRITE_SIZE passes (rocprofv3 --pmc, one counter per pass) -> profiles/pmc_traffic.json, the file bench.py
reads `roofline.traffic` from at run time.  HBM bytes per launch = FETCH_SIZE (KB) x 1024 x 2 (the gfx950 correction of
MI355X_MICROARCH.md: the counter ticks per 64-byte request on this part, documented as 32) + WRITE_SIZE (KB) x 1024.
    python tools/pmc_traffic.py <dir with pmc_FETCH_SIZE/ pmc_WRITE_SIZE/> <kernel-name substring> <shape tag> <out json> [key name]
(key name: what the record is filed under instead of the substring - "audio_cosine_hl2_kernel<1" -> "audio_cosine_hl1")"""
import collections
import csv
import glob
import json
import os
import sys


def mean_counter(d, name, kernel):
    per = collections.defaultdict(float)
    dur = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"] and r["Counter_Name"] == name:
                per[r["Dispatch_Id"]] += float(r["Counter_Value"])
                dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if not per:
        raise SystemExit("no %s rows for %s under %s" % (name, kernel, d))
    v = sorted(per.values())
    return sum(v) / len(v), len(v), sum(dur.values()) / len(dur)


def main(root, kernel, tag, out, key=None):
    fetch, n1, us1 = mean_counter(os.path.join(root, "pmc_FETCH_SIZE"), "FETCH_SIZE", kernel)
    write, n2, us2 = mean_counter(os.path.join(root, "pmc_WRITE_SIZE"), "WRITE_SIZE", kernel)
    rec = {"kernel": kernel, "shape": tag, "fetch_size_kb": round(fetch, 1), "write_size_kb": round(write, 1),
           "dispatches": [n1, n2], "us_under_the_passes": [round(us1, 1), round(us2, 1)],
           "hbm_bytes_per_launch": int(fetch * 1024 * 2 + write * 1024),
           "recipe": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python tools/bench_audio_hl.py; "
                     "bytes = FETCH_SIZE x 1024 x 2 (gfx950 correction) + WRITE_SIZE x 1024"}
    data = json.load(open(out)) if os.path.exists(out) else {}
    data[(key or kernel) + "|" + tag] = rec
    json.dump(data, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps(rec))


if __name__ == "__main__":
    main(*sys.argv[1:6])
