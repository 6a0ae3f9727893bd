#!/usr/bin/env python
"""Sum FETCH_SIZE x 2 + WRITE_SIZE over the kernels of one cfg-3 step (query normalise, column pack, prefilter GEMM,
sorted select, finish) from the two PMC passes of tools/pmc_cfg3.sh."""
import collections
import csv
import glob
import json
import os
import sys

STEP = ("hl_gemm32_kernel", "percode_select_sorted_kernel", "hl_gemm64h_kernel", "percode_select_bycode_kernel", "perm32_kernel",
        "sorted_finish_kernel", "hl_pack_cols_kernel", "l2_normalize_rows_kernel<false>")


def per_kernel(d, name):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                for k in STEP:
                    if k in r["Kernel_Name"]:
                        acc[k][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {k: sum(v.values()) / len(v) for k, v in acc.items() if v}


root, out = sys.argv[1], sys.argv[2]
fe = per_kernel(os.path.join(root, "pmc_FETCH_SIZE"), "FETCH_SIZE")
wr = per_kernel(os.path.join(root, "pmc_WRITE_SIZE"), "WRITE_SIZE")
rec = {"per_kernel_kb": {k: {"fetch": round(fe.get(k, 0.0), 1), "write": round(wr.get(k, 0.0), 1)} for k in STEP},
       "hbm_bytes_per_step": int(sum(fe.values()) * 1024 * 2 + sum(wr.values()) * 1024),
       "recipe": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python bench.py --workload cfg3; "
                 "per-dispatch means per kernel; bytes = FETCH_SIZE x 1024 x 2 (gfx950 correction) + WRITE_SIZE x 1024"}
key = "cfg3_step_bycode|100000x512 Q=1000" if fe.get("hl_gemm64h_kernel") else "cfg3_step|100000x512 Q=1000"
json.dump({key: rec}, open(out, "w"), indent=1, sort_keys=True)
print(json.dumps(rec))
