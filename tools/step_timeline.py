#!/usr/bin/env python
"""Timeline of one matching step from a rocprofv3 --kernel-trace (+ --memory-copy-trace) CSV directory: every kernel
/ copy of the LAST `steps` iterations folded by position, with start offsets relative to the step's first launch.

    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/tl -- python tools/step_loop.py 30
    python tools/step_timeline.py gpurun_out/tl 30
"""
import csv
import glob
import os
import sys


def main(d, steps):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:52], "q%s" % r.get("Queue_Id", "?")))
    for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", ""), "copy"))
    rows.sort()
    # a step ends with its device-to-host copy of the codes (the only D2H copy of the loop)
    cuts, armed = [], False
    for i, r in enumerate(rows):       # the codes' D2H copy = the first copyBuffer after the step's gate_chase_kernel
        if "gate_chase" in r[2]:
            armed = True
        elif armed and "copyBuffer" in r[2]:
            cuts.append(i)
            armed = False
    if len(cuts) < steps + 1:
        # no copy behind the walk: the results went straight to pinned host memory (walk(sync="ints")) - a step ends
        # with its gate_chase_kernel
        cuts = [i for i, r in enumerate(rows) if "gate_chase" in r[2]]
    if len(cuts) < steps + 1:
        print("no step markers found (%d); events:" % len(cuts), sorted(set(r[2] for r in rows))[:40])
        return
    cuts = cuts[-(steps + 1):]
    segs = [rows[cuts[k] + 1:cuts[k + 1] + 1] for k in range(steps)]
    per = min(len(x) for x in segs)
    if any(len(x) != per for x in segs):
        print("uneven steps:", sorted(set(len(x) for x in segs)))
    print("events per step: %d" % per)
    print("| # | event | queue | start us | end us | dur us |")
    print("|---|---|---|---|---|---|")
    # events are matched across steps by (kernel name, occurrence within the step), not by position: under a graph replay
    # the two branches' launch order is not the same in every step
    acc, order = {}, []
    for x in segs:
        seen = {}
        for r in x:
            k = (r[2], seen.get(r[2], 0))
            seen[r[2]] = k[1] + 1
            if k not in acc:
                acc[k] = [[], [], r[3]]
                order.append(k)
            acc[k][0].append((r[0] - x[0][0]) / 1e3)
            acc[k][1].append((r[1] - r[0]) / 1e3)
    order.sort(key=lambda k: sum(acc[k][0]) / len(acc[k][0]))
    for i, k in enumerate(order):
        st, du, qn = acc[k]
        a, d = sum(st) / len(st), sum(du) / len(du)
        print("| %d | %s | %s | %.1f | %.1f | %.1f |" % (i, k[0], qn, a, a + d, d))
    ends = [(max(r[1] for r in x) - x[0][0]) / 1e3 for x in segs]
    nxt = [(segs[k + 1][0][0] - segs[k][0][0]) / 1e3 for k in range(steps - 1)]
    print("GPU-side span of a step (first start -> last end): %.1f us;  step period: %.1f us" % (sum(ends) / steps, sum(nxt) / len(nxt)))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]))
