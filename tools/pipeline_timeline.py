#!/usr/bin/env python
"""Timeline of GraphPipeline replays from a rocprofv3 --kernel-trace CSV directory: the kernels of ~2 replay periods near the
end of the run, by start time, with their queue (lane), and how much of the post-sweep chain of one lane runs while a sweep
kernel of ANOTHER lane is resident (the overlap VERDICT r5 #2 asked to see).
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pl -- python tools/bench_graph_pipeline.py 4:2
    python tools/pipeline_timeline.py gpurun_out/pl [window_us]"""
import csv
import glob
import os
import sys


def main(d, window_us=1600.0):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], "q%s" % r.get("Queue_Id", "?")))
    rows.sort()
    if not rows:
        raise SystemExit("no kernel trace under %s" % d)
    # the pipeline's timed runs are the LAST thing the tool does: the window starts at a sweep launch ~6 ms before the end
    t_end = rows[-1][1]
    t_lo = t_end - 6e6
    starts = [r for r in rows if r[0] >= t_lo and "audio_cosine_hl2" in r[2]]
    if not starts:
        raise SystemExit("no sweep kernel in the steady-state part of the trace")
    t0 = starts[0][0]
    win = [r for r in rows if t0 <= r[0] < t0 + window_us * 1e3]
    sweeps = [r for r in win if "audio_cosine_hl2" in r[2]]
    tails = [r for r in win if any(k in r[2] for k in ("mixed_stream", "percode_select", "select_refine", "fuse_best",
                                                        "gate_table", "gate_chase", "sorted_finish"))]
    under = total = 0.0
    for s0, s1, _, q in tails:
        total += s1 - s0
        for a0, a1, _, qa in sweeps:
            if qa != q:
                under += max(0, min(s1, a1) - max(s0, a0))
    print("# GraphPipeline timeline (%s), %.0f us window in steady state\n" % (os.path.basename(d.rstrip("/")), window_us))
    print("post-sweep kernels (selects, fusion, gate, chase) in the window: %.1f us of kernel time, %.1f us (%.0f %%) of it while a "
          "sweep of ANOTHER lane is running\n" % (total / 1e3, under / 1e3, 100.0 * under / max(total, 1)))
    lanes = sorted({r[3] for r in win})
    print("queues (lanes + their side streams): %s\n" % ", ".join(lanes))
    print("| start us | end us | dur us | queue | kernel |")
    print("|---|---|---|---|---|")
    for s0, s1, name, q in win:
        print("| %.1f | %.1f | %.1f | %s | %s |" % ((s0 - t0) / 1e3, (s1 - t0) / 1e3, (s1 - s0) / 1e3, q, name[:60]))


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1600.0)
