"""experiments: the launches of ONE clip decode from a rocprofv3 --kernel-trace CSV directory (tools/bench_decode.py)."""
import csv, glob, sys, os
rows=[]
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        gx = int(r.get("Grid_Size", r.get("Grid_Size_X", 0))) * int(r.get("Grid_Size_Y", 1) or 1) if "Grid_Size" not in r else int(r["Grid_Size"])
        wx = int(r.get("Workgroup_Size", r.get("Workgroup_Size_X", 1)))
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], gx // max(wx, 1)))
rows.sort()
# find the last decode: sequence beginning with vq_gather
idx=[i for i,r in enumerate(rows) if "gather" in r[2]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else -4
i0=idx[k]; i1=idx[k+1]
t0=rows[i0][0]
for r in rows[i0:i1]:
    print("%7.1f %6.1f  blocks=%-5d %s" % ((r[0]-t0)/1e3, (r[1]-r[0])/1e3, r[3], r[2]))
print("span %.1f us, sum of kernels %.1f us" % ((rows[i1-1][1]-t0)/1e3, sum(r[1]-r[0] for r in rows[i0:i1])/1e3))
