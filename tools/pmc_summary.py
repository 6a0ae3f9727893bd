"""Summarise a rocprofv3 --pmc counter_collection.csv: mean per-dispatch counter value per kernel."""
import csv, sys, collections, glob
path = sys.argv[1]
files = glob.glob(path + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        per[(r["Dispatch_Id"], r["Kernel_Name"][:60], r["Counter_Name"])] += float(r["Counter_Value"])
    for (d, k, c), v in per.items():
        acc[k][c].append(v)
for k, cs in acc.items():
    n = max(len(v) for v in cs.values())
    print("%-62s n=%d" % (k, n), " ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(cs.items())))
