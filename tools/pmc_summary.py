"""Summarise rocprofv3 counter_collection CSVs per kernel (mean per dispatch)."""
import csv, glob, collections, sys
root = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
fs = glob.glob(root + "/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for f in fs:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:48]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k].add(r["Dispatch_Id"])
for k, d in sorted(agg.items()):
    if pat and pat not in k:
        continue
    n = max(len(cnt[k]), 1)
    print(f"{k:48s} disp {n:4d} " + " ".join(f"{c}={v / n:.0f}" for c, v in sorted(d.items())))
