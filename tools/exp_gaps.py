"""Idle time between consecutive kernels of the roll-out, from a rocprofv3 --kernel-trace CSV.
    python tools/exp_gaps.py <dir with *_kernel_trace.csv>
Prints, per (previous kernel -> next kernel) pair of the simulation kernels, the median gap and how often it occurs."""
import collections
import csv
import glob
import statistics
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:28], r.get("Stream_Id", "0")))
rows.sort()
sim = ("k_material", "k_p2g", "k_g2p", "k_grid", "k_clear")
gaps = collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    if a[2].startswith(sim) and b[2].startswith(sim):
        gaps[(a[2], b[2])].append((b[0] - a[1]) / 1e3)
tot = 0.0
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1])):
    if len(v) < 20:
        continue
    print(f"{k[0]:30s} -> {k[1]:30s} n={len(v):4d} median {statistics.median(v):6.2f} us  mean {statistics.mean(v):6.2f}")
    tot += statistics.median(v)
print("sum of medians over listed transitions: %.1f us" % tot)
