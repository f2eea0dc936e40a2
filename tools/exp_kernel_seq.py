"""Durations of the launches of one kernel, in launch order, from a rocprofv3 kernel trace (who is long, who is empty).
    python tools/exp_kernel_seq.py DIR kernel_name_prefix [count]"""
import csv, glob, sys
d, name = sys.argv[1], sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 36
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if k == name or k.startswith(name + "<"):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Grid_Size", r.get("Grid_Size_X", "?")), r.get("Stream_Id", "?")))
rows.sort()
t0 = rows[-n][0]
for s, e, g, st in rows[-n:]:
    print(f"  start {1e-3 * (s - t0):9.1f} us  duration {1e-3 * (e - s):7.1f} us  grid {g} stream {st}")
