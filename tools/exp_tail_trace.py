"""Kernel-by-kernel timeline of one steady-state frame's tail (last forward substep .. first reverse launch) from a rocprofv3
--kernel-trace CSV: start, duration, queue / stream, gap to the previous END on the same queue.   python tools/exp_tail_trace.py DIR"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if "at::native" in n:
        n = "torch:" + n.split("at::native::")[-1].split("<")[0][:36]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n[:44], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
# last complete frame: find the last k_material_bwd<1 (first reverse launch) that has a k_material_fwd<1 (last forward launch) before it
idx_b = [i for i, r in enumerate(rows) if r[2].startswith("k_material_bwd<1")]
idx_f = [i for i, r in enumerate(rows) if r[2].startswith("k_material_fwd<1")]
pick = len(idx_b) // 2
b = idx_b[pick]
a = max(i for i in idx_f if i < b)
t0 = rows[a][0]
last_end = {}
for s, e, n, q, st in rows[a - 2:b + 2]:
    gap = (s - last_end[q]) / 1e3 if q in last_end else float("nan")
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  queue {q:>3} stream {st:>3}  gap on queue {gap:7.1f}  {n}")
    last_end[q] = e
