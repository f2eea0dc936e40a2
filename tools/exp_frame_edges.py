"""First and last launches of one steady-state frame on the frame's own queue (rocprofv3 --kernel-trace CSV): what sits between the
reverse sweep's last launch and the next frame's first substep.   python tools/exp_frame_edges.py DIR"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if "at::native" in n:
        n = "torch:" + n.split("at::native::")[-1].split("<")[0][:36]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n[:44], r.get("Queue_Id", "?")))
rows.sort()
idx = [i for i, r in enumerate(rows) if r[2].startswith("k_material_bwd<0")]      # last constitutive launch of a reverse sweep
a = idx[len(idx) // 2]
b = next(i for i in range(a + 1, len(rows)) if rows[i][2].startswith("k_material_fwd_pair"))
t0 = rows[a][0]
prev_end = None
for s, e, n, q in rows[a - 4:b + 1]:
    gap = (s - prev_end) / 1e3 if prev_end is not None else float("nan")
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  queue {q:>3}  gap {gap:7.1f}  {n}")
    prev_end = e
