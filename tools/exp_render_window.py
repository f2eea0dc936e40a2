"""The render window of one metric frame (between the forward sweep's last launch and the reverse sweep's first) from a rocprofv3
kernel trace: every launch with start / duration / stream, and how busy the window is.
    python tools/exp_render_window.py DIR"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:34], r.get("Stream_Id", "?")))
rows.sort()
# frames: a k_material_fwd<1 (last plasticity of the forward sweep) followed later by k_material_bwd<1 (first launch of the reverse sweep)
idx = [i for i, r in enumerate(rows) if r[2].startswith("k_material_fwd<1")]
pick = idx[len(idx) * 3 // 4]
j = next(i for i in range(pick, len(rows)) if rows[i][2].startswith("k_material_bwd<1"))
win = rows[pick:j + 1]
t0 = win[0][1]
busy, cs, ce = 0, None, None
for s, e, n, st in win[1:-1]:
    if ce is None or s > ce:
        if ce is not None:
            busy += ce - cs
        cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
span = win[-1][0] - t0
print(f"render window {1e-3 * span:.1f} us, some kernel running {1e-3 * busy:.1f} us, {len(win) - 2} launches, sum of durations {1e-3 * sum(e - s for s, e, _, _ in win[1:-1]):.1f} us")
for s, e, n, st in win:
    print(f"  {1e-3 * (s - t0):8.1f} +{1e-3 * (e - s):7.1f}  s{st}  {n}")
