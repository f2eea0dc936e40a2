"""Per-substep timeline of the fused roll-out from a rocprofv3 kernel trace: every microsecond between the first and the last
kernel of one forward + backward roll-out attributed to a kernel or to the gap in front of it.
    rocprofv3 --kernel-trace -d DIR -- python tools/run_rollout.py metric 6
    python tools/timeline.py DIR [S]      -> markdown on stdout"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
S = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()


def short(n):
    n = n.split("(")[0]
    for p in ("void ", "at::native::", "(anonymous namespace)::"):
        n = n.replace(p, "")
    return n[:48]


# a roll-out starts with the LoRA merges of its two nets (the host synchronises between roll-outs, but a fast host leaves
# gaps too short to tell apart from the ones inside); take the last complete one
groups, cur = [], [rows[0]]
for a, b in zip(rows, rows[1:]):
    if "k_lora_merge_layers" in b[2] and "bwd" not in b[2] and any("k_material" in r[2] for r in cur):
        groups.append(cur); cur = []
    cur.append(b)
groups.append(cur)
groups = [g for g in groups if sum(1 for r in g if "k_material_bwd_pair" in r[2]) >= S - 1]
g = groups[-1]
t0, t1 = g[0][0], g[-1][1]
busy = sum(e - s for s, e, _ in g)
print(f"# roll-out timeline ({len(g)} kernels, S = {S}): first kernel start -> last kernel end {1e-3 * (t1 - t0):.1f} us, "
      f"inside kernels {1e-3 * busy:.1f} us ({100.0 * busy / (t1 - t0):.1f} %), between kernels {1e-3 * (t1 - t0 - busy):.1f} us\n")
agg = defaultdict(lambda: [0, 0.0, 0.0])
prev_end = None
for s, e, n in g:
    k = short(n)
    agg[k][0] += 1
    agg[k][1] += 1e-3 * (e - s)
    if prev_end is not None:
        agg[k][2] += 1e-3 * max(0, s - prev_end)
    prev_end = max(prev_end or e, e)
print("| kernel | launches | us inside (total) | avg us | gap in front (total us) | avg gap us |\n|---|---:|---:|---:|---:|---:|")
for k, (c, t, gp) in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
    print(f"| {k} | {c} | {t:.1f} | {t / c:.2f} | {gp:.1f} | {gp / c:.2f} |")
# one forward and one backward substep from the middle of the sweeps, launch by launch
names = [short(n) for _, _, n in g]


def show(title, first_pred, count_pred):
    idx = [i for i, n in enumerate(names) if first_pred(n)]
    if len(idx) < 3:
        return
    a, b = idx[len(idx) // 2], idx[len(idx) // 2 + 1]
    print(f"\n## {title}: launches {a}..{b - 1}, {1e-3 * (g[b][0] - g[a][0]):.1f} us start to start\n")
    print("| launch | start (us) | inside (us) | gap to next (us) |\n|---|---:|---:|---:|")
    for i in range(a, b):
        s, e, n = g[i]
        print(f"| {short(n)} | {1e-3 * (s - g[a][0]):.1f} | {1e-3 * (e - s):.1f} | {1e-3 * (g[i + 1][0] - e):.2f} |")


show("one forward substep", lambda n: n.startswith("k_material_fwd_pair"), None)
show("one reverse substep", lambda n: n.startswith("k_material_bwd_pair"), None)
