"""GPU busy / idle of a traced run: union of the kernels' [start, end) intervals over the steady part of the run.
    rocprofv3 --kernel-trace --output-format csv -d DIR -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline
    python tools/busy.py DIR [skip_fraction]"""
import csv, glob, sys
d = sys.argv[1]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40]))
rows.sort()
# steady part: between the first and the last k_material_bwd_pair of the second half of the run
pair = [r for r in rows if "k_material_bwd_pair" in r[2]]
t_lo = pair[int(len(pair) * skip)][0]
t_hi = pair[-1][1]
sel = [r for r in rows if r[0] >= t_lo and r[1] <= t_hi]
busy, cur_s, cur_e = 0, None, None
gaps = []
for s, e, n in sel:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = t_hi - t_lo
print(f"span {span / 1e6:.2f} ms, GPU busy (union of kernels) {busy / 1e6:.2f} ms = {100.0 * busy / span:.1f} %, idle {1e-6 * (span - busy):.2f} ms in {len(gaps)} gaps")
from collections import defaultdict
by = defaultdict(lambda: [0, 0])
for g, n in gaps:
    by[n][0] += 1; by[n][1] += g
print("idle time by the kernel that follows the gap (top 12):")
for n, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"  {n:42s} {c:5d} gaps {t / 1e3:9.1f} us total {t / c / 1e3:7.2f} us each")
