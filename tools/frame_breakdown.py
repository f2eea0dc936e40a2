"""Per-frame kernel breakdown and GPU busy fraction of a traced bench run (any workload; frames are counted by their
k_pixel_loss launches, V per frame).
    rocprofv3 --kernel-trace --output-format csv -d DIR -- python bench.py --workload bb --steps 60 --warmup 10 --no-cpu-baseline
    python tools/frame_breakdown.py DIR V"""
import csv, glob, sys
from collections import defaultdict
d, V = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:46]))
rows.sort()
loss = [r for r in rows if "k_pixel_loss" in r[2]]
lo, hi = loss[len(loss) // 2][0], loss[-1][1]
sel = [r for r in rows if r[0] >= lo and r[1] <= hi]
nfr = sum(1 for r in sel if "k_pixel_loss" in r[2]) / V
busy, cs, ce = 0, None, None
for s, e, n in sel:
    if ce is None or s > ce:
        if ce is not None:
            busy += ce - cs
        cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
print(f"frames {nfr:.1f}, {1e-3 * (hi - lo) / nfr:.1f} us per frame, GPU busy (union of kernels) {100.0 * busy / (hi - lo):.1f} % = "
      f"{1e-3 * busy / nfr:.1f} us per frame, {len(sel) / nfr:.1f} launches per frame")
agg = defaultdict(lambda: [0, 0])
for s, e, n in sel:
    agg[n][0] += 1; agg[n][1] += e - s
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:36]:
    print(f"  {n:46s} {c / nfr:6.1f} per frame {t / 1e3 / nfr:8.1f} us per frame {t / c / 1e3:7.1f} us each")
print(f"sum of kernel durations {sum(t for c, t in agg.values()) / 1e3 / nfr:.1f} us per frame")
