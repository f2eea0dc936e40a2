"""Tighten the bounds of the GPU tests to <= 3x their measured errors.
    python tools/parity_tighten.py gpurun_out/parity_measured.jsonl [--apply]
Reads the log that tests/gpu_util.py writes (one record per comparison: test item, source line, measured, bound), takes the
largest measured value per (source line, bound literal) over all cases and runs in the log, and proposes
bound = 3 x that, rounded up to {1, 1.5, 2, 3, 5, 7} x 10^k, never below FLOOR (a few fp32 ulps).  --apply rewrites the
literal in the test source when the line holds that literal exactly once; everything else is printed for a human."""
import collections
import json
import math
import re
import sys
from pathlib import Path

FLOOR = 2e-7
ROOT = Path(__file__).resolve().parent.parent / "tests"


def nice_ceil(x):
    k = math.floor(math.log10(x))
    for m in (1, 1.5, 2, 3, 5, 7, 10):
        if m * 10 ** k >= x * (1 - 1e-12):
            return m * 10 ** k
    return 10 ** (k + 1)


def lit(v):
    s = f"{v:.1e}"
    m, e = s.split("e")
    m = m.rstrip("0").rstrip(".")
    return f"{m}e{int(e)}"


rows = []
for path in sys.argv[1:]:
    if path.startswith("--"):
        continue
    rows += [json.loads(l) for l in open(path)]
apply = "--apply" in sys.argv
sites = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    m = re.match(r"(\w[\w ,'/-]*) @ (test_[\w]+\.py):(\d+): ", r["tensor"])
    if not m:
        continue
    sites[(m.group(2), int(m.group(3)))][r["bound"]] = max(sites[(m.group(2), int(m.group(3)))][r["bound"]], r["measured"])
edits, manual = collections.defaultdict(list), []
for (f, ln), bounds in sorted(sites.items()):
    src = (ROOT / f).read_text().split("\n")[ln - 1]
    lits = re.findall(r"(?<![\w.])(\d+(?:\.\d+)?e-\d+)(?![\w.])", src)
    for b, mx in bounds.items():
        want = max(FLOOR, nice_ceil(3 * mx)) if mx > 0 else FLOOR
        if want >= b * 0.7:
            continue
        cands = [t for t in lits if abs(float(t) - b) <= 1e-9 * b]
        if len(cands) == 1 and src.count(cands[0]) == 1:
            edits[f].append((ln, cands[0], lit(want), mx, b))
        else:
            manual.append((f, ln, mx, b, want, src.strip()[:120]))
for f, es in edits.items():
    lines = (ROOT / f).read_text().split("\n")
    for ln, old, new, mx, b in es:
        print(f"{f}:{ln}: {old} -> {new}   (measured {mx:.2e})")
        if apply:
            lines[ln - 1] = re.sub(r"(?<![\w.])" + re.escape(old) + r"(?![\w.])", new, lines[ln - 1]) + (f"      # measured {mx:.1e}" if "# measured" not in lines[ln - 1] and len(lines[ln - 1]) < 120 else "")
    if apply:
        (ROOT / f).write_text("\n".join(lines))
print("\n-- by hand (bound is computed, or the literal occurs more than once on the line):")
for f, ln, mx, b, want, src in manual:
    print(f"{f}:{ln}: measured {mx:.2e} bound {b:.2e} -> {want:.1e} | {src}")
