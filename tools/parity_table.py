"""DESIGN.md section 2's parity table from the errors the GPU tests measured (tests/gpu_util.parity appends one JSON line per
comparison to gpurun_out/parity_measured.jsonl):  python tools/parity_table.py [file] > table.md
One row per (case group, tensor): the largest measured error over the group's cases and runs, the bound the test holds it to,
and - where the fixture carries the reference's own fp32 run - the distance of that run from the reference's fp64 run."""
import collections
import json
import re
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/parity_measured.jsonl"
rows = [json.loads(l) for l in open(path) if l.strip()]


def group(case: str) -> str:
    case = re.sub(r"rank \d+, ", "", case)
    case = re.sub(r"\(reorder=[A-Za-z]+\)", "", case)
    case = re.sub(r"n\d+_g\d+_(noslip|freeslip)", "<4 fixtures>", case)
    case = re.sub(r"N=\d+ G=\d+ (noslip|freeslip)", "<4 cases>", case)
    case = re.sub(r"(jelly|plasticine|sand) checkpoint", "<3 checkpoints>", case)
    case = re.sub(r", (elasticity|plasticity),", ",", case)
    case = re.sub(r"world=\d, cap=\w+, ", "<world 2, 3>, ", case)
    return re.sub(r"\s+", " ", case).strip()


auto = [r for r in rows if re.match(r".+? @ test_\w+\.py:\d+: ", r["tensor"])]
rows = [r for r in rows if r not in auto]
agg = collections.OrderedDict()
for r in rows:
    k = (group(r["case"]), r["tensor"])
    a = agg.setdefault(k, {"m": 0.0, "b": 0.0, "n": None})
    a["m"] = max(a["m"], r["measured"])
    a["b"] = max(a["b"], r["bound"])
    if r.get("reference_fp32_vs_fp64") is not None:
        a["n"] = max(a["n"] or 0.0, r["reference_fp32_vs_fp64"])
print("| comparison | tensor | measured (max) | bound in the test | reference's own fp32 vs fp64 |\n|---|---|---:|---:|---:|")
for (c, t), a in agg.items():
    noise = "-" if a["n"] is None else f"{a['n']:.1e}"
    print(f"| {c} | {t} | {a['m']:.1e} | {a['b']:.1e} | {noise} |")


# ---- comparisons that publish themselves (tests/gpu_util.Measured): one row per source line of a test
if auto:
    sites = collections.OrderedDict()
    for r in auto:
        m = re.match(r"(.+?) @ (test_\w+\.py):(\d+): (.*)", r["tensor"])
        test = r["case"].split("::")[-1].split("[")[0]
        k = (m.group(2), int(m.group(3)), r["bound"])
        a = sites.setdefault(k, {"m": 0.0, "n": 0, "kind": m.group(1), "test": test, "src": m.group(4)})
        a["m"] = max(a["m"], r["measured"]); a["n"] += 1
    print("\n| test (file:line) | quantity | comparisons | measured (max) | bound in the test |\n|---|---|---:|---:|---:|")
    for (f, ln, b), a in sorted(sites.items()):
        src = re.sub(r"\s*#.*$", "", a["src"]).replace("|", "/").replace("assert ", "")
        src = re.sub(r"\s+", " ", src)[:70]
        print(f"| `{a['test']}` ({f.replace('test_gpu_', '').replace('.py', '')}:{ln}) | {a['kind']}: `{src}` | {a['n']} | {a['m']:.1e} | {b:.1e} |")
