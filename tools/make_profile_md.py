"""Build profiles/<tag>_kernel_stats.{md,csv} from rocprofv3 outputs: --kernel-trace --stats CSV plus two --pmc passes.
    python tools/make_profile_md.py gpurun_out/p2 r01b "description of the command"
"""
import csv, glob, sys, collections, shutil
root, tag, desc = sys.argv[1], sys.argv[2], sys.argv[3]
stats = glob.glob(root + "/stats/**/*kernel_stats.csv", recursive=True)[0]
shutil.copy(stats, f"profiles/{tag}_bench_metric_kernel_stats.csv")
rows = list(csv.DictReader(open(stats)))
def short(n):
    n = n.split("(")[0]
    n = n.replace("void ", "")
    if n.startswith("rocprim"):
        for key in ("radix_sort_onesweep_iteration", "radix_sort_onesweep_global_offsets", "lookback_scan", "radix_sort_block_sort", "transform", "init_lookback"):
            if key in n: return "rocprim::" + key
        return "rocprim::kernel"
    if n.startswith("at::native"):
        return "torch " + n.split("<")[0].split("::")[-1] + (" " + n.split("at::native::")[2].split("<")[0] if n.count("at::native::") > 1 else "")
    return n[:60]
agg = collections.OrderedDict()
for r in rows:
    k = short(r["Name"])
    a = agg.setdefault(k, [0, 0.0])
    a[0] += int(r["Calls"]); a[1] += float(r["TotalDurationNs"]) / 1e6
tot = sum(v[1] for v in agg.values())
def pmc(sub, counter):
    out = collections.defaultdict(lambda: [0.0, set()])
    for f in glob.glob(root + f"/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter: continue
            k = short(r["Kernel_Name"])
            out[k][0] += float(r["Counter_Value"]); out[k][1].add(r["Dispatch_Id"])
    return {k: v[0] / max(len(v[1]), 1) for k, v in out.items()}
fetch, write = pmc("fetch", "FETCH_SIZE"), pmc("write", "WRITE_SIZE")
with open(f"profiles/{tag}_bench_metric_kernel_stats.md", "w") as f:
    f.write(f"# {tag} - rocprofv3 summaries of `bench.py` (metric workload: 100k particles / 128^3 / 200k Gaussians / 1080p, S=20, V=3), 1x MI355X\n\n")
    f.write(f"## `{desc}`\n\n(kernel time over the whole run incl. warm-up, kernel-selection pass and ground-truth renders; full CSV: {tag}_bench_metric_kernel_stats.csv)\n\n")
    f.write("| kernel | calls | total ms | avg us | % |\n|---|---:|---:|---:|---:|\n")
    for k, (c, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        f.write(f"| {k} | {c} | {ms:.3f} | {1e3 * ms / c:.1f} | {100 * ms / tot:.2f} |\n")
    f.write(f"\ntotal kernel time {tot:.1f} ms\n\n")
    f.write("## HBM traffic per launch from PMC (separate passes: `--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`), KiB per dispatch\n\n")
    f.write("FETCH_SIZE is the raw counter; per MI355X_MICROARCH.md (HBM / rocprofv3 section) it under-reports wide coalesced streaming reads on gfx950 by 2x, so `2 x FETCH` is the upper estimate of read traffic.  WRITE_SIZE matches known byte counts.\n\n")
    f.write("| kernel | FETCH_SIZE KiB | WRITE_SIZE KiB |\n|---|---:|---:|\n")
    for k in sorted(set(fetch) | set(write)):
        if k.startswith(("torch", "Cijk", "rocprim", "__amd")): continue
        f.write(f"| {k} | {fetch.get(k, 0):.0f} | {write.get(k, 0):.0f} |\n")
# ---- MFMA / issue counters (one more pass: --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU
#      SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU)
mf = {}
for f in glob.glob(root + "/mfma/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        d = mf.setdefault(k, collections.defaultdict(float))
        d[r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_BUSY_CYCLES":
            d["_n"] += 1
            d["_ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
mfma_json = {}
if mf:
    with open(f"profiles/{tag}_bench_metric_kernel_stats.md", "a") as f:
        f.write("\n## Matrix-pipe and issue counters (separate pass: `--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 "
                "SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU`), per dispatch\n\n")
        f.write("MFMA busy % = SQ_VALU_MFMA_BUSY_CYCLES / (dispatch duration x ~2.4 GHz x 1024 SIMDs... the counter sums the busy cycles of "
                "all SIMDs): reported here against the kernel's own duration x 1024 SIMDs at the clock implied by SQ_BUSY_CYCLES.  "
                "MOPS_F32 x 512 = f32 MFMA flops (one MOPS unit = 512 flop).  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* count quad-cycles "
                "summed over waves.\n\n")
        f.write("| kernel | dispatches | avg us | MFMA busy cycles | MFMA flops (MOPS x 512) | achieved f32 MFMA TFLOP/s | MFMA busy % of SIMD-cycles | "
                "VALU insts | wait-LDS / wave-cycles | wait-inst-any / wave-cycles | active-VALU / wave-cycles |\n|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|\n")
        for k, d in sorted(mf.items(), key=lambda kv: -kv[1]["_ns"]):
            if k.startswith(("torch", "Cijk", "rocprim", "__amd")) or d["_n"] == 0:
                continue
            n = d["_n"]; us = d["_ns"] / n / 1e3
            busy = d["SQ_VALU_MFMA_BUSY_CYCLES"] / n
            flops = d["SQ_INSTS_VALU_MFMA_MOPS_F32"] / n * 512.0
            simd_cycles = us * 1e-6 * 2.4e9 * 1024.0
            wc = max(d["SQ_WAVE_CYCLES"], 1.0)
            f.write(f"| {k} | {int(n)} | {us:.1f} | {busy:.3g} | {flops:.3g} | {flops / (us * 1e-6) / 1e12:.1f} | {100 * busy / simd_cycles:.1f} | "
                    f"{d['SQ_INSTS_VALU'] / n:.3g} | {d['SQ_WAIT_INST_LDS'] / wc:.3f} | {d['SQ_WAIT_INST_ANY'] / wc:.3f} | {d['SQ_ACTIVE_INST_VALU'] / wc:.3f} |\n")
            mfma_json[k.split("<")[0]] = {"avg_us": round(us, 1), "mfma_busy_cycles": busy, "mfma_flops": flops,
                                          "mfma_busy_pct_of_simd_cycles": round(100 * busy / simd_cycles, 2)}
import json
import sys as _sys
from pathlib import Path as _Path
_sys.path.insert(0, str(_Path(__file__).resolve().parent.parent))
from neuma_amd._lib import csrc_digest as _csrc_digest      # noqa: E402
traffic = {}
for k in sorted(set(fetch) | set(write)):
    if k.startswith(("torch", "Cijk", "rocprim", "__amd")): continue
    base = k.split("<")[0]
    t = traffic.setdefault(base, {"fetch_kib": 0.0, "write_kib": 0.0, "n": 0})
    t["fetch_kib"] += fetch.get(k, 0.0); t["write_kib"] += write.get(k, 0.0); t["n"] += 1
for base, t in traffic.items():
    n = max(t.pop("n"), 1)
    t["fetch_kib"] = round(t["fetch_kib"] / n, 1); t["write_kib"] = round(t["write_kib"] / n, 1)
    # MI355X_MICROARCH.md (HBM): FETCH_SIZE reports half of the bytes of wide coalesced reads on gfx950 -> doubled
    t["hbm_bytes_per_launch"] = int((2.0 * t["fetch_kib"] + t["write_kib"]) * 1024)
json.dump({"source": f"profiles/{tag}_bench_metric_kernel_stats.md (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes)",
           "correction": "bytes = 2 x FETCH_SIZE + WRITE_SIZE (KiB counters; FETCH under-reports wide coalesced reads by 2x on gfx950)",
           "csrc_digest": _csrc_digest(),
           "csrc_digest_note": "neuma_amd._lib.csrc_digest() of the sources the profiled library was built from; bench.py drops the "
                               "copied counters and prints pmc_stale when the checkout's digest differs",
           "kernels": traffic, "mfma": mfma_json}, open("profiles/pmc_traffic.json", "w"), indent=1)
print(open(f"profiles/{tag}_bench_metric_kernel_stats.md").read())
