mkdir -p /root/repo/gpurun_out/r4
for v in 0 1 0 1; do NEUMA_SORT_GAUSSIANS=$v python bench.py --steps 60 --warmup 10 --no-cpu-baseline --epoch-frames 0 2>/dev/null | tail -1 > gpurun_out/r4/bench_gsort_$v.json; python - <<P
import json
d=json.load(open("gpurun_out/r4/bench_gsort_$v.json")); print("sorted=$v", d["value"], d["ms_per_step"], [(k["kernel"], k["avg_us"]) for k in d["kernel_rooflines"] if "render" in k["kernel"] or "bin" in k["kernel"]], {k:v for k,v in d["kernel_breakdown_ms_per_frame"].items() if "bin" in k or "sort" in k or "preproc" in k})
P
done > gpurun_out/r4/gsort.txt 2>&1
