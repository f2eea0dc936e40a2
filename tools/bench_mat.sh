#!/bin/bash
# short metric bench; prints frames/s and the constitutive kernels' average durations
python bench.py --steps ${1:-60} --warmup 10 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/b_mat.json
python - <<'PY'
import json
d = json.loads(open("gpurun_out/b_mat.json").read())
print("fps", d["value"], "ms", d["ms_per_step"], "pair_us", d["roofline"]["avg_us"], "frac", d["roofline"]["frac"])
for v in d.get("kernel_rooflines", []):
    if "material" in v["kernel"]: print(v["kernel"], v["avg_us"], v["frac"])
PY
