mkdir -p /root/repo/gpurun_out/r4
python -m pytest tests/test_gpu_bind_raster.py -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r4/t_bc.txt; cat gpurun_out/r4/t_bc.txt
python tools/exp_bincount_phases.py 2>&1 | tail -10 > gpurun_out/r4/bincount_phases2.txt
for i in 1 2; do python bench.py --steps 60 --warmup 10 --no-cpu-baseline --epoch-frames 0 2>/dev/null | tail -1 > gpurun_out/r4/bench_bc.json; python - <<P
import json
d=json.load(open("gpurun_out/r4/bench_bc.json")); print(d["value"], d["ms_per_step"], {k:v for k,v in d["kernel_breakdown_ms_per_frame"].items() if "bin" in k or "sort" in k})
P
done > gpurun_out/r4/bc.txt 2>&1
