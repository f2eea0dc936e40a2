mkdir -p /root/repo/gpurun_out/r4
python -m pytest tests/test_gpu_bind_raster.py tests/test_gpu_entrypoints.py tests/test_gpu_train.py tests/test_gpu_infer.py tests/test_gpu_configs.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r4/t_gorder.txt; cat gpurun_out/r4/t_gorder.txt
for v in 1 0 1 0; do NEUMA_GAUSSIAN_ORDER=$v python bench.py --steps 60 --warmup 10 --no-cpu-baseline --epoch-frames 0 2>/dev/null | tail -1 > gpurun_out/r4/bench_gorder_$v.json; python - <<P
import json
d=json.load(open("gpurun_out/r4/bench_gorder_$v.json")); print("order=$v", d["value"], d["ms_per_step"], {k:v for k,v in d["kernel_breakdown_ms_per_frame"].items() if "bin" in k or "sort" in k or "render" in k})
P
done > gpurun_out/r4/gorder.txt 2>&1
