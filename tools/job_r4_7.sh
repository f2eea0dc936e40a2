mkdir -p gpurun_out/r4
python -m pytest tests/test_gpu_mpm.py tests/test_gpu_pinned.py tests/test_gpu_rollout.py -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r4/t_f64e.txt
python bench.py --steps 40 --warmup 5 > gpurun_out/r4/bench_f64e.json 2> gpurun_out/r4/bench_f64e.err
cat gpurun_out/r4/t_f64e.txt
