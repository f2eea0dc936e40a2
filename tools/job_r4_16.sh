mkdir -p /root/repo/gpurun_out/r4
python -m pytest tests/test_gpu_mpm.py tests/test_gpu_pinned.py tests/test_gpu_rollout.py -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r4/t_pair.txt
cat gpurun_out/r4/t_pair.txt
python bench.py --steps 60 --warmup 10 --no-cpu-baseline --epoch-frames 0 > gpurun_out/r4/bench_pair.json 2> gpurun_out/r4/bench_pair.err
