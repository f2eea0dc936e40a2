mkdir -p /root/repo/gpurun_out/r4
python -m pytest tests/test_gpu_svd_material.py tests/test_gpu_rollout.py tests/test_gpu_pinned.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r4/t_pref.txt
cat gpurun_out/r4/t_pref.txt
python bench.py --steps 60 --warmup 10 --no-cpu-baseline --epoch-frames 0 > gpurun_out/r4/bench_pref.json 2> gpurun_out/r4/bench_pref.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o tr -- python /root/repo/tools/run_rollout.py metric 6 > /tmp/tl.log 2>&1; python /root/repo/tools/timeline.py /tmp/tl 20 > /root/repo/gpurun_out/r4/timeline_pref.md 2>&1
