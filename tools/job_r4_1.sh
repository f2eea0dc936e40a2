mkdir -p gpurun_out/r4
python -m pytest tests/test_gpu_mpm.py tests/test_gpu_pinned.py tests/test_gpu_rollout.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r4/t_f64.txt
python bench.py --steps 40 --warmup 5 > gpurun_out/r4/bench_f64.json 2> gpurun_out/r4/bench_f64.err
NEUMA_SCATTER=sort python bench.py --steps 40 --warmup 5 > gpurun_out/r4/bench_sort.json 2> gpurun_out/r4/bench_sort.err
cd /tmp && export TMPDIR=/tmp
for n in 12500 25000 50000 100000; do rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$n -o tr -- python /root/repo/tools/run_rollout.py metric 6 $n > /tmp/tr_$n.log 2>&1; python /root/repo/tools/timeline.py /tmp/tr_$n 20 > /root/repo/gpurun_out/r4/cliff_timeline_$n.md 2>&1; done
cat /root/repo/gpurun_out/r4/t_f64.txt
