mkdir -p /root/repo/gpurun_out/r4
python -m pytest tests/test_gpu_mpm.py tests/test_gpu_rollout.py -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r4/t_mpm3.txt
cd /tmp && export TMPDIR=/tmp
for n in 12500 25000 50000 100000; do rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$n -o tr -- python /root/repo/tools/run_rollout.py metric 6 $n > /tmp/tr_$n.log 2>&1; python /root/repo/tools/timeline.py /tmp/tr_$n 20 > /root/repo/gpurun_out/r4/cliff3_timeline_$n.md 2>&1; done
cd /root/repo; for n in 100000 50000 25000 12500; do python tools/exp_shard_overhead.py metric $n 2>&1 | grep "us/substep"; done > gpurun_out/r4/shard_overhead_r04.txt 2>&1
cat gpurun_out/r4/t_mpm3.txt
