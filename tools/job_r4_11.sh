mkdir -p /root/repo/gpurun_out/r4
cd /tmp && export TMPDIR=/tmp
for n in 12500 25000 50000 100000; do rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$n -o tr -- python /root/repo/tools/run_rollout.py metric 6 $n > /tmp/tr_$n.log 2>&1; python /root/repo/tools/timeline.py /tmp/tr_$n 20 > /root/repo/gpurun_out/r4/cliff2_timeline_$n.md 2>&1; done
