mkdir -p /root/repo/gpurun_out/r4
python -m pytest tests/test_gpu_mpm.py tests/test_gpu_pinned.py tests/test_gpu_rollout.py -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r4/t_dpp.txt; cat gpurun_out/r4/t_dpp.txt
( echo "=== p2g f64"; python tools/exp_scatter_phases.py metric 2>&1 | grep -v "after rollout" | tail -19
echo "=== g2p_bwd f64"; python tools/exp_scatter_phases.py metric bwd 2>&1 | grep -v "after rollout" | tail -19 ) > gpurun_out/r4/scatter_phases_dpp.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o tr -- python /root/repo/tools/run_rollout.py metric 6 > /tmp/tl.log 2>&1; python /root/repo/tools/timeline.py /tmp/tl 20 > /root/repo/gpurun_out/r4/timeline_dpp.md 2>&1
