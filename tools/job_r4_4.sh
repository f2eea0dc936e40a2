mkdir -p gpurun_out/r4
python -m pytest tests/test_gpu_mpm.py tests/test_gpu_pinned.py tests/test_gpu_rollout.py -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r4/t_f64b.txt
python bench.py --steps 40 --warmup 5 > gpurun_out/r4/bench_f64b.json 2> gpurun_out/r4/bench_f64b.err
( for v in "f64 0" "f64 128"; do set -- $v; echo "=== NEUMA_SCATTER=$1 NM_DBG=$2" ; NEUMA_SCATTER=$1 NM_DBG=$2 python tools/exp_scatter_phases.py metric 2>&1 | grep -v "after rollout" | tail -20; done
echo "=== bwd f64"; python tools/exp_scatter_phases.py metric bwd 2>&1 | grep -v "after rollout" | tail -20 ) > gpurun_out/r4/scatter_phases_b.txt 2>&1
cat gpurun_out/r4/t_f64b.txt
