mkdir -p gpurun_out/r4
( for n in 25000 50000; do echo "=== N=$n"; NM_EXP_N=$n python tools/exp_scatter_phases.py metric 2>&1 | grep -v "after rollout" | tail -18; done ) > gpurun_out/r4/scatter_phases_25k_b.txt 2>&1
