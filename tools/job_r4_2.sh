mkdir -p gpurun_out/r4
for v in "f64 0" "f64 64" "f64 1" "f64 3" "sort 0"; do set -- $v; echo "=== NEUMA_SCATTER=$1 NM_DBG=$2" ; NEUMA_SCATTER=$1 NM_DBG=$2 python tools/exp_scatter_phases.py metric 2>&1 | grep -v "after rollout" | tail -22; done > gpurun_out/r4/scatter_phases.txt 2>&1
for v in "f64 0" "sort 0"; do set -- $v; echo "=== bwd NEUMA_SCATTER=$1 NM_DBG=$2" ; NEUMA_SCATTER=$1 NM_DBG=$2 python tools/exp_scatter_phases.py metric bwd 2>&1 | grep -v "after rollout" | tail -22; done >> gpurun_out/r4/scatter_phases.txt 2>&1
