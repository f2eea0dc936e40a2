mkdir -p /root/repo/gpurun_out/r4
( for v in "0" "128"; do echo "=== p2g f64 NM_DBG=$v"; NM_DBG=$v python tools/exp_scatter_phases.py metric 2>&1 | grep -v "after rollout" | tail -19; done
echo "=== g2p_bwd f64"; python tools/exp_scatter_phases.py metric bwd 2>&1 | grep -v "after rollout" | tail -19 ) > gpurun_out/r4/scatter_phases_final.txt 2>&1
