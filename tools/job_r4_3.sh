mkdir -p gpurun_out/r4
( for v in "f64 25000" "sort 25000"; do set -- $v; echo "=== NEUMA_SCATTER=$1 N=$2" ; NM_EXP_N=$2 NEUMA_SCATTER=$1 python tools/exp_scatter_phases.py metric 2>&1 | grep -v "after rollout" | tail -20; done ) > gpurun_out/r4/scatter_phases_25k.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for v in "f64 0" "f64 4" "sort 4"; do set -- $v; NEUMA_HIP_LIB=/root/repo/tools/libneuma_phases.so NEUMA_SCATTER=$1 NM_DBG=$2 rocprofv3 --kernel-trace --output-format csv -d /tmp/trd_$1_$2 -o tr -- python /root/repo/tools/run_rollout.py metric 4 > /tmp/trd.log 2>&1; python /root/repo/tools/timeline.py /tmp/trd_$1_$2 20 2>&1 | head -14 > /root/repo/gpurun_out/r4/dbg_timeline_$1_$2.md; done
