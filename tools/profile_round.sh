#!/bin/bash
# The four rocprofv3 passes + bench lines behind profiles/<tag>_* (run on the GPU box from the repo root: bash tools/profile_round.sh r03c)
tag=${1:-rXX}
R=$PWD; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
CMD="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --epoch-frames 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $CMD > $O/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $CMD > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- $CMD > $O/write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/mfma -- $CMD > $O/mfma.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$tag -- python $R/tools/run_rollout.py metric 6 > /dev/null 2>&1
cd $R
python tools/timeline.py /tmp/tl_$tag 20 > $O/rollout_timeline.md
# keep only the small summaries (the merge back is capped at 64 MiB)
find $O -name "*kernel_trace.csv" -size +8M -delete
for w in metric bb jd sf burger stress; do python bench.py --workload $w --no-cpu-baseline --epoch-frames 0 2>/dev/null | tail -1 > $O/bench_$w.json; done
python bench.py --state deformed --no-cpu-baseline --epoch-frames 0 2>/dev/null | tail -1 > $O/bench_metric_deformed.json
python bench.py --state impact --no-cpu-baseline --epoch-frames 0 2>/dev/null | tail -1 > $O/bench_metric_impact.json
python bench.py 2>/dev/null | tail -1 > $O/bench_metric_cpu.json
du -sh $O
# memory-side atomics (round 6: what the reverse compositing's per-(tile, Gaussian) flush costs at the fabric) - own pass
cd /tmp
rocprofv3 -L 2>/dev/null | grep -i -o "TCC_EA0_ATOMIC[A-Z_a-z0-9]*\|TCC_ATOMIC[A-Z_a-z0-9]*\|TCC_EA0_WRREQ[A-Z_a-z0-9]*\|TCC_EA0_RDREQ[A-Z_a-z0-9]*" | sort -u > $O/atomic_counters_available.txt
rocprofv3 --pmc TCC_EA0_ATOMIC_sum TCC_ATOMIC_sum --output-format csv -d $O/atomic -- $CMD > $O/atomic.log 2>&1
cd $R
