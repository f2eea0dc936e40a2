#!/bin/bash
# round 6: gloo dry runs of the multi-rank bench path on one GPU (code-path checks, not rates), the phase counters of the final
# sources and the driver's own bench invocation -> gpurun_out/r06c (copied to profiles/r06_*)
O=gpurun_out/r06c; mkdir -p $O
export NEUMA_DIST_BACKEND=gloo
run() { name=$1; shift; timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --epoch-frames 0 "$@" > $O/$name.json 2> $O/$name.err; echo "$name rc=$?"; }
run gpus2_on_allreduce --gpus 2 --workload tiny --shard-sim on
NEUMA_SHARD_EXCHANGE=peers run gpus2_on_peers --gpus 2 --workload tiny --shard-sim on
run gpus2_off --gpus 2 --workload tiny --shard-sim off
run gpus2_bb_auto --gpus 2 --workload bb --shard-sim auto
run gpus8_on --gpus 8 --workload tiny --shard-sim on
run gpus8_off --gpus 8 --workload tiny --shard-sim off
unset NEUMA_DIST_BACKEND
run gpus1_tiny --workload tiny
python tools/exp_phases_rollout.py > $O/phases_rollout.txt 2>&1
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$?"
