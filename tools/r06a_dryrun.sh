set -x
mkdir -p gpurun_out/r06a
export NEUMA_DIST_BACKEND=gloo
for mode in on off; do
  for ex in allreduce peers; do
    timeout 300 python bench.py --gpus 2 --steps 3 --warmup 1 --workload tiny --no-cpu-baseline --epoch-frames 0 --shard-sim $mode > gpurun_out/r06a/g2_${mode}_${ex}.json 2> gpurun_out/r06a/g2_${mode}_${ex}.err; echo "rc=$? $mode $ex"
  done
done
NEUMA_SHARD_EXCHANGE=peers timeout 300 python bench.py --gpus 2 --steps 3 --warmup 1 --workload tiny --no-cpu-baseline --epoch-frames 0 --shard-sim on > gpurun_out/r06a/g2_on_peersenv.json 2> gpurun_out/r06a/g2_on_peersenv.err; echo "rc=$?"
timeout 300 python bench.py --gpus 2 --steps 3 --warmup 1 --workload bb --no-cpu-baseline --epoch-frames 0 --shard-sim auto > gpurun_out/r06a/g2_bb_auto.json 2> gpurun_out/r06a/g2_bb_auto.err; echo "rc=$?"
timeout 600 python bench.py --gpus 8 --steps 3 --warmup 1 --workload tiny --no-cpu-baseline --epoch-frames 0 --shard-sim on > gpurun_out/r06a/g8_on.json 2> gpurun_out/r06a/g8_on.err; echo "rc=$?"
timeout 600 python bench.py --gpus 8 --steps 3 --warmup 1 --workload tiny --no-cpu-baseline --epoch-frames 0 --shard-sim off > gpurun_out/r06a/g8_off.json 2> gpurun_out/r06a/g8_off.err; echo "rc=$?"
unset NEUMA_DIST_BACKEND
timeout 300 python bench.py --workload tiny --steps 3 --warmup 1 --no-cpu-baseline --epoch-frames 0 > gpurun_out/r06a/g1_tiny.json 2> gpurun_out/r06a/g1_tiny.err
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r06a/bench_head.json 2> gpurun_out/r06a/bench_head.err; echo "rc=$?"
tail -c 600 gpurun_out/r06a/*.err | tail -80
