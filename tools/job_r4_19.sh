mkdir -p /root/repo/gpurun_out/r4
timeout 900 python bench.py --gpus 2 --steps 10 --warmup 3 --epoch-frames 0 > gpurun_out/r4/bench_gpus2_auto.json 2> gpurun_out/r4/bench_gpus2_auto.err; echo rc=$?; tail -c 1500 gpurun_out/r4/bench_gpus2_auto.json; tail -5 gpurun_out/r4/bench_gpus2_auto.err
timeout 900 python bench.py --gpus 2 --steps 10 --warmup 3 --epoch-frames 0 --shard-sim on > gpurun_out/r4/bench_gpus2_on.json 2> gpurun_out/r4/bench_gpus2_on.err; echo rc=$?; tail -c 600 gpurun_out/r4/bench_gpus2_on.json; tail -5 gpurun_out/r4/bench_gpus2_on.err
