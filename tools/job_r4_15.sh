mkdir -p /root/repo/gpurun_out/r4
python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r4/bench_epoch.json 2> gpurun_out/r4/bench_epoch.err
tail -5 gpurun_out/r4/bench_epoch.err
python tools/exp_shard_overhead.py stress 2>&1 | grep "us/substep" > gpurun_out/r4/shard_overhead_1m.txt
