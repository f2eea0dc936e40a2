mkdir -p gpurun_out/r4
python -m pytest tests/test_gpu_shard.py tests/test_gpu_pinned.py tests/test_gpu_mpm.py tests/test_gpu_svd_material.py -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r4/t_shard2.txt
cat gpurun_out/r4/t_shard2.txt
for c in rccl python; do NEUMA_COMM=$c python tools/exp_shard_overhead.py metric 2>&1 | grep "us/substep"; done > gpurun_out/r4/shard_overhead_comm.txt 2>&1
for n in 50000 25000 12500; do python tools/exp_shard_overhead.py metric $n 2>&1 | grep "us/substep"; done >> gpurun_out/r4/shard_overhead_comm.txt 2>&1
NEUMA_SHARD_FORCE=1 python bench.py --shard-sim on --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r4/bench_shard1.json 2> gpurun_out/r4/bench_shard1.err
python bench.py --steps 40 --warmup 5 > gpurun_out/r4/bench_g.json 2> gpurun_out/r4/bench_g.err
