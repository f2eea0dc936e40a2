mkdir -p gpurun_out/r4; rm -f gpurun_out/parity_measured.jsonl
for i in 1 2; do python -m pytest tests/test_gpu_pinned.py tests/test_gpu_mpm.py tests/test_gpu_svd_material.py tests/test_gpu_shard.py -x -q -m gpu 2>&1 | tail -3; done > gpurun_out/r4/t_parity.txt
cat gpurun_out/r4/t_parity.txt
