mkdir -p /root/repo/gpurun_out/r4
python -m pytest tests/test_gpu_mpm.py -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r4/t_modes.txt; cat gpurun_out/r4/t_modes.txt
grep "NEUMA_SCATTER" gpurun_out/parity_measured.jsonl | tail -20
