mkdir -p /root/repo/gpurun_out/r4
python -m pytest tests/test_gpu_train.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r4/t_train.txt
cat gpurun_out/r4/t_train.txt
