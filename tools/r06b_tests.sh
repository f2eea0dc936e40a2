set -x
mkdir -p gpurun_out/r06b
rm -f gpurun_out/parity_measured.jsonl
for i in 1 2 3 4 5; do
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_rollout.py -q -x -k "second_stream or native_epoch or one_node_frame or skipping_the_zero" 2>&1 | tail -5
done
cp gpurun_out/parity_measured.jsonl gpurun_out/r06b/parity_far_5runs.jsonl
rm -f gpurun_out/parity_measured.jsonl
timeout 3000 python -m pytest tests -m gpu -q -x 2>&1 | tail -15
cp gpurun_out/parity_measured.jsonl gpurun_out/r06b/parity_full.jsonl
