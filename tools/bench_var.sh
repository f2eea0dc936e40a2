#!/bin/bash
# repeat the metric bench: frames/s and per-frame GPU time percentiles (run-to-run stability)
for i in $(seq 1 ${1:-4}); do
python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['ms_per_step'], d['frame_ms_gpu'], d['roofline']['avg_us'])"
done
