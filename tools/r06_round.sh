#!/bin/bash
# round 6: the full GPU suite on the shipped library, then the profile passes of tools/profile_round.sh under gpurun_out/<tag>
tag=${1:-r06a}
set -x
mkdir -p gpurun_out/$tag
rm -f gpurun_out/parity_measured.jsonl
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > gpurun_out/$tag/gputest.log
cat gpurun_out/$tag/gputest.log
cp gpurun_out/parity_measured.jsonl gpurun_out/$tag/parity_measured.jsonl
bash tools/profile_round.sh $tag
python tools/exp_frame_events.py > gpurun_out/$tag/frame_events.txt 2>&1
