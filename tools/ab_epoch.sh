#!/bin/bash
# single-frame and epoch rates of bench.py under environment settings (same box).  bash tools/ab_epoch.sh "A=1" "B=0" ...
for cfg in "$@"; do
  echo "== $cfg $(env $cfg python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); e=d['epoch']; print(round(d['value'],1), e['metric']['frames_per_s'], e['metric recompute']['frames_per_s'], e['bb']['frames_per_s'])")"
done
