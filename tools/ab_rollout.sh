#!/bin/bash
# A/B of the roll-out's kernels under environment switches: one rocprofv3 kernel trace of tools/run_rollout.py per setting,
# the per-substep timeline of each (tools/timeline.py) under gpurun_out/<tag>_<i>.md.   bash tools/ab_rollout.sh tag "A=1" "A=2 B=3" ...
tag=$1; shift
R=$PWD; mkdir -p $R/gpurun_out
cd /tmp; export TMPDIR=/tmp
i=0
for setting in "$@"; do
  rm -rf /tmp/ab_$tag_$i
  env $setting rocprofv3 --kernel-trace --output-format csv -d /tmp/ab_${tag}_$i -- python $R/tools/run_rollout.py metric 6 > /dev/null 2>&1
  (cd $R; echo "# $setting"; python tools/timeline.py /tmp/ab_${tag}_$i 20) > $R/gpurun_out/${tag}_$i.md
  echo "== $setting"; grep -E "^\| k_|timeline" $R/gpurun_out/${tag}_$i.md | head -14
  i=$((i+1))
done
