#!/bin/bash
# A/B kernel stats of the primary bench under an environment switch: bash tools/ab_kernel_stats.sh VAR a b TAG
set -u
VAR=$1; A=$2; B=$3; TAG=${4:-ab}
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for v in $A $B; do
  export $VAR=$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_ab_$v -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-swt2net --no-launch-timer > /dev/null 2>&1
  cp $(ls $OUT/prof_ab_$v/*/*kernel_stats.csv | head -1) $OUT/${TAG}_${VAR}_$v.csv
  rm -rf $OUT/prof_ab_$v
done
