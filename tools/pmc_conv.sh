#!/bin/bash
# usage: pmc_conv.sh TAG [bench --tune string]: FETCH_SIZE / WRITE_SIZE passes + kernel trace over two eager primary steps -> per-launch table
TAG=$1; TUNE=${2:-}
OUT=$GRAFT_REPO_ROOT/gpurun_out; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export NNZ_BENCH_LIVE_DICE=0 NNZ_UNET_GRAPH=0
B="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-swt2net --no-launch-timer --no-h2d-leg"
[ -n "$TUNE" ] && B="$B --tune $TUNE"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pf -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pw -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/pt -- $B > /dev/null 2>&1
python3 $R/tools/pmc_per_launch.py $(ls $OUT/pf/*/*counter_collection.csv | head -1) $(ls $OUT/pw/*/*counter_collection.csv | head -1) conv_box_kernel 49 $(ls $OUT/pt/*/*kernel_trace.csv | head -1) > $OUT/${TAG}_conv_per_launch.txt
rm -rf $OUT/pf $OUT/pw $OUT/pt
cat $OUT/${TAG}_conv_per_launch.txt
