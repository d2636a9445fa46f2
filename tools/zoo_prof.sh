#!/bin/bash
# usage: zoo_prof.sh TAG  -> gpurun_out/TAG_{m2net,swt2net}_graph_kernels.txt + small-op sources
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for M in M2Net SwT2Net; do
  rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_zoo -- python3 $R/tools/bench_zoo.py --models $M --steps 3 --warmup 3 > $OUT/${TAG}_${M,,}_bench.txt 2>&1
  python3 $R/tools/kernel_summary.py $(ls $OUT/prof_zoo/*/*kernel_trace.csv | head -1) 70 0.75 > $OUT/${TAG}_${M,,}_graph_kernels.txt 2>&1
  rm -rf $OUT/prof_zoo
  python3 $R/tools/probes/m2net_small_op_sources.py $M 2>/dev/null | head -70 > $OUT/${TAG}_${M,,}_small_op_sources.txt
done
cd $R
python3 tools/bench_zoo.py --models M2Net,SwT2Net --steps 6 --warmup 8 2>&1 | grep '"model"' > $OUT/${TAG}_zoo_bench.txt
cat $OUT/${TAG}_zoo_bench.txt
