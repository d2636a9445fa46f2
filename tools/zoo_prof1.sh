#!/bin/bash
# usage: zoo_prof1.sh TAG MODEL  -> gpurun_out/TAG_<model>_graph_kernels.txt (kernel trace of the replayed step, last quarter aggregated)
TAG=$1; M=$2
OUT=$GRAFT_REPO_ROOT/gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_zoo -- python3 $R/tools/bench_zoo.py --models $M --steps 3 --warmup 3 > $OUT/${TAG}_${M,,}_bench.txt 2>&1
python3 $R/tools/kernel_summary.py $(ls $OUT/prof_zoo/*/*kernel_trace.csv | head -1) 70 0.75 > $OUT/${TAG}_${M,,}_graph_kernels.txt 2>&1
rm -rf $OUT/prof_zoo
head -40 $OUT/${TAG}_${M,,}_graph_kernels.txt | cut -c1-150
