#!/bin/bash
# Refresh the bench line of record and the zoo numbers without redoing the counter passes: bash tools/refresh_bench.sh r03
set -u
TAG=${1:-r03}
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_zoo -- python3 $GRAFT_REPO_ROOT/tools/bench_zoo.py --models SwT2Net --steps 3 --warmup 3 > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/tools/kernel_summary.py $(ls $OUT/prof_zoo/*/*kernel_trace.csv | head -1) 45 0.75 > $OUT/${TAG}_swt2net_graph_kernels.txt 2>&1
rm -rf $OUT/prof_zoo
rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_zoo -- python3 $GRAFT_REPO_ROOT/tools/bench_zoo.py --models M2Net --steps 3 --warmup 3 > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/tools/kernel_summary.py $(ls $OUT/prof_zoo/*/*kernel_trace.csv | head -1) 45 0.75 > $OUT/${TAG}_m2net_graph_kernels.txt 2>&1
python3 $GRAFT_REPO_ROOT/tools/kernel_histogram.py $(ls $OUT/prof_zoo/*/*kernel_trace.csv | head -1) 0.4 22 > $OUT/${TAG}_m2net_kernel_histogram.txt 2>&1
rm -rf $OUT/prof_zoo
cd $GRAFT_REPO_ROOT
python3 tools/bench_zoo.py --models M2NetP,M2Net,SwT2Net,SSND2Net,MambaND2Net,UNETR2Net,LightMamba2Net,LightMamba2NetP --steps 5 --warmup 3 2>&1 | grep '"model"' > $OUT/${TAG}_zoo_bench.txt
python3 bench.py > $OUT/${TAG}_bench_n1.json 2>$OUT/${TAG}_bench_n1.err
tail -c 600 $OUT/${TAG}_bench_n1.json
