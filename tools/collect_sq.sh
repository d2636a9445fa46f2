#!/bin/bash
# The two SQ-counter passes of tools/collect_profiles.sh (steps 2a and 4) on their own:  bash tools/collect_sq.sh r06
set -u
TAG=${1:-r06}
OUT=$GRAFT_REPO_ROOT/gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
CTR="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_ANY"
export NNZ_BENCH_LIVE_DICE=0 NNZ_UNET_GRAPH=0
rocprofv3 --pmc $CTR --output-format csv -d $OUT/prof_sq -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-swt2net --no-launch-timer > /dev/null 2>&1
python3 $R/tools/pmc_kernel_sums.py $(ls $OUT/prof_sq/*/*counter_collection.csv | head -1) conv_box_kernel conv_wgrad_kernel norm_kernel > $OUT/${TAG}_primary_pmc_sq_summary.json 2>&1
rm -rf $OUT/prof_sq
unset NNZ_BENCH_LIVE_DICE NNZ_UNET_GRAPH
rocprofv3 --pmc $CTR --output-format csv -d $OUT/prof_wa_pmc -- python3 $R/tools/bench_zoo.py --models SwT2Net --steps 1 --warmup 1 --graph 0 > /dev/null 2>&1
python3 $R/tools/pmc_kernel_sums.py $(ls $OUT/prof_wa_pmc/*/*counter_collection.csv | head -1) win_attn dense32 > $OUT/${TAG}_swt2net_pmc_sq_summary.json 2>&1
rm -rf $OUT/prof_wa_pmc
head -30 $OUT/${TAG}_primary_pmc_sq_summary.json
