#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh r04
# Writes under gpurun_out/ (copy what is to be judged into profiles/).
set -u
TAG=${1:-r04}
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-swt2net"
# 1. kernel trace + stats of the primary bench command
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -- $BENCH > $OUT/${TAG}_bench_n1_profiled_run.json 2>/dev/null
cp $(ls $OUT/prof_stats/*/*kernel_stats.csv | head -1) $OUT/${TAG}_bench_n1_kernel_stats.csv
# 2. HBM traffic counters, separate passes (MI355X_MICROARCH.md, HBM / rocprofv3 section)
# (eager steps for the counter passes: NNZ_UNET_GRAPH=0 - the same kernels, dispatched one by one)
BENCH2="python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-swt2net --no-launch-timer"
export NNZ_UNET_GRAPH=0
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_f -- $BENCH2 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_w -- $BENCH2 > /dev/null 2>&1
cp $(ls $OUT/prof_f/*/*counter_collection.csv | head -1) $OUT/${TAG}_pmc_fetch_size.csv
cp $(ls $OUT/prof_w/*/*counter_collection.csv | head -1) $OUT/${TAG}_pmc_write_size.csv
unset NNZ_UNET_GRAPH
# 3. the zoo steps as they run (hipGraph replay): kernel trace, second half of the run aggregated by kernel
for M in M2Net SwT2Net SSND2Net LightMamba2Net UNETR2Net; do
  rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_zoo -- python3 $GRAFT_REPO_ROOT/tools/bench_zoo.py --models $M --steps 3 --warmup 3 > /dev/null 2>&1
  python3 $GRAFT_REPO_ROOT/tools/kernel_summary.py $(ls $OUT/prof_zoo/*/*kernel_trace.csv | head -1) 45 0.75 > $OUT/${TAG}_${M,,}_graph_kernels.txt 2>&1
  rm -rf $OUT/prof_zoo
done
# 4. one SQ counter pass over the scan kernels (both generations)
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/prof_zoo_pmc -- python3 $GRAFT_REPO_ROOT/tools/bench_scan.py --xs-only > /dev/null 2>&1
cp $(ls $OUT/prof_zoo_pmc/*/*counter_collection.csv | head -1) $OUT/${TAG}_scan_pmc_sq.csv
rm -rf $OUT/prof_stats $OUT/prof_f $OUT/prof_w $OUT/prof_zoo_pmc
cd $GRAFT_REPO_ROOT
python3 tools/bench_scan.py > $OUT/${TAG}_scan_bench.txt 2>&1
python3 tools/bench_conv_layers.py > $OUT/${TAG}_conv_layers.txt 2>&1
# (warm-up 14: the GradScaler of the autocast nets backs off for ~10 steps on the seeded SSND2Net - DESIGN section 2 - and the
#  timed steps should be applied ones)
python3 tools/bench_zoo.py --models M2NetP,M2Net,SwT2Net,SSND2Net,SSND2NetP,MambaND2Net,UNETR2Net,LightMamba2Net,LightMamba2NetP,LM2Net,SegMamba --steps 6 --warmup 14 2>&1 | grep '"model"' > $OUT/${TAG}_zoo_bench.txt
# round 4: same-box A/B of the consumer-side norm (kernel stats of the primary bench under both settings); conv experiments
bash tools/ab_kernel_stats.sh NNZ_CONSUMER_NORM 0 1 ${TAG}_ab
cd $GRAFT_REPO_ROOT
( for T in "7=0" "7=1" "8=128" "8=256"; do python3 tools/bench_conv_layers.py --tuning $T 2>&1 | grep -E "tuning|enc1.0|enc2.0|enc3.1|dec3.0|TOTAL"; done
  echo "--- forward with a raw (consumer-normalised) input and flipped weight gradient: --innorm 0 / 1"
  python3 tools/bench_conv_layers.py --innorm 0 2>&1 | grep -E "enc0.1|dec0.0|enc1.0|enc1.1|dec1.0|TOTAL"
  python3 tools/bench_conv_layers.py --innorm 1 2>&1 | grep -E "enc0.1|dec0.0|enc1.0|enc1.1|dec1.0|TOTAL" ) > $OUT/${TAG}_conv_experiments.txt
# round 4: phase timestamps of a conv_box workgroup (instrumented library tools/probes/_ts/libnnuzoo_hip_ts.so, built on this box if
# the snapshot does not carry it); knob 11 = 0 / 1: the forward statistics with VALU sums / on the matrix cores
for K in 0 1; do python3 tools/probes/conv_phase_probe.py --only enc0.1 --tuning 11=$K 2>&1 | grep -v -E "amdgpu|Warning|_benchmark" > $OUT/${TAG}_conv_phases_moments$K.txt; done
python3 tools/probes/conv_phase_probe.py --only enc1.1 2>&1 | grep -v -E "amdgpu|Warning|_benchmark" > $OUT/${TAG}_conv_phases_enc1_1.txt
# weight-gradient kernel: phases per tile (timestamp build tools/probes/_ts/libnnuzoo_hip_wts.so) and the in-kernel clocks of both kernels
python3 tools/probes/conv_phase_probe.py --wgrad --clock 1.5 2>&1 | grep -v -E "amdgpu|Warning|_benchmark" > $OUT/${TAG}_wgrad_phases_now.txt
python3 tools/probes/conv_phase_probe.py --only enc0.1 --clock 1.5 2>&1 | grep clock > $OUT/${TAG}_conv_box_clocks_now.txt
python3 tools/bench_zoo.py --models SwinUMambaD,SwinUMamba --steps 6 --warmup 10 2>&1 | grep '"model"' >> $OUT/${TAG}_zoo_bench.txt
python3 tools/probes/ssnd2net_loss_probe.py --size 512 --steps 14 2>/dev/null | grep '^{' | cut -c1-260 > $OUT/${TAG}_ssnd2net_loss_probe.txt
# 4b. the window-attention kernels: SQ counters over one SwT2Net run (eager: one dispatch per kernel)
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d $OUT/prof_wa_pmc -- python3 $GRAFT_REPO_ROOT/tools/bench_zoo.py --models SwT2Net --steps 1 --warmup 1 --graph 0 > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/tools/pmc_kernel_sums.py $(ls $OUT/prof_wa_pmc/*/*counter_collection.csv | head -1) win_attn dense32 > $OUT/${TAG}_swt2net_pmc_sq_summary.json 2>&1
rm -rf $OUT/prof_wa_pmc
# 4c. window attention per stage shape; per-kernel duration distributions of the M2Net step
python3 $GRAFT_REPO_ROOT/tools/bench_window_attention.py > $OUT/${TAG}_window_attention_bench.txt 2>/dev/null
rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_zoo -- python3 $GRAFT_REPO_ROOT/tools/bench_zoo.py --models M2Net --steps 3 --warmup 3 > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/tools/kernel_histogram.py $(ls $OUT/prof_zoo/*/*kernel_trace.csv | head -1) 0.4 22 > $OUT/${TAG}_m2net_kernel_histogram.txt 2>&1
rm -rf $OUT/prof_zoo
# 5. the bench line of record (defaults: both legs, cpu_baseline)
python3 bench.py > $OUT/${TAG}_bench_n1.json 2>$OUT/${TAG}_bench_n1.err
ls -la $OUT | tail -20
