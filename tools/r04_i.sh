cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04i
python3 -m pytest tests/test_fused_adamw_gpu.py tests/test_graph_replay_gpu.py -q -m gpu 2>&1 | grep -v GridwiseOp | tail -8 > gpurun_out/r04i/t.log
tail -5 gpurun_out/r04i/t.log
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04i
cd /tmp && export TMPDIR=/tmp
for M in SSND2Net M2Net; do
  rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_zoo -- python3 $GRAFT_REPO_ROOT/tools/bench_zoo.py --models $M --steps 3 --warmup 14 > $OUT/bench_$M.txt 2>&1
  python3 $GRAFT_REPO_ROOT/tools/kernel_summary.py $(ls $OUT/prof_zoo/*/*kernel_trace.csv | head -1) 30 0.8 > $OUT/${M}_graph_kernels.txt 2>&1
  rm -rf $OUT/prof_zoo
done
cd $GRAFT_REPO_ROOT
python3 tools/bench_zoo.py --models SSND2Net,M2Net,SwT2Net --steps 8 --warmup 14 2>/dev/null | grep '"model"' | cut -c1-120
tail -12 $OUT/SSND2Net_graph_kernels.txt
