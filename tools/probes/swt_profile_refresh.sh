OUT=$GRAFT_REPO_ROOT/gpurun_out; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_zoo -- python3 $R/tools/bench_zoo.py --models SwT2Net --steps 3 --warmup 3 > /dev/null 2>&1
python3 $R/tools/kernel_summary.py $(ls $OUT/prof_zoo/*/*kernel_trace.csv | head -1) 45 0.75 > $OUT/r05_swt2net_graph_kernels.txt 2>&1
rm -rf $OUT/prof_zoo
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d $OUT/prof_wa_pmc -- python3 $R/tools/bench_zoo.py --models SwT2Net --steps 1 --warmup 1 --graph 0 > /dev/null 2>&1
python3 $R/tools/pmc_kernel_sums.py $(ls $OUT/prof_wa_pmc/*/*counter_collection.csv | head -1) win_attn dense32 > $OUT/r05_swt2net_pmc_sq_summary.json 2>&1
rm -rf $OUT/prof_wa_pmc
for M in SwT2Net; do
  ZB="python3 $R/tools/bench_zoo.py --models $M --steps 1 --warmup 1 --graph 0"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_zf -- $ZB > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_zw -- $ZB > /dev/null 2>&1
  F=$(ls $OUT/prof_zf/*/*counter_collection.csv | head -1); W=$(ls $OUT/prof_zw/*/*counter_collection.csv | head -1)
  python3 $R/tools/pmc_traffic.py $F $W "win_attn" "" "tools/bench_zoo.py --models SwT2Net --steps 1 --warmup 1 --graph 0" > $OUT/win_attn_hbm_traffic.json
  rm -rf $OUT/prof_zf $OUT/prof_zw
done
head -n 6 $OUT/r05_swt2net_graph_kernels.txt | cut -c1-120
