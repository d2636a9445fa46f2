set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/p_kt -- python3 $R/tools/bench_scan.py --xs-only > /dev/null 2>&1
python3 $R/tools/kernel_summary.py $(ls $OUT/p_kt/*/*kernel_trace.csv | head -1) 12 0 > $OUT/s19_pf_kt.txt 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/p_a -- python3 $R/tools/bench_scan.py --xs-only > /dev/null 2>&1
python3 $R/tools/pmc_kernel_sums.py $(ls $OUT/p_a/*/*counter_collection.csv | head -1) xs_rl_bwd_kernel xs_rl_fwd_final > $OUT/s19_pf_pmc_a.json 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_IFETCH SQ_INST_LEVEL_VMEM --output-format csv -d $OUT/p_b -- python3 $R/tools/bench_scan.py --xs-only > /dev/null 2>&1
python3 $R/tools/pmc_kernel_sums.py $(ls $OUT/p_b/*/*counter_collection.csv | head -1) xs_rl_bwd_kernel xs_rl_fwd_final > $OUT/s19_pf_pmc_b.json 2>&1
rm -rf $OUT/p_kt $OUT/p_a $OUT/p_b
cat $OUT/s19_pf_kt.txt | cut -c1-150
