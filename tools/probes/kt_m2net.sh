OUT=$GRAFT_REPO_ROOT/gpurun_out; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_zoo -- python3 $R/tools/bench_zoo.py --models M2Net --steps 3 --warmup 3 > /dev/null 2>&1
python3 $R/tools/kernel_summary.py $(ls $OUT/prof_zoo/*/*kernel_trace.csv | head -1) 60 0.75 > $OUT/s22_m2net_kernels.txt 2>&1
rm -rf $OUT/prof_zoo
grep -i "copy\|dense32\|kernels " $OUT/s22_m2net_kernels.txt | cut -c1-150
