cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04c
python3 -m pytest tests/test_plain_unet_gpu.py tests/test_rebnconv_gpu.py tests/test_determinism_gpu.py tests/test_conv_kernels_gpu.py -x -q -m gpu 2>&1 | grep -v GridwiseOp > gpurun_out/r04c/t.log
tail -5 gpurun_out/r04c/t.log
for m in 0 1; do
NNZ_CONSUMER_NORM=$m python3 bench.py --no-swt2net --no-secondary --no-cpu-baseline > gpurun_out/r04c/bench_cn$m.json 2> gpurun_out/r04c/bench_cn$m.err
done
python3 -c "
import json
for f in ['bench_cn0','bench_cn1']:
    try:
        d=json.load(open('gpurun_out/r04c/%s.json'%f)); print(f, d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['ms_per_step'], d['roofline']['wgrad_ms_per_step'])
    except Exception as e: print(f, 'ERR', e)
"
cd /tmp && export TMPDIR=/tmp
for m in 0 1; do
NNZ_CONSUMER_NORM=$m rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04c/prof$m -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-swt2net > /dev/null 2>&1
cp $(ls $GRAFT_REPO_ROOT/gpurun_out/r04c/prof$m/*/*kernel_stats.csv | head -1) $GRAFT_REPO_ROOT/gpurun_out/r04c/stats_cn$m.csv
rm -rf $GRAFT_REPO_ROOT/gpurun_out/r04c/prof$m
done
