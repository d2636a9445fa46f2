#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd $R
python -m pytest tests/test_conv_kernels_gpu.py tests/test_determinism_gpu.py tests/test_plain_unet_gpu.py -m gpu -q -x -k "not full_size" 2>&1 | grep -v GridwiseOp | tail -3 > $OUT/r06_brick_tests.log
for i in 1 2; do
for v in 0 1; do
  python bench.py --steps 60 --warmup 15 --no-secondary --no-cpu-baseline --no-h2d-leg --tune conv14=$v 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; print('conv14=$v', j['value'], 'patches/s', j['ms_per_step'], 'ms  conv_box frac', r['frac'], 'avg_us', r['avg_launch_us'], 'conv ms/step', r['ms_per_step'], 'wgrad', r.get('wgrad_ms_per_step'))
" >> $OUT/r06_brick_ab.txt
done; done
python tools/bench_conv_layers.py --tuning 14=0 > $OUT/r06_conv_layers_brick0.txt 2>&1
python tools/bench_conv_layers.py --tuning 14=1 > $OUT/r06_conv_layers_brick1.txt 2>&1
cat $OUT/r06_brick_tests.log $OUT/r06_brick_ab.txt
