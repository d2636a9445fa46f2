cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04l
python3 -m pytest tests/test_plain_unet_gpu.py tests/test_determinism_gpu.py tests/test_conv_kernels_gpu.py tests/test_rebnconv_gpu.py -x -q -m gpu 2>&1 | grep -v GridwiseOp | tail -12 > gpurun_out/r04l/t.log
tail -6 gpurun_out/r04l/t.log | cut -c1-250
for T in 1 2 4; do python3 tools/bench_conv_layers.py --tuning 9=$T --only 0. 2>&1 | grep -E "tuning|enc0.1|dec0.0"; python3 tools/bench_conv_layers.py --tuning 9=$T --only 1.1 2>&1 | grep -E "enc1.1"; done > gpurun_out/r04l/persist.txt
cat gpurun_out/r04l/persist.txt | cut -c1-160
for T in 1 4; do python3 bench.py --no-swt2net --no-secondary --no-cpu-baseline --no-h2d-leg --tune conv9=$T 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('T=$T', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['ms_per_step'])"; done
