cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04n
python3 tools/bench_conv_layers.py 2>&1 | grep -v amdgpu > gpurun_out/r04n/layers.txt; cat gpurun_out/r04n/layers.txt | cut -c1-150
python3 -m pytest tests/test_plain_unet_gpu.py tests/test_conv_kernels_gpu.py tests/test_rebnconv_gpu.py tests/test_determinism_gpu.py tests/test_ssnd2net.py -x -q -m gpu 2>&1 | grep -v GridwiseOp | tail -8 > gpurun_out/r04n/t.log
tail -4 gpurun_out/r04n/t.log | cut -c1-250
python3 bench.py --no-swt2net --no-secondary --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['ms_per_step'], d['roofline']['wgrad_ms_per_step'], d['h2d_inclusive'])"
