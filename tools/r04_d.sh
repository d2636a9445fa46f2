cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04e
python3 -m pytest tests/test_plain_unet_gpu.py -x -q -m gpu -k "consumer_side" 2>&1 | grep -v GridwiseOp | tail -30 > gpurun_out/r04e/t.log
cat gpurun_out/r04e/t.log
python3 tools/bench_conv_layers.py --innorm 0 > gpurun_out/r04e/layers_in0.txt 2>&1
python3 tools/bench_conv_layers.py --innorm 1 > gpurun_out/r04e/layers_in1.txt 2>&1
paste -d'\n' gpurun_out/r04e/layers_in0.txt gpurun_out/r04e/layers_in1.txt | grep -v amdgpu | cut -c1-150
for m in 0 1; do
NNZ_CONSUMER_NORM=$m python3 bench.py --no-swt2net --no-secondary --no-cpu-baseline > gpurun_out/r04e/bench_cn$m.json 2> gpurun_out/r04e/bench_cn$m.err
done
python3 -c "
import json
for f in ['bench_cn0','bench_cn1']:
    try:
        d=json.load(open('gpurun_out/r04e/%s.json'%f)); print(f, d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['ms_per_step'], d['roofline']['wgrad_ms_per_step'])
    except Exception as e: print(f, 'ERR', e)
"
